"""One in-process run of the device tokeniser over a synthetic FASTA and FASTQ (for rocprofv3 --kernel-trace --stats).
Usage: python scripts/tok_profile.py [n_reads]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metacherchant_amd as m  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
L = 150
rng = np.random.default_rng(0)
lut = np.frombuffer(b"AGCT", dtype=np.uint8)
genome = rng.integers(0, 4, 5_000_000).astype(np.uint8)
starts = rng.integers(0, len(genome) - L, n)
tmp = os.environ.get("TMPDIR", "/tmp")
fa, fq = os.path.join(tmp, "tokp.fasta"), os.path.join(tmp, "tokp.fastq")
with open(fa, "wb") as f, open(fq, "wb") as g:
    qual = b"I" * L
    for i in range(n):
        s = lut[genome[starts[i]:starts[i] + L]].tobytes()
        f.write(b">r%d\n%s\n" % (i, s))
        g.write(b"@r%d\n%s\n+\n%s\n" % (i, s, qual))
os.environ["MC_TOKENIZER"] = "device"
os.environ["MC_INGEST_DEBUG"] = "1"
for path in (fa, fq):
    ctx = m.Context(31, m.KEY_PACKED, 0, 6_000_000)
    for rep in range(3):
        t0 = time.time()
        r = ctx.add_reads_file(path)
        d = ctx.finalize()
        print("%s: %d reads, %d distinct, %.3f s" % (path, r, d, time.time() - t0), flush=True)
        ctx.clear()
    ctx.close()
