"""Wall-clock of mc_add_reads_file for a synthetic FASTA / FASTQ: the host parser on one core and on all of them, and
the device tokeniser (csrc/tokenizer.h); counting on the GPU in all three.  Usage: python scripts/ingest_bench.py [n_reads]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
L = 150
rng = np.random.default_rng(0)
lut = np.frombuffer(b"AGCT", dtype=np.uint8)
genome = rng.integers(0, 4, 5_000_000).astype(np.uint8)
starts = rng.integers(0, len(genome) - L, n)
tmp = os.environ.get("TMPDIR", "/tmp")
fa, fq = os.path.join(tmp, "ingest.fasta"), os.path.join(tmp, "ingest.fastq")
with open(fa, "wb") as f, open(fq, "wb") as g:
    qual = b"I" * L
    for i in range(n):
        s = lut[genome[starts[i]:starts[i] + L]].tobytes()
        f.write(b">r%d\n%s\n" % (i, s))
        g.write(b"@r%d\n%s\n+\n%s\n" % (i, s, qual))
code = r'''
import sys, time
sys.path.insert(0, %r)
import metacherchant_amd as m
ctx = m.Context(31, m.KEY_PACKED, 0, 6_000_000)
ctx.add_reads_file(sys.argv[1]); ctx.finalize(); ctx.clear()   # warm up (page cache, allocations)
t0 = time.time(); n = ctx.add_reads_file(sys.argv[1]); d = ctx.finalize(); t1 = time.time()
print("%%s: %%d reads, %%d distinct k-mers, %%.3f s = %%.1f Mbases/s" %% (sys.argv[1].rsplit(".", 1)[1], n, d, t1 - t0, n * %d / (t1 - t0) / 1e6))
''' % (ROOT, L)
for path in (fa, fq):
    for label, extra in (("host, 1 thread", {"MC_TOKENIZER": "host", "MC_INGEST_THREADS": "1"}), ("host, all cores", {"MC_TOKENIZER": "host"}),
                         ("device tokeniser", {"MC_TOKENIZER": "device", "MC_INGEST_DEBUG": "1"})):
        env = dict(os.environ)
        env.pop("MC_INGEST_THREADS", None)
        env.update(extra)
        out = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, env=env)
        print("%-17s" % label, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:])
        if "MC_INGEST_DEBUG" in extra:
            print("\n".join(l for l in out.stderr.splitlines() if "device tokeniser" in l))
