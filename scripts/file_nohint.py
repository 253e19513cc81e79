"""mc_add_reads_file of a FASTA with sequencing errors (configs[1]'s reads) into a context without a capacity hint: the
device tokeniser's chunks, the table sized after the first one.  Usage: python scripts/file_nohint.py [n_reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import metacherchant_amd as m
R = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
k, L = 31, 150
dev = torch.device("cuda:0")
tmp = os.environ.get("TMPDIR", "/tmp")
path = os.path.join(tmp, "nohint_reads.fasta")
d_words = torch.empty((R * L + 31) // 32 + 1, dtype=torch.int64, device=dev)
d_off = torch.empty(R + 1, dtype=torch.int64, device=dev)
g = m.Context(k, m.KEY_PACKED, 0, 1 << 20)
g.synth_reads_dev(20240531, 10, 5_000_000, 42, 0, R, L, 100, d_words, d_off)
g.close()
w = d_words.cpu().numpy().view(np.uint64)
lut = np.frombuffer(b"AGCT", dtype=np.uint8)
with open(path, "wb") as f:
    B = 500_000
    for a in range(0, R, B):
        n = min(B, R - a)
        pos = (np.arange(n * L, dtype=np.uint64) + np.uint64(a * L))
        codes = ((w[pos >> np.uint64(5)] >> (np.uint64(62) - np.uint64(2) * (pos & np.uint64(31)))) & np.uint64(3)).astype(np.uint8)
        seqs = lut[codes].reshape(n, L)
        rec = np.empty((n, L + 4), dtype=np.uint8)
        rec[:, 0] = ord(">"); rec[:, 1] = ord("r"); rec[:, 2] = ord("\n"); rec[:, 3:3 + L] = seqs; rec[:, 3 + L] = ord("\n")
        f.write(rec.tobytes())
del w, d_words, d_off
os.environ["MC_INGEST_DEBUG"] = "1"
for hint in (0, int(50e6 + R * 120 * 0.27) + (1 << 20)):
    ctx = m.Context(k, m.KEY_PACKED, 0, hint)
    ctx.set_coverage_hint(5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = ctx.add_reads_file(path)
    nd = ctx.finalize()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    st = ctx.stats()
    print("hint %d: %d reads, %d distinct, %.1f ms (counting kernels %.1f ms, %d table rebuilds, table %.1f GB)" % (hint, n, nd, 1e3 * (t1 - t0), st.count_ms, st.grows, st.table_bytes / 1e9), flush=True)
    ctx.close()
