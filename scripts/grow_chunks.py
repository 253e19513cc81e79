"""configs[1]'s reads added in several calls without a capacity hint (what mc_add_reads_file does with a large file): cost of a
table that has to grow under later batches.  Usage: python scripts/grow_chunks.py [n_reads] [n_calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import metacherchant_amd as m
R = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 6
k, L = 31, 150
dev = torch.device("cuda:0")
d_words = torch.empty((R * L + 31) // 32 + 1, dtype=torch.int64, device=dev)
d_off = torch.empty(R + 1, dtype=torch.int64, device=dev)
ctx = m.Context(k, m.KEY_PACKED, 0, 0)
ctx.set_coverage_hint(5)
ctx.synth_reads_dev(20240531, 10, 5_000_000, 42, 0, R, L, 100, d_words, d_off)
per = R // C // 32 * 32  # (reads per call: whole words)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(C):
    a, b = i * per, (R if i == C - 1 else (i + 1) * per)
    t1 = time.perf_counter()
    ctx.add_reads_packed_dev(d_words[a * L // 32:], d_off[a:b + 1] - a * L, b - a, (b - a) * L)
    torch.cuda.synchronize()
    st = ctx.stats()
    print("call %d: %.1f ms, kernels so far %.1f ms, rebuilds %d, table %.1f GB" % (i, 1e3 * (time.perf_counter() - t1), st.count_ms, st.grows, st.table_bytes / 1e9), flush=True)
nd = ctx.finalize()
torch.cuda.synchronize()
print("total %.1f ms, %d distinct" % (1e3 * (time.perf_counter() - t0), nd))
