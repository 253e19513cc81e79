"""Order-independent checksums of the table after counting R synthetic reads (k = 31, E1), for comparing one large
pipeline run against several smaller ones (MC_MAX_RUN_BASES).  Usage: python scripts/bigrun_check.py n_reads"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import metacherchant_amd as m
R = int(sys.argv[1]); k, L = 31, 150
dev = torch.device("cuda:0")
n_bases = R * L
d_words = torch.empty((n_bases + 31) // 32 + 1, dtype=torch.int64, device=dev)
d_off = torch.empty(R + 1, dtype=torch.int64, device=dev)
hint = int(50e6 + R * 120 * 0.27) + (1 << 20)
ctx = m.Context(k, m.KEY_PACKED, 0, hint)
ctx.set_coverage_hint(5)
ctx.synth_reads_dev(20240531, 10, 5_000_000, 42, 0, R, L, 100, d_words, d_off)
for rep in range(2):
    ctx.clear()
    torch.cuda.synchronize(); t0 = time.time()
    ctx.add_reads_packed_dev(d_words, d_off, R, n_bases)
    nd = ctx.finalize()
    torch.cuda.synchronize(); t1 = time.time()
st = ctx.stats()
gk = torch.empty(nd, dtype=torch.int64, device=dev); gc = torch.empty(nd, dtype=torch.int16, device=dev)
assert ctx.export_dev(0, gk, gc, nd) == nd
gc64 = gc.to(torch.int64)
M = (1 << 64) - 1
x = gk * (0x9E3779B97F4A7C15 - (1 << 64)) + gc64 * (0xC2B2AE3D27D4EB4F - (1 << 64))
x = x ^ ((x >> 31) & ((1 << 33) - 1))
x = x * (0xD6E8FEB86659FD93 - (1 << 64))
print("CHECK distinct=%d sum=%d kc=%d mix=%d solid=%d windows=%d | %.1f ms, launches %d, p1 %.1f p2 %.1f p3 %.1f" % (
    nd, int(gc64.sum().item()), int((gk * gc64).sum().item()) & M, int(x.sum().item()) & M, int((gc >= 5).sum().item()), st.windows // 2,
    1e3 * (t1 - t0), st.count_launches, st.p1_ms, st.p2_ms, st.p3_ms), flush=True)
