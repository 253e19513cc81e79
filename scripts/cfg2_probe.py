"""configs[2]-shaped counting at a chosen size, with the memory picture after every batch.
Usage: python scripts/cfg2_probe.py n_reads [contigs] [err] [hint]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import metacherchant_amd as m
R = int(sys.argv[1]); contigs = int(sys.argv[2]) if len(sys.argv) > 2 else 100
err = int(sys.argv[3]) if len(sys.argv) > 3 else 100
hint = int(sys.argv[4]) if len(sys.argv) > 4 else 0
k, L = 63, 150
dev = torch.device("cuda:0")
def free():
    f, t = torch.cuda.mem_get_info()
    return "%.1f GB used" % ((t - f) / 1e9)
ctx = m.Context(k, m.KEY_POLY, 0, hint)
ctx.set_coverage_hint(3)
print("created:", free(), flush=True)
B = min(R, 10_000_000)
d_words = torch.empty((B * L + 31) // 32 + 1, dtype=torch.int64, device=dev)
d_off = torch.empty(B + 1, dtype=torch.int64, device=dev)
t0 = time.time()
for first in range(0, R, B):
    n = min(B, R - first)
    ctx.synth_reads_dev(20240531, contigs, 5_000_000, 42, first, n, L, err, d_words, d_off)
    torch.cuda.synchronize()
    t1 = time.time()
    try:
        ctx.add_reads_packed_dev(d_words, d_off, n, n * L)
    except Exception as e:
        print("FAILED at read", first, e, free(), flush=True)
        raise
    torch.cuda.synchronize()
    st = ctx.stats()
    print("reads %d..%d: %.3f s, windows %d, %s, table_bytes %.1f GB" % (first, first + n, time.time() - t1, st.windows, free(), getattr(st, "table_bytes", 0) / 1e9), flush=True)
nd = ctx.finalize()
print("distinct", nd, "total %.2f s" % (time.time() - t0), free(), flush=True)
if os.environ.get("PROBE_STEPS"):
    import numpy as np
    seed = m.native.synth_genome(20240531, 100000, 1000)
    sv = []
    for i in range(len(seed) - k + 1):
        v = 0
        for c in seed[i:i + k]:
            v = (v << 2) | int(c)
        sv.append(v)
    hi = np.array([v >> 64 for v in sv], dtype=np.uint64); lo = np.array([v & (2**64 - 1) for v in sv], dtype=np.uint64)
    for s in range(int(os.environ["PROBE_STEPS"])):
        t1 = time.time()
        ctx.clear()
        print("step", s, "cleared", free(), flush=True)
        ctx.add_reads_packed_dev(d_words, d_off, B, B * L)
        nd = ctx.finalize()
        st = ctx.stats()
        print("step", s, "counted %.3f s" % (time.time() - t1), free(), "p1 %.1f p2 %.1f p3 %.1f spill %d grows %d launches %d" % (st.p1_ms, st.p2_ms, st.p3_ms, st.spill_keys, st.grows, st.count_launches), flush=True)
        ctx.reset_stats()
        res = ctx.bfs_batch([(hi, lo, -1), (hi, lo, 1), (hi, lo, 0)], 3, 100000, -1)
        print("step", s, "bfs done %.3f s" % (time.time() - t1), [len(r["hi"]) if r else 0 for r in res], free(), flush=True)
