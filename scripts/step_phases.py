"""Host-side wall clock of the phases of one bench step (configs[1]): where the time between the kernels goes.
Usage: python scripts/step_phases.py [n_reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import metacherchant_amd as m
R = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
k, L = 31, 150
dev = torch.device("cuda:0")
d_words = torch.empty((R * L + 31) // 32 + 1, dtype=torch.int64, device=dev)
d_off = torch.empty(R + 1, dtype=torch.int64, device=dev)
ctx = m.Context(k, m.KEY_PACKED, 0, int(50e6 + R * 120 * 0.27) + (1 << 20))
ctx.set_coverage_hint(5)
ctx.synth_reads_dev(20240531, 10, 5_000_000, 42, 0, R, L, 100, d_words, d_off)
seed = m.native.synth_genome(20240531, 100000, 1000)
sv = []
for i in range(len(seed) - k + 1):
    v = 0
    for c in seed[i:i + k]:
        v = (v << 2) | int(c)
    sv.append(v)
hi = np.array([v >> 64 for v in sv], dtype=np.uint64); lo = np.array([v & (2**64 - 1) for v in sv], dtype=np.uint64)
acc = {}
def lap(name, t0):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1
N = 6
for rep in range(N + 1):
    if rep == 1:
        acc.clear(); ctx.reset_stats()
    torch.cuda.synchronize(); t = time.perf_counter()
    ctx.clear(); t = lap("clear", t)
    ctx.add_reads_packed_dev(d_words, d_off, R, R * L); t = lap("add_reads", t)
    ctx.finalize(); t = lap("finalize", t)
    res = ctx.bfs_batch([(hi, lo, -1), (hi, lo, 1)], 5, 100000, -1); t = lap("bfs_batch", t)
st = ctx.stats()
print("per step, ms:", {k_: round(1e3 * v / N, 3) for k_, v in acc.items()}, "sum %.3f" % (1e3 * sum(acc.values()) / N))
print("kernel ms per step: count %.3f (p1 %.3f p2 %.3f p3 %.3f); bfs device_ms %s" % (st.count_ms / N, st.p1_ms / N, st.p2_ms / N, st.p3_ms / N, [round(r["device_ms"], 3) for r in res]))
