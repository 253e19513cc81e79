"""Tuning: configs[2] scaled to 10 M reads (k = 63 polynomial keys) counted a few times; prints the levels' times.
MC_LIB selects a tuning build.  python scripts/long_probe.py [reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import metacherchant_amd as m
R = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L, k = 150, 63
m.native.load()
dev = torch.device("cuda:0")
n_bases = R * L
words = torch.empty((n_bases + 31) // 32 + 1, dtype=torch.int64, device=dev)
off = torch.empty(R + 1, dtype=torch.int64, device=dev)
gen = m.Context(31, m.KEY_PACKED, 0, 1 << 20)
from bench import GENOME_SEED, READ_SEED
gen.synth_reads_dev(GENOME_SEED, 10, R // 2, READ_SEED, 0, R, L, 100, words, off)
gen.close()
windows = R * (L - k + 1)
NO_HINT = os.environ.get("PROBE_NO_HINT") == "1"   # a fresh context without a capacity hint per iteration (the CLI's default)
ctx = m.Context(k, m.KEY_POLY, 0, 0 if NO_HINT else int(windows * 0.53) + (1 << 20))
ctx.set_coverage_hint(3)
for it in range(3):
    if NO_HINT and it:
        ctx.close()
        ctx = m.Context(k, m.KEY_POLY, 0, 0)
        ctx.set_coverage_hint(3)
    ctx.clear()
    ctx.reset_stats()
    ctx.add_reads_packed_dev(words, off, R, n_bases)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nd = ctx.finalize()
    t_fin = (time.perf_counter() - t0) * 1e3
    st = ctx.stats()
    # the walk of configs[2] (direction 0 from the seed gene, coverage 3): host time of the call, its set-up included
    import numpy as np
    seed = m.native.synth_genome(GENOME_SEED, 100000, 1000)
    sv = []
    for i in range(len(seed) - k + 1):
        v = 0
        for c in seed[i:i + k]:
            v = (v << 2) | int(c)
        sv.append(v)
    hi = np.array([v >> 64 for v in sv], dtype=np.uint64)
    lo = np.array([v & 0xFFFFFFFFFFFFFFFF for v in sv], dtype=np.uint64)
    t0 = time.perf_counter()
    res = ctx.bfs(hi, lo, 0, 3, 100000, -1)
    t_bfs = (time.perf_counter() - t0) * 1e3
    print("lib %s: distinct %d long_runs %d p1 %.2f p2 %.2f p3 %.2f ms (count %.2f), spills %d grows %d; finalize %.2f ms (key join %.2f ms, %d checks, %d keys in several regions, unchecked %d); walk %.2f ms host / %.2f device, %d vertices" % (
        os.path.basename(os.environ.get("MC_LIB", "product")), nd, st.long_runs, st.p1_ms, st.p2_ms, st.p3_ms, st.count_total_ms, st.spill_keys, st.grows,
        t_fin, st.dup_ms, st.dup_checks, st.dup_keys, st.dup_unchecked, t_bfs, res["device_ms"], len(res["hi"])), flush=True)
ctx.close()
