#!/usr/bin/env python3
"""BFS statistics of a small synthetic case (MC_BFS_STATS=1 prints the scout's counters): scripts/bfs_probe.py [err] [reads]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MC_BFS_STATS", "1")
import torch
import metacherchant_amd as m
err = int(sys.argv[1]) if len(sys.argv) > 1 else 100
R = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
k, L = 31, 150
dev = torch.device("cuda:0")
ctx = m.Context(k, m.KEY_PACKED, 0, 0)
ctx.set_coverage_hint(5)
d_words = torch.empty((R * L + 31) // 32 + 1, dtype=torch.int64, device=dev)
d_off = torch.empty(R + 1, dtype=torch.int64, device=dev)
ctx.synth_reads_dev(20240531, 1, 2000000, 42, 0, R, L, err, d_words, d_off)
ctx.add_reads_packed_dev(d_words, d_off, R, R * L)
print("distinct", ctx.finalize())
seed = m.native.synth_genome(20240531, 100000, 1000)
sv = []
for i in range(len(seed) - k + 1):
    v = 0
    for c in seed[i:i + k]:
        v = (v << 2) | int(c)
    sv.append(v)
hi = np.zeros(len(sv), dtype=np.uint64)
lo = np.array(sv, dtype=np.uint64)
for rep in range(2):
    res = ctx.bfs_batch([(hi, lo, -1), (hi, lo, 1)], 5, 90000, -1)
    print("device_ms", res[0]["device_ms"], "levels", res[0]["levels"], res[1]["levels"], "rounds", res[0]["rounds"], res[1]["rounds"])
