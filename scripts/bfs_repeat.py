#!/usr/bin/env python3
"""Repeats the same BFS many times and compares every result with the oracle's (a race shows as a run that differs)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import metacherchant_amd as mc
from oracle import pyoracle as po
from helpers import synth_case, oracle_table, seed_windows

genome, reads, off = synth_case(2, 200000, 80000, 150, 50)
half = 40000
seed = genome[10000:10500]
hi, lo = seed_windows(seed, 31)
ctx = mc.Context(31, mc.KEY_PACKED, 0, 3_000_000)
ctx.set_coverage_hint(5)
ctx.add_reads_packed(po.pack(reads[:off[half]]), off[:half + 1])
ctx.finalize()
t1, _ = oracle_table(reads[:off[half]], off[:half + 1], 31, po.KEY_PACKED)
bad = 0
for d in (-1, 1):
    want = po.bfs(t1, 31, po.KEY_PACKED, [seed], d, 5, 3000, -1)
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
        got = ctx.bfs(hi, lo, d, 5, 3000, -1)
        for f in ("hi", "lo", "dist", "cov", "last"):
            if not np.array_equal(got[f], want[f]):
                idx = np.nonzero(np.asarray(got[f]) != np.asarray(want[f]))[0]
                print("d=%d it=%d field %s differs at %s (of %d): got %s want %s  dist there %s" % (
                    d, it, f, idx[:10], len(want[f]), np.asarray(got[f])[idx[:10]], np.asarray(want[f])[idx[:10]], np.asarray(want["dist"])[idx[:10]]), flush=True)
                bad += 1
                break
print("bad runs:", bad)
