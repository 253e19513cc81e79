"""Repeats the three walks of one small case many times and compares each with the oracle's (a rare, timing-dependent
difference in the walk shows here): python scripts/bfs_stress.py [repeats] [k] [L] [reads] [contigs] [clen] [cov] [err]"""
import os
import sys
import time

os.environ.setdefault("MC_BFS_SELFCHECK", "1")  # every walk also checked on the device (bfs_device.h k_bfs_check)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import metacherchant_amd as mc
from oracle import pyoracle as po
from tests.helpers import GENOME_SEED, assert_bfs_equal, seed_windows

a = [int(x) for x in sys.argv[1:]]
reps, k, L, n_reads, contigs, clen, cov, err = (a + [2000, 29, 45, 193530, 3, 20000, 3, 0][len(a):])[:8]
genome = po.synth_genome(GENOME_SEED + 36, contigs * clen)
reads = po.synth_reads(genome, contigs, clen, 274713999, 0, n_reads, L, err)
off = np.arange(n_reads + 1, dtype=np.uint64) * L
t = po.Table()
t.count_reads(reads, off, k, po.KEY_PACKED)
FRESH = int(os.environ.get("STRESS_FRESH", 0))  # a new context every FRESH repeats (0: one context for all)
packed0, packed1 = po.pack(reads[:off[n_reads // 2]]), po.pack(reads[off[n_reads // 2]:])


def make_ctx():
    c = mc.Context(k, mc.KEY_PACKED, 0, 6_000_000)
    c.set_coverage_hint(cov)
    h = n_reads // 2
    c.add_reads_packed(packed0, off[:h + 1])
    c.add_reads_packed(packed1, off[h:] - off[h])
    assert c.finalize() == t.size()
    return c


ctx = make_ctx()
rng = np.random.default_rng(5)
bad = 0
t0 = time.time()
for rep in range(reps):
    if FRESH and rep and rep % FRESH == 0:
        ctx.close()
        ctx = make_ctx()
    if rep % 50 == 0:
        s0 = int(rng.integers(0, clen - 600))
        seed = genome[s0:s0 + 400]
        hi, lo = seed_windows(seed, k)
        want = {d: po.bfs(t, k, po.KEY_PACKED, [seed], d, cov, 20000, -1) for d in (1, -1, 0)}
    for d in (1, -1, 0):
        got = ctx.bfs(hi, lo, d, cov, 20000, -1)
        try:
            assert_bfs_equal(got, want[d])
        except AssertionError as e:
            bad += 1
            w = want[d]
            n = min(len(got["hi"]), len(w["hi"]))
            print("rep %d dir %d seed at %d: %s" % (rep, d, s0, str(e)[:200]), flush=True)
            for j in range(n, len(got["hi"])):
                key = int(got["lo"][j])
                dup = np.nonzero(np.asarray(got["lo"][:n]) == got["lo"][j])[0]
                print("   device only: entry %d lo=%x dist=%d cov=%d last=%d; same k-mer earlier in the result at %s; oracle table count %s" % (
                    j, key, got["dist"][j], got["cov"][j], got["last"][j], dup, t.get(np.array([key], dtype=np.uint64))), flush=True)
print("%d walks, %d differ (%.0f s)" % (3 * reps, bad, time.time() - t0))
