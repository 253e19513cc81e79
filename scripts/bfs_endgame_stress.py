"""The regime of the one wrong walk of round 3 (gpurun_out/soak_r3.log:80), made dense: direction-0 walks (two walkers, then
one) over error-free contigs whose component ends `delta` vertices under / over --maxkmers, several jobs per launch, every
result compared with the oracle's and (MC_BFS_SELFCHECK=1, set here) checked on the device; optionally while another
context counts reads on the same GPU, so that the walk's workgroups share the chip.

  python scripts/bfs_endgame_stress.py [--walks N] [--jobs J] [--k K] [--clen LEN] [--contend 0|1] [--dirs 0,1,-1] [--seed S]

Variants of the library (metacherchant_amd/build.py build_lib(variant=...)) are chosen with MC_LIB; MC_BFS_COMPANION=0 and
MC_BFS_DIRECT=0 bisect the companion workgroup and the walk over the counting table."""
import argparse
import os
import sys
import threading
import time

os.environ.setdefault("MC_BFS_SELFCHECK", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import metacherchant_amd as mc
from metacherchant_amd.native import McError
from oracle import pyoracle as po
from tests.helpers import GENOME_SEED, seed_windows

ap = argparse.ArgumentParser()
ap.add_argument("--walks", type=int, default=20000)
ap.add_argument("--jobs", type=int, default=3)
ap.add_argument("--k", type=int, default=29)
ap.add_argument("--clen", type=int, default=20000)
ap.add_argument("--contigs", type=int, default=3)
ap.add_argument("--L", type=int, default=45)
ap.add_argument("--cov", type=int, default=3)
ap.add_argument("--contend", type=int, default=0)
ap.add_argument("--dirs", default="0")
ap.add_argument("--seed", type=int, default=7)
ap.add_argument("--max-bad", type=int, default=5)
a = ap.parse_args()
k, clen, L = a.k, a.clen, a.L
dirs = [int(x) for x in a.dirs.split(",")]
n_reads = max(2000, int(40 * a.contigs * clen / L))  # ~40-fold
rng = np.random.default_rng(a.seed)
genome = po.synth_genome(GENOME_SEED + 36, a.contigs * clen)
reads = po.synth_reads(genome, a.contigs, clen, 274713999, 0, n_reads, L, 0)
off = np.arange(n_reads + 1, dtype=np.uint64) * L
t = po.Table()
t.count_reads(reads, off, k, po.KEY_PACKED)
ctx = mc.Context(k, mc.KEY_PACKED, 0, 6_000_000)
ctx.set_coverage_hint(a.cov)
ctx.add_reads_packed(po.pack(reads), off)
assert ctx.finalize() == t.size()
comp = clen - k + 1  # vertices of a whole contig, one orientation

stop = threading.Event()
busy = [0]


def contender():
    """counts a read set of its own over and over (the partitioned pipeline: thousands of workgroups per launch)"""
    g2 = po.synth_genome(GENOME_SEED + 99, 2_000_000)
    r2 = po.synth_reads(g2, 1, 2_000_000, 99, 0, 200_000, 150, 100)
    o2 = np.arange(200_001, dtype=np.uint64) * 150
    w2 = po.pack(r2)
    c2 = mc.Context(31, mc.KEY_PACKED, 0, 40_000_000)
    while not stop.is_set():
        c2.clear()
        c2.add_reads_packed(w2, o2)
        c2.finalize()
        busy[0] += 1
    c2.close()


th = None
if a.contend:
    th = threading.Thread(target=contender, daemon=True)
    th.start()

walks = bad = 0
t0 = time.time()
last = t0
want_cache = {}
while walks < a.walks and bad < a.max_bad:
    # a seed inside one contig; the cap lands delta under / over the component
    contig = int(rng.integers(0, a.contigs))
    s0 = contig * clen + int(rng.integers(0, clen - 400))
    seed = genome[s0:s0 + 400]
    hi, lo = seed_windows(seed, k)
    delta = int(rng.integers(-64, 65))
    cap = comp + delta
    jobs, wants = [], []
    for j in range(a.jobs):
        d = dirs[int(rng.integers(0, len(dirs)))]
        key = (s0, d, cap)
        if key not in want_cache:
            want_cache[key] = po.bfs(t, k, po.KEY_PACKED, [seed], d, a.cov, cap, -1)
        jobs.append((hi, lo, d))
        wants.append(want_cache[key])
    for rep in range(8):  # the same batch a few times: the oracle's walks are the expensive part of a batch
        try:
            got = ctx.bfs_batch(jobs, a.cov, cap, -1)
        except McError as e:
            bad += 1
            print("walk %d (seed at %d, cap %d = component %+d): %s" % (walks, s0, cap, delta, e), flush=True)
            walks += len(jobs)
            continue
        for j, (g, w) in enumerate(zip(got, wants)):
            walks += 1
            same = g is not None and w is not None and all(np.array_equal(g[f], w[f]) for f in ("hi", "lo", "dist", "cov", "last")) and g["levels"] == w["levels"]
            if not same:
                bad += 1
                n = min(len(g["lo"]), len(w["lo"]))
                first = np.nonzero((np.asarray(g["lo"][:n]) != np.asarray(w["lo"][:n])) | (np.asarray(g["last"][:n]) != np.asarray(w["last"][:n])))[0][:4]
                print("walk %d job %d dir %d (seed at %d, cap %d = component %+d): %d entries, oracle %d; first differences at %s" % (
                    walks, j, jobs[j][2], s0, cap, delta, len(g["lo"]), len(w["lo"]), first), flush=True)
                for i in range(n, len(g["lo"])):
                    dup = np.nonzero(np.asarray(g["lo"][:n]) == g["lo"][i])[0]
                    print("   device only: entry %d lo=%x dist=%d cov=%d last=%d; the same k-mer earlier at %s" % (i, int(g["lo"][i]), g["dist"][i], g["cov"][i], g["last"][i], dup), flush=True)
    want_cache.clear()
    if time.time() - last > 30:
        last = time.time()
        print("... %d walks, %d bad, %.0f s%s" % (walks, bad, last - t0, ", %d counting runs beside them" % busy[0] if a.contend else ""), flush=True)
stop.set()
if th:
    th.join()
print("%d walks, %d bad (%.0f s)%s" % (walks, bad, time.time() - t0, ", %d counting runs beside them" % busy[0] if a.contend else ""))
sys.exit(1 if bad else 0)
