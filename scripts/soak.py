"""Randomised parity soak on the GPU: random k, error rate, read count, genome size and coverage threshold; the
partitioned pipeline (forced) counts the reads in one or two batches, and the table and three BFS walks are compared
with the oracle each time.  Usage: python scripts/soak.py [iterations] [seed]"""
import os
import sys
import time

os.environ["MC_COUNT_PATH"] = "partition"
os.environ.setdefault("MC_BFS_SELFCHECK", "1")  # every walk also checked on the device (bfs_device.h k_bfs_check)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import metacherchant_amd as mc
from oracle import pyoracle as po
from tests.helpers import GENOME_SEED, assert_bfs_equal, seed_windows

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
only = int(os.environ["SOAK_ONLY"]) if "SOAK_ONLY" in os.environ else None  # replay one iteration of a run (the others only draw their numbers)
first = int(os.environ.get("SOAK_FROM", 0))                                     # ... or all from this one on
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
# SOAK_KS=33,41,47,55,63: other k-mer lengths (above 31: polynomial keys -- with a capacity hint, the long-record pipeline)
KS = [int(x) for x in os.environ.get("SOAK_KS", "").split(",") if x] or [21, 23, 24, 25, 27, 28, 29, 30, 31, 31, 31, 41]
SEEDS = int(os.environ.get("SOAK_SEEDS", 6))  # seed sequences per table, three walks each
# SOAK_COLLIDE=1: where k has constructed vectors (33, 41, 47, 55, 63), a pair of k-mers with one hash is written into the genome
COLLIDE = os.environ.get("SOAK_COLLIDE") == "1"
VECTORS = {}
if COLLIDE:
    import json
    for _v in json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "poly_collisions.json")))["vectors"]:
        VECTORS.setdefault(_v["k"], []).append(_v)
walks = 0
t0 = time.time()
for it in range(iters):
    k = int(rng.choice(KS))
    mode, omode = (mc.KEY_PACKED, po.KEY_PACKED) if k <= 31 else (mc.KEY_POLY, po.KEY_POLY)
    err = int(rng.choice([0, 50, 100, 200, 500]))
    L = int(rng.choice([45, 70, 100, 150, 250]))
    n_reads = int(rng.integers(30_000, 250_000))
    contigs = int(rng.integers(1, 4))
    clen = int(rng.choice([3_000, 20_000, 60_000, 150_000, 400_000]))
    cov = int(rng.integers(1, 7))
    hint = bool(rng.integers(0, 2))
    cap = int(rng.choice([0, 0, 2_000_000, 6_000_000]))
    rseed = int(rng.integers(1, 1 << 30))
    if (only is not None and it != only) or it < first:
        if bool(rng.integers(0, 4) == 0):
            rng.integers(0, L + 1, n_reads)
        rng.integers(0, 2)
        rng.integers(0, max(1, clen - 600))
        continue
    print("it %d: k=%d err=%d L=%d reads=%d genome=%dx%d cov=%d hint=%d cap=%d rseed=%d" % (it, k, err, L, n_reads, contigs, clen, cov, hint, cap, rseed), flush=True)
    genome = po.synth_genome(GENOME_SEED + it, contigs * clen)
    planted = None
    if COLLIDE and k in VECTORS and clen >= 3000:
        # two different k-mers with one PolynomialHash key (tests/golden/poly_collisions.json) written into the genome before the reads
        # are drawn: x in contig 0 where the first seed's walk passes, y (or its reverse complement) somewhere else -- the reference
        # counts both in one counter (csrc/dup_check.h), and the neighbours of x that nobody counted may have the hash of y's neighbours
        crng = np.random.default_rng(rseed ^ 0xC0111DE)  # (a generator of its own: the iterations' draws stay what they were)
        v = VECTORS[k][int(crng.integers(0, len(VECTORS[k])))]
        x, y = po.encode(v["x"]), po.encode(v["y"])
        if crng.integers(0, 2):
            y = (3 - y[::-1]).astype(np.uint8)
        px = int(crng.integers(300, clen - 600 - k))
        py = int(crng.integers(100, contigs * clen - k - 100))
        while abs(py - px) < 2 * k + 400 or (py % clen) + k > clen:
            py = int(crng.integers(100, contigs * clen - k - 100))
        genome = genome.copy()
        genome[px:px + k] = x
        genome[py:py + k] = y
        planted = (px, py, v["key"])
    reads = po.synth_reads(genome, contigs, clen, rseed, 0, n_reads, L, err)
    off = np.arange(n_reads + 1, dtype=np.uint64) * L
    ragged = bool(rng.integers(0, 4) == 0)
    if ragged:  # reads cut to random lengths 0..L (some shorter than k, some empty)
        lens = rng.integers(0, L + 1, n_reads)
        keep = (np.arange(L)[None, :] < lens[:, None]).ravel()
        reads = reads[keep]
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    t = po.Table()
    t.count_reads(reads, off, k, omode)
    ctx = mc.Context(k, mode, 0, cap)
    if hint:
        ctx.set_coverage_hint(cov)
    two = bool(rng.integers(0, 2))
    if two:
        h = n_reads // 2
        ctx.add_reads_packed(po.pack(reads[:off[h]]), off[:h + 1])
        ctx.add_reads_packed(po.pack(reads[off[h]:]), off[h:] - off[h])
    else:
        ctx.add_reads_packed(po.pack(reads), off)
    nd = ctx.finalize()
    assert nd == t.size(), (it, nd, t.size())
    gk, gc = ctx.export(0)
    ok, oc = t.dump()
    assert np.array_equal(gk, ok) and np.array_equal(gc, oc), it
    a = int(rng.integers(0, max(1, clen - 600)))
    if planted:
        a = max(0, planted[0] - 250)  # the first seed sequence ends in front of x
        assert ctx.get(np.array([planted[2]], dtype=np.int64))[0] == t.get(planted[2]), it
    srng = np.random.default_rng(rseed)  # (a generator of its own: the iterations' draws stay what they were)
    for si, d in [(si, d) for si in range(SEEDS) for d in (1, -1, 0)]:
        if d == 1:
            if si:
                a = int(srng.integers(0, max(1, contigs * clen - 600)))
            seed = genome[a:a + 400]
            hi, lo = seed_windows(seed, k)
        # --maxkmers: 20000 as ever for the first seed, then around the size of a contig (the cap bites a few vertices from the end), small, or out of reach
        maxk = 20000 if si == 0 else int(srng.choice([20000, clen - k + 1 + int(srng.integers(-40, 41)), int(srng.integers(400, 3000)), 1 << 20]))
        got = ctx.bfs(hi, lo, d, cov, maxk, -1)
        want = po.bfs(t, k, omode, [seed], d, cov, maxk, -1)
        walks += 1
        if got is not None and want is not None and len(got["hi"]) != len(want["hi"]):  # what the replayed iteration got too much / too little
            n = min(len(got["hi"]), len(want["hi"]))
            for name, r in (("device", got), ("oracle", want)):
                for j in range(n, len(r["hi"])):
                    key = int(r["lo"][j])
                    print("dir %d: %s only: entry %d lo=%x dist=%d cov=%d last=%d | oracle table count of that k-mer (as a packed key, both strands): %s | device table: %s" % (
                        d, name, j, key, r["dist"][j], r["cov"][j], r["last"][j], t.get(np.array([key], dtype=np.uint64)) if hasattr(t, "get") else "?",
                        ctx.get(np.array([key], dtype=np.uint64))), flush=True)
        assert_bfs_equal(got, want)
    st = ctx.stats()
    ctx.close()
    print("it %d ok%s: k=%d err=%d L=%d reads=%d genome=%dx%d cov=%d hint=%d cap=%d batches=%d distinct=%d list=%d sweeps=%d spills=%d long=%d grows=%d (%.0f s)" % (
        it, " (ragged)" if ragged else "", k, err, L, n_reads, contigs, clen, cov, hint, cap, 2 if two else 1, nd, st.solid_list_builds, st.solid_sweeps, st.spill_keys,
        st.long_runs, st.grows, time.time() - t0) + (" planted x at %d, y at %d: %d keys in several regions" % (planted[0], planted[1], st.dup_keys) if planted else ""), flush=True)
print("soak ok: %d iterations, %d walks" % (iters, walks))
