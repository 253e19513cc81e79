"""A model of the BFS scout's hops (metacherchant_amd/csrc/bfs_device.h scout_eval / scout_companion) on a linear contig, to
tell what bounds the levels a hop adds: reads of L bases at `cov`-fold depth with substitution errors at rate `err`, every
k-mer's slot pointing at one random error-free occurrence of it, a hop following `ncand` candidate reads over `lanes` levels
and ending at the first level that none of them holds error-free (`mode`: other rules).  No GPU, no library: numpy only.

  python scripts/hop_model.py            configs[1]'s kind (150 bp, 30-fold, 1 %): the shipped rule and the alternatives

Round 4 result (levels per team hop; the GPU measures 47.9 on configs[1] E1 with MC_BFS_STATS=1):
  2 candidates nearest to the tip (shipped)         48.6      3: 54.8    4: 57.6    8: 62.3   (per-hop cost grows with the waves)
  union of the two reads' solid levels              48.7      -- nothing: where one read has an error the other one decides anyway
  candidates whose reads END furthest (no errors known)  1: 42.5   2: 55.0   2 over 128 levels: 76.0
  candidates that get furthest (errors known: an oracle) 1: 63.5 of 64;  over 128 levels: 91.2
so a hop of 64 levels is cut by the ERRORS of the reads it follows (one read: 99 (1 - 0.99^64) = 47 levels), and only more
candidates or knowing the errors beforehand lengthen it."""
import bisect
import sys

import numpy as np

rng = np.random.default_rng(5)
G, L, k, cov, err = 400000, 150, 31, 30, 0.01
W = L - k + 1
n_reads = G * cov // L
starts = np.sort(rng.integers(0, G - L, n_reads))
bad = np.zeros((n_reads, W), dtype=bool)  # window w of read r holds an error
for r in range(n_reads):
    for p in np.nonzero(rng.random(L) < err)[0]:
        bad[r, max(0, p - k + 1):min(W - 1, p) + 1] = True

_ptr = {}


def pointer(w):
    """the read the slot of the k-mer at genome window w points into: one of its error-free occurrences"""
    if w not in _ptr:
        i0, i1 = bisect.bisect_left(starts, w - W + 1), bisect.bisect_right(starts, w)
        c = [r for r in range(i0, i1) if not bad[r, w - starts[r]]]
        _ptr[w] = c[rng.integers(0, len(c))] if c else -1
    return _ptr[w]


def solid_levels(r, tip, lanes):
    """which of the levels 1..lanes past the tip read r holds error-free (nothing when it does not hold the tip itself)"""
    m = np.zeros(lanes, dtype=bool)
    if r < 0:
        return m
    o = tip - starts[r]
    if o < 0 or o >= W or bad[r, o]:
        return m
    n = min(lanes, W - 1 - o)
    m[:n] = ~bad[r, o + 1:o + 1 + n]
    return m


def prefix(m):
    z = np.nonzero(~m)[0]
    return len(m) if len(z) == 0 else int(z[0])


def walk(ncand=2, lanes=64, mode="nearest", levels=60000, within=48):
    """mode: nearest -- the pointers nearest to the tip that lead into other reads (the shipped rule); union -- the same reads,
    a level counts when either holds it; ends -- the reads that END furthest past the new tip; oracle -- the reads that get furthest"""
    tip, cands, hops, total = 1000, [pointer(1000)], 0, 0
    while tip < 1000 + levels:
        masks = [(solid_levels(r, tip, lanes), r) for r in cands[:ncand]]
        hops += 1
        if mode == "union":
            u = masks[0][0].copy()
            for mm, _ in masks[1:]:
                u |= mm
            m = prefix(u)
            ender_mask, ender = next(((mm, r) for mm, r in masks if m and mm[m - 1]), masks[0])
        else:
            ender_mask, ender = max(masks, key=lambda x: prefix(x[0]))
            m = prefix(ender_mask)
        if m == 0:  # the walk asks again from its own vertex
            tip += 1
            cands = [pointer(tip)]
            continue
        found, seen = [], {ender}
        for lv in range(m, max(0, m - within), -1):
            if mode == "union" and not ender_mask[lv - 1]:
                continue
            p = pointer(tip + lv)
            if p >= 0 and p not in seen:
                seen.add(p)
                found.append(p)
        tip += m
        total += m
        if mode == "ends":
            found.sort(key=lambda r: -(starts[r] + W - 1 - tip))
        elif mode == "oracle":
            found.sort(key=lambda r: -prefix(solid_levels(r, tip, lanes)))
        cands = found[:ncand] if found else [pointer(tip)]
    return total / hops


if __name__ == "__main__":
    rows = [("2 candidates nearest to the tip (shipped)", dict()), ("3", dict(ncand=3)), ("4", dict(ncand=4)), ("8", dict(ncand=8)),
            ("union of 2", dict(mode="union")), ("ends furthest, 1", dict(ncand=1, mode="ends")), ("ends furthest, 2", dict(mode="ends")),
            ("ends furthest, 2, 128 levels", dict(mode="ends", lanes=128)), ("oracle, 1", dict(ncand=1, mode="oracle")),
            ("oracle, 1, 128 levels", dict(ncand=1, mode="oracle", lanes=128))]
    for name, kw in rows:
        print("%-45s %.1f levels a hop" % (name, walk(**kw)), flush=True)
