"""CPU model of how evenly the k-mers of a read set fill the regions of a minimizer-bin table at k = 63 (DESIGN.md section 3.1 "Long
records", section 7): a random genome, reads at 30-fold depth with 1 % substitutions on both strands, every window's canonical
polynomial-style key (any 64-bit function of the canonical k-mer does for counting distinct keys) and the bin it goes to under

  a  the rule the long-record pipeline uses: bin = mix(smallest sk_order over the window's canonical 15-mers)
  b  ... x the GROUP of where that 15-mer sits: min(offset, 48 - offset) >> 3, capped at 2 -- the same from either strand; where
     the smallest hash occurs twice in a window the offset nearest an end counts (also the same from either strand)
  c  ... x the second-smallest hash of the window (as a multiset: strand-symmetric by construction)

and prints, for a few table loads, how full the fullest regions of 4096 slots get.  python scripts/bin_model.py [genome bases]"""
import sys
import numpy as np

rng = np.random.default_rng(1)
G = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
k, m, cov, L, err = 63, 15, 30, 150, 0.01
w = k - m + 1
nreads = G * cov // L
genome = rng.integers(0, 4, G, dtype=np.uint8)
starts = rng.integers(0, G - L, nreads)
reads = genome[starts[:, None] + np.arange(L)[None, :]].copy()
e = rng.random(reads.shape) < err
reads[e] = (reads[e] + rng.integers(1, 4, e.sum())) % 4
strand = rng.random(nreads) < 0.5
reads[strand] = (3 - reads[strand])[:, ::-1]
M32 = np.uint64(0xFFFFFFFF)


def order(x):  # kmer_device.h sk_order
    x = (x * np.uint64(0x9E3779B1)) & M32
    return x ^ (x >> np.uint64(15))


def mix(x):  # kmer_device.h sk_bin
    x = x ^ (x >> np.uint64(16)); x = (x * np.uint64(0x7FEB352D)) & M32
    x = x ^ (x >> np.uint64(15)); x = (x * np.uint64(0x846CA68B)) & M32
    return x ^ (x >> np.uint64(16))


wts = rng.integers(1, 2 ** 62, k, dtype=np.uint64)
keys, bins = [], {"a": [], "b": [], "c": []}
for b0 in range(0, nreads, 20000):
    a = reads[b0:b0 + 20000].astype(np.uint64)
    n15 = L - m + 1
    f = np.zeros((a.shape[0], n15), dtype=np.uint64)
    r = np.zeros((a.shape[0], n15), dtype=np.uint64)
    for i in range(m):
        f = (f << np.uint64(2)) | a[:, i:i + n15]
        r = r | ((np.uint64(3) - a[:, i:i + n15]) << np.uint64(2 * i))
    h = order(np.minimum(f, r))
    hw = np.lib.stride_tricks.sliding_window_view(h, w, axis=1)           # (reads, windows, w)
    h1 = hw.min(axis=2)
    is_min = hw == h1[:, :, None]
    pos = np.arange(w)[None, None, :]
    c = np.where(is_min, np.minimum(pos, w - 1 - pos), w).min(axis=2)     # the smallest hash's distance from the nearer end
    grp = np.minimum(c >> 3, 2).astype(np.uint64)
    srt = np.sort(hw, axis=2)[:, :, :2]
    h2 = srt[:, :, 1]
    sw = np.lib.stride_tricks.sliding_window_view(a, k, axis=1)
    fw = (sw * wts[None, None, :]).sum(axis=2)
    rc = ((np.uint64(3) - sw[:, :, ::-1]) * wts[None, None, :]).sum(axis=2)
    keys.append(np.minimum(fw, rc).ravel())
    bins["a"].append(mix(h1).ravel())
    bins["b"].append(mix(h1 ^ (grp * np.uint64(0x5BD1E995))).ravel())
    bins["c"].append(mix(h1 ^ ((h2 * np.uint64(0x85EBCA6B)) & M32)).ravel())
keys = np.concatenate(keys)
u, idx = np.unique(keys, return_index=True)
print("%d windows, %d distinct keys" % (keys.size, u.size))
for name in "abc":
    ub = np.concatenate(bins[name])[idx]
    for load in (0.36, 0.45, 0.53):
        nreg = int(u.size / load / 4096)
        cnt = np.bincount(((ub * np.uint64(nreg)) >> np.uint64(32)).astype(np.int64), minlength=nreg)
        print("rule %s load %.2f: %5d regions, mean %4.0f keys, p99 %4.0f, p99.9 %4.0f, fullest %4d; above 80 %%: %.2f %% of the regions, above 90 %%: %.3f %%; keys beyond 90 %% of "
              "their region: %.4f %% of all" % (name, load, nreg, cnt.mean(), np.percentile(cnt, 99), np.percentile(cnt, 99.9), cnt.max(), 100 * (cnt > 3276).mean(),
                                               100 * (cnt > 3686).mean(), 100.0 * np.maximum(cnt - 3686, 0).sum() / u.size))
