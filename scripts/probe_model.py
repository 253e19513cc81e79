#!/usr/bin/env python3
"""How many steps does the merge kernel's probe loop take?  (numpy only, no GPU.)

A leaf of configs[1]: ~1 435 distinct keys arrive as ~2 330 probes (every distinct key once, the rest repeats) in a region
image of 4 096 slots, linear probing from a random home slot.  A wave settles 128 keys a step of its pair loop (64 lanes x
2 keys) and the loop runs until the SLOWEST of them has found its slot: 4.6 steps, although a key needs 1.23 on average and
95 % of them are settled after two -- which is why k_p3_dedup runs two steps for both keys of a lane and leaves the rest to
a loop of one key a lane."""
import numpy as np

rng = np.random.default_rng(1)
S = 4096


def leaf(n_keys=1435, n_probes=2330):
    keys = np.concatenate([np.arange(n_keys), rng.integers(0, n_keys, n_probes - n_keys)])
    rng.shuffle(keys)
    home = rng.integers(0, S, n_keys)
    tab = -np.ones(S, dtype=np.int64)
    lens = []
    for k in keys:
        s, n = home[k], 1
        while tab[s] != -1 and tab[s] != k:
            s = (s + 1) % S
            n += 1
        tab[s] = k
        lens.append(n)
    return np.array(lens)


if __name__ == "__main__":
    per_step, hist = [], np.zeros(40)
    for _ in range(200):
        l = leaf()
        per_step += [l[i:i + 128].max() for i in range(0, len(l), 128)]
        for x in l:
            hist[min(x, 39)] += 1
    p = hist / hist.sum()
    print("steps until the slowest of 128 keys is settled: %.2f" % np.mean(per_step))
    print("steps a key needs on average: %.2f;  P(1..6 steps): %s" % ((p * np.arange(40)).sum(), np.round(p[1:7], 4)))
    for k in (2, 3):
        print("keys not settled after %d steps: %.1f %%" % (k, 100 * p[k + 1:].sum()))
