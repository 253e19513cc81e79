#!/usr/bin/env python3
"""Constructs pairs of DIFFERENT k-mers with the same PolynomialHash key (src/utils/PolynomialHash.java:19-28) and writes
them to tests/golden/poly_collisions.json.

The reference counts such k-mers in ONE counter (src/io/LargeKIOUtils.java:46-49: addAndBound(hash, 1) per window, whatever
the bases); a table that files hash keys under the minimizer bin of their BASES (csrc/count_long.h) meets the two in different
regions, so these vectors are what the parity tests plant (tests/test_poly_collisions.py, tests/test_gpu_collisions.py).

Construction: fw(s) = 5^k + sum b_i 5^(k-1-i) mod 2^64, so two k-mers collide on their forward hashes when the
difference of their digit strings, read as a number in base 5 with digits -3 .. 3, is a multiple of 2^64.  m * 2^64 is written
in BALANCED base 5 (digits -2 .. 2), spread over the last positions of the k-mer, and x, y = x + delta are drawn so that every
digit stays a base.  The free positions are searched until
  * both k-mers take the forward strand (key = Math.min(fw, rc) on signed longs), so the KEYS are equal, and
  * their minimizer bin words differ widely under both bin rules of the long-record table (smallest sk_order hash alone, and
    the two smallest: kmer_device.h sk_hmin_of_kmer2, count_long.h skl_bin), so that they land in different regions of any
    table of 16 regions or more.
No FNV-1a pair: FNV1AHash (src/utils/FNV1AHash.java:33-42) has no such lattice structure; a 64-bit birthday search costs
~5e9 hashes and as many stored values -- not done here (FNV-1a keys are counted per window, by key, on every path anyway).
"""
import json
import os
import random
import sys

M64 = (1 << 64) - 1
CODES = "AGCT"  # itmo!/dna/DnaTools.java:31
SK_M = 15


def fw_hash(b):
    h = 1
    for x in b:
        h = (h * 5 + x) & M64
    return h


def rc_codes(b):
    return [3 - x for x in reversed(b)]


def signed(h):
    return h - (1 << 64) if h >> 63 else h


def key_poly(b):
    return min(signed(fw_hash(b)), signed(fw_hash(rc_codes(b))))


def sk_order(x):
    x = (x * 0x9E3779B1) & 0xFFFFFFFF
    return x ^ (x >> 15)


def sk_bin(h):
    x = h
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def bin_words(b):
    """(bin word by the smallest hash, bin word by the two smallest) of a k-mer's canonical SK_M-mers, low byte cleared"""
    hs = []
    for i in range(len(b) - SK_M + 1):
        f = 0
        for x in b[i:i + SK_M]:
            f = (f << 2) | x
        r = 0
        for x in rc_codes(b[i:i + SK_M]):
            r = (r << 2) | x
        hs.append(sk_order(min(f, r)))
    hs.sort()
    one = hs[0]
    two = hs[0] ^ ((hs[1] * 0x85EBCA6B) & 0xFFFFFFFF)
    return sk_bin(one) & 0xFFFFFF00, sk_bin(two) & 0xFFFFFF00


def balanced5(n):
    """digits d_e in -2 .. 2 with sum d_e 5^e = n"""
    d = []
    while n:
        r = n % 5
        if r > 2:
            r -= 5
        d.append(r)
        n = (n - r) // 5
    return d


def make_pair(k, m, rng, tries=200000):
    delta = balanced5(m << 64)  # exponent e <-> position k - 1 - e
    if len(delta) > k:
        return None
    delta += [0] * (k - len(delta))
    for _ in range(tries):
        x = [0] * k
        for e in range(k):
            d = delta[e]
            lo, hi = max(0, -d), min(3, 3 - d)
            x[k - 1 - e] = rng.randint(lo, hi)
        y = [x[i] + delta[k - 1 - i] for i in range(k)]
        hx, hy = fw_hash(x), fw_hash(y)
        assert hx == hy
        if not (signed(hx) < signed(fw_hash(rc_codes(x))) and signed(hy) < signed(fw_hash(rc_codes(y)))):
            continue
        bx, by = bin_words(x), bin_words(y)
        if abs(bx[0] - by[0]) < (1 << 28) or abs(bx[1] - by[1]) < (1 << 28):
            continue
        if x == y or x == rc_codes(y):
            continue
        return x, y, signed(hx)
    return None


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden",
                                                              "poly_collisions.json")
    rng = random.Random(20261005)
    vectors = []
    for k, ms in ((63, (1, 2, 3, 7)), (47, (1, 5)), (33, (1, 11)), (41, (1,)), (55, (3,))):
        for m in ms:
            p = make_pair(k, m, rng)
            if p is None:
                print("no pair for k=%d m=%d" % (k, m), file=sys.stderr)
                continue
            x, y, key = p
            sx, sy = "".join(CODES[c] for c in x), "".join(CODES[c] for c in y)
            bx, by = bin_words(x), bin_words(y)
            vectors.append({"k": k, "multiple_of_2_64": m, "x": sx, "y": sy, "key": key,
                            "bin_word_smallest": [bx[0], by[0]], "bin_word_two_smallest": [bx[1], by[1]]})
            print("k=%d m=%d key=%d\n  %s\n  %s" % (k, m, key, sx, sy))
    with open(out, "w") as f:
        json.dump({"hash": "PolynomialHash (src/utils/PolynomialHash.java:19-28)", "made_by": "scripts/poly_collisions.py",
                   "vectors": vectors}, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
