"""Different k-mers with one PolynomialHash key share a counter (src/io/LargeKIOUtils.java:46-49 adds every window's HASH to the
map, src/utils/PolynomialHash.java:19-28): the constructed vectors of tests/golden/poly_collisions.json (scripts/poly_collisions.py)
against the oracle.  The GPU side of the same vectors: tests/test_gpu_collisions.py."""
import importlib.util
import json
import os

import numpy as np

from oracle import pyoracle as po

HERE = os.path.dirname(os.path.abspath(__file__))
VECTORS = json.load(open(os.path.join(HERE, "golden", "poly_collisions.json")))["vectors"]
# the pair VERDICT r5 constructed (balanced base-5 digits of 2^64 over the last 28 positions of a 63-mer)
REVIEW_PAIR = ("GACATTTTGATATTATCGACAAAATGTAGTTGCGGCTCGTTTGCAGCCATTTAATTCCGAAGC",
               "GACATTTTGATATTATCGACAAAATGTAGTTGCGGAGACGGCTGCAACGTCTCGCGGATCATG", -7762169406344464911)


def _script():
    spec = importlib.util.spec_from_file_location("poly_collisions", os.path.join(HERE, "..", "scripts", "poly_collisions.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_vectors_collide_in_the_oracle_and_in_the_scripts_own_hash():
    s = _script()
    assert {v["k"] for v in VECTORS} >= {33, 47, 63}
    for v in VECTORS + [dict(k=63, x=REVIEW_PAIR[0], y=REVIEW_PAIR[1], key=REVIEW_PAIR[2])]:
        k, x, y = v["k"], po.encode(v["x"]), po.encode(v["y"])
        assert len(x) == len(y) == k and v["x"] != v["y"]
        assert v["x"] != po.decode(3 - y[::-1])  # (not each other's reverse complement either)
        assert po.key(x, k, po.KEY_POLY) == po.key(y, k, po.KEY_POLY) == v["key"]
        assert s.key_poly(x.tolist()) == s.key_poly(y.tolist()) == v["key"]  # (a second, pure-Python account of the Java loop)
        # the other strand of either gives the same key, as any k-mer's does
        assert po.key(3 - y[::-1], k, po.KEY_POLY) == v["key"]


def test_bins_differ_under_both_rules():
    s = _script()
    for v in VECTORS:
        bx, by = s.bin_words(po.encode(v["x"]).tolist()), s.bin_words(po.encode(v["y"]).tolist())
        assert [bx[0], by[0]] == v["bin_word_smallest"] and [bx[1], by[1]] == v["bin_word_two_smallest"]
        assert abs(bx[0] - by[0]) >= 1 << 28 and abs(bx[1] - by[1]) >= 1 << 28


def test_oracle_counts_colliding_kmers_in_one_counter():
    for v in VECTORS:
        k = v["k"]
        x, y = po.encode(v["x"]), po.encode(v["y"])
        codes = np.concatenate([x, y, 3 - y[::-1], x])
        off = np.array([0, k, 2 * k, 3 * k, 4 * k], dtype=np.uint64)
        t = po.Table()
        assert t.count_reads(codes, off, k, po.KEY_POLY) == 4
        assert t.size() == 1 and t.get(v["key"]) == 4
