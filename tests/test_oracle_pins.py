"""Pins that constrain the oracle's ARITHMETIC, not only its formats ("What's weak" 1).  CPU only.

1. Adjacency of the shipped example: the L lines of the fixture's graph.gfa and the Neighbors sets of its seqs.fasta,
   re-expressed through unitig SEQUENCES (node ids come from an older revision and differ), must equal what the oracle
   writes for the same graph.txt -- this pins neighbour generation, the (k-1)-overlap rule, the orientation signs of
   GFAWriter (including its `>=`) and the canonical orientation of unitigs jointly, on 15 of the 16 unitigs.
2. A string-level model written straight from the Java sources with LETTERS (src/utils/StringUtils.java:8-41,
   src/algo/OneSequenceCalculator.java:154-214, src/algo/TerminationMode.java:31-47: no 2-bit codes, no packed words,
   a Python dict as the map) run on the fixture's own reads (the 1047 Hi-C sequences of
   tests/EnvironmentFinderMainTest.java:38-44): counts per k-mer, BFS discovery order, distances, coverages and
   lastKmers must equal the C oracle's.  A wrong base code, complement, neighbour letter order or saturation rule in
   oracle/mc_oracle.c would show here.

What stays citation-only (no vector of the reference can reach it): the numeric value of a
table key (the layout of the map never reaches an output), hence also which k-mers collide under the
64-bit hashes of k > 31, and the intra-bucket order of the fixture's graph.txt (written by an older revision).
"""
import lzma
import os

import numpy as np
import pytest

from oracle import host_oracle as ho
from oracle import pyoracle as po

SEED = "ATTTCCAGCCCCTTCTGTGCGGCTTTCAGCGAGTTTTCCCACTGCTGTACAAATGTGGGCC"
COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}
NUCLEOTIDES = "AGCT"  # itmo!/dna/DnaTools.java:31 -- the order neighbours are generated in


def rc(s):
    return "".join(COMP[c] for c in reversed(s))


def normalize(s):  # StringUtils.normalizeDna: s.compareTo(rc) < 0 ? s : rc
    r = rc(s)
    return s if s < r else r


# ---------------------------------------------------------------- 1. adjacency of the fixture

def _gfa_links_by_sequence(text):
    seq_of, links = {}, set()
    for l in text.splitlines():
        f = l.split("\t")
        if f[0] == "S":
            seq_of[f[1].replace("_start", "")] = f[2]
    for l in text.splitlines():
        f = l.split("\t")
        if f[0] == "L":
            links.add((seq_of[f[1].replace("_start", "")], f[2], seq_of[f[3].replace("_start", "")], f[4], f[5]))
    return seq_of, links


def _seqs_neighbours_by_sequence(text):
    lines = text.splitlines()
    seq_of, nb = {}, {}
    for h, s in zip(lines[0::2], lines[1::2]):
        seq_of[h.split()[1][2:].replace("_start", "")] = s
    for h, s in zip(lines[0::2], lines[1::2]):
        inside = h[h.index("Neighbors:[") + 11:-1]
        nb[s] = {x for x in inside.split(", ") if x}
    return seq_of, nb


@pytest.fixture(scope="module")
def fixture_files(golden_dir, tmp_path_factory):
    g = os.path.join(golden_dir, "ref_example")
    fix = [l.split(" ") for l in lzma.open(os.path.join(g, "graph.txt.xz"), "rt").read().splitlines()]
    t = po.Table()
    for s, c in fix:
        t.add(po.key(po.encode(s), 31, po.KEY_PACKED), int(c))
    seqs, comments = ho.rich_fasta_read(os.path.join(g, "seq.fasta"))
    out = str(tmp_path_factory.mktemp("pins"))
    res = ho.environment_finder(t, 31, po.KEY_PACKED, seqs, comments, out, coverage=5, max_radius=100000,
                                bothdirs=False, chunk_length=10, merge=True)
    (_, files), = res.items()
    return g, files


def test_fixture_gfa_links_between_common_unitigs(fixture_files):
    g, files = fixture_files
    ref_seq, ref_links = _gfa_links_by_sequence(open(os.path.join(g, "graph.gfa")).read())
    our_seq, our_links = _gfa_links_by_sequence(files["graph.gfa"])
    common = set(ref_seq.values()) & set(our_seq.values())
    assert len(common) == 15
    ref_c = {l for l in ref_links if l[0] in common and l[2] in common}
    our_c = {l for l in our_links if l[0] in common and l[2] in common}
    assert len(ref_c) >= 20  # most of the fixture's 41 L lines join unitigs both revisions have
    assert ref_c == our_c    # same pairs, same orientation signs, same overlap field
    # the links that touch the 16th unitig (merged with its flanks by the older revision) are the only other ones
    lost = set(ref_seq.values()) - common
    assert all(l[0] in lost or l[2] in lost for l in ref_links - ref_c)


def test_fixture_seqs_fasta_neighbour_sets(fixture_files):
    """Neighbors:[...] of seqs.fasta, through sequences.  The fixture's file is from the older revision whose ids (and the
    stale rc pointers getNeighborIds follows, OneSequenceCalculator.java:375-385) differ, so only this much can be asked:
    every neighbour relation it prints between unitigs that both revisions print is one the oracle prints too."""
    g, files = fixture_files
    ref_ids, ref_nb = _seqs_neighbours_by_sequence(open(os.path.join(g, "seqs.fasta")).read())
    our_ids, our_nb = _seqs_neighbours_by_sequence(files["seqs.fasta"])
    both = set(ref_nb) & set(our_nb)
    assert len(both) >= 14
    checked = 0
    for s in both:
        ref_set = {ref_ids[i] for i in ref_nb[s] if i in ref_ids} & both
        our_set = {our_ids[i] for i in our_nb[s] if i in our_ids} & both
        assert ref_set <= our_set, s[:40]
        checked += len(ref_set)
    assert checked >= 10


# ---------------------------------------------------------------- 2. the string-level model

def model_count(reads, k):
    table = {}
    for r in reads:
        for i in range(len(r) - k + 1):
            w = normalize(r[i:i + k])
            table[w] = min(32767, table.get(w, 0) + 1)  # NumUtils.addAndBound(short, short)
    return table


def model_bfs(table, seeds, k, direction, cov, max_kmers=None, max_radius=None):
    """OneSequenceCalculator.runBfs on strings.  Returns (queue order of distanceToKmer, dist, cov, lastKmers)."""
    def get(s):
        return table.get(normalize(s), -1)  # BigLong2ShortHashMap.get: -1 when absent

    def neighbours(s):
        left = [c + s[:-1] for c in NUCLEOTIDES]   # StringUtils.leftNeighbors
        right = [s[1:] + c for c in NUCLEOTIDES]   # StringUtils.rightNeighbors
        if direction < 0:
            return left
        if direction > 0:
            return right
        out = []
        for a, b in zip(left, right):              # StringUtils.allNeighbors: L0, R0, L1, R1, ...
            out += [a, b]
        return out

    queue, dist, last = [], {}, set()
    for s in seeds:
        for i in range(len(s) - k + 1):
            w = s[i:i + k]
            if get(w) >= cov:
                queue.append(w)
                dist.setdefault(w, 0)
    if not queue:
        return None
    head = 0
    while head < len(queue):
        v = queue[head]
        head += 1
        for n in neighbours(v):
            if get(n) >= cov:
                ok = n not in dist
                if ok and max_kmers is not None and len(dist) >= max_kmers:
                    ok = False
                if ok and max_radius is not None and dist[v] + 1 > max_radius:
                    ok = False
                if ok:
                    dist[n] = dist[v] + 1
                    queue.append(n)
                else:
                    last.add(v)
    order = list(dist)  # insertion order
    return order, [dist[s] for s in order], [get(s) for s in order], [1 if s in last else 0 for s in order]


@pytest.fixture(scope="module")
def hic_reads(golden_dir):
    reads, _ = ho.rich_fasta_read(os.path.join(golden_dir, "ref_example", "selected_reads.fasta"))
    assert len(reads) == 1047
    return reads


@pytest.mark.parametrize("k", [31, 21])
def test_counts_of_the_fixture_reads_equal_the_string_model(hic_reads, k):
    model = model_count(hic_reads, k)
    codes = np.concatenate([po.encode(r) for r in hic_reads])
    off = np.zeros(len(hic_reads) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in hic_reads])
    t = po.Table()
    n = t.count_reads(codes, off, k, po.KEY_PACKED)
    assert n == sum(max(0, len(r) - k + 1) for r in hic_reads)
    assert t.size() == len(model)
    keys, counts = t.dump()
    by_key = dict(zip(keys.tolist(), counts.tolist()))
    for w, c in model.items():
        assert by_key[po.key(po.encode(w), k, po.KEY_PACKED)] == c
    # the table key is the smaller of the two strands as NUMBERS with A0 G1 C2 T3 (ShortKmer.toLong): pinned here only
    # through the two strands sharing one counter, whichever of them the reads held
    w = next(iter(model))
    assert po.key(po.encode(w), k, po.KEY_PACKED) == po.key(po.encode(rc(w)), k, po.KEY_PACKED)


@pytest.mark.parametrize("direction,cov,max_kmers,max_radius", [(-1, 2, None, 40), (1, 2, 300, None), (0, 2, 500, 30), (0, 1, None, 12),
                                                                (1, 3, 100000, None)])
def test_bfs_on_the_fixture_reads_equals_the_string_model(hic_reads, direction, cov, max_kmers, max_radius):
    k = 31
    model = model_count(hic_reads, k)
    codes = np.concatenate([po.encode(r) for r in hic_reads])
    off = np.zeros(len(hic_reads) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in hic_reads])
    t = po.Table()
    t.count_reads(codes, off, k, po.KEY_PACKED)
    # seeds: the most covered read region -- take the read holding the most frequent k-mer, and the shipped seed too
    best = max(model, key=model.get)
    holder = next(r for r in hic_reads if best in r or rc(best) in r)
    for seeds in ([holder], [SEED, holder[:60]]):
        want = model_bfs(model, seeds, k, direction, cov, max_kmers, max_radius)
        got = po.bfs(t, k, po.KEY_PACKED, [po.encode(s) for s in seeds], direction, cov,
                     -1 if max_kmers is None else max_kmers, -1 if max_radius is None else max_radius)
        assert (want is None) == (got is None)
        if want is None:
            continue
        order, dist, covs, last = want
        assert [po.kmer_string(h, l, k) for h, l in zip(got["hi"], got["lo"])] == order
        assert got["dist"].tolist() == dist and got["cov"].tolist() == covs and got["last"].tolist() == last
        assert len(order) > 31  # the walk left the seed


def test_hash_keys_are_strand_symmetric_on_the_fixture_reads(hic_reads):
    """src/utils/PolynomialHash.java:19-28 / FNV1AHash.java:33-42 (k > 31): hash(s) == hash(rc(s)) and distinct k-mers of
    this read set do not collide, so the hash-key table has the model's size and counts."""
    k = 45
    model = model_count(hic_reads, k)
    codes = np.concatenate([po.encode(r) for r in hic_reads])
    off = np.zeros(len(hic_reads) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in hic_reads])
    for mode in (po.KEY_POLY, po.KEY_FNV1A):
        t = po.Table()
        t.count_reads(codes, off, k, mode)
        assert t.size() == len(model)
        some = list(model)[:200]
        for w in some:
            assert po.key(po.encode(w), k, mode) == po.key(po.encode(rc(w)), k, mode)
            assert t.get(po.key(po.encode(w), k, mode)) == model[w]
