"""A second, independent pin for `environment-finder-multi`.  CPU only.

The reference ships no output of this tool, so the C++ host (`environment_finder_multi`, csrc/host/envfinder.cpp) and
oracle/host_oracle.py were only ever compared with each other -- both written by this build, from the same reading of the
Java.  Here a THIRD restatement, written straight from src/algo/MultiSequenceCalculator.java:51-159,
src/algo/MultiNode.java, src/io/writers/GFAWriterMulti.java:39-133, src/io/graph/DeBruijnGraphUtils.java:13-27 and
src/utils/StringUtils.java:35-42 with LETTERS, Python lists and dicts (no packed words, no shared helper of the oracle,
and its own, much simpler account of java.util.HashMap's iteration order: with list bins only, the keys come out by
bucket, and inside a bucket in the order they were first put -- a resize splits a bin into two that keep that order, a
re-put keeps the place; the model REFUSES an input on which a bin would have been treeified), is run on joins built
from the reference's own fixture graph.txt (Hi-C_pipline/example_work_dir/output/1/merged/graph.txt, k = 31: the file
with itself, overlapping subsets, disjoint subsets, three and four environments) and on the even-k palindrome hazard of
MultiSequenceCalculator.java:59-80; `seqs.fasta` and `graph.gfa` must equal the oracle's and the C++ host's byte for byte.
"""
import lzma
import os
import subprocess

import pytest

from oracle import host_oracle as ho

SEED = "ATTTCCAGCCCCTTCTGTGCGGCTTTCAGCGAGTTTTCCCACTGCTGTACAAATGTGGGCC"  # Hi-C_pipline/example/seq.fasta
NUCLEOTIDES = "AGCT"          # itmo!/dna/DnaTools.java:31
COMP = {"A": "T", "G": "C", "C": "G", "T": "A"}


def reverse_complement(s):    # itmo!/dna/DnaTools.java reverseComplement(String)
    return "".join(COMP[c] for c in reversed(s))


def normalize_dna(s):         # src/utils/StringUtils.java:35-42
    r = reverse_complement(s)
    return s if s < r else r


class Treeified(Exception):
    pass


def java_string_hash(s):      # String.hashCode, then HashMap.hash's spread
    h = 0
    for ch in s:
        h = (31 * h + ord(ch)) & 0xFFFFFFFF
    return h ^ (h >> 16)


def hashmap_order(puts):
    """The keys of a java.util.HashMap<String, ?> in iteration order after `puts` (keys, in the order they were put; putting
    a key again changes nothing).  JDK 8+: 16 buckets, doubled whenever ++size > 0.75 * buckets; a resize splits every bin
    into a low and a high one that keep their order, new keys go to the tail of their bin.  So, as long as every bin is a
    list: bucket by bucket, first-put order inside a bucket.  A bin that would have been treeified (a ninth key arriving in
    a bin of eight when there are >= 64 buckets) is beyond this model: Treeified."""
    first = {}
    cap, size = 16, 0
    fill = {}
    for key in puts:
        if key in first:
            continue
        h = java_string_hash(key)
        first[key] = (h, len(first))
        b = h & (cap - 1)
        n = fill.get(b, 0)
        if n >= 8 and cap >= 64:
            raise Treeified(key)
        # (below 64 buckets a ninth key makes the map resize instead of treeifying: the split below covers the order)
        fill[b] = n + 1
        size += 1
        if size > (cap * 3) // 4 or (n >= 8 and cap < 64):
            cap *= 2
            fill = {}
            for hh, _ in first.values():
                bb = hh & (cap - 1)
                fill[bb] = fill.get(bb, 0) + 1
    return sorted(first, key=lambda key: (first[key][0] & (cap - 1), first[key][1]))


def load_graph(text):         # src/io/graph/DeBruijnGraphUtils.java:13-27: "<kmer> <depth>" lines into a HashMap
    depth = {}
    puts = []
    for line in text.splitlines():
        tokens = line.split(" ")
        depth[tokens[0]] = int(tokens[1])
        puts.append(tokens[0])
    return depth, hashmap_order(puts)   # (the map, and its keySet() in iteration order)


class Node:                   # src/algo/MultiNode.java
    def __init__(self, sequence, ident, is_gene):
        self.sequence, self.id, self.is_gene = sequence, ident, is_gene
        self.deleted = False
        self.rc = None
        self.neighbors = []
        self.graphs = set()


def model(env_texts, sequence):
    """(seqs.fasta, graph.gfa) as src/algo/MultiSequenceCalculator.java and src/io/writers/GFAWriterMulti.java write them"""
    graphs = [load_graph(t) for t in env_texts]
    k = len(graphs[0][1][0])
    # initializeStructures, :51-100
    puts = []
    for _, keys in graphs:
        for kmer in keys:
            puts.append(kmer)
            puts.append(reverse_complement(kmer))
    order = hashmap_order(puts)
    size = len(order)
    node_by_kmer = {}
    nodes = []
    for kmer in order:
        r = reverse_complement(kmer)
        if kmer > r:                       # kmer.compareTo(rc) > 0
            continue
        if len(nodes) + 2 > size:          # nodes = new MultiNode[size]: a palindrome is one key but two nodes
            raise IndexError("ArrayIndexOutOfBoundsException")
        gene = kmer in sequence or r in sequence
        a, b = Node(kmer, len(nodes), gene), Node(r, len(nodes) + 1, gene)
        a.rc, b.rc = b, a
        nodes += [a, b]
        node_by_kmer[a.sequence] = a
        node_by_kmer[b.sequence] = b       # (a palindrome: the second put replaces the first)
    for i, (_, keys) in enumerate(graphs):
        for kmer in keys:
            n = node_by_kmer[kmer]
            n.graphs.add(i)
            n.rc.graphs.add(i)
    for n in nodes:
        for j in range(4):
            nb = node_by_kmer.get(n.sequence[1:] + NUCLEOTIDES[j])
            if nb is not None:
                n.rc.neighbors.append(nb)
    # doMerge, :102-139
    while True:
        acted = False
        for n in nodes:
            if not n.deleted and len(n.neighbors) == 1:
                other = n.neighbors[0]
                if len(n.neighbors) == 1 and len(other.neighbors) == 1 and n.is_gene == other.is_gene and n.graphs == other.graphs:
                    first_plus, second_minus = n, other
                    first_minus, second_plus = first_plus.rc, second_minus.rc
                    assert second_plus.sequence[-(k - 1):] == first_plus.sequence[:k - 1]
                    assert first_minus.sequence[-(k - 1):] == second_minus.sequence[:k - 1]
                    new_seq = second_plus.sequence + first_plus.sequence[k - 1:]
                    new_rc = first_minus.sequence + second_minus.sequence[k - 1:]
                    second_plus.sequence, first_minus.sequence = new_seq, new_rc
                    second_plus.rc, first_minus.rc = first_minus, second_plus
                    first_plus.deleted = second_minus.deleted = True
                    acted = True
        if not acted:
            break
    # outputNodeSequences, :141-178
    seqs = []
    for n in nodes:
        if not n.deleted and n.id < n.rc.id:
            ids = set()
            for x in n.neighbors:
                ids.add(min(x.id, x.rc.id) + 1)
            for x in n.rc.neighbors:
                ids.add(min(x.id, x.rc.id) + 1)
            ids.discard(min(n.id, n.rc.id) + 1)
            seqs.append("> Id%d%s Length:%d Neighbors:[%s]\n%s\n" % (min(n.id, n.rc.id) + 1, "_start" if n.is_gene else "", len(n.sequence),
                                                                     ", ".join(str(x) for x in sorted(ids)), n.sequence))
    # GFAWriterMulti.printGraph, :39-133
    n_graphs = len(graphs)

    def colour(n):
        if n.is_gene:
            return "#00ff00"
        s = len(n.graphs)
        if n_graphs == 2:
            return {1: "#ff0000", 2: "#0000ff"}.get(s, "#000000")
        if n_graphs == 3:
            return {1: "#ff0000", 2: "#0000ff", 3: "#ff00ff", 4: "#ffff00", 5: "#ffaa00", 6: "#00ffff"}.get(s, "#000000")
        value = 256 * s // n_graphs
        return "#%02X%02X%02X" % (value, value, value)

    gfa = []
    for n in nodes:
        if not n.deleted and n.id < n.rc.id:
            coverage = 0
            for depth, _ in graphs:
                for i in range(len(n.sequence) - k + 1):
                    coverage += depth.get(normalize_dna(n.sequence[i:i + k]), 0)
            ident = ("" if n.id < n.rc.id else "-") + str(min(n.rc.id, n.id) + 1) + ("_start" if n.is_gene else "")
            gfa.append("S\t%s\t%s\tLN:i:%d\tKC:i:%d\tCL:Z:%s\tC2:Z:%s\n" % (ident, n.sequence, len(n.sequence), coverage, colour(n), colour(n)))
    for a in nodes:
        if not a.deleted:
            for b in a.neighbors:
                gfa.append("L\t%d%s\t%s\t%d%s\t%s\t%dM\n" % (min(a.rc.id, a.id) + 1, "_start" if a.is_gene else "", "+" if a.id < a.rc.id else "-",
                                                                 min(b.rc.id, b.id) + 1, "_start" if b.is_gene else "", "+" if b.id > b.rc.id else "-", k - 1))
    return "".join(seqs), "".join(gfa)


@pytest.fixture(scope="module")
def hosttest():
    if os.environ.get("MC_HOSTTEST"):  # (tests/test_host_sanitizers.py: the same tests on a sanitizer build)
        return os.environ["MC_HOSTTEST"]
    from metacherchant_amd import build
    build.build_host()
    return build.HOSTTEST


@pytest.fixture(scope="module")
def fixture_lines(golden_dir):
    return lzma.open(os.path.join(golden_dir, "ref_example", "graph.txt.xz"), "rt").read().splitlines()


def _three_ways(hosttest, tmp_path, env_texts, sequence):
    envs = []
    for i, t in enumerate(env_texts):
        p = tmp_path / ("env%d.txt" % i)
        p.write_text(t)
        envs.append(str(p))
    seq = tmp_path / "gene.fa"
    seq.write_text(">gene\n%s\n" % sequence)
    want_seqs, want_gfa = model(env_texts, sequence)
    oracle, _ = ho.environment_finder_multi(envs, str(seq), str(tmp_path / "unused"))
    assert oracle["seqs.fasta"] == want_seqs
    assert oracle["graph.gfa"] == want_gfa
    out = tmp_path / "out"
    subprocess.check_output([hosttest, "multi", str(out), str(seq), "1"] + envs)
    assert (out / "seqs.fasta").read_text() == want_seqs
    assert (out / "graph.gfa").read_text() == want_gfa
    return want_seqs, want_gfa


def _stretch(fixture, a, b):
    """The fixture's lines (its own depths, its own file order) for the k-mers of the plasmid's bases [a, b): environments
    that are connected pieces of the reference's graph.  The seed gene is RC(plasmid[106:167])."""
    lines, plasmid = fixture
    want = set()
    for i in range(a, b - 31 + 1):
        want.add(normalize_dna(plasmid[i:i + 31]))
    return "".join(l + "\n" for l in lines if l.split(" ")[0] in want)


@pytest.mark.parametrize("case", ["itself", "overlap", "disjoint", "three", "four"])
def test_joins_of_the_reference_fixture(hosttest, tmp_path, fixture_lines, golden_dir, case):
    plasmid = "".join(l.strip() for l in open(os.path.join(golden_dir, "ref_example", "salmonella_pls.fasta")) if not l.startswith(">")).upper()
    assert reverse_complement(plasmid[106:167]) == SEED
    fx = (fixture_lines, plasmid)
    if case == "itself":       # one environment twice: every node in both graphs, coverages doubled
        envs = [_stretch(fx, 0, 6000)] * 2
    elif case == "overlap":
        envs = [_stretch(fx, 0, 5000), _stretch(fx, 3000, 9000)]
    elif case == "disjoint":   # (the second one holds nothing of the gene)
        envs = [_stretch(fx, 0, 4000), _stretch(fx, 20000, 25000)]
    elif case == "three":
        envs = [_stretch(fx, 0, 4000), _stretch(fx, 2000, 6000), _stretch(fx, 5000, 9000)]
    else:                      # four and more: the grey scale
        envs = [_stretch(fx, 1000 * i, 4000 + 1500 * i) for i in range(4)]
    seqs, gfa = _three_ways(hosttest, tmp_path, envs, SEED)
    assert "_start" in seqs and "#00ff00" in gfa          # the gene's k-mers are in
    n_s, n_l = gfa.count("\nS\t") + 1, gfa.count("\nL\t")
    assert n_s >= 3 and n_l >= 4                          # (compacted: the gene's unitig and what lies on either side, per colour)
    assert max(len(l.split("\t")[2]) for l in gfa.splitlines() if l.startswith("S\t")) > 1000   # long unitigs: doMerge ran
    if case == "itself":
        assert "#ff0000" not in gfa                       # nothing belongs to one graph only
    if case in ("overlap", "disjoint"):
        assert "#ff0000" in gfa and ("#0000ff" in gfa) == (case == "overlap")


def test_a_consecutive_stretch_compacts_to_one_unitig(hosttest, tmp_path):
    """(a case small enough to check by eye: 40 bases, k = 5, two environments that share the middle)"""
    s = "ATGGCGTACGTTAGCCATGAACTTGGACCTAGGATTCAGA"
    k = 5

    def env(a, b, depth):
        seen, out = set(), []
        for i in range(a, b - k + 1):
            c = normalize_dna(s[i:i + k])
            if c not in seen:
                seen.add(c)
                out.append("%s %d\n" % (c, depth))
        return "".join(out)

    seqs, gfa = _three_ways(hosttest, tmp_path, [env(0, 28, 3), env(12, 40, 7)], s[18:26])
    assert gfa.count("\nS\t") + gfa.startswith("S\t") >= 3   # only-first, shared (+ the gene's), only-second


def test_even_k_palindrome_is_where_the_reference_dies(hosttest, tmp_path):
    """MultiSequenceCalculator.java:59-80: a palindromic k-mer (even k) is ONE key of nodeByKmer but takes TWO places of
    nodes[size] -- the last pair written falls off the array.  The model raises where the JVM would; both restatements
    refuse the input (tests/test_host_cpp.py checks their messages)."""
    env = "ACGT 3\nAAAA 2\n"
    with pytest.raises(IndexError):
        model([env], "AAAACGT")
    p = tmp_path / "pal.txt"
    p.write_text(env)
    s = tmp_path / "s.fa"
    s.write_text(">g\nAAAACGT\n")
    with pytest.raises(ValueError, match="palindromic"):
        ho.environment_finder_multi([str(p)], str(s), str(tmp_path / "o"))
    assert subprocess.call([hosttest, "multi", str(tmp_path / "o"), str(s), "1", str(p)], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL) == 1


def test_the_order_model_itself():
    """hashmap_order against the one fact of the reference that pins it: the fixture's graph.txt is a HashMap's keySet()
    written out, so re-putting its keys in file order must give them back in file order (the bucket sequence is
    non-decreasing and equal buckets keep their order)."""
    keys = ["AAAAC", "CAGTT", "GGGTA", "TTTAA", "ACGTA", "CCCCC", "GATTA", "TGCAT", "AGAGA", "CTCTC", "AAAAC"]
    order = hashmap_order(keys)
    assert sorted(order) == sorted(set(keys)) and len(order) == 10
    assert hashmap_order(order) == order
    buckets = [java_string_hash(x) & 15 for x in order]
    assert buckets == sorted(buckets)
