"""Pins the CPU oracle (oracle/) against everything the reference ships for this path:
SURVEY.md section 8(c) items 1-6 and Appendix B.  CPU only."""
import lzma
import os

import numpy as np
import pytest

from oracle import host_oracle as ho
from oracle import pyoracle as po

SEED = "ATTTCCAGCCCCTTCTGTGCGGCTTTCAGCGAGTTTTCCCACTGCTGTACAAATGTGGGCC"


def _signed(h):
    return h - (1 << 32) if h >= 1 << 31 else h


def test_appendix_b_known_answers():
    c = po.encode(SEED)
    assert po.key(c[:31], 31, po.KEY_PACKED) == 0x0FE86ABEDD96FE19
    assert int(po.lib().mco_rc_packed(0x0FE86ABEDD96FE19, 31)) == 0x26D01A621055B503
    assert po.key(c[:31], 31, po.KEY_POLY) == -2974806478792698801
    assert po.key(c[:31], 31, po.KEY_FNV1A) == -6084461599936088689
    assert po.key(c, 61, po.KEY_POLY) == 342666523779128550
    assert po.key(po.encode(ho.reverse_complement(SEED)), 61, po.KEY_POLY) == 342666523779128550
    assert po.key(c, 61, po.KEY_FNV1A) == -3112663821774459406
    h = ho.java_string_hash(SEED[:31])
    assert _signed(h) == -1013967415
    assert ((h ^ (h >> 16)) & 131071) == 54873


def test_encoding_order_is_AGCT():
    assert [po.lib().mco_code(ord(c)) for c in "AGCTagctN"] == [0, 1, 2, 3, 0, 1, 2, 3, -1]
    assert ho.normalize_dna("TTTT") == "AAAA"
    assert ho.normalize_dna("ACGT") == "ACGT"  # palindrome: s.compareTo(rc) == 0 -> rc == s


@pytest.fixture(scope="module")
def fixture_env(golden_dir, tmp_path_factory):
    g = os.path.join(golden_dir, "ref_example")
    fix = [l.split(" ") for l in lzma.open(os.path.join(g, "graph.txt.xz"), "rt").read().splitlines()]
    t = po.Table()
    for s, c in fix:
        t.add(po.key(po.encode(s), 31, po.KEY_PACKED), int(c))
    seqs, comments = ho.rich_fasta_read(os.path.join(g, "seq.fasta"))
    out = str(tmp_path_factory.mktemp("fixture_env"))
    # Hi-C_pipline/HiCEnvironmentFinder.sh:57
    res = ho.environment_finder(t, 31, po.KEY_PACKED, seqs, comments, out, coverage=5, max_radius=100000,
                                bothdirs=False, chunk_length=10, merge=True)
    (prefix, files), = res.items()
    return g, fix, seqs, comments, prefix, files


def test_seed_reader_headerless_record(fixture_env):
    _, _, seqs, comments, prefix, _ = fixture_env
    assert seqs == [SEED] and comments == [""]
    assert prefix.endswith("/merged/")


def test_fixture_graph_txt_reproduced_from_itself(fixture_env):
    """graph.txt of the shipped example, used as the k-mer table, must come back: same k-mers,
    same coverages, same java.util.HashMap bucket sequence, all keys ASCII-canonical.
    (Intra-bucket order of the fixture is from an older revision: not compared.)"""
    _, fix, _, _, _, files = fixture_env
    ours = [l.split(" ") for l in files["graph.txt"].splitlines()]
    assert len(ours) == 93572
    assert sorted(map(tuple, ours)) == sorted(map(tuple, fix))

    def bucket(s):
        h = ho.java_string_hash(s)
        return (h ^ (h >> 16)) & 131071

    assert [bucket(s) for s, _ in ours] == [bucket(s) for s, _ in fix]
    assert all(s == ho.normalize_dna(s) for s, _ in ours)
    assert min(int(c) for _, c in ours) == 5
    assert files["env.txt"] == files["graph.txt"]


def _s_lines(text):
    d = {}
    for l in text.splitlines():
        f = l.split("\t")
        if f[0] == "S":
            d[ho.normalize_dna(f[2])] = tuple(f[3:5])
    return d


def test_fixture_gfa_formats_and_kc(fixture_env):
    g, _, _, _, _, files = fixture_env
    ours, ref = _s_lines(files["graph.gfa"]), _s_lines(open(os.path.join(g, "graph.gfa")).read())
    common = set(ours) & set(ref)
    assert len(ref) == 16 and len(common) == 15  # 15 of 16 unitigs, all LN/KC
    for s in common:
        assert ours[s] == ref[s]
    # the 16th: the older revision had no gene-node barrier; ours splits it in gene + two flanks
    (lost,) = set(ref) - set(ours)
    parts = sorted(set(ours) - set(ref), key=len)
    assert [len(p) for p in parts] == [61, 95, 3619] and len(lost) == 61 + 95 + 3619 - 2 * 30
    assert ref["GCCGCAACAGCCGCAACAGCCGCAACAGCCGCAAC"] == ("LN:i:35", "KC:i:3565")  # Appendix B
    gene = [l for l in files["graph.gfa"].splitlines() if l.endswith("CL:Z:GREEN")]
    assert len(gene) == 1 and "_start\t" in gene[0]
    for l in files["graph.gfa"].splitlines():
        if l.startswith("L"):
            assert l.endswith("\t30M") and len(l.split("\t")) == 6


def test_fixture_seqs_and_tsv_formats(fixture_env):
    g, _, _, _, _, files = fixture_env
    ref_seqs = open(os.path.join(g, "seqs.fasta")).read().splitlines()
    our_seqs = files["seqs.fasta"].splitlines()
    assert set(ref_seqs[1::2]) - set(our_seqs[1::2]) == {s for s in ref_seqs[1::2] if len(s) == 3715}
    import re
    for h in our_seqs[0::2]:
        assert re.fullmatch(r"> Id\d+(_start)? Length:\d+ Neighbors:\[(\d+(, \d+)*)?\]", h), h
    assert files["tsvs/nodes.tsv"].splitlines()[0] == "id\tlength\tseq"
    assert files["tsvs/edges.tsv"].splitlines()[0] == "source\ttarget"
    assert open(os.path.join(g, "tsvs", "edges.tsv")).read().splitlines()[1] == \
        files["tsvs/edges.tsv"].splitlines()[1]


def test_hic_selected_reads_count(golden_dir):
    """tests/EnvironmentFinderMainTest.java:38-44 asserts 1047 Hi-C sequences."""
    d, c = ho.rich_fasta_read(os.path.join(golden_dir, "ref_example", "selected_reads.fasta"))
    assert len(d) == 1047 and len(c) == 1047


def test_plasmid_known_answer(golden_dir):
    """error-free reads tiling the plasmid at depth >= 3 give back the plasmid's
    canonical 31-mers; the seed is RC(plasmid[106:167])."""
    recs = ho.read_fasta_reads(os.path.join(golden_dir, "ref_example", "salmonella_pls.fasta"))
    assert len(recs) == 1
    pls = recs[0]
    assert ho.reverse_complement(pls[106:167]) == SEED
    L, step = 150, 20  # every 31-mer covered by 6 reads except near the linear ends
    reads = [pls[i:i + L] for i in range(0, len(pls) - L + 1, step)] + [pls[-L:]]
    codes = np.concatenate([po.encode(r) for r in reads])
    off = np.arange(len(reads) + 1, dtype=np.uint64) * L
    t = po.Table()
    n = t.count_reads(codes, off, 31, po.KEY_PACKED)
    assert n == len(reads) * (L - 30)
    canon = {ho.normalize_dna(pls[i:i + 31]) for i in range(len(pls) - 30)}
    assert len(canon) == 93644
    assert t.size() == len(canon)
    r = po.bfs(t, 31, po.KEY_PACKED, [po.encode(SEED)], -1, 3, -1, 100000)
    got = {ho.normalize_dna(po.kmer_string(h, l, 31)) for h, l in zip(r["hi"], r["lo"])}
    assert got <= canon and len(got) > 93000


def test_strand_symmetry_and_read_order_invariance():
    rng = np.random.default_rng(7)
    for k, mode in [(31, po.KEY_PACKED), (21, po.KEY_PACKED), (63, po.KEY_POLY), (41, po.KEY_FNV1A),
                    (31, po.KEY_POLY)]:
        for _ in range(20):
            c = rng.integers(0, 4, k).astype(np.uint8)
            rc = (3 - c[::-1]).astype(np.uint8)
            assert po.key(c, k, mode) == po.key(rc, k, mode)
    L, n = 100, 300
    genome = rng.integers(0, 4, 2000).astype(np.uint8)
    starts = rng.integers(0, 2000 - L, n)
    reads = [genome[s:s + L] for s in starts]
    off = np.arange(n + 1, dtype=np.uint64) * L
    t1, t2 = po.Table(), po.Table()
    t1.count_reads(np.concatenate(reads), off, 31, po.KEY_PACKED)
    perm = rng.permutation(n)
    flipped = [(3 - reads[i][::-1]).astype(np.uint8) if i % 2 else reads[i] for i in perm]
    t2.count_reads(np.concatenate(flipped), off, 31, po.KEY_PACKED)
    k1, c1 = t1.dump()
    k2, c2 = t2.dump()
    assert np.array_equal(k1, k2) and np.array_equal(c1, c2)


def test_count_saturates_and_absent_is_minus_one():
    t = po.Table()
    assert t.get(12345) == -1 and t.get(0) == -1
    for _ in range(5):
        t.add(0)  # key 0 = poly-A, the reference's FREE marker, stored out of band
    assert t.get(0) == 5 and t.size() == 1
    t.add(77, 32760)
    t.add(77, 5)
    t.add(77, 5)
    assert t.get(77) == 32767


def test_packed_layout_matches_bytes():
    rng = np.random.default_rng(3)
    lens = [0, 5, 30, 31, 32, 64, 150, 33, 1, 97]
    reads = [rng.integers(0, 4, n).astype(np.uint8) for n in lens]
    codes = np.concatenate(reads)
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    words = po.pack(codes)
    for k, mode in [(31, po.KEY_PACKED), (5, po.KEY_PACKED), (33, po.KEY_POLY), (31, po.KEY_FNV1A)]:
        a, b = po.Table(), po.Table()
        na = a.count_reads(codes, off, k, mode)
        nb = b.count_reads_packed(words, off, k, mode)
        assert na == nb == sum(max(0, n - k + 1) for n in lens)
        ka, ca = a.dump()
        kb, cb = b.dump()
        assert np.array_equal(ka, kb) and np.array_equal(ca, cb)


def test_mt_baseline_equals_single_thread():
    g = po.synth_genome(20240531, 20000)
    L = 150
    reads = po.synth_reads(g, 1, 20000, 42, 0, 2000, L, 100)
    off = np.arange(2001, dtype=np.uint64) * L
    words = po.pack(reads)
    t = po.Table()
    n = t.count_reads_packed(words, off, 31, po.KEY_PACKED)
    w, nd, sec, tt = po.count_reads_packed_mt(words, off, 31, po.KEY_PACKED, 4, want_table=True)
    assert w == n and nd == t.size() and sec > 0
    k1, c1 = t.dump()
    k2, c2 = tt.dump()
    assert np.array_equal(k1, k2) and np.array_equal(c1, c2)
