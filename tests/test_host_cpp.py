"""The C++ host side (metacherchant_amd/csrc/host: seed reader, read ingest, trim, subgraph map,
compaction, writers) against the Python restatement in oracle/host_oracle.py, byte for byte.
CPU only: the BFS passes come from the oracle through a dump file (mc_hosttest)."""
import lzma
import os
import subprocess

import numpy as np
import pytest

from oracle import host_oracle as ho
from oracle import pyoracle as po


@pytest.fixture(scope="module")
def hosttest():
    if os.environ.get("MC_HOSTTEST"):  # (tests/test_host_sanitizers.py: the same tests on a sanitizer build)
        return os.environ["MC_HOSTTEST"]
    from metacherchant_amd import build
    build.build_host()
    assert os.path.exists(build.HOSTTEST)
    return build.HOSTTEST


def _dump(path, k, chunk, trim, genes, passes):
    with open(path, "w") as f:
        f.write("%d %d %d %d\n" % (k, chunk, 1 if trim else 0, len(genes)))
        for g in genes:
            f.write(g + "\n")
        f.write("%d\n" % len(passes))
        for d, r in passes:
            f.write("%d %d\n" % (d, len(r["lo"])))
            for h, l, dist, cov, last in zip(r["hi"], r["lo"], r["dist"], r["cov"], r["last"]):
                f.write("%s %d %d %d\n" % (po.kmer_string(h, l, k), dist, cov, last))


def _oracle_files(k, genes, passes, trim, chunk):
    env = ho.Environment(k, genes, False)
    for _, r in passes:
        kmers = [po.kmer_string(h, l, k) for h, l in zip(r["hi"], r["lo"])]
        env.add_pass(kmers, r["dist"], r["cov"], r["kept"] if trim else None)
    files = env.files(chunk)
    files["env.txt"] = files["graph.txt"]
    return files


def _compare(hosttest, tmp_path, k, genes, passes, trim, chunk):
    dump = str(tmp_path / "dump.txt")
    out = str(tmp_path / "out")
    _dump(dump, k, chunk, trim, genes, passes)
    subprocess.check_call([hosttest, "env", dump, out])
    want = _oracle_files(k, genes, passes, trim, chunk)
    for name, text in want.items():
        with open(os.path.join(out, name)) as f:
            assert f.read() == text, name
    return want


def test_fixture_example_all_files(hosttest, golden_dir, tmp_path):
    """The shipped example (Hi-C_pipline/HiCEnvironmentFinder.sh:57): 93 572 k-mers, 2 passes, merge."""
    g = os.path.join(golden_dir, "ref_example")
    fix = [l.split(" ") for l in lzma.open(os.path.join(g, "graph.txt.xz"), "rt").read().splitlines()]
    t = po.Table()
    for s, c in fix:
        t.add(po.key(po.encode(s), 31, po.KEY_PACKED), int(c))
    seqs, _ = ho.rich_fasta_read(os.path.join(g, "seq.fasta"))
    passes = [(d, po.bfs(t, 31, po.KEY_PACKED, [po.encode(s) for s in seqs], d, 5, -1, 100000)) for d in (-1, 1)]
    want = _compare(hosttest, tmp_path, 31, seqs, passes, False, 10)
    assert len(want["graph.txt"].splitlines()) == 93572


from tests.helpers import hic_case_reads  # noqa: E402


def test_merge_with_hic_seeds(hosttest, golden_dir, tmp_path):
    """The reference's only integration-tested combination (tests/EnvironmentFinderMainTest.java:23-45): --merge true with
    --hicseq.  The Hi-C sequences are BFS seeds AFTER the --seq sequences (src/algo/OneSequenceCalculator.java:181-191)
    and never gene nodes (:421-432 looks at the --seq sequences only).  Reads: the example's plasmid tiled without errors."""
    g = os.path.join(golden_dir, "ref_example")
    reads = hic_case_reads(g)
    codes = np.concatenate([po.encode(r) for r in reads])
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads])
    k = 31
    t = po.Table()
    t.count_reads(codes, off, k, po.KEY_PACKED)
    seqs, comments = ho.rich_fasta_read(os.path.join(g, "seq.fasta"))
    hic, hic_comments = ho.rich_fasta_read(os.path.join(g, "selected_reads.fasta"))
    assert len(hic) == 1047 and set(hic_comments) == {"1"}
    want_dir = str(tmp_path / "want")
    res = ho.environment_finder(t, k, po.KEY_PACKED, seqs, comments, want_dir, coverage=5, max_radius=40, bothdirs=False,
                                chunk_length=10, merge=True, hic_seqs=hic)
    want = res[os.path.join(want_dir, "merged") + "/"]
    assert want is not None
    # the same passes through the C++ host: seeds = --seq then Hi-C sequences, genes = --seq only
    seeds = [po.encode(s) for s in seqs + hic]
    passes = [(d, po.bfs(t, k, po.KEY_PACKED, seeds, d, 5, -1, 40)) for d in (-1, 1)]
    dump, out = str(tmp_path / "dump.txt"), str(tmp_path / "out")
    _dump(dump, k, 10, False, seqs, passes)
    subprocess.check_call([hosttest, "env", dump, out])
    for name, text in want.items():
        with open(os.path.join(out, name)) as f:
            assert f.read() == text, name
    # the Hi-C seeds did reach vertices the --seq seed alone does not, and none of theirs is a gene node
    alone = ho.environment_finder(t, k, po.KEY_PACKED, seqs, comments, str(tmp_path / "alone"), coverage=5, max_radius=40,
                                  bothdirs=False, chunk_length=10, merge=True)
    n_alone = len(alone[os.path.join(str(tmp_path / "alone"), "merged") + "/"]["graph.txt"].splitlines())
    assert len(want["graph.txt"].splitlines()) > n_alone + 2000
    assert want["graph.gfa"].count("CL:Z:GREEN") == alone[os.path.join(str(tmp_path / "alone"), "merged") + "/"]["graph.gfa"].count("CL:Z:GREEN") > 0


@pytest.mark.parametrize("trim", [False, True])
@pytest.mark.parametrize("bothdirs", [False, True])
def test_branching_graph_trim_and_bothdirs(hosttest, tmp_path, trim, bothdirs):
    rng = np.random.default_rng(17)
    unit = rng.integers(0, 4, 300).astype(np.uint8)
    parts = []
    for _ in range(25):
        u = unit.copy()
        pos = rng.integers(0, 300, 5)
        u[pos] = rng.integers(0, 4, 5)
        parts.append(u)
    genome = np.concatenate(parts)
    L, n, k = 90, 2500, 21
    starts = rng.integers(0, len(genome) - L, n)
    reads = np.concatenate([genome[s:s + L] for s in starts])
    off = np.arange(n + 1, dtype=np.uint64) * L
    t = po.Table()
    t.count_reads(reads, off, k, po.KEY_PACKED)
    gene = po.decode(genome[650:760])
    passes = []
    for d in ([0] if bothdirs else [-1, 1]):
        r = po.bfs(t, k, po.KEY_PACKED, [genome[650:760]], d, 2, 600, 80, trim)
        assert r is not None
        passes.append((d, r))
    _compare(hosttest, tmp_path, k, [gene], passes, trim, 25)


@pytest.mark.parametrize("k,mode", [(5, po.KEY_PACKED), (32, po.KEY_POLY), (47, po.KEY_FNV1A), (63, po.KEY_POLY)])
def test_packed_labels_at_k_extremes(hosttest, tmp_path, k, mode):
    """The host keeps node labels 2-bit packed in 128 bits: shortest labels (every bin of the subgraph map
    crowded), the first k past one word, and the longest k the ABI takes."""
    rng = np.random.default_rng(k)
    genome = rng.integers(0, 4, 1500).astype(np.uint8)
    genome[700:760] = genome[200:260]  # a repeat, so the graph branches
    L, n = 100, 600
    starts = rng.integers(0, len(genome) - L, n)
    reads = np.concatenate([genome[s:s + L] for s in starts])
    off = np.arange(n + 1, dtype=np.uint64) * L
    t = po.Table()
    t.count_reads(reads, off, k, mode)
    seed = genome[400:520]
    for trim in (False, True):
        passes = []
        for d in (-1, 1):
            r = po.bfs(t, k, mode, [seed], d, 2, 500, 200, trim)
            assert r is not None
            passes.append((d, r))
        _compare(hosttest, tmp_path, k, [po.decode(seed)], passes, trim, 1)


def test_seed_reader(hosttest, golden_dir, tmp_path):
    def run(path):
        out = subprocess.check_output([hosttest, "seeds", path]).decode().splitlines()
        nd, nc = map(int, out[0].split())
        return [l[2:] for l in out[1:1 + nd]], [l[2:] for l in out[1 + nd:1 + nd + nc]]

    g = os.path.join(golden_dir, "ref_example")
    for name in ("seq.fasta", "selected_reads.fasta"):
        d, c = run(os.path.join(g, name))
        wd, wc = ho.rich_fasta_read(os.path.join(g, name))
        assert d == wd and c == wc
    assert len(run(os.path.join(g, "selected_reads.fasta"))[0]) == 1047
    p = tmp_path / "s.fasta"
    p.write_text(">a b c\nACGTN\nnacg\n;second\n>more\nTTTT\r\nGG\n>only comment\n")
    d, c = run(str(p))
    wd, wc = ho.rich_fasta_read(str(p))
    assert d == wd == ["ACGTAAACG", "TTTTGG"] and c == wc == ["a b c", "secondmore", "only comment"]
    assert subprocess.call([hosttest, "seeds", str(tmp_path / "missing.fasta")], stderr=subprocess.DEVNULL) == 1


def test_read_ingest_policies(hosttest, tmp_path):
    def run(path):
        return subprocess.check_output([hosttest, "reads", path], stderr=subprocess.DEVNULL).decode().splitlines()

    fa = tmp_path / "r.fasta"
    fa.write_text(">r1\nACGTACGT\nACGT\n>r2 has N\nACGNACGT\n>r3\nacgtTTGA\n;comment\n>r4\n\nGGGG\n>r5 n\nACGTn\n")
    assert run(str(fa)) == ho.read_fasta_reads(str(fa)) == ["ACGTACGTACGT", "ACGTTTGA", "GGGG"]
    fq = tmp_path / "r.fastq"
    # Sanger offset (a char < 64 in the first record); '!' = phred 0 splits, N splits
    fq.write_text("@a\nACGTACGTAC\n+\nIIII!IIII5\n@b\nACGNNACG\n+b\nIIIIIIII\n\n@c\nTTTT\n+\n!!!!\n@d\nGATTACA\n+\nIIIIII!\n")
    assert run(str(fq)) == ho.read_fastq_reads(str(fq)) == ["ACGT", "CGTAC", "ACG", "ACG", "GATTAC"]
    fq2 = tmp_path / "i.fq"
    # all qualities >= '@' in the first 1000 records: read as Illumina+64, '@' = phred 0
    fq2.write_text("@a\nACGTAC\n+\nhhh@hh\n")
    assert run(str(fq2)) == ho.read_fastq_reads(str(fq2)) == ["ACG", "AC"]
    # gzip: format by the inner extension, same policies on the decompressed text (two members, no final newline)
    import gzip
    fqz = tmp_path / "R.FQ.gz"
    with open(fqz, "wb") as f:
        f.write(gzip.compress(fq.read_bytes()))
        f.write(gzip.compress(b"@e\nGGCC\n+\nIIII"))
    assert run(str(fqz)) == ho.read_fastq_reads(str(fqz)) == ["ACGT", "CGTAC", "ACG", "ACG", "GATTAC", "GGCC"]
    faz = tmp_path / "r.fna.gz"
    faz.write_bytes(gzip.compress(fa.read_bytes()))
    assert run(str(faz)) == ho.read_fasta_reads(str(faz)) == ["ACGTACGTACGT", "ACGTTTGA", "GGGG"]
    # bzip2: two streams back to back (Hadoop's codec reads on), FASTA and FASTQ
    import bz2
    fab = tmp_path / "r.fa.bz2"
    fab.write_bytes(bz2.compress(fa.read_bytes()) + bz2.compress(b">r6\nTTGACA"))
    assert run(str(fab)) == ho.read_fasta_reads(str(fab)) == ["ACGTACGTACGT", "ACGTTTGA", "GGGG", "TTGACA"]
    fqb = tmp_path / "r.fastq.bz2"
    fqb.write_bytes(bz2.compress(fq.read_bytes()))
    assert run(str(fqb)) == ho.read_fastq_reads(str(fqb)) == ["ACGT", "CGTAC", "ACG", "ACG", "GATTAC"]
    (tmp_path / "cut.fa.bz2").write_bytes(bz2.compress(fa.read_bytes() * 50)[:-20])
    assert subprocess.call([hosttest, "reads", str(tmp_path / "cut.fa.bz2")], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL) == 1
    # Sanger phred 64 ('a'): the reference keeps the phred in 6 bits, so it reads back as 0 and splits the read
    fq3 = tmp_path / "wrap.fastq"
    fq3.write_text("@a\nACGTACGT\n+\n5III~aIb\n")
    assert run(str(fq3)) == ho.read_fastq_reads(str(fq3)) == ["ACGTA", "GT"]
    # BINQ: 4-byte big-endian length + bytes (phred << 2 | nuc), 0xFF padding between records, phred 0 splits
    def rec(seq, quals):
        body = bytes((q << 2) | "AGCT".index(c) for c, q in zip(seq, quals))
        return len(body).to_bytes(4, "big") + body
    bq = tmp_path / "lib.binq"
    bq.write_bytes(rec("ACGTAC", [30, 30, 0, 30, 30, 63]) + b"\xff\xff" + rec("GGTT", [1, 2, 3, 4]) + rec("", []) + rec("AAAA", [0, 0, 0, 0]) + b"\xff")
    assert run(str(bq)) == ho.read_binq_reads(str(bq)) == ["AC", "TAC", "GGTT"]
    (tmp_path / "cut.binq").write_bytes(rec("ACGTAC", [30] * 6)[:-2])
    assert subprocess.call([hosttest, "reads", str(tmp_path / "cut.binq")], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL) == 1
    for unsupported in ("r.binq.gz", "r.binq.bz2", "r.txt.gz"):
        (tmp_path / unsupported).write_bytes(b"x")
        assert subprocess.call([hosttest, "reads", str(tmp_path / unsupported)], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL) == 1
    bad = tmp_path / "r.txt"
    bad.write_text(">x\nACGT\n")
    assert subprocess.call([hosttest, "reads", str(bad)], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL) == 1
    iupac = tmp_path / "y.fa"
    iupac.write_text(">x\nACGRT\n")
    assert subprocess.call([hosttest, "reads", str(iupac)], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL) == 1


def _env_text(rng, genome, k, start, length, cov_lo, cov_hi, drop=0.0):
    """graph.txt of a pretend environment: the canonical k-mers of genome[start:start+length] (some dropped), random depths."""
    seq = "".join("AGCT"[c] for c in genome[start:start + length])
    seen, lines = set(), []
    for i in range(len(seq) - k + 1):
        s = ho.normalize_dna(seq[i:i + k])
        if s in seen or rng.random() < drop:
            continue
        seen.add(s)
        lines.append("%s %d\n" % (s, int(rng.integers(cov_lo, cov_hi))))
    order = rng.permutation(len(lines))  # file order decides java.util.HashMap bin order
    return "".join(lines[i] for i in order)


@pytest.mark.parametrize("n_env,k,gene_id", [(2, 21, 1), (3, 15, 2), (5, 31, 1), (1, 9, 1)])
def test_environment_finder_multi_cpp_equals_oracle(hosttest, tmp_path, n_env, k, gene_id):
    """--tool environment-finder-multi: the C++ host and the Python restatement of
    MultiSequenceCalculator / GFAWriterMulti / printProbability write the same five files, byte for byte."""
    rng = np.random.default_rng(100 + n_env)
    genome = rng.integers(0, 4, 3000).astype(np.uint8)
    # a variant with a few substitutions: environments that share most of a region and differ in bubbles
    variant = genome.copy()
    for p in rng.integers(200, 2800, 12):
        variant[p] = (variant[p] + 1) & 3
    envs = []
    for e in range(n_env):
        src = variant if e % 2 else genome
        p = tmp_path / ("env%d.txt" % e)
        p.write_text(_env_text(rng, src, k, 100 + 150 * e, 1800, 1, 40, drop=0.02 * e))
        envs.append(str(p))
    seq = tmp_path / "genes.fasta"
    g1 = "".join("AGCT"[c] for c in genome[700:760])
    g2 = "".join("AGCT"[c] for c in genome[900:990])
    seq.write_text(">first gene\n%s\n>second\n%s\n" % (g1, g2))
    out = tmp_path / "out"
    log = subprocess.check_output([hosttest, "multi", str(out), str(seq), str(gene_id)] + envs).decode().splitlines()
    want, want_log = ho.environment_finder_multi(envs, str(seq), str(out), gene_id)
    assert log == want_log
    for name, text in want.items():
        assert (out / name).read_text() == text, name
    assert want["graph.gfa"].count("\nS\t") > 5 and "CL:Z:#" in want["graph.gfa"]
    if n_env >= 2:
        assert "#00ff00" in want["graph.gfa"]  # the gene's unitigs


def test_environment_finder_multi_edge_cases(hosttest, tmp_path):
    def fmt(x):
        return subprocess.check_output([hosttest, "fmt", repr(float(x))]).decode()
    # String.format("%6.2f"): HALF_UP on the float's exact value, NaN / infinities as Java prints them
    for x in (0.125, 0.375, 0.005, 0.63, 1.0, 0.995, -0.004, float("nan"), float("-inf"), 12345.678):
        assert fmt(np.float32(x)) == ho.java_format_6_2f(np.float32(x)), x
    assert ho.java_format_6_2f(np.float32(0.125)) == "  0.13" and ho.java_format_6_2f(float("nan")) == "   NaN"
    # even k with a palindromic k-mer: the reference dies with an ArrayIndexOutOfBoundsException; both restatements refuse
    e = tmp_path / "pal.txt"
    e.write_text("ACGT 3\nAAAA 2\n")
    s = tmp_path / "s.fa"
    s.write_text(">g\nAAAACGT\n")
    assert subprocess.call([hosttest, "multi", str(tmp_path / "o"), str(s), "1", str(e)], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL) == 1
    with pytest.raises(ValueError, match="palindromic"):
        ho.environment_finder_multi([str(e)], str(s), str(tmp_path / "o"))
    # k-mers of different lengths
    e2 = tmp_path / "mixed.txt"
    e2.write_text("ACGTA 3\nAAAC 2\n")
    assert subprocess.call([hosttest, "multi", str(tmp_path / "o"), str(s), "1", str(e2)], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL) == 1


def test_parallel_ingest_equals_serial(hosttest, tmp_path):
    """Uncompressed files are cut at record starts and parsed on several threads; the reads, and their order, must
    be those of the serial reader (which restates the reference's readers) -- also for quality lines that start with
    '@', blank lines, CRLF, multi-line FASTA records, and files the cutting rejects (then the serial reader runs)."""
    rng = np.random.default_rng(9)

    def run(path, threads):
        env = dict(os.environ, MC_INGEST_THREADS=str(threads), MC_INGEST_CHUNK_BYTES="700")
        return subprocess.check_output([hosttest, "reads", str(path)], stderr=subprocess.DEVNULL, env=env).decode().splitlines()

    def dna(n):
        return "".join("ACGT"[c] for c in rng.integers(0, 4, n))

    fa = tmp_path / "p.fasta"
    with open(fa, "w") as f:
        for i in range(400):
            s = dna(int(rng.integers(1, 200)))
            if i % 17 == 0:
                s = s[: len(s) // 2] + "N" + s[len(s) // 2:]
            nl = "\r\n" if i % 5 == 0 else "\n"
            f.write(">r%d some text%s" % (i, nl))
            for j in range(0, len(s), 60):
                f.write(s[j:j + 60] + nl)
            if i % 23 == 0:
                f.write(";a comment line\n")
    want = ho.read_fasta_reads(str(fa))
    assert run(fa, 1) == want
    assert run(fa, 4) == want and run(fa, 7) == want

    fq = tmp_path / "p.fastq"
    with open(fq, "w") as f:
        for i in range(600):
            n = int(rng.integers(1, 120))
            s, q = dna(n), ["I"] * n
            if i % 3 == 0:
                q[0] = "@"  # a quality line that starts like a header (phred 31 at offset 33)
            if i % 7 == 0 and n > 5:
                q[n // 2] = "!"
            if i % 13 == 0 and n > 3:
                s = s[:2] + "N" + s[3:]
            f.write("@r%d\n%s\n+%s\n%s\n%s" % (i, s, "r%d" % i if i % 2 else "", "".join(q), "\n" if i % 29 == 0 else ""))
    want = ho.read_fastq_reads(str(fq))
    assert run(fq, 1) == want
    assert run(fq, 4) == want and run(fq, 6) == want

    # all qualities >= '@' in the first 1000 records: Illumina+64 for the whole file, also in the chunks that come later
    fq64 = tmp_path / "i.fq"
    with open(fq64, "w") as f:
        for i in range(1500):
            n = int(rng.integers(5, 80))
            q = ["h"] * n
            q[n // 3] = "@"  # phred 0 at offset 64
            f.write("@x%d\n%s\n+\n%s\n" % (i, dna(n), "".join(q)))
    want = ho.read_fastq_reads(str(fq64))
    assert run(fq64, 5) == want == run(fq64, 1)

    # "+" and "@" markers swapped in one record: the reference's reader does not care, the cutting does -> serial result
    odd = tmp_path / "odd.fastq"
    text = open(fq).read().replace("@r300\n", "+r300\n", 1)
    odd.write_text(text)
    assert run(odd, 4) == ho.read_fastq_reads(str(odd))


def _spread(s):
    h = ho.java_string_hash(s)
    return h ^ (h >> 16)


def test_treeified_bins_same_order_in_all_three_maps(hosttest, tmp_path):
    """java.util.HashMap bins of 9 and more colliding keys: the treeified bin's order -- root moved to
    the front, a new node linked behind its tree parent, split / untreeify at resizes -- comes out the same from the Python
    restatement, the C++ string map and the C++ packed-k-mer map, and differs from plain insertion order.  (The three are
    restatements of the JDK 8 sources; no JVM here to run the real class.)"""
    rng = np.random.default_rng(2024)

    def rand_kmer(k):
        return "".join("ACGT"[c] for c in rng.integers(0, 4, k))

    for k, n_fill in ((31, 30), (21, 700), (45, 3000)):
        keys = []
        # three crowded buckets (index taken at a large table, so they stay together through several resizes; two of them
        # differ only in a high bit and part ways at a later split), filled one by one between other keys
        crowd = {5: [], 5 + 4096: [], 77: []}
        while min(len(v) for v in crowd.values()) < 14:
            s = rand_kmer(k)
            b = _spread(s) & 8191
            if b in crowd and len(crowd[b]) < 14:
                crowd[b].append(s)
        filler = [rand_kmer(k) for _ in range(n_fill)]
        order = []
        for i in range(14):
            for b in crowd:
                order.append(crowd[b][i])
            order.extend(filler[i * len(filler) // 14:(i + 1) * len(filler) // 14])
        seen = set()
        keys = [s for s in order if not (s in seen or seen.add(s))]
        m = ho.JavaHashMap()
        for i, s in enumerate(keys):
            m.put(s, i)
        assert m.n_treeified >= 2 and not m.order_unknown
        want = list(m.items())
        by_bucket = {}
        for s, v in want:
            by_bucket.setdefault(_spread(s) & (m.cap - 1), []).append(v)
        assert any(vs != sorted(vs) for vs in by_bucket.values())  # some bin is not in insertion order: a tree bin
        path = tmp_path / ("keys%d.txt" % k)
        path.write_text("\n".join(keys) + "\n")
        out = subprocess.check_output([hosttest, "hashmap", str(path)]).decode().splitlines()
        s_hdr = out[0].split()
        n = int(s_hdr[1])
        assert n == len(keys) and int(s_hdr[2]) == m.n_treeified and s_hdr[3] == "0"
        got_s = [(l.split()[1], int(l.split()[2])) for l in out[1:1 + n]]
        k_hdr = out[1 + n].split()
        assert int(k_hdr[1]) == n and int(k_hdr[2]) == m.n_treeified and k_hdr[3] == "0"
        got_k = [(l.split()[1], int(l.split()[2])) for l in out[2 + n:2 + 2 * n]]
        assert got_s == want and got_k == want

    # removals from treeified bins (removeTreeNode + balanceDeletion): runTrimPaths' retainAll goes through the key set's
    # iterator (movable = false: the chain only loses the node), HashMap.remove moves the root to the front afterwards; a bin
    # whose tree was too small before the removal goes back to a plain list.  Puts after removals land where the REBALANCED
    # tree says, so a wrong tree shows in the order.  The three maps agree after every mix; the Python tree keeps the
    # red-black invariants throughout.
    def check_rb(bin_):
        root = bin_.first
        while root.parent is not None:
            root = root.parent
        assert not root.red

        def walk(n):
            if n is None:
                return 1
            if n.red:
                assert not (n.left is not None and n.left.red) and not (n.right is not None and n.right.red)
            for c in (n.left, n.right):
                assert c is None or c.parent is n
            hl, hr = walk(n.left), walk(n.right)
            assert hl == hr
            return hl + (0 if n.red else 1)

        walk(root)
        assert sum(1 for _ in bin_.entries()) == count(root)

    def count(n):
        return 0 if n is None else 1 + count(n.left) + count(n.right)

    for trial in range(12):
        pool = []
        while len(pool) < 60:  # two buckets of 30 keys each at any capacity up to 1024
            s = rand_kmer(25)
            if _spread(s) & 1023 in (9, 600):
                pool.append(s)
        ops, live = [], []
        m = ho.JavaHashMap()
        val = 0
        n_tree_removals = 0
        for step in range(400):
            r = rng.random()
            if live and (r < 0.35 or len(live) == len(pool)):
                s = live.pop(int(rng.integers(0, len(live))))
                movable = bool(rng.integers(0, 2))
                was_tree = isinstance(m.bins[_spread(s) & (m.cap - 1)], ho._TreeBin)
                ops.append(("~" if movable else "-") + s)
                m.remove(s, movable)
                n_tree_removals += was_tree
            elif r < 0.9 or not live:
                cand = [s for s in pool if s not in live]
                if not cand:
                    continue
                s = cand[int(rng.integers(0, len(cand)))]
                live.append(s)
                ops.append(s)
                m.put(s, val)
                val += 1
            else:  # a filler key now and then: resizes split and rebuild the bins
                s = rand_kmer(25)
                if s in m:
                    continue
                ops.append(s)
                m.put(s, val)
                val += 1
            for bn in m.bins:
                if isinstance(bn, ho._TreeBin):
                    check_rb(bn)
        assert n_tree_removals > 20 and not m.order_unknown
        path = tmp_path / ("ops%d.txt" % trial)
        path.write_text("\n".join(ops) + "\n")
        out = subprocess.check_output([hosttest, "hashmap", str(path)]).decode().splitlines()
        want = list(m.items())
        n = len(want)
        assert out[0].split()[1] == str(n) and out[0].split()[3] == "0" and out[1 + n].split()[3] == "0"
        assert [(l.split()[1], int(l.split()[2])) for l in out[1:1 + n]] == want
        assert [(l.split()[1], int(l.split()[2])) for l in out[2 + n:2 + 2 * n]] == want


def test_cli_refuses_k_above_63_with_a_clear_message(tmp_path):
    """The reference takes any k > 31 (it hashes k-mer strings, src/io/LargeKIOUtils.java:41-54); this build keeps oriented
    k-mers in 128 bits and says so instead of miscounting.  The check comes before any device is touched: CPU test."""
    from metacherchant_amd import build
    build.build_all()
    if not os.path.exists(build.CLI):
        pytest.skip("the CLI needs libmcgpu.so (hipcc)")
    seq = tmp_path / "s.fasta"
    seq.write_text(">s\n" + "ACGT" * 40 + "\n")
    for k in (64, 127):
        p = subprocess.run([build.CLI, "-k", str(k), "-i", str(seq), "--seq", str(seq), "-o", str(tmp_path / "o"), "-w", str(tmp_path / "wd"),
                            "--maxkmers", "10", "--force"], capture_output=True, text=True, timeout=120)
        assert p.returncode == 1
        assert "k = %d is not supported: this build handles k <= 31 (packed keys) and 32 <= k <= 63 (hash keys)" % k in p.stderr
