"""End to end on the GPU: the native `metacherchant --tool environment-finder` CLI (C++ host + HIP
library) against the oracle pipeline (oracle/ C counting + BFS, Python host restatement) on the
same FASTA/FASTQ inputs: every output file byte-identical.  Needs a real MI355X: -m gpu."""
import os
import subprocess

import numpy as np
import pytest

from oracle import host_oracle as ho
from oracle import pyoracle as po
from tests.helpers import synth_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cli():
    from metacherchant_amd import build
    build.build_all()
    assert os.path.exists(build.CLI)
    return build.CLI


def _write_fasta(path, reads, L, names=True, n_every=0):
    with open(path, "w") as f:
        for i in range(len(reads) // L):
            s = po.decode(reads[i * L:(i + 1) * L])
            if n_every and i % n_every == 3:
                s = s[:40] + "N" + s[41:]  # FASTA records with N are dropped whole
            f.write(">r%d\n%s\n%s\n" % (i, s[:70], s[70:]))  # multi-line records


def _oracle_run(read_files, k, mode, seqs, comments, out_dir, **kw):
    t = po.Table()
    for p in read_files:
        inner = p[:-3] if p.endswith(".gz") else p
        reads = ho.read_fastq_reads(p) if inner.endswith((".fastq", ".fq")) else ho.read_fasta_reads(p)
        codes = np.concatenate([po.encode(r) for r in reads])
        off = np.zeros(len(reads) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(r) for r in reads])
        t.count_reads(codes, off, k, mode)
    return t, ho.environment_finder(t, k, mode, seqs, comments, out_dir, **kw)


def _assert_same_tree(want_results, got_root, want_root):
    for prefix, files in want_results.items():
        rel = os.path.relpath(prefix, want_root)
        gdir = os.path.join(got_root, rel)
        if files is None:
            assert not os.path.exists(os.path.join(gdir, "graph.txt"))
            continue
        for name, text in files.items():
            with open(os.path.join(gdir, name)) as f:
                assert f.read() == text, (rel, name)


def test_cli_config1_both_passes_merge(cli, tmp_path):
    """BASELINE.json configs[0]: 10k x 150 bp, k=31, coverage=5, maxkmers=100000, bothdirs=False."""
    genome, reads, _ = synth_case(1, 50000, 10000, 150, 100)
    r1, r2 = str(tmp_path / "reads_1.fasta"), str(tmp_path / "reads_2.fa.gz")
    _write_fasta(r1, reads[:6000 * 150], 150, n_every=50)
    _write_fasta(r2[:-3], reads[6000 * 150:], 150)
    import gzip
    with open(r2[:-3], "rb") as f, open(r2, "wb") as z:  # the second file gzip-compressed (read through zlib)
        z.write(gzip.compress(f.read()))
    os.remove(r2[:-3])
    seq = str(tmp_path / "seed.fasta")
    with open(seq, "w") as f:
        f.write(">seed\n%s\n" % po.decode(genome[10000:10500]))
    out, want = str(tmp_path / "out"), str(tmp_path / "want")
    cmd = [cli, "--tool", "environment-finder", "-k", "31", "--coverage", "5", "--reads", r1, r2, "--seq", seq,
           "--output", out, "--work-dir", str(tmp_path / "wd"), "--maxkmers", "100000", "--bothdirs", "False",
           "--chunklength", "10", "--merge", "true", "--force"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    for line in ("Loading file reads_1.fasta...", "reads added", "Hashtable size: ", "Finding single environment for 1 sequences",
                 "Extending endings by 0 kmers", "Finished processing all sequences!"):
        assert line in p.stderr
    seqs, comments = ho.rich_fasta_read(seq)
    t, res = _oracle_run([r1, r2], 31, po.KEY_PACKED, seqs, comments, want, coverage=5, max_kmers=100000,
                         bothdirs=False, chunk_length=10, merge=True)
    assert "Hashtable size: %d kmers" % t.size() in p.stderr
    _assert_same_tree(res, out, want)
    assert os.path.exists(os.path.join(str(tmp_path / "wd"), "SUCCESS"))
    assert len(res[os.path.join(want, "merged") + "/"]["graph.txt"].splitlines()) > 10000


@pytest.mark.parametrize("extra,kw", [
    (["--maxradius", "150", "--bothdirs", "--trim"], dict(max_radius=150, bothdirs=True, trim=True)),
    (["--maxkmers=700", "--coverage=3", "--chunklength", "40"], dict(max_kmers=700, coverage=3, chunk_length=40)),
    (["--maxkmers", "900", "--maxradius", "60", "--trim", "true", "--coverage", "2"],
     dict(max_kmers=900, max_radius=60, trim=True, coverage=2)),
])
def test_cli_multi_sequence_dirs_and_flags(cli, tmp_path, extra, kw):
    """One output directory per FASTA comment (no --merge), a sequence absent from the reads, FASTQ input."""
    genome, reads, _ = synth_case(2, 20000, 6000, 150, 50)
    fq = str(tmp_path / "reads.fastq")
    with open(fq, "w") as f:
        for i in range(6000):
            s = po.decode(reads[i * 150:(i + 1) * 150])
            q = ["I"] * 150
            if i % 7 == 0:
                q[60] = "!"  # phred 0: the read is split here and the base dropped
            if i % 11 == 0:
                s = s[:100] + "N" + s[101:]
            f.write("@r%d\n%s\n+\n%s\n" % (i, s, "".join(q)))
    seq = str(tmp_path / "genes.fasta")
    rng = np.random.default_rng(4)
    with open(seq, "w") as f:
        f.write(">geneA\n%s\n>geneB some text\n%s\n>absent\n%s\n" % (
            po.decode(genome[3000:3300]), po.decode(genome[25000:25200]), po.decode(rng.integers(0, 4, 120).astype(np.uint8))))
    out, want = str(tmp_path / "out"), str(tmp_path / "want")
    cmd = [cli, "-k", "25", "-i", fq, "--seq", seq, "-o", out, "-w", str(tmp_path / "wd"), "--force"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    assert "Could not find any k-mers of the target gene in the input, halting." in p.stderr
    seqs, comments = ho.rich_fasta_read(seq)
    _, res = _oracle_run([fq], 25, po.KEY_PACKED, seqs, comments, want, **kw)
    assert res[os.path.join(want, "absent") + "/"] is None
    _assert_same_tree(res, out, want)


@pytest.mark.parametrize("k,hash_name,mode", [(41, "poly", po.KEY_POLY), (63, "fnv1a", po.KEY_FNV1A), (21, "poly", po.KEY_POLY)])
def test_cli_hash_key_modes(cli, tmp_path, k, hash_name, mode):
    """k > 31 (or --forcehash): the table key is the reference's 64-bit hash (src/utils/*Hash.java)."""
    genome, reads, _ = synth_case(1, 30000, 5000, 150, 30)
    r1 = str(tmp_path / "reads.fna")
    _write_fasta(r1, reads, 150)
    seq = str(tmp_path / "seed.fasta")
    with open(seq, "w") as f:
        f.write(">s\n%s\n" % po.decode(genome[15000:15300]))
    out, want = str(tmp_path / "out"), str(tmp_path / "want")
    cmd = [cli, "-k", str(k), "-i", r1, "--seq", seq, "-o", out, "-w", str(tmp_path / "wd"), "--force", "--maxkmers", "3000",
           "--coverage", "3", "--bothdirs", "True", "--hash", hash_name] + (["--forcehash"] if k <= 31 else [])
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    assert "Reading hashes of k-mers instead" in p.stderr
    seqs, comments = ho.rich_fasta_read(seq)
    _, res = _oracle_run([r1], k, mode, seqs, comments, want, coverage=3, max_kmers=3000, bothdirs=True)
    _assert_same_tree(res, out, want)


def _write_reads_fasta(path, reads):
    with open(path, "w") as f:
        for i, r in enumerate(reads):
            f.write(">r%d\n%s\n" % (i, r))


def test_cli_merge_with_hicseq(cli, golden_dir, tmp_path):
    """The reference's integration test, tests/EnvironmentFinderMainTest.java:23-45: --merge true --hicseq selected_reads.fasta
    --maxradius ... --bothdirs False --chunklength 10 (its WGS reads are not shipped: tests/helpers.py hic_case_reads).  The
    1047 Hi-C sequences seed the walk after the --seq sequence (src/algo/OneSequenceCalculator.java:181-191)."""
    from tests.helpers import hic_case_reads
    g = os.path.join(golden_dir, "ref_example")
    r1 = str(tmp_path / "wgs.fasta")
    _write_reads_fasta(r1, hic_case_reads(g))
    seq, hicf = os.path.join(g, "seq.fasta"), os.path.join(g, "selected_reads.fasta")
    out, want = str(tmp_path / "out"), str(tmp_path / "want")
    cmd = [cli, "--k", "31", "--coverage", "5", "--reads", r1, "--seq", seq, "--output", out, "--work-dir", str(tmp_path / "wd"),
           "--maxradius", "100000", "--bothdirs", "False", "--chunklength", "10", "--merge", "true", "--hicseq", hicf, "--force"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    assert "hicSequences = 1047" in p.stderr
    assert "Finding single environment for 1 sequences" in p.stderr
    seqs, comments = ho.rich_fasta_read(seq)
    hic, _ = ho.rich_fasta_read(hicf)
    _, res = _oracle_run([r1], 31, po.KEY_PACKED, seqs, comments, want, coverage=5, max_radius=100000, bothdirs=False,
                         chunk_length=10, merge=True, hic_seqs=hic)
    _assert_same_tree(res, out, want)
    files = res[os.path.join(want, "merged") + "/"]
    assert os.path.exists(os.path.join(out, "merged", "graph.gfa"))
    # the plasmid piece the --seq gene sits in, plus what the Hi-C seeds reach in their loci; only the gene's nodes are green
    _, alone = _oracle_run([r1], 31, po.KEY_PACKED, seqs, comments, str(tmp_path / "alone"), coverage=5, max_radius=100000,
                           bothdirs=False, chunk_length=10, merge=True)
    alone = alone[os.path.join(str(tmp_path / "alone"), "merged") + "/"]
    assert len(files["graph.txt"].splitlines()) > len(alone["graph.txt"].splitlines()) + 2000
    assert files["graph.gfa"].count("CL:Z:GREEN") == alone["graph.gfa"].count("CL:Z:GREEN") > 0


def test_cli_hicseq_without_merge_names_directories_from_the_hic_file(cli, tmp_path):
    """Without --merge the Hi-C sequences seed nothing, but their FASTA comments REPLACE those of --seq as output directory
    names (src/tools/EnvironmentFinderMain.java:149 overwrites `comments`; :245-248 uses them); with fewer Hi-C records
    than sequences the reference throws on the missing comment, here a clear error."""
    genome, reads, _ = synth_case(1, 30000, 5000, 150, 50)
    r1 = str(tmp_path / "reads.fasta")
    _write_fasta(r1, reads, 150)
    seq, hicf = str(tmp_path / "genes.fasta"), str(tmp_path / "hic.fasta")
    with open(seq, "w") as f:
        f.write(">geneA\n%s\n>geneB\n%s\n" % (po.decode(genome[3000:3200]), po.decode(genome[20000:20150])))
    with open(hicf, "w") as f:
        f.write(">first hic\n%s\n>second\n%s\n>third\n%s\n" % (po.decode(genome[9000:9100]), po.decode(genome[12000:12100]), po.decode(genome[15000:15100])))
    out, want = str(tmp_path / "out"), str(tmp_path / "want")
    cmd = [cli, "-k", "25", "-i", r1, "--seq", seq, "--hicseq", hicf, "-o", out, "-w", str(tmp_path / "wd"), "--force", "--maxkmers", "800",
           "--coverage", "3"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    seqs, _ = ho.rich_fasta_read(seq)
    _, hic_comments = ho.rich_fasta_read(hicf)
    assert hic_comments[:2] == ["first hic", "second"]
    _, res = _oracle_run([r1], 25, po.KEY_PACKED, seqs, hic_comments, want, coverage=3, max_kmers=800)  # (no hic_seqs: not merged)
    assert set(os.listdir(out)) == {"first hic", "second"}
    _assert_same_tree(res, out, want)
    # fewer Hi-C records than sequences
    with open(hicf, "w") as f:
        f.write(">only\n%s\n" % po.decode(genome[9000:9100]))
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 1 and "has no FASTA comment" in p.stderr


def test_cli_error_paths(cli, tmp_path):
    """tests/EnvironmentFinderMainTest.java:47-94 pins the two messages."""
    r1 = str(tmp_path / "reads.fasta")
    open(r1, "w").write(">r\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n")
    seq = str(tmp_path / "seed.fasta")
    open(seq, "w").write(">s\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n")
    base = [cli, "-k", "21", "-i", r1, "-o", str(tmp_path / "o"), "-w", str(tmp_path / "wd"), "--force"]
    p = subprocess.run(base + ["--seq", str(tmp_path / "nope.fasta"), "--maxkmers", "10"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 1 and "Could not load sequences from " + str(tmp_path / "nope.fasta") in p.stderr
    p = subprocess.run(base + ["--seq", seq, "--hicseq", str(tmp_path / "nohic.fasta"), "--maxkmers", "10"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 1 and "Could not load Hi-C sequences from " + str(tmp_path / "nohic.fasta") in p.stderr
    p = subprocess.run(base + ["--seq", seq], capture_output=True, text=True, timeout=600)
    assert p.returncode == 1 and "At least one of --maxkmers and --maxradius parameters should be set" in p.stderr


@pytest.mark.parametrize("k,mode", [(25, po.KEY_PACKED), (40, po.KEY_POLY)])
def test_cli_kmer_counter_and_reload(cli, tmp_path, k, mode):
    """--tool kmer-counter: <name>.stat.txt byte-identical with the oracle, <name>.kmers.bin the same records;
    a table reloaded from the file (mc_load_kmers) equals the original and walks to the same environment."""
    import metacherchant_amd as m
    from tests.helpers import assert_bfs_equal, seed_windows
    genome, reads, off = synth_case(1, 30000, 4000, 150, 80)
    fa = str(tmp_path / "Sample_A.fasta")
    _write_fasta(fa, reads, 150, n_every=40)
    wd = str(tmp_path / "wd")
    p = subprocess.run([cli, "-t", "kmer-counter", "-k", str(k), "-i", fa, "-w", wd, "--force"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    t = po.Table()
    rr = ho.read_fasta_reads(fa)
    codes = np.concatenate([po.encode(r) for r in rr])
    roff = np.zeros(len(rr) + 1, dtype=np.uint64)
    roff[1:] = np.cumsum([len(r) for r in rr])
    t.count_reads(codes, roff, k, mode)
    want_recs, want_stat = ho.kmer_counter_files(t)
    assert ho.library_name(fa) == "Sample_A"
    bin_path = os.path.join(wd, "kmers", "Sample_A.kmers.bin")
    assert open(os.path.join(wd, "kmers", "Sample_A.stat.txt")).read() == want_stat
    assert ho.read_kmers_bin(bin_path) == want_recs
    assert "Hashtable size: %d kmers" % t.size() in p.stderr
    assert "k-mers found, " in p.stderr and "(100.0%) of them is good (not erroneous)" in p.stderr
    assert "k-mers printed to " + bin_path in p.stderr
    # reload: every record (threshold 0), then only the solid ones
    ctx = m.Context(k, mode, 0, 0)
    assert ctx.load_kmers(bin_path) == (len(want_recs), len(want_recs))
    assert ctx.finalize() == t.size()
    gk, gc = ctx.export(0)
    ok, oc = t.dump()
    assert np.array_equal(gk, ok) and np.array_equal(gc, oc)
    seed = genome[8000:8200]
    hi, lo = seed_windows(seed, k)
    assert_bfs_equal(ctx.bfs(hi, lo, 0, 4, 3000, -1), po.bfs(t, k, mode, [seed], 0, 4, 3000, -1))
    ctx.close()
    ctx = m.Context(k, mode, 0, 0)
    n_rec, n_add = ctx.load_kmers(bin_path, 3)
    assert n_rec == len(want_recs) and n_add == sum(1 for _, c in want_recs if c > 3) == ctx.finalize()
    bad = tmp_path / "bad.kmers.bin"
    bad.write_bytes(b"123456789012345")
    with pytest.raises(m.native.McError, match="multiple of the 10-byte record"):
        ctx.load_kmers(str(bad))
    ctx.close()


def test_cli_more_seed_sequences_than_one_bfs_batch(cli, tmp_path):
    """300 seed sequences without --merge: the reference runs one OneSequenceCalculator per sequence with no limit on
    their number (EnvironmentFinderMain.java:218-225); the CLI sends their 600 passes to the GPU in chunks."""
    genome, reads, _ = synth_case(2, 40000, 9000, 150, 40)
    fa = str(tmp_path / "reads.fasta")
    _write_fasta(fa, reads, 150)
    seq = str(tmp_path / "genes.fasta")
    with open(seq, "w") as f:
        for i in range(300):
            f.write(">gene%03d\n%s\n" % (i, po.decode(genome[200 + 250 * i:200 + 250 * i + 90])))
    out, want = str(tmp_path / "out"), str(tmp_path / "want")
    cmd = [cli, "-k", "31", "-i", fa, "--seq", seq, "-o", out, "-w", str(tmp_path / "wd"), "--force", "--maxkmers", "400",
           "--coverage", "3"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    seqs, comments = ho.rich_fasta_read(seq)
    _, res = _oracle_run([fa], 31, po.KEY_PACKED, seqs, comments, want, coverage=3, max_kmers=400)
    assert len(res) == 300
    _assert_same_tree(res, out, want)


def test_cli_config5_four_environments_then_multi_join(cli, tmp_path):
    """BASELINE.json configs[4] at small scale: four read sets drawn from overlapping subsets of the genomes (all hold
    the contig with the seed gene) -> four environments built by the HIP path -> `--tool environment-finder-multi` joins
    their graph.txt files (src/tools/EnvironmentFinderMultiMain.java:84-170).  Every file of every stage equals the oracle
    pipeline's.  (Parity of the join itself is unpinned: the reference ships no output of that tool.)"""
    rng = np.random.default_rng(77)
    contigs = [rng.integers(0, 4, 30000).astype(np.uint8) for _ in range(4)]
    variant = contigs[0].copy()  # two of the four samples carry a variant of the gene's neighbourhood: coloured bubbles
    for pos in (9800, 10650, 11200):
        variant[pos] = (variant[pos] + 1) & 3
    gene = po.decode(contigs[0][10000:10300])
    seq = str(tmp_path / "gene.fasta")
    with open(seq, "w") as f:
        f.write(">thegene\n%s\n" % gene)  # (no blank in the name: array options are re-tokenised on [, ], Tool.java:888-895)
    envs_got, envs_want = [], []
    for s in range(4):
        src = [variant if s % 2 else contigs[0], contigs[1 + s % 3]]
        L, n = 150, 4000
        reads = []
        for i in range(n):
            g = src[int(rng.integers(0, 2))]
            a = int(rng.integers(0, len(g) - L))
            r = g[a:a + L].copy()
            if rng.integers(0, 2):
                r = (3 - r[::-1]).astype(np.uint8)  # the other strand (complement = 3 - code)
            reads.append(r)
        fa = str(tmp_path / ("sample%d.fasta" % s))
        _write_fasta(fa, np.concatenate(reads), L)
        out, want = str(tmp_path / ("out%d" % s)), str(tmp_path / ("want%d" % s))
        cmd = [cli, "-k", "31", "-i", fa, "--seq", seq, "-o", out, "-w", str(tmp_path / ("wd%d" % s)), "--force", "--maxradius", "400",
               "--coverage", "3", "--bothdirs", "true"]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        seqs, comments = ho.rich_fasta_read(seq)
        _, res = _oracle_run([fa], 31, po.KEY_PACKED, seqs, comments, want, coverage=3, max_radius=400, bothdirs=True)
        _assert_same_tree(res, out, want)
        envs_got.append(os.path.join(out, "thegene", "graph.txt"))
        envs_want.append(os.path.join(want, "thegene", "graph.txt"))
        os.makedirs(os.path.dirname(envs_want[-1]), exist_ok=True)
        with open(envs_want[-1], "w") as f:
            f.write(res[os.path.join(want, "thegene") + "/"]["graph.txt"])
    joined = str(tmp_path / "joined")
    p = subprocess.run([cli, "--tool", "environment-finder-multi", "--env"] + envs_got + ["--seq", seq, "-o", joined, "-w",
                       str(tmp_path / "wdj"), "--force"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    # (the Jaccard tables print the environments' paths: the oracle joins the very files the CLI wrote, which the
    # comparison above found byte-identical with its own)
    for g, w in zip(envs_got, envs_want):
        assert open(g).read() == open(w).read()
    want_files, _ = ho.environment_finder_multi(envs_got, seq, joined, 1)
    for name, text in want_files.items():
        with open(os.path.join(joined, name)) as f:
            assert f.read() == text, name
    gfa = want_files["graph.gfa"]
    assert gfa.count("\nS\t") > 3 and "#00ff00" in gfa  # the gene's unitigs are marked, the samples' bubbles coloured


@pytest.mark.parametrize("devices,k,extra,env", [("0,0", 31, ["--maxkmers", "4000", "--coverage", "3"], {}),
                                                ("0,0,0", 25, ["--maxradius", "200", "--coverage", "2", "--bothdirs", "true"], {}),
                                                ("0,0", 41, ["--maxkmers", "2500", "--coverage", "3", "--bothdirs", "true"], {}),
                                                # round 3's way of walking (the solid k-mers gathered into a second table on the first device);
                                                # by default the first device reads the others' tables in place
                                                ("0,0,0", 31, ["--maxkmers", "4000", "--coverage", "3"], {"MC_GROUP_WALK": "gather"}),
                                                ("0,0", 41, ["--maxkmers", "2500", "--coverage", "3"], {"MC_GROUP_WALK": "gather"}),
                                                # exchanges of 3 x 833 reads: the FASTQ's last one holds TWO reads, so a share extracts
                                                # (into the buffer its list of solid k-mers sat in) and then owns nothing of the batch:
                                                # the list must not be taken for valid
                                                ("0,0,0", 31, ["--maxkmers", "4000", "--coverage", "3"], {"MC_GROUP_BATCH_READS": "833", "MC_TOKENIZER": "host"}),
                                                # the files tokenised on the first device in chunks of ~1000 reads, the other devices
                                                # fetching their shares from its read store: several exchanges a file
                                                ("0,0,0", 31, ["--maxkmers", "4000", "--coverage", "3"],
                                                 {"MC_GROUP_BATCH_READS": "1024", "MC_TOKENIZER_CHUNK_BYTES": "160000", "MC_INGEST_DEBUG": "1"})])
def test_cli_several_devices_equal_one(cli, tmp_path, devices, k, extra, env):
    """`--devices a,b,...`: reads dealt to the devices, super-k-mer records (keys for k < 23 and hash keys) exchanged by owner
    with peer copies, every device counts what it owns, the solid shards are gathered on the first for the BFS -- the
    native counterpart of distributed.py.  Output byte-identical with one device's and with the oracle's.
    This box has one GPU, so the devices are shares of it (the same ordinal several times): everything but the xGMI hop
    itself is exercised."""
    genome, reads, _ = synth_case(2, 40000, 12000, 150, 60)
    fa1, fa2 = str(tmp_path / "a.fasta"), str(tmp_path / "b.fq")
    _write_fasta(fa1, reads[:7000 * 150], 150, n_every=60)
    with open(fa2, "w") as f:
        for i in range(7000, 12000):
            f.write("@r%d\n%s\n+\n%s\n" % (i, po.decode(reads[i * 150:(i + 1) * 150]), "I" * 150))
    seq = str(tmp_path / "genes.fasta")
    with open(seq, "w") as f:
        f.write(">g1\n%s\n>g2\n%s\n" % (po.decode(genome[5000:5400]), po.decode(genome[52000:52300])))
    hashed = k > 31
    mode = po.KEY_POLY if hashed else po.KEY_PACKED
    outs = {}
    for name, dev in (("one", None), ("many", devices)):
        out = str(tmp_path / ("out_" + name))
        cmd = [cli, "-k", str(k), "-i", fa1, fa2, "--seq", seq, "-o", out, "-w", str(tmp_path / ("wd_" + name)), "--force"] + extra
        if dev:
            cmd += ["--devices", dev]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr[-2000:]
        outs[name] = (out, p.stderr)
    assert "Counting on %d devices" % len(devices.split(",")) in outs["many"][1]
    if "MC_INGEST_DEBUG" in env:  # (the device path ran, in several exchanges)
        lines = [l for l in outs["many"][1].splitlines() if "[ingest] group:" in l]
        assert len(lines) == 2 and all(int(l.split(" reads in ")[1].split()[0]) >= 2 for l in lines), outs["many"][1][-2000:]
    size_line = [l for l in outs["one"][1].splitlines() if "Hashtable size" in l][0].split("Hashtable size")[1]
    assert ("Hashtable size" + size_line) in outs["many"][1]  # owners are disjoint: the shards add up to the one table
    # packed k-mers of 23 bases and more travel as super-k-mer records in the binned form (include/mcgpu.h
    # mc_extract_superkmers_binned_dev): the devices' counting runs start at their second level
    import json
    metrics = json.load(open(str(tmp_path / "wd_many" / "metrics.json")))
    assert (metrics["binned_runs"] > 0) == (23 <= k <= 31), metrics
    # every device's reads sit in the first device's store and its records' pointers lead there (mc_set_read_pointers mode 2): the
    # walk's look-ahead does as well as one device's over all the reads -- about as many verification rounds for the same levels
    one = json.load(open(str(tmp_path / "wd_one" / "metrics.json")))
    # (with pointers from the first device's reads alone these walks took 2.2 to 3.4 times the rounds; the rounds of a walk vary a
    # little from run to run -- scouts and verifier race --, and for hash keys, which travel as one key a window and leave their
    # pointers by another rule, by more: checked for packed keys)
    assert metrics["bfs_levels"] == one["bfs_levels"], (metrics, one)
    assert hashed or metrics["bfs_rounds"] <= 2 * one["bfs_rounds"] + 32, (metrics, one)
    kw = {}
    it = iter(extra)
    for a in it:
        v = next(it)
        kw[{"--maxkmers": "max_kmers", "--maxradius": "max_radius", "--coverage": "coverage", "--bothdirs": "bothdirs"}[a]] = (v == "true") if a == "--bothdirs" else int(v)
    seqs, comments = ho.rich_fasta_read(seq)
    want = str(tmp_path / "want")
    _, res = _oracle_run([fa1, fa2], k, mode, seqs, comments, want, **kw)
    _assert_same_tree(res, outs["many"][0], want)
    _assert_same_tree(res, outs["one"][0], want)


def test_cli_devices_batch_that_falls_back_to_keys_keeps_its_owners(cli, tmp_path):
    """a batch whose records overflow an owner's piece (low-complexity reads: one minimizer, one owner) travels as
    keys instead.  Keys used to be dealt by their own hash, records by their minimizer: a k-mer counted as records in one
    batch and as keys in the next had its count split over two shards, each perhaps under --coverage.  Here the middle one
    of three exchanges is two thirds poly-A / (AC)n reads and takes the key form on eight shares of the GPU, the genome's
    reads are spread over all three, and the output must be the one device's and the oracle's, byte for byte."""
    genome, reads, _ = synth_case(2, 40000, 60000, 150, 60)
    L = 150
    low = np.zeros(L, dtype=np.uint8), np.tile(np.array([0, 2], dtype=np.uint8), L // 2)  # AAAA..., ACAC...
    # 20 000 genome reads, 14 000 low-complexity ones in runs of 1 000 (a workgroup of the extraction then holds nothing else,
    # and all its records go to one owner), 40 000 genome reads: exchanges of 8 x 4 200 reads -- the first two meet the runs
    # and fall back to keys, the third (6 800 genome reads) travels as records
    parts = [reads[:20000 * L]] + [np.tile(low[(i // 1000) & 1], 1) for i in range(14000)] + [reads[20000 * L:]]
    allr = np.concatenate(parts)
    n = len(allr) // L
    assert n == 74000
    fa = str(tmp_path / "reads.fasta")
    _write_fasta(fa, allr, L)
    seq = str(tmp_path / "genes.fasta")
    with open(seq, "w") as f:
        f.write(">g1\n%s\n" % po.decode(genome[5000:5400]))
    outs = {}
    for name, dev in (("one", None), ("many", "0,0,0,0,0,0,0,0")):
        out = str(tmp_path / ("out_" + name))
        cmd = [cli, "-k", "31", "-i", fa, "--seq", seq, "-o", out, "-w", str(tmp_path / ("wd_" + name)), "--force", "--maxkmers", "6000", "--coverage", "40"]
        if dev:
            cmd += ["--devices", dev]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, MC_GROUP_BATCH_READS="4200", MC_TOKENIZER="host", MC_INGEST_DEBUG="1"))
        assert p.returncode == 0, p.stderr[-2000:]
        outs[name] = (out, p.stderr)
    assert "this batch travels as keys" in outs["many"][1], outs["many"][1][-3000:]
    assert outs["many"][1].count("this batch travels as keys") == 2  # (not every batch: the k-mers of the genome are counted in both forms)
    size_line = [l for l in outs["one"][1].splitlines() if "Hashtable size" in l][0].split("Hashtable size")[1]
    assert ("Hashtable size" + size_line) in outs["many"][1]
    seqs, comments = ho.rich_fasta_read(seq)
    want = str(tmp_path / "want")
    # --coverage 40 of ~110-fold: a k-mer whose count were split over two shards would fall under it
    _, res = _oracle_run([fa], 31, po.KEY_PACKED, seqs, comments, want, max_kmers=6000, coverage=40)
    _assert_same_tree(res, outs["many"][0], want)
    _assert_same_tree(res, outs["one"][0], want)


def test_cli_rccl_transport_wants_a_gpu_per_rank(cli, tmp_path):
    """The RCCL transport of the native driver (ncclCommInitAll, grouped ncclSend / ncclRecv) cannot run on shares of one GPU
    -- a communicator takes every device once -- and says so instead of hanging in the rendezvous; test_cli_two_real_devices
    runs it where there are two GPUs."""
    genome, reads, _ = synth_case(1, 5000, 300, 100, 0)
    fa, seq = str(tmp_path / "r.fasta"), str(tmp_path / "s.fasta")
    _write_fasta(fa, reads, 100)
    with open(seq, "w") as f:
        f.write(">g\n%s\n" % po.decode(genome[1000:1200]))
    p = subprocess.run([cli, "-k", "31", "-i", fa, "--seq", seq, "-o", str(tmp_path / "o"), "-w", str(tmp_path / "w"), "--maxkmers", "1000",
                        "--devices", "0,0"], capture_output=True, text=True, timeout=300, env=dict(os.environ, MC_GROUP_TRANSPORT="rccl"))
    assert p.returncode != 0 and "RCCL wants every rank on a GPU of its own" in (p.stderr + p.stdout)


def test_cli_two_real_devices(cli, tmp_path):
    """The same over two different GPUs, with both transports of the native driver (peer copies, RCCL), and bench.py's
    two-rank path over RCCL (`nccl`) -- skipped on a one-GPU box."""
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    genome, reads, _ = synth_case(1, 40000, 9000, 150, 40)
    fa = str(tmp_path / "reads.fasta")
    _write_fasta(fa, reads, 150)
    seq = str(tmp_path / "gene.fasta")
    with open(seq, "w") as f:
        f.write(">g\n%s\n" % po.decode(genome[9000:9400]))
    trees = []
    for dev, transport in ((None, None), ("0,1", "peer"), ("0,1", "rccl")):
        out = str(tmp_path / ("out" + (transport or "")))
        cmd = [cli, "-k", "31", "-i", fa, "--seq", seq, "-o", out, "-w", str(tmp_path / ("wd" + (transport or ""))), "--force", "--maxkmers", "5000", "--coverage", "3"]
        p = subprocess.run(cmd + (["--devices", dev] if dev else []), capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, **({"MC_GROUP_TRANSPORT": transport} if transport else {})))
        assert p.returncode == 0, p.stderr[-2000:]
        trees.append({n: open(os.path.join(out, "g", n)).read() for n in ("graph.txt", "graph.gfa", "seqs.fasta")})
    assert trees[0] == trees[1] == trees[2]
    # bench.py --gpus 2: one process per GPU, the exchange as all-to-alls over RCCL; its step checks the walk against the
    # one-GPU result itself (distinct k-mers and vertices reached are printed: compare with a one-rank run of twice the reads)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29571", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "1000000",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=root)
    assert two.returncode == 0, two.stderr[-2000:]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--reads", "2000000", "--no-cpu-baseline",
                          "--skip-no-hint"], capture_output=True, text=True, timeout=900, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    import json
    j2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert j2["n_gpus"] == 2 and j2["distinct_kmers"] == j1["distinct_kmers"] and j2["bfs"]["reached"] == j1["bfs"]["reached"]
