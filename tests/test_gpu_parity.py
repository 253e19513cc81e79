"""Parity of the HIP path (through the C ABI of include/mcgpu.h) against the CPU oracle.
Bit-exact: integer / byte / index work.  Needs a real MI355X: run with -m gpu."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests.helpers import (GENOME_SEED, READ_SEED, assert_bfs_equal, oracle_table, ragged_case, seed_windows,
                           synth_case)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mc():
    import metacherchant_amd as m
    m.native.load()
    return m


@pytest.fixture(params=["direct", "partition"])
def count_path(request, monkeypatch):
    """Both counting kernels must give the same table: the direct one (an atomic per occurrence) and the
    partitioned pipeline (scatter by hash prefix, merge regions in LDS).  Read at mc_create."""
    monkeypatch.setenv("MC_COUNT_PATH", request.param)
    return request.param


def _gpu_table(mc, words, offsets, k, mode, hint=0):
    ctx = mc.Context(k, mode, 0, hint)
    ctx.add_reads_packed(words, offsets)
    n = ctx.finalize()
    return ctx, n


def _assert_tables_equal(ctx, n_distinct, t):
    assert n_distinct == t.size()
    gk, gc = ctx.export(0)
    ok, oc = t.dump()
    assert np.array_equal(gk, ok)
    assert np.array_equal(gc, oc)


@pytest.mark.parametrize("err", [0, 100])
def test_count_config1_k31(mc, err, count_path):
    """BASELINE.json configs[0]: 10k x 150 bp, k=31 -- every (key, count) pair equal."""
    _, reads, off = synth_case(1, 50000, 10000, 150, err)
    t, n = oracle_table(reads, off, 31, po.KEY_PACKED)
    ctx, nd = _gpu_table(mc, po.pack(reads), off, 31, mc.KEY_PACKED)
    assert ctx.stats().windows == n == 10000 * 120
    _assert_tables_equal(ctx, nd, t)
    # get(): present, absent, key 0
    gk, _ = ctx.export(0)
    rng = np.random.default_rng(1)
    q = np.concatenate([gk[:1000], rng.integers(1, 1 << 61, 1000), [0]]).astype(np.int64)
    assert np.array_equal(ctx.get(q), t.get_many(q))
    ctx.close()


@pytest.mark.parametrize("k,mode", [(31, 0), (23, 0), (27, 0), (30, 0), (21, 0), (5, 0), (1, 0), (63, 1), (33, 1), (47, 2), (31, 1), (64 - 1, 2)])
def test_count_ragged_reads_all_key_modes(mc, k, mode, count_path):
    """Empty reads, reads shorter than k, k-1, k, ragged lengths; packed key, poly and fnv1a hashes."""
    rng = np.random.default_rng(100 + k + mode)
    _, codes, off = ragged_case(rng, 700)
    t, n = oracle_table(codes, off, k, mode)
    ctx, nd = _gpu_table(mc, po.pack(codes), off, k, mode)
    assert ctx.stats().windows == n
    _assert_tables_equal(ctx, nd, t)
    ctx.close()


def test_count_saturates_at_32767_and_key_zero(mc, count_path):
    """poly-A / poly-T reads: key 0 (the reference's FREE marker) with > 32767 occurrences."""
    L, n = 150, 400
    codes = np.zeros(n * L, dtype=np.uint8)
    codes[(n // 2) * L:] = 3  # second half poly-T: same canonical k-mer
    off = np.arange(n + 1, dtype=np.uint64) * L
    t, _ = oracle_table(codes, off, 31, po.KEY_PACKED)
    ctx, nd = _gpu_table(mc, po.pack(codes), off, 31, mc.KEY_PACKED)
    assert nd == 1 and t.get(0) == 32767
    assert list(ctx.get(np.array([0, 1], dtype=np.int64))) == [32767, -1]
    _assert_tables_equal(ctx, nd, t)
    ctx.close()


def test_add_reads_file(mc, tmp_path):
    """mc_add_reads_file: FASTQ.gz through the library's own reader == the oracle reader + oracle table."""
    import gzip
    from oracle import host_oracle as ho
    _, reads, _ = synth_case(1, 20000, 3000, 100, 100)
    fq = tmp_path / "x.fq.gz"
    lines = []
    for i in range(3000):
        s = po.decode(reads[i * 100:(i + 1) * 100])
        q = ["I"] * 100
        if i % 5 == 0:
            q[37] = "!"
        lines.append("@r%d\n%s\n+\n%s\n" % (i, s, "".join(q)))
    fq.write_bytes(gzip.compress("".join(lines).encode()))
    want_reads = ho.read_fastq_reads(str(fq))
    codes = np.concatenate([po.encode(r) for r in want_reads])
    off = np.zeros(len(want_reads) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in want_reads])
    t, _ = oracle_table(codes, off, 27, po.KEY_PACKED)
    ctx = mc.Context(27, mc.KEY_PACKED, 0, 0)
    assert ctx.add_reads_file(str(fq)) == len(want_reads) == 3600
    _assert_tables_equal(ctx, ctx.finalize(), t)
    with pytest.raises(mc.native.McError, match="Can't detect file format"):
        ctx.add_reads_file(str(tmp_path / "reads.txt"))
    with pytest.raises(mc.native.McError, match="Failed to read from file"):
        ctx.add_reads_file(str(tmp_path / "missing.fasta"))
    ctx.close()


def test_empty_input_and_state_errors(mc):
    ctx = mc.Context(31, mc.KEY_PACKED, 0, 0)
    with pytest.raises(mc.McError):
        ctx.get(np.array([1], dtype=np.int64))  # before finalize
    ctx.add_reads_packed(np.zeros(1, dtype=np.uint64), np.zeros(1, dtype=np.uint64))
    assert ctx.finalize() == 0
    assert list(ctx.get(np.array([0, 5], dtype=np.int64))) == [-1, -1]
    assert ctx.bfs(None, np.array([5], dtype=np.uint64), -1, 1, 10, -1) is None
    with pytest.raises(mc.McError):
        mc.Context(32, mc.KEY_PACKED)
    with pytest.raises(mc.McError):
        mc.Context(64, mc.KEY_POLY)
    ctx.close()


def test_table_grows_from_small(mc, count_path):
    """No capacity hint: the table starts at 4 M slots and is rebuilt as it fills."""
    _, reads, off = synth_case(4, 2_000_000, 120_000, 150, 100)
    t, n = oracle_table(reads, off, 31, po.KEY_PACKED)
    ctx, nd = _gpu_table(mc, po.pack(reads), off, 31, mc.KEY_PACKED, 0)
    assert ctx.stats().grows >= 1
    assert (ctx.stats().p3_ms > 0) == (count_path == "partition")
    _assert_tables_equal(ctx, nd, t)
    # batches in a different order and split over several calls give the same table
    ctx2 = mc.Context(31, mc.KEY_PACKED, 0, t.size())
    words = po.pack(reads)
    half = 60_000
    w2 = po.pack(reads[half * 150:])
    ctx2.add_reads_packed(w2, off[: 120_000 - half + 1])
    ctx2.add_reads_packed(words, off[: half + 1])
    assert ctx2.stats().grows == 0
    _assert_tables_equal(ctx2, ctx2.finalize(), t)
    ctx.close()
    ctx2.close()


def _bfs_both(mc, ctx, t, k, mode, seed_codes, d, cov, mk, mr):
    hi, lo = seed_windows(seed_codes, k)
    got = ctx.bfs(hi, lo, d, cov, mk, mr)
    want = po.bfs(t, k, mode, [seed_codes], d, cov, mk, mr)
    assert_bfs_equal(got, want)
    return got


@pytest.fixture(scope="module")
def bfs_case(mc):
    genome, reads, off = synth_case(2, 30000, 12000, 150, 50)
    t, _ = oracle_table(reads, off, 31, po.KEY_PACKED)
    ctx, _ = _gpu_table(mc, po.pack(reads), off, 31, mc.KEY_PACKED)
    yield genome, t, ctx
    ctx.close()


@pytest.mark.parametrize("d", [-1, 1, 0])
@pytest.mark.parametrize("mk,mr", [(100000, -1), (777, -1), (-1, 50), (1500, 300), (3, -1)])
def test_bfs_parity(mc, bfs_case, d, mk, mr):
    """Discovery order, distances, coverages and lastKmers flags equal, incl. maxkmers biting mid-level."""
    genome, t, ctx = bfs_case
    seed = genome[10000:10500]
    got = _bfs_both(mc, ctx, t, 31, po.KEY_PACKED, seed, d, 5, mk, mr)
    assert got is not None and len(got["lo"]) >= 1


def test_bfs_branching_graph_and_duplicate_seeds(mc):
    """A repeat-rich genome (many branches, wide frontiers, cycles) + a seed whose windows repeat."""
    rng = np.random.default_rng(5)
    unit = rng.integers(0, 4, 400).astype(np.uint8)
    parts = []
    for i in range(60):
        u = unit.copy()
        pos = rng.integers(0, 400, 6)
        u[pos] = rng.integers(0, 4, 6)
        parts.append(u)
    genome = np.concatenate(parts)
    L, n = 100, 6000
    starts = rng.integers(0, len(genome) - L, n)
    reads = np.concatenate([genome[s:s + L] for s in starts])
    off = np.arange(n + 1, dtype=np.uint64) * L
    for k, mode in [(21, po.KEY_PACKED), (31, po.KEY_PACKED), (33, po.KEY_POLY)]:
        t, _ = oracle_table(reads, off, k, mode)
        ctx, nd = _gpu_table(mc, po.pack(reads), off, k, mode)
        _assert_tables_equal(ctx, nd, t)
        seed = np.concatenate([genome[100:200], genome[100:200], genome[500:560]])  # repeated windows
        for d in (-1, 1, 0):
            for mk, mr in [(-1, 40), (2000, -1), (5000, 25), (100000, -1)]:
                _bfs_both(mc, ctx, t, k, mode, seed, d, 3, mk, mr)
        ctx.close()


def test_bfs_k63_hash_key(mc):
    """BASELINE.json configs[2] at small scale: k=63, poly hash key, coverage 3, bothdirs."""
    genome, reads, off = synth_case(1, 40000, 9000, 150, 30)
    t, _ = oracle_table(reads, off, 63, po.KEY_POLY)
    ctx, nd = _gpu_table(mc, po.pack(reads), off, 63, mc.KEY_POLY)
    _assert_tables_equal(ctx, nd, t)
    for d, mk, mr in [(0, 100000, -1), (0, 5000, -1), (-1, -1, 1000), (1, 300, 100)]:
        _bfs_both(mc, ctx, t, 63, po.KEY_POLY, genome[20000:20400], d, 3, mk, mr)
    ctx.close()


def test_bfs_batch_equals_single_passes(mc, bfs_case):
    """Both passes of --bothdirs False (and a second seed) in one launch == one pass at a time."""
    genome, t, ctx = bfs_case
    rng = np.random.default_rng(2)
    seeds = [genome[10000:10500], genome[40000:40100], rng.integers(0, 4, 100).astype(np.uint8)]
    jobs, want = [], []
    for s in seeds:
        hi, lo = seed_windows(s, 31)
        for d in (-1, 1, 0):
            jobs.append((hi, lo, d))
            want.append(po.bfs(t, 31, po.KEY_PACKED, [s], d, 5, 3000, 700))
    got = ctx.bfs_batch(jobs, 5, 3000, 700)
    assert got[-1] is None and want[-1] is None
    for g, w in zip(got, want):
        assert_bfs_equal(g, w)


def test_coverage_hint_tracks_solid_count_over_batches(mc, monkeypatch):
    """mc_set_coverage_hint: the merge kernel keeps #(count >= min_cov) current over several batches (the
    second batch lifts keys over the threshold that the first one left below it); the BFS set-up then
    needs no counting sweep, and the walk is the same as without the hint.  A different threshold, or an
    addition through another kernel, falls back to the sweep."""
    monkeypatch.setenv("MC_COUNT_PATH", "partition")
    monkeypatch.setenv("MC_BFS_DIRECT", "0")  # (the walk on a copy of the solid k-mers; by default it reads the counting table itself)
    genome, reads, off = synth_case(2, 30000, 12000, 150, 50)
    t, _ = oracle_table(reads, off, 31, po.KEY_PACKED)
    half = 6000
    seed = genome[10000:10500]
    hi, lo = seed_windows(seed, 31)
    want = po.bfs(t, 31, po.KEY_PACKED, [seed], 1, 5, 3000, -1)
    ctx = mc.Context(31, mc.KEY_PACKED, 0, 0)
    ctx.set_coverage_hint(5)
    for rep in range(2):  # the second round checks mc_clear keeps the hint working
        ctx.add_reads_packed(po.pack(reads[:off[half]]), off[:half + 1])
        ctx.add_reads_packed(po.pack(reads[off[half]:]), off[half:] - off[half])
        assert ctx.finalize() == t.size()
        ctx.reset_stats()
        assert_bfs_equal(ctx.bfs(hi, lo, 1, 5, 3000, -1), want)
        st = ctx.stats()
        assert st.solid_sweeps == 0
        assert st.solid_kmers == ctx.export_count(5) == int((t.dump()[1] >= 5).sum())
        assert_bfs_equal(ctx.bfs(hi, lo, 1, 3, 3000, -1), po.bfs(t, 31, po.KEY_PACKED, [seed], 1, 3, 3000, -1))
        st = ctx.stats()
        assert st.solid_sweeps == 1 and st.solid_kmers == int((t.dump()[1] >= 3).sum())
        ctx.clear()
    ctx.close()


def test_compact_records_of_the_counting_pipeline(mc, monkeypatch, capfd):
    """Reads into a table whose size is vouched for (capacity hint) travel through the two scatter levels as 16-byte units
    that carry the leaf number and, in place of a pointer array, the RELATIVE position of their first window
    (count_pipeline.h k_sk1w_extract<false, true>, k_sk2_scatter_compact).  Same (key, count) pairs as the oracle and as the
    two-array form, over two batches (the second one starts in the middle of a tile of the read store), and the read
    pointers made the trip: the walk needs few round trips."""
    monkeypatch.delenv("MC_COUNT_PATH", raising=False)
    monkeypatch.setenv("MC_INGEST_DEBUG", "1")
    genome, reads, off = synth_case(2, 200000, 80000, 150, 50)
    t, _ = oracle_table(reads, off, 31, po.KEY_PACKED)
    ok, oc = t.dump()
    seed = genome[10000:10500]
    hi, lo = seed_windows(seed, 31)
    want = po.bfs(t, 31, po.KEY_PACKED, [seed], 1, 5, 3000, -1)
    cut = 40001
    for compact in ("1", "0"):
        monkeypatch.setenv("MC_SK_COMPACT", compact)
        capfd.readouterr()
        ctx = mc.Context(31, mc.KEY_PACKED, 0, 3_000_000)  # ~3000 regions: two scatter levels
        ctx.set_coverage_hint(5)
        ctx.add_reads_packed(po.pack(reads[:off[cut]]), off[:cut + 1])
        ctx.add_reads_packed(po.pack(reads[off[cut]:]), off[cut:] - off[cut])
        assert ctx.finalize() == t.size()
        err = capfd.readouterr().err
        assert (err.count("[count] compact records") == 2) == (compact == "1"), err
        gk, gc = ctx.export(1)
        assert np.array_equal(gk, ok) and np.array_equal(gc, oc)
        got = ctx.bfs(hi, lo, 1, 5, 3000, -1)
        assert_bfs_equal(got, want)
        assert got["rounds"] * 4 < got["levels"]
        ctx.close()


@pytest.mark.parametrize("mode", ["vleaf", "two_arrays", "beyond_reach"])
def test_first_batch_without_a_hint_sizes_its_table_behind_compact_records(mc, monkeypatch, capfd, mode):
    """No capacity hint, an empty table: the first level writes compact records whose leaf field is the ten bits of the bin word below the
    level-1 bucket's (1024 leaves a bucket, whatever the table), the sample of the first bucket then makes a table of a power of two
    of regions and the second level keeps the bits that table needs (mcgpu.hip PipePlan::vleaf).  Same pairs, same walk as the oracle
    and as the two-array form (MC_SK_VLEAF=0); and where the wanted table lies beyond the reach of the ten bits
    (MC_SK_VLEAF_MAX_REGIONS plays that at a small size) the first level runs again the two-array way."""
    monkeypatch.delenv("MC_COUNT_PATH", raising=False)
    monkeypatch.setenv("MC_INGEST_DEBUG", "1")
    if mode == "two_arrays":
        monkeypatch.setenv("MC_SK_VLEAF", "0")
    if mode == "beyond_reach":
        monkeypatch.setenv("MC_SK_VLEAF_MAX_REGIONS", "512")
    genome, reads, off = synth_case(4, 500000, 320000, 150, 100)  # (38 M windows: enough records for the sample's scratch set)
    t, _ = oracle_table(reads, off, 31, po.KEY_PACKED)
    ok, oc = t.dump()
    seed = genome[10000:10500]
    hi, lo = seed_windows(seed, 31)
    ctx = mc.Context(31, mc.KEY_PACKED, 0, 0)
    ctx.set_coverage_hint(5)
    capfd.readouterr()
    ctx.add_reads_packed(po.pack(reads), off)
    assert ctx.finalize() == t.size()
    err = capfd.readouterr().err
    n_v = err.count("(whatever the table: it is sized behind the first level)")
    assert n_v == (0 if mode == "two_arrays" else 1), err
    st = ctx.stats()
    regions = st.table_bytes // (4096 * 16)
    if mode == "vleaf":
        assert regions & (regions - 1) == 0 and 0.2 < len(ok) / (regions * 4096) < 0.5, (regions, len(ok))  # (a power of two, load 0.23 .. 0.45 if the sample is right)
    assert ("the first level again, two arrays" in err) == (mode == "beyond_reach"), err
    gk, gc = ctx.export(1)
    assert np.array_equal(gk, ok) and np.array_equal(gc, oc)
    got = ctx.bfs(hi, lo, 1, 5, 3000, -1)
    assert_bfs_equal(got, po.bfs(t, 31, po.KEY_PACKED, [seed], 1, 5, 3000, -1))
    assert got["rounds"] * 4 < got["levels"]  # (the read pointers made the trip)
    # a second batch into the same table (no longer empty: the ordinary plan)
    ctx.add_reads_packed(po.pack(reads[:off[30000]]), off[:30001])
    t.count_reads(reads[:off[30000]], off[:30001], 31, po.KEY_PACKED)
    ok, oc = t.dump()
    assert ctx.finalize() == t.size()
    gk, gc = ctx.export(1)
    assert np.array_equal(gk, ok) and np.array_equal(gc, oc)
    ctx.close()


def test_capacity_hint_a_quarter_of_what_the_reads_hold(mc, monkeypatch):
    """A capacity hint vouches for the table's size, so the pipeline merges into it without sampling -- and finds most regions
    full: leaves hand occurrences on to the next regions, the direct kernel that drains that list parks what finds no room in
    the whole chain, the table doubles (several times) under the run.  Every (key, count) pair must still be there (the drain
    once read and appended to the same list in one launch and wiped what it had parked itself: scripts/soak.py found 15 000
    of 21 M keys missing)."""
    monkeypatch.delenv("MC_COUNT_PATH", raising=False)
    genome, reads, off = synth_case(3, 150000, 100000, 250, 200)
    t, _ = oracle_table(reads, off, 29, po.KEY_PACKED)
    ok, oc = t.dump()
    assert t.size() > 8_000_000
    ctx = mc.Context(29, mc.KEY_PACKED, 0, 2_400_000)
    half = 50000
    ctx.add_reads_packed(po.pack(reads[:off[half]]), off[:half + 1])
    ctx.add_reads_packed(po.pack(reads[off[half]:]), off[half:] - off[half])
    assert ctx.finalize() == t.size()
    gk, gc = ctx.export(0)
    assert np.array_equal(gk, ok) and np.array_equal(gc, oc)
    st = ctx.stats()
    assert st.spill_keys > 0 and st.grows > 0
    # ... and every key must be FOUND where the probing rule looks (mc_get probes; the export above sweeps): round 5's merge
    # kernel once placed keys two slots beyond the TABLE_MAX_PROBES a look-up examines
    assert np.array_equal(ctx.get(ok), oc)
    ctx.close()


def test_every_key_of_a_crowded_table_is_where_lookups_look(mc, monkeypatch):
    """scripts/soak.py, seed 51, iteration 0 (round 5): k = 27, 5 % errors, 250-base reads at 600-fold depth, a capacity hint a
    third of what the reads hold, coverage 2.  The regions are crowded enough for probe sequences of 128 slots; the merge kernel's
    pair loop ran its two unrolled steps OUTSIDE the count of TABLE_MAX_PROBES, so a key could land at slots 129-130 from its home:
    the export (a sweep) showed it, a look-up did not, and a walk of a million vertices missed one.  Look-ups of every key, and
    the walk that went wrong, against the oracle."""
    monkeypatch.setenv("MC_COUNT_PATH", "partition")
    k, cov = 27, 2
    genome = po.synth_genome(GENOME_SEED, 2 * 20000)
    reads = po.synth_reads(genome, 2, 20000, 844180353, 0, 97570, 250, 500)
    off = np.arange(97570 + 1, dtype=np.uint64) * 250
    t, _ = oracle_table(reads, off, k, po.KEY_PACKED)
    ok, oc = t.dump()
    ctx = mc.Context(k, mc.KEY_PACKED, 0, 6_000_000)
    ctx.add_reads_packed(po.pack(reads), off)
    assert ctx.finalize() == t.size()
    gk, gc = ctx.export(0)
    assert np.array_equal(gk, ok) and np.array_equal(gc, oc)
    assert np.array_equal(ctx.get(ok), oc)
    seed = genome[1000:1400]
    hi, lo = seed_windows(seed, k)
    for d in (1, 0):
        assert_bfs_equal(ctx.bfs(hi, lo, d, cov, 300000, -1), po.bfs(t, k, po.KEY_PACKED, [seed], d, cov, 300000, -1))
    ctx.close()


@pytest.mark.parametrize("k,mode", [(21, "packed"), (41, "poly")])
def test_solid_copy_of_a_crowded_hash_prefix_table_holds_the_keys_that_moved_on(mc, monkeypatch, k, mode):
    """tables whose regions are hash prefixes (k < 23, hash keys) hand an addition that finds its stretch full on to
    the next region (kmer_device.h TABLE_CHAIN).  The builder of the walk's own copy of the solid k-mers (MC_BFS_DIRECT=0) scanned
    only the counting regions a solid region's keys are AT HOME in and lost the ones that had moved on: the walk then took a
    solid k-mer for absent.  A capacity hint far too small crowds the table; walks in all directions against the oracle."""
    monkeypatch.delenv("MC_COUNT_PATH", raising=False)
    monkeypatch.setenv("MC_BFS_DIRECT", "0")
    omode, gmode = (po.KEY_PACKED, mc.KEY_PACKED) if mode == "packed" else (po.KEY_POLY, mc.KEY_POLY)
    genome, reads, off = synth_case(3, 150000, 100000, 250, 200)
    t, _ = oracle_table(reads, off, k, omode)
    ctx = mc.Context(k, gmode, 0, t.size() // 4)
    half = 50000
    ctx.add_reads_packed(po.pack(reads[:off[half]]), off[:half + 1])
    ctx.add_reads_packed(po.pack(reads[off[half]:]), off[half:] - off[half])
    assert ctx.finalize() == t.size()
    assert ctx.stats().spill_keys > 0  # (occurrences were handed on / parked)
    for a in (20000, 170000, 400000):
        seed = genome[a:a + 300]
        hi, lo = seed_windows(seed, k)
        for d in (1, -1, 0):
            assert_bfs_equal(ctx.bfs(hi, lo, d, 3, 8000, -1), po.bfs(t, k, omode, [seed], d, 3, 8000, -1))
    ctx.close()


def test_trim_gives_the_scratch_back_and_the_context_carries_on(mc, monkeypatch):
    """mc_trim: the pipeline's scratch and the pools' idle blocks go back to the driver; the table, the read store and a valid
    list of solid k-mers stay, so walks before and after are the same, and the next batch allocates its scratch again."""
    import torch
    monkeypatch.delenv("MC_COUNT_PATH", raising=False)
    monkeypatch.setenv("MC_BFS_DIRECT", "0")  # (the walk's table is built from the merge kernel's list: it must survive the trim)
    genome, reads, off = synth_case(2, 200000, 80000, 150, 50)
    t, _ = oracle_table(reads, off, 31, po.KEY_PACKED)
    half = 40000
    t1, _ = oracle_table(reads[:off[half]], off[:half + 1], 31, po.KEY_PACKED)
    seed = genome[10000:10500]
    hi, lo = seed_windows(seed, 31)
    ctx = mc.Context(31, mc.KEY_PACKED, 0, 3_000_000)
    ctx.set_coverage_hint(5)
    ctx.add_reads_packed(po.pack(reads[:off[half]]), off[:half + 1])
    ctx.finalize()
    free0 = torch.cuda.mem_get_info()[0]
    ctx.trim()
    assert torch.cuda.mem_get_info()[0] > free0  # something came back
    assert_bfs_equal(ctx.bfs(hi, lo, 1, 5, 3000, -1), po.bfs(t1, 31, po.KEY_PACKED, [seed], 1, 5, 3000, -1))
    assert ctx.stats().solid_list_builds == 1  # the list survived
    ctx.add_reads_packed(po.pack(reads[off[half]:]), off[half:] - off[half])
    assert ctx.finalize() == t.size()
    ctx.trim()
    gk, gc = ctx.export(1)
    ok, oc = t.dump()
    assert np.array_equal(gk, ok) and np.array_equal(gc, oc)
    assert_bfs_equal(ctx.bfs(hi, lo, -1, 5, 3000, -1), po.bfs(t, 31, po.KEY_PACKED, [seed], -1, 5, 3000, -1))
    ctx.close()


def test_solid_list_from_the_merge_kernel(mc, monkeypatch):
    """With the threshold known while counting, the merge kernel lists the keys at or above it as it writes each
    region back (tables of more than 512 regions), and the BFS set-up builds its table from that list instead of
    sweeping the counting table: same walks; the list is rebuilt by every batch, dropped when another kernel adds
    keys, and not used for another threshold."""
    monkeypatch.delenv("MC_COUNT_PATH", raising=False)  # batches of 4.8 M windows take the pipeline by themselves
    monkeypatch.setenv("MC_BFS_DIRECT", "0")  # (the BFS table as a copy built from the list; by default the walk reads the counting table itself)
    genome, reads, off = synth_case(2, 200000, 80000, 150, 50)  # > 2^18 solid k-mers: a two-level build of the BFS table
    t, _ = oracle_table(reads, off, 31, po.KEY_PACKED)
    half = 40000
    seed = genome[10000:10500]
    hi, lo = seed_windows(seed, 31)
    ctx = mc.Context(31, mc.KEY_PACKED, 0, 3_000_000)  # ~3000 regions
    ctx.set_coverage_hint(5)
    ctx.add_reads_packed(po.pack(reads[:off[half]]), off[:half + 1])
    ctx.finalize()
    t1, _ = oracle_table(reads[:off[half]], off[:half + 1], 31, po.KEY_PACKED)
    for d in (-1, 1):
        assert_bfs_equal(ctx.bfs(hi, lo, d, 5, 3000, -1), po.bfs(t1, 31, po.KEY_PACKED, [seed], d, 5, 3000, -1))
    st = ctx.stats()
    assert st.solid_list_builds == 1 and st.solid_sweeps == 0  # (the second walk reuses the table)
    ctx.add_reads_packed(po.pack(reads[off[half]:]), off[half:] - off[half])  # every region is rewritten: a new list
    assert ctx.finalize() == t.size()
    ok, oc = t.dump()
    gk, gc = ctx.export(5)  # the multi-GPU gather's export reads the list too (and leaves it in place)
    assert np.array_equal(gk, ok[oc >= 5]) and np.array_equal(gc, oc[oc >= 5])
    assert ctx.stats().solid_list_builds == 2
    want = po.bfs(t, 31, po.KEY_PACKED, [seed], 1, 5, 3000, -1)
    assert_bfs_equal(ctx.bfs(hi, lo, 1, 5, 3000, -1), want)
    st = ctx.stats()
    assert st.solid_list_builds == 3 and st.solid_sweeps == 0 and st.solid_kmers == int((t.dump()[1] >= 5).sum())
    assert_bfs_equal(ctx.bfs(hi, lo, 1, 3, 3000, -1), po.bfs(t, 31, po.KEY_PACKED, [seed], 1, 3, 3000, -1))
    assert ctx.stats().solid_list_builds == 3  # another threshold: swept
    assert_bfs_equal(ctx.bfs(hi, lo, 1, 5, 3000, -1), want)  # the list was consumed by the first build: swept too
    assert ctx.stats().solid_list_builds == 3
    # keys added by another kernel (a small batch goes through the direct path) end the list's validity
    extra = reads[:off[10]]
    ctx.add_reads_packed(po.pack(extra), off[:11])
    ctx.finalize()
    t.count_reads(extra, off[:11], 31, po.KEY_PACKED)
    assert_bfs_equal(ctx.bfs(hi, lo, 1, 5, 3000, -1), po.bfs(t, 31, po.KEY_PACKED, [seed], 1, 5, 3000, -1))
    assert ctx.stats().solid_list_builds == 3
    ctx.close()


def test_deep_coverage_of_a_small_genome(mc, monkeypatch):
    """450-fold coverage of a 69 kb genome with 2 % errors (found by scripts/soak.py): the error variants of a locus
    share its minimizer bin, so the table's regions and the pipeline's leaves fill very unevenly.  Records that do
    not fit their leaf go through the direct kernel, additions that find their region full are parked until the
    table has been enlarged (mc_finalize_counts) -- nothing is lost, nothing fails."""
    monkeypatch.setenv("MC_COUNT_PATH", "partition")
    k, L, n_reads, clen, rseed = 30, 150, 205367, 69072, 947706781
    genome = po.synth_genome(GENOME_SEED + 36, clen)
    reads = po.synth_reads(genome, 1, clen, rseed, 0, n_reads, L, 200)
    off = np.arange(n_reads + 1, dtype=np.uint64) * L
    t, _ = oracle_table(reads, off, k, po.KEY_PACKED)
    for batches in (1, 2):
        ctx = mc.Context(k, mc.KEY_PACKED, 0, 6_000_000)  # (12 M distinct k-mers come)
        ctx.set_coverage_hint(6)
        if batches == 1:
            ctx.add_reads_packed(po.pack(reads), off)
        else:
            h = n_reads // 2
            ctx.add_reads_packed(po.pack(reads[:off[h]]), off[:h + 1])
            ctx.add_reads_packed(po.pack(reads[off[h]:]), off[h:] - off[h])
        _assert_tables_equal(ctx, ctx.finalize(), t)
        seed = genome[30000:30400]
        hi, lo = seed_windows(seed, k)
        for d in (1, -1):
            assert_bfs_equal(ctx.bfs(hi, lo, d, 6, 20000, -1), po.bfs(t, k, po.KEY_PACKED, [seed], d, 6, 20000, -1))
        ctx.close()


@pytest.mark.parametrize("k,err,L,n_reads,contigs,clen,cap,rseed", [
    (31, 500, 250, 180342, 2, 3000, 6_000_000, 52959798),  # ~47 000 distinct k-mers per minimizer: no region holds that
    (21, 100, 150, 242878, 3, 3000, 0, 171492290),         # 9 000 k-mers make up most of 32 M occurrences
])
def test_thousandfold_coverage(mc, monkeypatch, k, err, L, n_reads, contigs, clen, cap, rseed):
    """Read sets that cover a few kilobases thousands of times (scripts/soak.py found both).  Minimizer-bin regions
    cannot hold what shares one minimizer, however many regions there are: the context falls back to regions by the
    key's own hash, for good, and what the merge kernel had not merged goes through the direct kernel.  With hash-prefix
    regions a few heavy keys overflow their buckets' spill lists: the batch is counted by the direct kernel instead."""
    monkeypatch.setenv("MC_COUNT_PATH", "partition")
    genome = po.synth_genome(GENOME_SEED + 1, contigs * clen)
    reads = po.synth_reads(genome, contigs, clen, rseed, 0, n_reads, L, err)
    off = np.arange(n_reads + 1, dtype=np.uint64) * L
    t, _ = oracle_table(reads, off, k, po.KEY_PACKED)
    ctx = mc.Context(k, mc.KEY_PACKED, 0, cap)
    ctx.add_reads_packed(po.pack(reads), off)
    _assert_tables_equal(ctx, ctx.finalize(), t)
    h = n_reads // 3
    ctx.add_reads_packed(po.pack(reads[:off[h]]), off[:h + 1])  # and it keeps counting afterwards
    t.count_reads(reads[:off[h]], off[:h + 1], k, po.KEY_PACKED)
    _assert_tables_equal(ctx, ctx.finalize(), t)
    seed = genome[1000:1400]
    hi, lo = seed_windows(seed, k)
    assert_bfs_equal(ctx.bfs(hi, lo, 1, 5, 20000, -1), po.bfs(t, k, po.KEY_PACKED, [seed], 1, 5, 20000, -1))
    if k >= 23:
        # a rank that gave up minimizer bins still takes part in the multi-GPU exchange: it cuts its reads into
        # super-k-mer records like the others and counts the records it receives (directly)
        import torch
        dev = torch.device("cuda:0")
        m = 5000
        d_words = torch.from_numpy(po.pack(reads[:off[m]]).view(np.int64)).to(dev)
        d_off = torch.from_numpy(off[:m + 1].view(np.int64)).to(dev)
        cap2 = ctx.superkmer_capacity(int(off[m]), m)
        assert cap2 > 0
        recs = torch.zeros((cap2, 2), dtype=torch.int64, device=dev)
        bins = torch.zeros(cap2, dtype=torch.int32, device=dev)
        ooff = ctx.extract_superkmers_dev(d_words, d_off, m, int(off[m]), 1, recs, bins, cap2)
        ctx.add_superkmers_dev(recs, bins, int(ooff[1]))
        t.count_reads(reads[:off[m]], off[:m + 1], k, po.KEY_PACKED)
        _assert_tables_equal(ctx, ctx.finalize(), t)
    ctx.close()


def test_bfs_no_seed_passes(mc, bfs_case):
    _, t, ctx = bfs_case
    rng = np.random.default_rng(9)
    seed = rng.integers(0, 4, 200).astype(np.uint8)  # random: not in the genome
    hi, lo = seed_windows(seed, 31)
    assert ctx.bfs(hi, lo, -1, 5, 1000, -1) is None
    assert po.bfs(t, 31, po.KEY_PACKED, [seed], -1, 5, 1000, -1) is None


def test_bfs_unbounded_radius_grows_buffers(mc):
    """Only --maxradius: distanceToKmer is not bounded up front and has to grow on the device."""
    genome, reads, off = synth_case(1, 1_200_000, 200_000, 150, 0)
    t, _ = oracle_table(reads, off, 31, po.KEY_PACKED)
    ctx, _ = _gpu_table(mc, po.pack(reads), off, 31, mc.KEY_PACKED, t.size())
    got = _bfs_both(mc, ctx, t, 31, po.KEY_PACKED, genome[600000:600200], 0, 1, -1, 10_000_000)
    assert len(got["lo"]) > (1 << 20)
    ctx.close()


def test_device_pointer_path_and_generator(mc, count_path):
    """Reads generated straight into HBM == the oracle's generator; counting from device pointers."""
    import torch
    n_contigs, contig_len, n_reads, L = 3, 20000, 5000, 150
    dev = torch.device("cuda:0")
    for err in (0, 100):
        ctx = mc.Context(31, mc.KEY_PACKED, 0, 0)
        n_words = (n_reads * L + 31) // 32 + 1
        d_words = torch.zeros(n_words, dtype=torch.int64, device=dev)
        d_off = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
        ctx.synth_reads_dev(GENOME_SEED, n_contigs, contig_len, READ_SEED, 7, n_reads, L, err, d_words, d_off)
        genome = po.synth_genome(GENOME_SEED, n_contigs * contig_len)
        assert np.array_equal(genome, mc.native.synth_genome(GENOME_SEED, 0, n_contigs * contig_len))
        reads = po.synth_reads(genome, n_contigs, contig_len, READ_SEED, 7, n_reads, L, err)
        want_words = po.pack(reads)
        assert np.array_equal(d_words.cpu().numpy().view(np.uint64)[:-1], want_words[:n_words - 1])
        assert np.array_equal(d_off.cpu().numpy().view(np.uint64), np.arange(n_reads + 1, dtype=np.uint64) * L)
        ctx.add_reads_packed_dev(d_words, d_off, n_reads, n_reads * L)
        off = np.arange(n_reads + 1, dtype=np.uint64) * L
        t, _ = oracle_table(reads, off, 31, po.KEY_PACKED)
        _assert_tables_equal(ctx, ctx.finalize(), t)
        # get_dev
        gk, gc = ctx.export(0)
        dk = torch.from_numpy(gk).to(dev)
        dout = torch.zeros(len(gk), dtype=torch.int16, device=dev)
        ctx.get_dev(dk, len(gk), dout)
        assert np.array_equal(dout.cpu().numpy(), gc)
        ctx.close()


@pytest.mark.parametrize("count_path2", ["direct", "partition"])
def test_extract_keys_by_owner_and_merge_pairs(mc, count_path2, monkeypatch):
    """Multi-GPU building blocks on one GPU: keys (+ hints) bucketed by owner, counted per owner, gathered
    as (key, count, hint) triples and merged == counting everything in one table; the merged table still
    walks with long look-ahead (the hints survived the trip)."""
    import torch
    monkeypatch.setenv("MC_COUNT_PATH", count_path2)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    _, codes, off = ragged_case(rng, 900)
    for k, mode in [(31, po.KEY_PACKED), (41, po.KEY_POLY)]:
        t, n = oracle_table(codes, off, k, mode)
        words = po.pack(codes)
        d_words = torch.from_numpy(words.view(np.int64)).to(dev)
        d_off = torch.from_numpy(off.view(np.int64)).to(dev)
        G = 3
        ex = mc.Context(k, mode, 0, 0)
        d_keys = torch.zeros(n, dtype=torch.int64, device=dev)
        d_hints = torch.zeros(n, dtype=torch.int32, device=dev)
        ooff = ex.extract_keys_dev(d_words, d_off, len(off) - 1, int(off[-1]), G, d_keys, n, d_hints)
        assert int(ooff[-1]) == n
        only_offsets = ex.extract_keys_dev(d_words, d_off, len(off) - 1, int(off[-1]), G, None, 0)
        assert np.array_equal(only_offsets, ooff)
        keys = d_keys.cpu().numpy()
        for o in range(G):
            seg = keys[int(ooff[o]):int(ooff[o + 1])]
            assert all(mc.native.key_owner(int(x), G) == o for x in seg[:200])
        merged = mc.Context(k, mode, 0, 0)
        for o in range(G):
            own = mc.Context(k, mode, 0, 0)
            a, b = int(ooff[o]), int(ooff[o + 1])
            own.add_keys_dev(d_keys[a:b], b - a, d_hints[a:b])
            nd = own.finalize()
            pk = torch.zeros(nd, dtype=torch.int64, device=dev)
            pc = torch.zeros(nd, dtype=torch.int16, device=dev)
            ph = torch.zeros(nd, dtype=torch.int32, device=dev)
            assert own.export_dev(0, pk, pc, nd, ph) == nd
            merged.add_pairs_dev(pk, pc, nd, ph)
            own.close()
        _assert_tables_equal(merged, merged.finalize(), t)
        # thresholded export
        gk, gc = merged.export(3)
        ok, oc = t.dump()
        assert np.array_equal(gk, ok[oc >= 3]) and np.array_equal(gc, oc[oc >= 3])
        ex.close()
        merged.close()


@pytest.mark.parametrize("k", [31, 23, 28])
def test_superkmer_records_by_owner(mc, k):
    """The compact form of the multi-GPU split: reads -> super-k-mer records bucketed by owner -> every owner
    counts its records; owners hold disjoint key sets whose union is the table of all reads, and the walk over
    the merged solid k-mers still has its read-context hints."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    genome, reads, off = synth_case(2, 40000, 9000, 150, 60)
    _, rcodes, roff = ragged_case(rng, 600)  # short and ragged reads too
    codes = np.concatenate([reads, rcodes])
    offs = np.concatenate([off, roff[1:] + off[-1]])
    t, n = oracle_table(codes, offs, k, po.KEY_PACKED)
    d_words = torch.from_numpy(po.pack(codes).view(np.int64)).to(dev)
    d_off = torch.from_numpy(offs.view(np.int64)).to(dev)
    n_reads = len(offs) - 1
    ex = mc.Context(k, mc.KEY_PACKED, 0, 0)
    assert mc.Context(21, mc.KEY_PACKED, 0, 0).superkmer_capacity(1000, 10) == 0  # short k-mers: key form only
    cap = ex.superkmer_capacity(n, n_reads)
    assert 0 < cap < n
    G = 3
    recs = torch.zeros((cap, 2), dtype=torch.int64, device=dev)
    bins = torch.zeros(cap, dtype=torch.int32, device=dev)
    ooff = ex.extract_superkmers_dev(d_words, d_off, n_reads, int(offs[-1]), G, recs, bins, cap)
    n_rec = int(ooff[-1])
    assert n_rec * 4 < n  # many windows per record
    seen = []
    merged = mc.Context(k, mc.KEY_PACKED, 0, 0)
    for o in range(G):
        own = mc.Context(k, mc.KEY_PACKED, 0, 0)
        a, b = int(ooff[o]), int(ooff[o + 1])
        own.add_superkmers_dev(recs[a:b], bins[a:b], b - a)
        nd = own.finalize()
        pk = torch.zeros(nd, dtype=torch.int64, device=dev)
        pc = torch.zeros(nd, dtype=torch.int16, device=dev)
        ph = torch.zeros(nd, dtype=torch.int32, device=dev)
        assert own.export_dev(0, pk, pc, nd, ph) == nd
        seen.append(pk.cpu().numpy())
        merged.add_pairs_dev(pk, pc, nd, ph)
        own.close()
    allk = np.concatenate(seen)
    assert len(np.unique(allk)) == len(allk) == t.size()  # owners are disjoint
    _assert_tables_equal(merged, merged.finalize(), t)
    merged.share_read_store(ex)  # the read pointers of the records refer to the extracting context's reads
    if k == 31:
        seed = genome[5000:5300]
        hi, lo = seed_windows(seed, k)
        got = merged.bfs(hi, lo, 1, 3, 4000, -1)
        assert_bfs_equal(got, po.bfs(t, k, po.KEY_PACKED, [seed], 1, 3, 4000, -1))
        assert got["rounds"] * 4 < got["levels"]  # long look-ahead: the read pointers made the trip
    ex.close()
    merged.close()


@pytest.mark.parametrize("k,mode_name", [(31, "KEY_PACKED"), (25, "KEY_PACKED"), (21, "KEY_PACKED"), (45, "KEY_POLY"), (31, "KEY_FNV1A")])
def test_bfs_table_straight_from_gathered_pairs(mc, k, mode_name):
    """Rank 0's side of the gather: the exported shards, side by side with padding between them, go straight into a
    BFS-only context; the walk equals the oracle's on the table of all reads and still has its hints."""
    import torch
    dev = torch.device("cuda:0")
    mode, omode = getattr(mc, mode_name), getattr(po, mode_name)
    genome, reads, off = synth_case(2, 30000, 8000, 150, 50)
    t, n = oracle_table(reads, off, k, omode)
    d_words = torch.from_numpy(po.pack(reads).view(np.int64)).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    full = mc.Context(k, mode, 0, 0)
    full.set_coverage_hint(3)
    full.add_reads_packed_dev(d_words, d_off, len(off) - 1, int(off[-1]))
    nd = full.finalize()
    n3 = full.export_count(3)
    ok, oc = t.dump()
    assert n3 == int((oc >= 3).sum())  # (the tracked number when the pipeline counted, a sweep otherwise)
    # two "shards" (below / from the median key), each padded to a common length like all_gather_into_tensor's input
    pk = torch.zeros(nd, dtype=torch.int64, device=dev)
    pc = torch.zeros(nd, dtype=torch.int16, device=dev)
    ph = torch.zeros(nd, dtype=torch.int32, device=dev)
    assert full.export_dev(2, pk, pc, nd, ph) <= nd  # a lower threshold than the walk's: filtered on the way in
    m = full.export_count(2)
    med = pk[:m].median()
    parts = [pk[:m] < med, pk[:m] >= med]
    mx = int(max(int(p.sum()) for p in parts)) + 5
    all_k = torch.zeros(2 * mx, dtype=torch.int64, device=dev)
    all_c = torch.full((2 * mx,), -1, dtype=torch.int16, device=dev)
    all_h = torch.zeros(2 * mx, dtype=torch.int32, device=dev)
    for r, p in enumerate(parts):
        cnt = int(p.sum())
        all_k[r * mx:r * mx + cnt] = pk[:m][p]
        all_c[r * mx:r * mx + cnt] = pc[:m][p]
        all_h[r * mx:r * mx + cnt] = ph[:m][p]
    solid = mc.Context(k, mode, 0, 0)
    solid.share_read_store(full)  # the pairs' read pointers refer to the reads `full` counted
    assert solid.solid_from_pairs_dev(all_k, all_c, 2 * mx, 3, all_h) == n3
    seed = genome[4000:4300]
    hi, lo = seed_windows(seed, k)
    for d in (-1, 0, 1):
        got = solid.bfs(hi, lo, d, 3, 3000, -1)
        assert_bfs_equal(got, po.bfs(t, k, omode, [seed], d, 3, 3000, -1))
        # the read pointers made the trip: long look-ahead in every key mode (both directions at once: two walkers,
        # each with a scout whose hops count as round trips)
        assert got["rounds"] * (2 if d == 0 else 4) < got["levels"]
    with pytest.raises(Exception):
        solid.bfs(hi, lo, 1, 4, 3000, -1)  # built for coverage 3 only
    solid.clear()
    assert solid.solid_from_pairs_dev(all_k, all_c, 0, 3, all_h) == 0  # nothing gathered: an empty graph
    assert solid.bfs(hi, lo, 1, 3, 3000, -1) is None
    with pytest.raises(Exception):
        full.solid_from_pairs_dev(all_k, all_c, 2 * mx, 3, all_h)  # holds counts
    full.close()
    solid.close()


def test_hints_survive_exchange_and_speed_up_the_walk(mc):
    """Same results with and without hints; with them the BFS needs far fewer memory round trips."""
    import torch
    dev = torch.device("cuda:0")
    genome, reads, off = synth_case(1, 60000, 20000, 150, 0)
    n = 20000 * 120
    words = po.pack(reads)
    d_words = torch.from_numpy(words.view(np.int64)).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    ex = mc.Context(31, mc.KEY_PACKED, 0, 0)
    d_keys = torch.zeros(n, dtype=torch.int64, device=dev)
    d_hints = torch.zeros(n, dtype=torch.int32, device=dev)
    ooff = ex.extract_keys_dev(d_words, d_off, 20000, 20000 * 150, 1, d_keys, n, d_hints)
    assert int(ooff[1]) == n
    hi, lo = seed_windows(genome[30000:30200], 31)
    rounds = {}
    res = {}
    for name, h in (("with", d_hints), ("without", None)):
        c = mc.Context(31, mc.KEY_PACKED, 0, 200000)
        c.share_read_store(ex)
        c.add_keys_dev(d_keys, n, h)
        c.finalize()
        r = c.bfs(hi, lo, -1, 3, 20000, -1)
        rounds[name], res[name] = r["rounds"], r
        c.close()
    assert_bfs_equal(res["with"], res["without"])
    assert rounds["with"] * 4 < rounds["without"]
    ex.close()


@pytest.mark.parametrize("k,mode,n_owners", [(31, "packed", 3), (27, "packed", 2), (41, "poly", 3), (21, "packed", 2)])
def test_walk_over_the_owners_tables_in_place(mc, k, mode, n_owners):
    """Several GPUs, the walk without a gather (include/mcgpu.h mc_shard_export / mc_shard_attach): the reads' super-k-mer
    records (keys for hash keys and k < 23) are dealt to `n_owners` contexts -- shares of the one GPU here --, every owner
    counts its own, and the first one, handed the others' tables, walks over all of them: each lookup goes to the table of
    the k-mer's owner (the owner of its minimizer for records, of its own hash for keys).  Walks in all directions equal the
    oracle's over ONE table; detached, the first context sees its own shard only."""
    import torch
    dev = torch.device("cuda:0")
    omode, gmode = (po.KEY_PACKED, mc.KEY_PACKED) if mode == "packed" else (po.KEY_POLY, mc.KEY_POLY)
    genome, reads, off = synth_case(2, 60000, 30000, 150, 60)
    t, n = oracle_table(reads, off, k, omode)
    d_words = torch.from_numpy(po.pack(reads).view(np.int64)).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    own = [mc.Context(k, gmode, 0, 0) for _ in range(n_owners)]
    n_reads, n_bases = len(off) - 1, int(off[-1])
    cap = own[0].superkmer_capacity(n, n_reads)
    records = cap != 0
    assert records == (mode == "packed" and k >= 23)
    if records:  # own[0] is the rank whose reads these are: it extracts (its read store is the one the pointers lead into) and counts its share
        d_recs = torch.zeros((cap, 2), dtype=torch.int64, device=dev)
        d_ptrs = torch.zeros(cap, dtype=torch.int32, device=dev)
        ooff = own[0].extract_superkmers_dev(d_words, d_off, n_reads, n_bases, n_owners, d_recs, d_ptrs, cap)
        for o in range(n_owners):
            a, b = int(ooff[o]), int(ooff[o + 1])
            assert b > a
            own[o].add_superkmers_dev(d_recs[a:b].contiguous(), d_ptrs[a:b].contiguous(), b - a)
    else:
        d_keys = torch.zeros(n, dtype=torch.int64, device=dev)
        d_hints = torch.zeros(n, dtype=torch.int32, device=dev)
        ooff = own[0].extract_keys_dev(d_words, d_off, n_reads, n_bases, n_owners, d_keys, n, d_hints)
        for o in range(n_owners):
            a, b = int(ooff[o]), int(ooff[o + 1])
            own[o].add_keys_dev(d_keys[a:b].contiguous(), b - a, d_hints[a:b].contiguous())
    assert sum(c.finalize() for c in own) == t.size()  # owners are disjoint
    handles = [c.shard_export() for c in own]
    own[0].shard_attach(handles, 0, records)
    seeds = [genome[5000:5400], genome[61000:61300]]
    for seed in seeds:
        hi, lo = seed_windows(seed, k)
        for d in (1, -1, 0):
            assert_bfs_equal(own[0].bfs(hi, lo, d, 3, 6000, -1), po.bfs(t, k, omode, [seed], d, 3, 6000, -1))
    # a second threshold on the same attachment (nothing is rebuilt: the walk reads the counts themselves)
    hi, lo = seed_windows(seeds[0], k)
    assert_bfs_equal(own[0].bfs(hi, lo, 0, 6, 3000, 200), po.bfs(t, k, omode, [seeds[0]], 0, 6, 3000, 200))
    own[0].shard_detach()
    alone = own[0].bfs(hi, lo, 0, 3, 6000, -1)
    want = po.bfs(t, k, omode, [seeds[0]], 0, 3, 6000, -1)
    assert alone is None or len(alone["lo"]) < len(want["lo"])  # (a third or half of the k-mers: the walk falls apart)
    with pytest.raises(mc.McError):
        own[1].shard_attach(handles, 0, records)  # handle 0 is not context 1's own table
    # an attachment names the tables as they were.  More reads into the walking context (its table may grow, i.e.
    # move) drop it, and the walk refuses to run on a view that is gone -- it used to read the freed block -- until the
    # tables are attached again; the walk then sees the new counts too.
    own[0].shard_attach(handles, 0, records)
    er = np.ascontiguousarray(genome[5000:5400], dtype=np.uint8)
    own[0].add_reads_packed(po.pack(er), np.array([0, len(er)], dtype=np.uint64))
    own[0].finalize()
    with pytest.raises(mc.McError, match="attach"):
        own[0].bfs(hi, lo, 0, 3, 6000, -1)
    handles = [c.shard_export() for c in own]
    own[0].shard_attach(handles, 0, records)
    assert own[0].bfs(hi, lo, 0, 3, 6000, -1) is not None
    for c in own:
        c.close()

