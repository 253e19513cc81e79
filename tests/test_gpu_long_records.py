"""The long-record form of the counting pipeline (csrc/count_long.h): polynomial-hash keys of 33 .. 63 bases, a table sized by
a capacity hint (or an empty table, sized by a sample of the batch), regions = minimizer bins of the k-mers' BASES.  Same (key, count) pairs and the same walks as the CPU oracle
(oracle/pyoracle.py follows src/utils/PolynomialHash.java:19-28, src/io/IOUtils.java:207-208 and
src/algo/OneSequenceCalculator.java) -- and as the per-window form, which MC_LONG_RECORDS=0 keeps.  Needs a real MI355X."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests.helpers import assert_bfs_equal, oracle_table, ragged_case, seed_windows, synth_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mc():
    import metacherchant_amd as m
    m.native.load()
    return m


def _walks(ctx, t, k, genome, cov):
    seed = genome[10000:10400]
    hi, lo = seed_windows(seed, k)
    for d, mk, mr in [(0, 100000, -1), (1, 3000, -1), (-1, -1, 200), (0, 777, -1)]:
        assert_bfs_equal(ctx.bfs(hi, lo, d, cov, mk, mr), po.bfs(t, k, po.KEY_POLY, [seed], d, cov, mk, mr))


def _bins(monkeypatch, bins):
    """MC_LONG_BINS=2 (read at mc_create): the bin word of a window from its TWO smallest minimizer hashes (count_long.h skl_word2) --
    what a context picks by itself when its table cannot be roomy (configs[2] at full size); the tests force it at any load."""
    if bins is None:
        monkeypatch.delenv("MC_LONG_BINS", raising=False)
    else:
        monkeypatch.setenv("MC_LONG_BINS", bins)


@pytest.mark.parametrize("bins", [None, "2"])
@pytest.mark.parametrize("k", [63, 33, 40, 41, 47, 48, 55, 62])
def test_long_records_count_and_walk(mc, monkeypatch, k, bins):
    """Two batches (the second one into a table that holds keys already, starting in the middle of a tile of the read store):
    every (key, count) pair of the oracle, the walks of the oracle straight on the minimizer-bin table (look-ups by the k-mers'
    bases), a look-up BY KEY of everything (one sweep of the table answers it: the table stays as it is), a third batch as long
    records again, and then a key stream (mc_add_keys_dev) -- which moves the table to hash-prefix regions -- and the walks again."""
    monkeypatch.setenv("MC_COUNT_PATH", "partition")  # (a batch of under 2^22 windows takes the direct kernel otherwise: by key)
    monkeypatch.delenv("MC_LONG_RECORDS", raising=False)
    _bins(monkeypatch, bins)
    n_reads = 80000 if k > 50 else 50000
    genome, reads, off = synth_case(2, 200000, n_reads, 150, 50)
    t, _ = oracle_table(reads, off, k, po.KEY_POLY)
    ok, oc = t.dump()
    ctx = mc.Context(k, mc.KEY_POLY, 0, int(t.size() * 1.3))
    ctx.set_coverage_hint(3)
    cut = n_reads // 2 + 1
    ctx.add_reads_packed(po.pack(reads[:off[cut]]), off[:cut + 1])
    ctx.add_reads_packed(po.pack(reads[off[cut]:]), off[cut:] - off[cut])
    assert ctx.finalize() == t.size()
    st = ctx.stats()
    assert st.long_runs == 2 and st.grows == 0, (st.long_runs, st.grows)
    gk, gc = ctx.export(0)
    assert np.array_equal(gk, ok) and np.array_equal(gc, oc)
    _walks(ctx, t, k, genome, 3)
    assert np.array_equal(ctx.get(ok), oc)       # by key: answered by a sweep
    absent = ok[:1000] ^ np.int64(0x5DEECE66D)   # keys that are (almost surely) not there, one of them asked for twice
    got = ctx.get(np.concatenate([absent, ok[:5], ok[:5]]))
    assert np.array_equal(got[1000:1005], oc[:5]) and np.array_equal(got[1005:], oc[:5])
    assert np.all((got[:1000] == -1) | np.isin(absent, ok))
    assert ctx.stats().grows == 0
    # ... a third batch, long records again into the same table
    ctx.add_reads_packed(po.pack(reads[:off[cut]]), off[:cut + 1])
    ctx.finalize()
    assert ctx.stats().long_runs == 3
    t2 = po.Table()
    t2.count_reads(reads, off, k, po.KEY_POLY)
    t2.count_reads(reads[:off[cut]], off[:cut + 1], k, po.KEY_POLY)
    ok2, oc2 = t2.dump()
    gk, gc = ctx.export(0)
    assert np.array_equal(gk, ok2) and np.array_equal(gc, oc2)
    _walks(ctx, t2, k, genome, 3)
    # ... and a stream of bare keys: the table leaves its minimizer bins for it (one rebuild), everything stays findable
    import torch
    extra = torch.from_numpy(ok2[:100000].copy()).to("cuda:0")
    ctx.add_keys_dev(extra, len(extra))
    ctx.finalize()
    assert ctx.stats().grows == 1
    oc3 = oc2.copy()
    oc3[:100000] = np.minimum(oc3[:100000].astype(np.int32) + 1, 32767).astype(np.int16)
    gk, gc = ctx.export(0)
    assert np.array_equal(gk, ok2) and np.array_equal(gc, oc3)
    assert np.array_equal(ctx.get(ok2[:200000]), oc3[:200000])
    ctx.close()


@pytest.mark.parametrize("bins", [None, "2"])
@pytest.mark.parametrize("k", [33, 63, 50])
def test_long_records_of_ragged_reads(mc, monkeypatch, k, bins):
    """Empty reads, reads of k - 1, k and k + 1 bases, reads of every length up to 220: windows never span two reads (the
    128-bit read-start mask of k_skl_extract), runs are cut at 32 windows, tiles end inside reads."""
    monkeypatch.setenv("MC_COUNT_PATH", "partition")
    monkeypatch.delenv("MC_LONG_RECORDS", raising=False)
    _bins(monkeypatch, bins)
    rng = np.random.default_rng(100 + k)
    genome, reads, off = ragged_case(rng, 6000, 220, 20000)
    # a few reads of exactly k - 1, k, k + 1 bases, and a long error-free one (runs of more than 32 windows)
    extra = [genome[5:5 + k - 1], genome[50:50 + k], genome[90:90 + k + 1], genome[1000:3000]]
    lens = np.diff(off).tolist() + [len(e) for e in extra]
    reads = np.concatenate([reads] + extra)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    t, _ = oracle_table(reads, off, k, po.KEY_POLY)
    ok, oc = t.dump()
    ctx = mc.Context(k, mc.KEY_POLY, 0, 100000)
    ctx.add_reads_packed(po.pack(reads), off)
    assert ctx.finalize() == t.size()
    assert ctx.stats().long_runs == 1
    gk, gc = ctx.export(0)
    assert np.array_equal(gk, ok) and np.array_equal(gc, oc)
    seed = genome[1200:1500]
    hi, lo = seed_windows(seed, k)
    for d in (0, 1, -1):
        assert_bfs_equal(ctx.bfs(hi, lo, d, 2, 5000, -1), po.bfs(t, k, po.KEY_POLY, [seed], d, 2, 5000, -1))
    ctx.close()


def test_long_records_into_a_table_a_quarter_the_size(mc, monkeypatch):
    """A capacity hint a quarter of what the reads hold: bins overflow, leaves hand occurrences on along the region chain or
    fail -- and a table of hash keys in minimizer bins cannot be rebuilt with more bins: it moves to hash-prefix regions under
    the run and the leaves that were left go in by key.  Nothing may be lost or counted twice."""
    monkeypatch.delenv("MC_COUNT_PATH", raising=False)
    monkeypatch.delenv("MC_LONG_RECORDS", raising=False)
    k = 47
    genome, reads, off = synth_case(2, 150000, 60000, 150, 150)
    t, _ = oracle_table(reads, off, k, po.KEY_POLY)
    ok, oc = t.dump()
    ctx = mc.Context(k, mc.KEY_POLY, 0, t.size() // 4)
    ctx.add_reads_packed(po.pack(reads), off)
    assert ctx.finalize() == t.size()
    gk, gc = ctx.export(0)
    assert np.array_equal(gk, ok) and np.array_equal(gc, oc)
    assert np.array_equal(ctx.get(ok), oc)
    seed = genome[10000:10300]
    hi, lo = seed_windows(seed, k)
    assert_bfs_equal(ctx.bfs(hi, lo, 0, 3, 20000, -1), po.bfs(t, k, po.KEY_POLY, [seed], 0, 3, 20000, -1))
    ctx.close()


def test_long_records_switched_off(mc, monkeypatch):
    """MC_LONG_RECORDS=0 (read at mc_create): the per-window pipeline, as before."""
    monkeypatch.delenv("MC_COUNT_PATH", raising=False)
    monkeypatch.setenv("MC_LONG_RECORDS", "0")
    genome, reads, off = synth_case(1, 100000, 50000, 150, 50)
    t, _ = oracle_table(reads, off, 63, po.KEY_POLY)
    ok, oc = t.dump()
    for hint in (3_000_000, 0):
        ctx = mc.Context(63, mc.KEY_POLY, 0, hint)
        ctx.add_reads_packed(po.pack(reads), off)
        assert ctx.finalize() == t.size()
        assert ctx.stats().long_runs == 0
        gk, gc = ctx.export(0)
        assert np.array_equal(gk, ok) and np.array_equal(gc, oc)
        ctx.close()


@pytest.mark.parametrize("k", [63, 45])
def test_long_records_without_a_capacity_hint(mc, monkeypatch, k):
    """No hint (the CLI's default): the first batch into the empty table still travels as long records -- the records carry their
    bin word, the distinct keys of the first level-1 bucket size the table between the two levels (pipe_resize_by_sample) -- and
    the second batch, into a table that now holds keys and that nothing vouches for, moves it to hash-prefix regions and takes the
    per-window form.  Pairs and walks as the oracle's after either."""
    monkeypatch.delenv("MC_COUNT_PATH", raising=False)
    monkeypatch.delenv("MC_LONG_RECORDS", raising=False)
    n_reads = 120000
    genome, reads, off = synth_case(2, 200000, n_reads, 150, 100)
    cut = 70000
    t1, _ = oracle_table(reads[:off[cut]], off[:cut + 1], k, po.KEY_POLY)
    ctx = mc.Context(k, mc.KEY_POLY, 0, 0)
    ctx.set_coverage_hint(2)
    ctx.add_reads_packed(po.pack(reads[:off[cut]]), off[:cut + 1])
    assert ctx.finalize() == t1.size()
    st = ctx.stats()
    assert st.long_runs == 1 and st.grows == 1, (st.long_runs, st.grows)  # (the 64 MB table it was created with was replaced once, empty)
    ok, oc = t1.dump()
    gk, gc = ctx.export(0)
    assert np.array_equal(gk, ok) and np.array_equal(gc, oc)
    _walks(ctx, t1, k, genome, 2)
    t, _ = oracle_table(reads, off, k, po.KEY_POLY)
    ctx.add_reads_packed(po.pack(reads[off[cut]:]), off[cut:] - off[cut])
    assert ctx.finalize() == t.size()
    assert ctx.stats().long_runs == 1
    ok, oc = t.dump()
    gk, gc = ctx.export(0)
    assert np.array_equal(gk, ok) and np.array_equal(gc, oc)
    _walks(ctx, t, k, genome, 2)
    ctx.close()


def test_long_records_without_a_read_store_and_through_a_kmers_file(mc, monkeypatch, tmp_path):
    """No read store (mc_set_read_pointers(0): the records carry no pointers, the walk gets no look-ahead) -- same pairs, same
    walks; and the table as a .kmers.bin file (a sweep) loaded into another context of the same kind (a key stream: that one
    leaves its minimizer bins for it) -- same pairs, same walks again."""
    monkeypatch.setenv("MC_COUNT_PATH", "partition")
    monkeypatch.delenv("MC_LONG_RECORDS", raising=False)
    monkeypatch.delenv("MC_LONG_BINS", raising=False)
    k = 51
    genome, reads, off = synth_case(2, 150000, 40000, 150, 50)
    t, _ = oracle_table(reads, off, k, po.KEY_POLY)
    ok, oc = t.dump()
    ctx = mc.Context(k, mc.KEY_POLY, 0, int(t.size() * 1.2))
    ctx.set_read_pointers(False)
    ctx.add_reads_packed(po.pack(reads), off)
    assert ctx.finalize() == t.size() and ctx.stats().long_runs == 1
    gk, gc = ctx.export(0)
    assert np.array_equal(gk, ok) and np.array_equal(gc, oc)
    _walks(ctx, t, k, genome, 3)
    path = str(tmp_path / "t.kmers.bin")
    ctx.save_kmers(path)
    other = mc.Context(k, mc.KEY_POLY, 0, int(t.size() * 1.2))
    other.load_kmers(path)
    assert other.finalize() == t.size()
    gk, gc = other.export(0)
    assert np.array_equal(gk, ok) and np.array_equal(gc, oc)
    assert np.array_equal(other.get(ok[:50000]), oc[:50000])
    _walks(other, t, k, genome, 3)
    other.close()
    ctx.close()


def test_a_run_cut_for_long_records_is_cut_again_when_the_long_form_declines(mc, monkeypatch):
    """a k = 33 .. 63 context cuts its batches for long records (2^34 bases a run).  When the long form declines at run
    time -- here: a second batch into a table no hint vouches for -- the table moves to hash-prefix regions and the batch takes one
    record a window, for which the run may be far too large (32-bit bucket indices, scratch).  The caller must cut it again.
    MC_MAX_RUN_BASES_PER_WINDOW plays the small limit: the second batch goes through in several runs, same counts as the oracle."""
    monkeypatch.setenv("MC_COUNT_PATH", "partition")
    monkeypatch.delenv("MC_LONG_RECORDS", raising=False)
    monkeypatch.setenv("MC_MAX_RUN_BASES_PER_WINDOW", str(1 << 21))
    k = 47
    genome, reads, off = synth_case(2, 200000, 60000, 150, 50)
    t, _ = oracle_table(reads, off, k, po.KEY_POLY)
    ctx = mc.Context(k, mc.KEY_POLY, 0, 0)
    cut = 20000
    ctx.add_reads_packed(po.pack(reads[:off[cut]]), off[:cut + 1])          # long records, table sized by the sample
    assert ctx.stats().long_runs == 1
    launches = ctx.stats().count_launches
    ctx.add_reads_packed(po.pack(reads[off[cut]:]), off[cut:] - off[cut])   # 6 M bases: declined, then three runs of <= 2 M bases
    st = ctx.stats()
    assert st.long_runs == 1 and st.count_launches - launches >= 3, (st.long_runs, st.count_launches - launches)
    assert st.left_bins == 3  # (mc_stats says why the table left its minimizer bins: nothing vouched for its size)
    assert ctx.finalize() == t.size()
    gk, gc = ctx.export(0)
    ok, oc = t.dump()
    assert np.array_equal(gk, ok) and np.array_equal(gc, oc)
    # ... and the same from device memory (mc_add_reads_packed_dev cuts by bisection on the device's offsets)
    import torch
    ctx2 = mc.Context(k, mc.KEY_POLY, 0, 0)
    ctx2.add_reads_packed(po.pack(reads[:off[cut]]), off[:cut + 1])
    w = torch.from_numpy(po.pack(reads[off[cut]:]).view(np.int64)).to("cuda:0")
    o = torch.from_numpy((off[cut:] - off[cut]).astype(np.int64)).to("cuda:0")
    ctx2.add_reads_packed_dev(w, o, len(off) - 1 - cut, int(off[-1] - off[cut]))
    assert ctx2.stats().count_launches >= 4
    assert ctx2.finalize() == t.size()
    gk, gc = ctx2.export(0)
    assert np.array_equal(gk, ok) and np.array_equal(gc, oc)
    ctx.close()
    ctx2.close()
