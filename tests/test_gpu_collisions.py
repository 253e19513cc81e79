"""Different k-mers with one PolynomialHash key through the long-record counting form (csrc/count_long.h), where the table's
regions are minimizer bins of the k-mers' BASES: the reference adds both to ONE counter (src/io/LargeKIOUtils.java:46-49,
src/utils/PolynomialHash.java:19-28, itmo!/structures/map/Long2ShortHashMap.java:119-157).  The vectors of
tests/golden/poly_collisions.json (scripts/poly_collisions.py: equal keys, different bins under both bin rules) are planted in
a read set; "Hashtable size", every (key, count) pair, look-ups by key, .kmers.bin and the walks -- coverages AND reached sets --
must equal the oracle's and the per-window form's, with a hint and with a table sized by the sample, MC_LONG_BINS 1 and 2,
after a second batch, and after the table has moved to hash-prefix regions.  Needs a real MI355X."""
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.helpers import assert_bfs_equal, oracle_table, seed_windows, synth_case

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
VECTORS = json.load(open(os.path.join(HERE, "golden", "poly_collisions.json")))["vectors"]


@pytest.fixture(scope="module")
def mc():
    import metacherchant_amd as m
    m.native.load()
    return m


def _rc(c):
    return (3 - c[::-1]).astype(np.uint8)


def _planted(k, pairs, flip):
    """Two contigs per pair: x inside the first, y (or its reverse complement) inside the second, both tiled with error-free reads of
    150 bases every 10 (both strands) -- and 50 000 reads over 2 x 200 kb as everybody else.  Returns codes, offsets, the contigs."""
    rng = np.random.default_rng(7 * k + len(pairs))
    genome, reads, off = synth_case(2, 200000, 50000, 150, 50)
    pieces, lens, contigs = [reads], np.diff(off).tolist(), []
    for i, v in enumerate(pairs):
        x, y = po.encode(v["x"]), po.encode(v["y"])
        if flip:
            y = _rc(y)
        for s in (x, y):
            c = rng.integers(0, 4, 4000).astype(np.uint8)
            c[2000:2000 + k] = s
            contigs.append(c)
            for j, st in enumerate(range(0, 4000 - 150 + 1, 10)):
                r = c[st:st + 150]
                pieces.append(_rc(r) if (j + i) & 1 else r)
                lens.append(150)
    codes = np.concatenate(pieces)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    return genome, codes, offs, contigs


def _check(ctx, t, k, pairs, contigs, genome, tmp_path, tag):
    ok, oc = t.dump()
    assert ctx.finalize() == t.size(), tag  # "Hashtable size" (src/io/LargeKIOUtils.java:86)
    gk, gc = ctx.export(0)
    assert np.array_equal(gk, ok) and np.array_equal(gc, oc), tag
    for cov in (3, 12):
        sel = oc >= cov
        gk, gc = ctx.export(cov)
        assert np.array_equal(gk, ok[sel]) and np.array_equal(gc, oc[sel]), (tag, cov)
        assert ctx.export_count(cov) == int(sel.sum()), (tag, cov)
    pk = np.array([v["key"] for v in pairs], dtype=np.int64)
    want = t.get_many(pk)
    assert np.array_equal(ctx.get(pk), want), tag          # by key, the colliding ones alone ...
    assert np.array_equal(ctx.get(ok), oc), tag            # ... and everything
    # .kmers.bin (src/io/IOUtils.java:39-65: big-endian key, big-endian count)
    path = str(tmp_path / ("c_%s.kmers.bin" % tag))
    ctx.save_kmers(path, None, 0)
    raw = np.fromfile(path, dtype=np.uint8).reshape(-1, 10)
    fk = raw[:, :8].copy().view(">i8").reshape(-1).astype(np.int64)
    fc = raw[:, 8:].copy().view(">i2").reshape(-1).astype(np.int16)
    o = np.argsort(fk, kind="stable")
    assert np.array_equal(fk[o], ok) and np.array_equal(fc[o], oc), tag
    # the walks: through x with everything around it solid (coverages), and with a threshold only the SUM of the two counters
    # reaches -- x as the seed: with either counter alone that is the reference's "fail" (no seed k-mer reaches the threshold)
    for i, v in enumerate(pairs):
        cx = contigs[2 * i]
        seed = cx[1700:1700 + 200]
        hi, lo = seed_windows(seed, k)
        for d in (0, 1):
            assert_bfs_equal(ctx.bfs(hi, lo, d, 3, 2500, -1), po.bfs(t, k, po.KEY_POLY, [seed], d, 3, 2500, -1))
        total = int(t.get(v["key"]))
        own = total // 2 + 1 if total < 32767 else 32767  # (above either k-mer's own count unless they are very uneven: checked below)
        seed1 = cx[2000:2000 + k]  # x itself: a seed is queued when ITS count reaches the threshold (OneSequenceCalculator.java:159-196)
        h1, l1 = seed_windows(seed1, k)
        want1 = po.bfs(t, k, po.KEY_POLY, [seed1], 1, own, 100, -1)
        assert want1 is not None and int(want1["cov"][0]) == min(total, 32767), tag
        assert_bfs_equal(ctx.bfs(h1, l1, 1, own, 100, -1), want1)
    seed = genome[10000:10300]
    hi, lo = seed_windows(seed, k)
    assert_bfs_equal(ctx.bfs(hi, lo, 0, 3, 20000, -1), po.bfs(t, k, po.KEY_POLY, [seed], 0, 3, 20000, -1))


@pytest.mark.parametrize("bins", [None, "2"])
@pytest.mark.parametrize("hinted", [True, False])
@pytest.mark.parametrize("k", sorted({v["k"] for v in VECTORS}))
def test_colliding_kmers_share_a_counter_in_minimizer_bins(mc, monkeypatch, tmp_path, k, hinted, bins):
    monkeypatch.setenv("MC_COUNT_PATH", "partition")
    monkeypatch.delenv("MC_LONG_RECORDS", raising=False)
    if bins is None:
        monkeypatch.delenv("MC_LONG_BINS", raising=False)
    else:
        monkeypatch.setenv("MC_LONG_BINS", bins)
    pairs = [v for v in VECTORS if v["k"] == k]
    genome, codes, offs, contigs = _planted(k, pairs, flip=bool(k & 2))
    t, _ = oracle_table(codes, offs, k, po.KEY_POLY)
    for v in pairs:  # the plant is what it is meant to be: x and y are covered alike, so either alone has half the counter
        covering = sum(1 for st in range(0, 4000 - 150 + 1, 10) if st <= 2000 and st + 150 >= 2000 + k)
        assert t.get(v["key"]) == 2 * covering >= 10
    ctx = mc.Context(k, mc.KEY_POLY, 0, int(t.size() * 1.3) if hinted else 0)
    ctx.set_coverage_hint(3)
    ctx.add_reads_packed(po.pack(codes), offs)
    tag = "k%d_%s_%s" % (k, "hint" if hinted else "sample", bins or "auto")
    if bins == "2":
        # the join's key streams given back before anything walks (mc_trim): the walks' look-ups are then checked by key through a
        # sweep of the table -- the same hits, the same second slots
        ctx.finalize()
        ctx.trim()
    _check(ctx, t, k, pairs, contigs, genome, tmp_path, tag)
    st = ctx.stats()
    assert st.long_runs == 1, "the batch was meant to travel as long records"
    # the join found the planted pairs -- and the walks' check by key, where a neighbour of x that nobody counted has the hash of y's
    # neighbour (the rolling hash goes on colliding while the same bases leave and enter), gave those keys a second slot as well
    assert st.dup_keys >= len(pairs), st.dup_keys
    if hinted:
        # a second batch into the same table (long records again): the counters of x's contig double, y's do not
        n0 = 50000
        sel_lo = int(offs[n0])
        ctx.add_reads_packed(po.pack(codes[sel_lo:]), offs[n0:] - offs[n0])
        t.count_reads(codes[sel_lo:], offs[n0:] - offs[n0], k, po.KEY_POLY)
        _check(ctx, t, k, pairs, contigs, genome, tmp_path, tag + "_2")
        assert ctx.stats().long_runs == 2
    # a stream of bare keys moves the table to hash-prefix regions: the two counters become one there
    import torch
    ok, _ = t.dump()
    extra = torch.from_numpy(ok[:50000].copy()).to("cuda:0")
    ctx.add_keys_dev(extra, len(extra))
    for key_ in ok[:50000]:
        t.add(int(key_), 1)
    _check(ctx, t, k, pairs, contigs, genome, tmp_path, tag + "_3")
    assert ctx.stats().dup_keys == 0 and ctx.stats().left_bins == 1  # (a key stream came)
    ctx.close()


@pytest.mark.parametrize("k", [63, 33])
def test_per_window_form_merges_them_too(mc, monkeypatch, tmp_path, k):
    """MC_LONG_RECORDS=0: regions by the key's own hash -- one counter by construction; the same checks."""
    monkeypatch.setenv("MC_COUNT_PATH", "partition")
    monkeypatch.setenv("MC_LONG_RECORDS", "0")
    pairs = [v for v in VECTORS if v["k"] == k]
    genome, codes, offs, contigs = _planted(k, pairs, flip=True)
    t, _ = oracle_table(codes, offs, k, po.KEY_POLY)
    ctx = mc.Context(k, mc.KEY_POLY, 0, int(t.size() * 1.3))
    ctx.set_coverage_hint(3)
    ctx.add_reads_packed(po.pack(codes), offs)
    _check(ctx, t, k, pairs, contigs, genome, tmp_path, "k%d_perwindow" % k)
    assert ctx.stats().long_runs == 0 and ctx.stats().dup_keys == 0
    ctx.close()


def test_key_streams_that_overflow_send_the_join_to_its_sweep(mc, monkeypatch, tmp_path, capfd):
    """The merge kernel's key streams are sized by the capacity hint (a hint that falls short of the keys, at configs[2]'s size, makes
    them overflow; MC_DUP_L1_SCALE plays that here): the join must notice and start again from a sweep of the table, whose streams
    are sized by the number of keys the table really holds -- same results."""
    monkeypatch.setenv("MC_COUNT_PATH", "partition")
    monkeypatch.setenv("MC_DUP_L1_SCALE", "0.05")
    monkeypatch.setenv("MC_INGEST_DEBUG", "1")
    monkeypatch.delenv("MC_LONG_RECORDS", raising=False)
    monkeypatch.delenv("MC_LONG_BINS", raising=False)
    k = 47
    pairs = [v for v in VECTORS if v["k"] == k]
    genome, codes, offs, contigs = _planted(k, pairs, flip=True)
    t, _ = oracle_table(codes, offs, k, po.KEY_POLY)
    ctx = mc.Context(k, mc.KEY_POLY, 0, int(t.size() * 0.8))
    ctx.set_coverage_hint(3)
    ctx.add_reads_packed(po.pack(codes), offs)
    _check(ctx, t, k, pairs, contigs, genome, tmp_path, "k47_short_hint")
    st = ctx.stats()
    assert st.long_runs == 1 and st.left_bins == 0 and st.dup_checks >= 1 and st.dup_keys >= len(pairs), (st.long_runs, st.left_bins, st.dup_checks, st.dup_keys)
    ctx.close()
    assert "[join] a segment of the merge kernel's key streams overflowed" in capfd.readouterr().err
