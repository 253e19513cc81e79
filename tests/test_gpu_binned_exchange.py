"""The binned form of the multi-GPU record exchange (include/mcgpu.h mc_extract_superkmers_binned_dev / mc_add_superkmers_binned_dev): the
sender puts every owner's super-k-mer records in the order of the level-1 buckets of the owner's counting run, which then starts at its
second level.  One process plays all ranks: two senders extract two chunks each for three (or eight) owners, every owner is handed its
parts back to back with their rows of counts, and the owners' tables together must hold exactly the oracle's (key, count) pairs
(src/io/IOUtils.java:201-214: one addAndBound(key, 1) a window), every key on the rank that owns its minimizer, and the same as the flat
form of the exchange gives.  Needs a real MI355X."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests.helpers import oracle_table, ragged_case, synth_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mc():
    import metacherchant_amd as m
    m.native.load()
    return m


def _case(n_reads=60000):
    genome, codes, offs = synth_case(2, 200000, n_reads, 150, 100)
    return genome, codes, offs


def _extract_all(mc, torch, senders, chunks, W, fine, k, hint, codes, offs, binned):
    """every (sender, chunk)'s records for every owner: list over owners of lists of (records, pointers, counts row, windows)"""
    dev = torch.device("cuda:0")
    n_reads = len(offs) - 1
    parts = [[] for _ in range(W)]
    for s in range(senders):
        ctx = mc.Context(k, mc.KEY_PACKED, 0, hint)
        if s:
            ctx.set_read_pointers(False)  # (only the first sender walks: the others' records carry no pointer)
        lo, hi = n_reads * s // senders, n_reads * (s + 1) // senders
        for c in range(chunks):
            a, b = lo + (hi - lo) * c // chunks, lo + (hi - lo) * (c + 1) // chunks
            sub = codes[int(offs[a]):int(offs[b])]
            o = offs[a:b + 1] - offs[a]
            d_words = torch.from_numpy(po.pack(sub).view(np.int64)).to(dev)
            d_off = torch.from_numpy(o.astype(np.uint64).view(np.int64)).to(dev)
            nb = int(o[-1])
            cap = ctx.superkmer_capacity(nb, b - a)
            send = torch.empty((cap, 2), dtype=torch.int64, device=dev)
            send_p = torch.empty(cap, dtype=torch.int32, device=dev)
            if binned:
                fc = torch.empty((W, fine), dtype=torch.int32, device=dev)
                off, win = ctx.extract_superkmers_binned_dev(d_words, d_off, b - a, nb, W, fine, send, send_p, cap, fc)
                assert np.array_equal(fc.sum(dim=1).cpu().numpy(), np.diff(off).astype(np.int64))
            else:
                off = ctx.extract_superkmers_dev(d_words, d_off, b - a, nb, W, send, send_p, cap)
                fc, win = None, np.zeros(W, dtype=np.uint64)
            for w in range(W):
                x, y = int(off[w]), int(off[w + 1])
                parts[w].append((send[x:y].clone(), send_p[x:y].clone(), fc[w].clone() if binned else None, int(win[w])))
        ctx.close()
    return parts


def _count_owner(mc, torch, parts, k, hint, fine, binned, spoil=False):
    recs = torch.cat([p[0] for p in parts])
    ptrs = torch.cat([p[1] for p in parts])
    n = recs.shape[0]
    ctx = mc.Context(k, mc.KEY_PACKED, 0, hint)
    ctx.set_coverage_hint(3)
    if binned:
        po_ = np.concatenate([[0], np.cumsum([p[0].shape[0] for p in parts])]).astype(np.uint64)
        pc = torch.stack([p[2] for p in parts])
        if spoil:
            pc[1, 3] += 1
        ctx.add_superkmers_binned_dev(recs, ptrs, n, sum(p[3] for p in parts), fine, po_, pc)
    else:
        ctx.add_superkmers_dev(recs, ptrs, n)
    return ctx


@pytest.mark.parametrize("k,W", [(31, 3), (31, 8), (25, 2), (27, 5)])
def test_binned_exchange_counts_what_the_oracle_counts(mc, k, W):
    import torch
    genome, codes, offs = _case()
    t, _ = oracle_table(codes, offs, k, po.KEY_PACKED)
    ok, oc = t.dump()
    hint = 40_000_000  # (a table with a second level: ~27 000 regions, 512 level-1 buckets)
    probe = mc.Context(k, mc.KEY_PACKED, 0, hint)
    fine = probe.superkmer_fine_buckets(W)
    probe.close()
    assert fine >= 256 and fine * W <= 16384
    parts = _extract_all(mc, torch, 2, 2, W, fine, k, hint, codes, offs, binned=True)
    flat = _extract_all(mc, torch, 2, 2, W, fine, k, hint, codes, offs, binned=False)
    all_k, all_c = [], []
    windows = 0
    for w in range(W):
        # the same records as the flat form deals to this owner, in another order
        assert sum(p[0].shape[0] for p in parts[w]) == sum(p[0].shape[0] for p in flat[w])
        windows += sum(p[3] for p in parts[w])
        ctx = _count_owner(mc, torch, parts[w], k, hint, fine, True)
        n = ctx.finalize()
        st = ctx.stats()
        assert st.binned_runs == 1 and st.grows == 0
        gk, gc = ctx.export(0)
        assert len(gk) == n
        ref = _count_owner(mc, torch, flat[w], k, hint, fine, False)
        ref.finalize()
        assert ref.stats().binned_runs == 0
        rk, rc = ref.export(0)
        assert np.array_equal(gk, rk) and np.array_equal(gc, rc), w
        for cov in (3,):
            a, b = ctx.export(cov), ref.export(cov)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and ctx.export_count(cov) == len(a[0])
        all_k.append(gk)
        all_c.append(gc)
        ctx.close()
        ref.close()
    assert windows == len(codes) - (len(offs) - 1) * (k - 1)  # the senders' window counts: every window of every read
    gk = np.concatenate(all_k)
    gc = np.concatenate(all_c)
    o = np.argsort(gk, kind="stable")
    assert np.array_equal(gk[o], ok) and np.array_equal(gc[o], oc)  # owners are disjoint: together, the oracle's table


def test_counts_that_do_not_add_up_are_refused(mc):
    import torch
    genome, codes, offs = _case(20000)
    k, W, hint = 31, 2, 40_000_000
    probe = mc.Context(k, mc.KEY_PACKED, 0, hint)
    fine = probe.superkmer_fine_buckets(W)
    probe.close()
    parts = _extract_all(mc, torch, 2, 2, W, fine, k, hint, codes, offs, binned=True)
    with pytest.raises(Exception, match="do not add up"):
        _count_owner(mc, torch, parts[0], k, hint, fine, True, spoil=True)


@pytest.mark.parametrize("scale,binned_runs", [(2, 1), (0.5, 0)])
def test_fine_buckets_that_are_not_the_tables_own(mc, scale, binned_runs):
    """Senders that planned for twice the owner's level-1 buckets: every bucket of the owner is two fine buckets, every part two
    segments of every bucket -- still a run from the second level.  For half as many: the owner's buckets are not unions of fine
    buckets, and it counts the same records as a flat stream (binned_runs 0).  The same table either way."""
    import torch
    genome, codes, offs = _case(20000)
    k, W, hint = 31, 2, 40_000_000
    t, _ = oracle_table(codes, offs, k, po.KEY_PACKED)
    ok, oc = t.dump()
    probe = mc.Context(k, mc.KEY_PACKED, 0, hint)
    fine = int(probe.superkmer_fine_buckets(W) * scale)
    probe.close()
    parts = _extract_all(mc, torch, 2, 2, W, fine, k, hint, codes, offs, binned=True)
    ks, cs = [], []
    for w in range(W):
        ctx = _count_owner(mc, torch, parts[w], k, hint, fine, True)
        ctx.finalize()
        assert ctx.stats().binned_runs == binned_runs
        a, b = ctx.export(0)
        ks.append(a)
        cs.append(b)
        ctx.close()
    gk, gc = np.concatenate(ks), np.concatenate(cs)
    o = np.argsort(gk, kind="stable")
    assert np.array_equal(gk[o], ok) and np.array_equal(gc[o], oc)


def test_ragged_reads_through_the_binned_exchange(mc):
    """Reads of 0 .. 220 bases (shorter than k, one window, many): the same table as the oracle's, three owners, two senders, two chunks."""
    import torch
    rng = np.random.default_rng(5)
    _, codes, offs = ragged_case(rng, 30000, max_len=220, genome_len=60000)
    k, W, hint = 27, 3, 40_000_000
    t, _ = oracle_table(codes, offs, k, po.KEY_PACKED)
    ok, oc = t.dump()
    probe = mc.Context(k, mc.KEY_PACKED, 0, hint)
    fine = probe.superkmer_fine_buckets(W)
    probe.close()
    parts = _extract_all(mc, torch, 2, 2, W, fine, k, hint, codes, offs, binned=True)
    ks, cs = [], []
    for w in range(W):
        ctx = _count_owner(mc, torch, parts[w], k, hint, fine, True)
        ctx.finalize()
        assert ctx.stats().binned_runs == 1
        a, b = ctx.export(0)
        ks.append(a)
        cs.append(b)
        ctx.close()
    gk, gc = np.concatenate(ks), np.concatenate(cs)
    o = np.argsort(gk, kind="stable")
    assert np.array_equal(gk[o], ok) and np.array_equal(gc[o], oc)
