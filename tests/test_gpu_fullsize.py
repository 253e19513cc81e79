"""BASELINE.json configs[1] at full size: 10 M x 150 bp synthetic reads (1 % substitutions), k = 31, coverage 5,
maxkmers 100000, both passes of bothdirs=False.  The oracle's multi-threaded counter builds the same table on the host
cores, and the two tables are compared through order-independent checksums (number of keys, sum of counts = number of
windows, wrapping sums of key*count and of a 64-bit mix of (key, count), number of keys at the threshold); the BFS
results are compared element by element.  Hosts with few cores check a fifth of the reads instead (same code paths:
still > 4 M windows per batch and > 512 table regions)."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.helpers import GENOME_SEED, READ_SEED, assert_bfs_equal, seed_windows

pytestmark = pytest.mark.gpu

M64 = (1 << 64) - 1


def _mix(keys_u64, counts_u64):
    """splitmix-style 64-bit mix of (key, count), elementwise, wrapping (numpy uint64 or torch int64 alike)."""
    x = keys_u64 * 0x9E3779B97F4A7C15 + counts_u64 * 0xC2B2AE3D27D4EB4F
    x = x ^ (x >> 31)
    return x * 0xD6E8FEB86659FD93


def test_config1_full_size_table_checksums_and_bfs():
    import torch
    import metacherchant_amd as mc
    cores = os.cpu_count() or 1
    R = 10_000_000 if cores >= 64 else 2_000_000
    k, L, cov, contigs, clen, err = 31, 150, 5, 10, 5_000_000, 100
    dev = torch.device("cuda:0")
    n_bases = R * L
    d_words = torch.empty((n_bases + 31) // 32 + 1, dtype=torch.int64, device=dev)
    d_off = torch.empty(R + 1, dtype=torch.int64, device=dev)
    ctx = mc.Context(k, mc.KEY_PACKED, 0, 0)  # no capacity hint: the table grows as the reference's does
    ctx.set_coverage_hint(cov)
    ctx.synth_reads_dev(GENOME_SEED, contigs, clen, READ_SEED, 0, R, L, err, d_words, d_off)
    ctx.add_reads_packed_dev(d_words, d_off, R, n_bases)
    nd = ctx.finalize()
    assert ctx.stats().windows == R * (L - k + 1)

    # ---- the same reads through the oracle (the generator itself is compared in test_gpu_parity.py)
    words = d_words.cpu().numpy().view(np.uint64)
    off = d_off.cpu().numpy().view(np.uint64)
    w, ond, _, t = po.count_reads_packed_mt(words, off, k, po.KEY_PACKED, min(cores, 256), want_table=True)
    assert w == R * (L - k + 1) and ond == nd
    ok, oc = t.dump()
    oku, ocu = ok.view(np.uint64), oc.astype(np.uint64)
    with np.errstate(over="ignore"):
        want = (len(ok), int(ocu.sum()), int((oku * ocu).sum()), int(_mix(oku, ocu).sum()), int((oc >= cov).sum()))
    del oku, ocu

    gk = torch.empty(nd, dtype=torch.int64, device=dev)
    gc = torch.empty(nd, dtype=torch.int16, device=dev)
    assert ctx.export_dev(0, gk, gc, nd) == nd
    gc64 = gc.to(torch.int64)

    def u(x):  # a wrapped int64 sum as the unsigned number numpy reports
        return int(x.item()) & M64

    # torch int64 arithmetic wraps like uint64; logical shift emulated on the non-negative range by masking
    x = gk * (0x9E3779B97F4A7C15 - (1 << 64)) + gc64 * (0xC2B2AE3D27D4EB4F - (1 << 64))
    x = x ^ ((x >> 31) & ((1 << 33) - 1))
    x = x * (0xD6E8FEB86659FD93 - (1 << 64))
    got = (nd, u(gc64.sum()), u((gk * gc64).sum()), u(x.sum()), int((gc >= cov).sum().item()))
    assert got == want
    assert got[1] == R * (L - k + 1)  # no count saturates in this workload: every occurrence is in the table
    assert ctx.export_count(cov) == want[4]
    del gk, gc, gc64, x

    # ---- BFS: seed gene = contig 0, bases [100000, 101000), both passes
    seed = mc.native.synth_genome(GENOME_SEED, 100000, 1000)
    hi, lo = seed_windows(seed, k)
    res = ctx.bfs_batch([(hi, lo, -1), (hi, lo, 1)], cov, 100000, -1)
    for d, r in zip((-1, 1), res):
        assert_bfs_equal(r, po.bfs(t, k, po.KEY_PACKED, [seed], d, cov, 100000, -1))
    # --maxkmers that BITES at full size: the right-hand pass is cut at 50 000 vertices in the middle of its walk (the cut
    # order is TerminationMode.java:31-47's: the cap is tested at every single insertion), and 17 000 cuts the left-hand one
    # 789 vertices before it would have ended by itself; --maxradius cutting both at once
    for cap_k, cap_r in ((50000, -1), (17000, -1), (100000, 12345)):
        res = ctx.bfs_batch([(hi, lo, -1), (hi, lo, 1)], cov, cap_k, cap_r)
        for d, r in zip((-1, 1), res):
            want = po.bfs(t, k, po.KEY_PACKED, [seed], d, cov, cap_k, cap_r)
            assert_bfs_equal(r, want)
            if cap_r < 0 and d == 1:
                assert len(want["lo"]) == cap_k  # (it did bite)
    assert ctx.stats().solid_sweeps == 0
    ctx.close()
