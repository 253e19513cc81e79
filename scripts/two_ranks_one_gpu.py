"""Both ranks of a world-2 job on ONE GPU (gloo carries the collectives, staging CUDA tensors through
the host): exercises ShardedCounter with real Contexts where only one GPU is available.
Usage: python scripts/two_ranks_one_gpu.py [k]"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sys.path.insert(0, ROOT)


def worker(rank, world, port, k, q):
    try:
        _worker(rank, world, port, k, q)
    except BaseException:
        import traceback
        q.put(("error", rank, traceback.format_exc()))  # the parent reports it at once instead of timing out
        raise


def _worker(rank, world, port, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import metacherchant_amd as m
    from metacherchant_amd.distributed import ShardedCounter, split_reads
    from oracle import pyoracle as po
    from tests.helpers import synth_case, seed_windows
    dev = torch.device("cuda:0")
    L, n_reads, contigs, clen, err, cov, hint = 150, 40000, 2, 100000, 80, 4, False
    if VARIANT:  # (soak: python scripts/two_ranks_one_gpu.py <k> <variant>)
        rng = np.random.default_rng(VARIANT)
        L = int(rng.choice([70, 100, 150, 250]))
        n_reads = int(rng.integers(20000, 90000))
        contigs, clen = int(rng.integers(1, 4)), int(rng.choice([3000, 30000, 100000, 300000]))
        err, cov, hint = int(rng.choice([0, 50, 100, 300])), int(rng.integers(2, 6)), bool(rng.integers(0, 2))
    genome, reads, off = synth_case(contigs, clen, n_reads, L, err)
    lo, hi = split_reads(n_reads, world, rank)
    if not VARIANT or VARIANT % 2:
        # the exchange in several chunks (configs[3] needs four a rank), the shares unequal: rank 1 has fewer reads than
        # chunks, so some of its chunks are empty while rank 0 still sends -- and rank 0's chunks start at absolute offsets
        share0 = n_reads - 3
        lo, hi = (0, share0) if rank == 0 else (share0, n_reads)
        os.environ["MC_EXCHANGE_CHUNK_READS"] = str(-(-share0 // 5))
    mine = reads[lo * L:hi * L]
    words = torch.from_numpy(po.pack(mine).view(np.int64)).to(dev)
    offs = torch.from_numpy((np.arange(hi - lo + 1, dtype=np.uint64) * L).view(np.int64)).to(dev)
    mode = m.KEY_PACKED if k <= 31 else m.KEY_POLY
    ctx = m.Context(k, mode, 0, 0)
    if hint:
        ctx.set_coverage_hint(cov)
    sc = ShardedCounter(ctx, dev)
    sc.add_reads_dev(words, offs, hi - lo, (hi - lo) * L, (hi - lo) * (L - k + 1))
    if not VARIANT or VARIANT % 2:
        assert sc.n_chunks == 5, sc.n_chunks
    total = sc.finalize()
    in_place = os.environ.get("TWO_RANKS_WALK", "shards") != "gather"
    if in_place:  # rank 0 maps rank 1's table (an IPC handle through the all-gather) and walks over both
        sc.attach_shards(dst=0)
        solid, n_solid = ctx, ctx.export_count(cov)
        nt = torch.tensor([n_solid], dtype=torch.int64, device=dev)
        dist.all_reduce(nt)
        n_solid = int(nt.item())
    else:
        solid = m.Context(k, mode, 0, 0) if rank == 0 else None
        n_solid = sc.gather_solid(solid, cov, dst=0)
    if rank == 0:
        t = po.Table()
        t.count_reads(reads, off, k, mode)
        ok, oc = t.dump()
        assert total == t.size(), (total, t.size())
        assert n_solid == int((oc >= cov).sum())
        # the BFS-only context holds no counts of its own (mc_solid_from_pairs_dev): what it knows shows in the walks
        seed = genome[min(30000, clen // 3):min(30000, clen // 3) + 300]
        shi, slo = seed_windows(seed, k)
        rounds = {}
        for d in (1, -1, 0):
            got = solid.bfs(shi, slo, d, cov, 20000, -1)
            want = po.bfs(t, k, mode, [seed], d, cov, 20000, -1)
            assert (got is None) == (want is None)
            if got is None:
                continue
            rounds[d] = (got["rounds"], got["levels"])
            assert np.array_equal(got["lo"], want["lo"]) and np.array_equal(got["dist"], want["dist"])
            assert np.array_equal(got["cov"], want["cov"]) and np.array_equal(got["hi"], want["hi"])
        if in_place and sc.gather_reads and VARIANT and VARIANT % 2 == 0 and rounds and mode == m.KEY_PACKED and sc.by_minimizer:
            # (records: k >= 23.  Keys -- one a window, k < 23 and hash keys -- leave their pointers by another rule than one context's
            # direct counting does, and the rounds differ by up to 2 x either way)
            # The other rank's packed reads were brought to this rank's store and its records' pointers lead there: the walk's
            # look-ahead is as good as one context's over all the reads -- about as many verification rounds for the same levels.
            # (Pointers that led anywhere else would only cost rounds, never a result: this is where it would show.)
            one = m.Context(k, mode, 0, 0)
            if hint:
                one.set_coverage_hint(cov)
            one.add_reads_packed(po.pack(reads), off)
            one.finalize()
            for d, (r_sh, lv) in rounds.items():
                ref = one.bfs(shi, slo, d, cov, 20000, -1)
                assert ref["levels"] == lv
                print("walk %d: %d levels, %d rounds over two ranks' tables, %d in one context" % (d, lv, r_sh, ref["rounds"]), flush=True)
                # (where the look-ahead works at all: a walk through thin, error-ridden coverage takes several rounds a level either
                # way, and how many varies from run to run)
                if 4 * ref["rounds"] < lv:
                    assert r_sh <= 1.5 * ref["rounds"] + 16, (d, lv, r_sh, ref["rounds"])
            one.close()
        if sc.fine_buckets:  # the binned form of the record exchange: every counting run of this rank started at its second level
            assert ctx.stats().binned_runs == sc.n_count_runs > 0, (ctx.stats().binned_runs, sc.n_count_runs)
        q.put(("ok", total, n_solid, sc.bytes_sent, sc.n_chunks, "fine buckets: %d" % sc.fine_buckets, ctx.superkmer_capacity(1000, 10) > 0))
    if in_place:
        sc.walk_done(dst=0)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 31
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, k, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    if res[0] != "ok":
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():
                p.terminate()
        raise SystemExit("rank %d failed:\n%s" % (res[1], res[2]))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, p.exitcode
    print("two ranks on one GPU, k=%d:" % k, res)
