"""The multi-GPU orchestration (metacherchant_amd/distributed.py) on CPU: world_size 2, gloo.
The ranks run the real exchange logic (bucket by owner -> all-to-all -> count owned keys ->
all-gather thresholded shards -> merge) over a stand-in for metacherchant_amd.Context built on the
oracle, and the result must equal counting everything in one table."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackedContext:
    """Implements the Context methods distributed.py uses, on CPU tensors, with oracle/ functions."""

    def __init__(self, k, mode, records=False, rank=0, binned=False):
        from oracle import pyoracle as po
        self.po, self.k, self.mode = po, k, mode
        self.binned = binned    # ... whose exchange has the binned form (fine buckets), as tables with a second level have
        self.n_binned_runs = 0
        self.ptr_mode, self.fill, self.seeks, self.imports = 1, 0, [], []
        self.t = po.Table()
        self.records = records  # stand in for a context that splits reads into super-k-mer records
        self.rank = rank        # stamped into what this context extracts, so that the receiver can tell the sources apart

    # the record form of the split, degenerate here: one window per record, the key in the record's first word
    def superkmer_capacity(self, n_windows, n_reads):
        return n_windows + 1 if self.records else 0

    def extract_superkmers_dev(self, d_words, d_off, n_reads, n_bases, n_owners, d_recs, d_bins, cap):
        assert d_recs.shape == (cap, 2) and d_bins.shape == (cap,)
        keys = torch.zeros(cap, dtype=torch.int64)
        out = self.extract_keys_dev(d_words, d_off, n_reads, n_bases, n_owners, keys, cap)
        d_recs[:, 0] = keys
        d_recs[:, 1] = 7 + 16 * self.rank
        d_bins[:] = 5
        return out

    def add_superkmers_dev(self, d_recs, d_bins, n):
        # the second words say which rank a record came from; read pointers travel only from the rank that walks (rank 0):
        # its records bring theirs (5), everybody else's arrive with none
        src = (d_recs[:n, 1] - 7) // 16
        assert bool(((d_recs[:n, 1] - 7) % 16 == 0).all())
        if self.ptr_mode & 0x10:  # (the reads of every rank are in the walking rank's store: every record brings its pointer)
            assert bool((d_bins[:n] == 5).all())
        else:
            assert bool((d_bins[:n][src == 0] == 5).all()) and bool((d_bins[:n][src != 0] == 0).all())
        self.add_keys_dev(d_recs[:n, 0].contiguous(), n)

    # the shared read store (ShardedCounter.gather_reads): pointers = where the walking rank's store holds the read, the words follow
    def set_read_pointers(self, mode):
        self.ptr_mode = int(mode)

    def read_store_seek(self, at_bases, reserve_bases=0):
        assert at_bases % 32 == 0
        self.fill = at_bases
        self.seeks.append(at_bases)

    def read_store_tell(self):
        return self.fill

    def read_store_import_dev(self, d_words, n_words, at_bases):
        assert (self.ptr_mode & 0xF) == 1 and at_bases % 32 == 0 and d_words.shape[0] == n_words
        self.imports.append((at_bases, d_words.numpy().copy().view(np.uint64)))

    def _append(self, d_off, n_reads, n_bases):
        """what mc_extract_*_dev does to the (real or deemed) store: the chunk's whole words behind the fill"""
        first = int(d_off.numpy().view(np.uint64)[0])
        self.fill += 32 * ((n_bases + 31) // 32 - first // 32)

    # the binned form: fine bucket of a record = a function of its key; an owner's records in bucket order, a row of counts beside them
    FINE = 8

    def superkmer_fine_buckets(self, n_owners):
        return self.FINE if self.records and self.binned else 0

    @staticmethod
    def _fine_of(keys, fine):
        return ((keys.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)).astype(np.int64) % fine

    def extract_superkmers_binned_dev(self, d_words, d_off, n_reads, n_bases, n_owners, n_fine, d_recs, d_bins, cap, d_fine_counts):
        assert n_fine == self.FINE and d_fine_counts.shape == (n_owners, n_fine) and d_fine_counts.dtype == torch.int32
        out = self.extract_superkmers_dev(d_words, d_off, n_reads, n_bases, n_owners, d_recs, d_bins, cap)
        win = np.zeros(n_owners, dtype=np.uint64)
        for o in range(n_owners):
            a, b = int(out[o]), int(out[o + 1])
            keys = d_recs[a:b, 0].numpy().copy()
            f = self._fine_of(keys, n_fine)
            order = np.argsort(f, kind="stable")
            d_recs[a:b, 0] = torch.from_numpy(keys[order])
            d_fine_counts[o] = torch.from_numpy(np.bincount(f, minlength=n_fine).astype(np.int32))
            win[o] = b - a  # (one window a record here)
        return out, win

    def add_superkmers_binned_dev(self, d_recs, d_bins, n, n_windows, n_fine, part_offsets, d_part_counts):
        po = [int(x) for x in part_offsets]
        assert po[0] == 0 and po[-1] == n and n_windows == n and d_part_counts.shape == (len(po) - 1, n_fine)
        for p in range(len(po) - 1):  # every part in fine-bucket order, its row of counts beside it
            keys = d_recs[po[p]:po[p + 1], 0].numpy()
            f = self._fine_of(keys, n_fine)
            assert bool((np.diff(f) >= 0).all())
            assert np.array_equal(np.bincount(f, minlength=n_fine), d_part_counts[p].numpy())
        self.n_binned_runs += 1
        self.add_superkmers_dev(d_recs, d_bins, n)

    def add_reads_packed_dev(self, d_words, d_off, n_reads, n_bases):
        self.t.count_reads_packed(d_words.numpy().view(np.uint64), d_off.numpy().view(np.uint64), self.k, self.mode)

    def _keys(self, d_words, d_off, n_reads=None):
        po = self.po
        words, off = d_words.numpy().view(np.uint64), d_off.numpy().view(np.uint64)
        if n_reads is not None:
            off = off[:n_reads + 1]  # (d_off may be a view into a longer offsets array: a piece of the reads)
        keys = []
        for r in range(len(off) - 1):
            codes = np.array([(int(words[p >> 5]) >> (62 - 2 * (p & 31))) & 3 for p in range(int(off[r]), int(off[r + 1]))],
                             dtype=np.uint8)
            for i in range(len(codes) - self.k + 1):
                keys.append(po.key(codes[i:i + self.k], self.k, self.mode))
        return np.array(keys, dtype=np.int64)

    def extract_keys_dev(self, d_words, d_off, n_reads, n_bases, n_owners, d_keys, cap, d_hints=None):
        from metacherchant_amd import native
        keys = self._keys(d_words, d_off, n_reads)
        assert int(d_off.numpy().view(np.uint64)[n_reads]) == n_bases
        if n_reads:
            self._append(d_off, n_reads, n_bases)
        owners = np.array([native.key_owner(int(x), n_owners) for x in keys], dtype=np.int64)
        order = np.argsort(owners, kind="stable")
        out = np.zeros(n_owners + 1, dtype=np.uint64)
        out[1:] = np.cumsum(np.bincount(owners, minlength=n_owners))
        d_keys[:len(keys)] = torch.from_numpy(keys[order])
        if d_hints is not None:
            d_hints[:len(keys)] = 0
        return out

    def add_keys_dev(self, d_keys, n, d_hints=None):
        for x in d_keys[:n].tolist():
            self.t.add(int(x), 1)

    def add_pairs_dev(self, d_keys, d_counts, n, d_hints=None):
        for x, c in zip(d_keys[:n].tolist(), d_counts[:n].tolist()):
            self.t.add(int(x), int(c))

    def solid_from_pairs_dev(self, d_keys, d_counts, n, min_cov, d_hints=None):
        kept = 0
        for x, c in zip(d_keys[:n].tolist(), d_counts[:n].tolist()):
            if c >= min_cov and c >= 0:
                self.t.add(int(x), int(c))
                kept += 1
        return kept

    def finalize(self):
        return self.t.size()

    def export_count(self, min_cov):
        return int((self.t.dump()[1] >= min_cov).sum())

    def export_dev(self, min_cov, d_keys, d_counts, cap, d_hints=None):
        k, c = self.t.dump()
        m = c >= min_cov
        n = int(m.sum())
        d_keys[:n] = torch.from_numpy(k[m])
        d_counts[:n] = torch.from_numpy(c[m])
        return n


def _worker(rank, world, port, k, mode, reads, L, n_reads, min_cov, q, records=False, chunk_reads=0, share0=None, count_every=0, pool_slack=None, keep_gb=None,
            binned=False):
    # binned: the contexts offer the binned form of the record exchange ("rank0": only rank 0's does -- the ranks must settle on the flat form)
    # chunk_reads: MC_EXCHANGE_CHUNK_READS (0: the default, one chunk here); share0: reads of rank 0 (None: equal shares)
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if chunk_reads:
            os.environ["MC_EXCHANGE_CHUNK_READS"] = str(chunk_reads)
        if count_every:
            os.environ["MC_EXCHANGE_COUNT_EVERY"] = str(count_every)  # (what bounds the memory of the kept chunks at configs[3]'s size)
        if keep_gb is not None:
            os.environ["MC_EXCHANGE_KEEP_GB"] = repr(keep_gb)  # (a counting run as soon as this much is held)
        if pool_slack is not None:
            os.environ["MC_EXCHANGE_POOL_SLACK"] = str(pool_slack)  # (the one receive buffer too small: later chunks get tensors of their own)
        from metacherchant_amd.distributed import ShardedCounter, split_reads
        from oracle import pyoracle as po
        lo, hi = split_reads(n_reads, world, rank)
        if share0 is not None:  # unequal shares of a world of two: the small one runs out of reads while the other still has chunks
            lo, hi = (0, share0) if rank == 0 else (share0, n_reads)
        mine = reads[lo * L:hi * L]
        words = torch.from_numpy(po.pack(mine).view(np.int64))
        off = torch.from_numpy((np.arange(hi - lo + 1, dtype=np.uint64) * L).view(np.int64))
        ctx = OracleBackedContext(k, mode, records, rank, binned=(binned is True) or (binned == "rank0" and rank == 0))
        sc = ShardedCounter(ctx, torch.device("cpu"))
        sc.add_reads_dev(words, off, hi - lo, (hi - lo) * L, (hi - lo) * (L - k + 1))
        total = sc.finalize()  # (... where the walking rank takes the other ranks' reads into its store)
        if sc.gather_reads:
            # the other rank's packed reads went to rank 0's store: every chunk's words (pad word included) to where that rank's
            # context deemed them -- stretches that follow one another from the position the ranks agreed on
            other = split_reads(n_reads, world, 1) if share0 is None else (share0, n_reads)
            theirs = po.pack(reads[other[0] * L:other[1] * L])
            if rank == 0:
                assert ctx.ptr_mode == 0x11 and len(ctx.seeks) == 1 and ctx.seeks[0] == 0
                at = 32 * ((hi - lo) * L // 32 + 1 + 2 * sc.n_chunks + 2)  # behind rank 0's own stretch (its words, two a chunk, a spare)
                for where, w in sorted(ctx.imports, key=lambda x: x[0]):
                    assert where == at, (where, at)
                    hit = [i for i in range(len(theirs) - len(w) + 1) if theirs[i] == w[0] and np.array_equal(theirs[i:i + len(w)], w)]
                    assert hit, "an imported stretch is not a piece of the other rank's packed reads"
                    at += 32 * (len(w) - 1)
                assert sum(len(w) - 1 for _, w in ctx.imports) >= len(theirs) - 1 or other[1] == other[0]
            else:
                assert ctx.ptr_mode == 0x12 and not ctx.imports and len(ctx.seeks) == 1 and ctx.seeks[0] > 0
        solid = OracleBackedContext(k, mode)
        n_solid = sc.gather_solid(solid, min_cov, dst=0)
        own_keys = ctx.t.dump()[0]
        from metacherchant_amd import native
        assert all(native.key_owner(int(x), world) == rank for x in own_keys[:500])
        if rank == 0:
            sk, scnt = solid.t.dump()
            q.put((total, n_solid, sk, scnt, sc.bytes_sent, sc.n_chunks, sc.n_count_runs, sc.n_pool_misses, sc.pool_rows, ctx.n_binned_runs, sc.fine_buckets))
    finally:
        dist.destroy_process_group()


def _attach_worker(rank, world, port, fails, q):
    # ShardedCounter.attach_shards on two ranks: rank 0 maps (or cannot map) rank 1's table, and BOTH ranks learn which
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from metacherchant_amd.distributed import ShardedCounter

        class Ctx(OracleBackedContext):
            attached = None

            def shard_export(self):
                return bytes([self.rank]) * 128

            def shard_attach(self, handles, own, by_minimizer):
                if fails:
                    raise RuntimeError("GPU 0 cannot read GPU 1's memory (no peer access)")
                self.attached = [h[0] for h in handles]

            def shard_detach(self):
                self.attached = None

        ctx = Ctx(31, 0, True, rank)
        sc = ShardedCounter(ctx, torch.device("cpu"))
        got = []
        for _ in range(2):  # (the second walk takes the first one's answer as read: no further broadcast)
            ok = sc.attach_shards(dst=0)
            got.append((ok, ctx.attached))
            sc.walk_done(dst=0)
        q.put((rank, got, sc.attach_error))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# (k, key mode, record form, MC_EXCHANGE_CHUNK_READS, reads of rank 0 of 300): the last three run the exchange in several chunks,
# with shares so unequal that rank 1 (6, 2 and 0 reads) has chunks without a read while rank 0 still sends
# ... and the last two with a receive buffer half of what arrives (the chunks that do not fit get tensors of their own, and the
# counting run's input is put together after all)
# binned: the record exchange in its binned form (every owner's records in fine-bucket order, a row of counts beside them: the stand-in
# checks order and counts of every part it is handed) -- one chunk, five chunks with one run, with a run every two chunks, with a
# receive buffer too small; "rank0": only one rank's table offers it, and both must take the flat form
@pytest.mark.parametrize("k,mode,records,chunk_reads,share0,count_every,pool_slack,binned", [
    (31, 0, False, 0, None, 0, None, False), (35, 1, False, 0, None, 0, None, False), (31, 0, True, 0, None, 0, None, False),
    (31, 0, True, 64, 294, 0, None, False), (27, 0, True, 50, 298, 0, None, False), (33, 1, False, 64, 300, 0, None, False), (31, 0, True, 64, 294, 2, None, False),
    (31, 0, True, 64, 294, 0, 0.5, False), (33, 1, False, 64, 300, 0, 0.5, False),
    (31, 0, True, 0, None, 0, None, True), (31, 0, True, 64, 294, 0, None, True), (27, 0, True, 50, 298, 2, None, True), (31, 0, True, 64, 294, 0, 0.5, True),
    (31, 0, True, 64, 294, 0, None, "rank0")])
def test_sharded_count_equals_single_table(k, mode, records, chunk_reads, share0, count_every, pool_slack, binned):
    from metacherchant_amd import build
    build.build_lib()  # key_owner comes from the C ABI (host function, no GPU needed)
    from oracle import pyoracle as po
    from tests.helpers import synth_case
    L, n_reads, min_cov = 100, 300, 2
    _, reads, off = synth_case(1, 3000, n_reads, L, 100)
    t = po.Table()
    t.count_reads(reads, off, k, mode)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, k, mode, reads, L, n_reads, min_cov, q, records, chunk_reads, share0, count_every, pool_slack, None, binned))
             for r in range(2)]
    for p in procs:
        p.start()
    total, n_solid, sk, scnt, sent, n_chunks, runs, pool_misses, _, binned_runs, fine = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ok, oc = t.dump()
    assert total == t.size()
    m = oc >= min_cov
    assert n_solid == int(m.sum())
    assert np.array_equal(sk, ok[m]) and np.array_equal(scnt, oc[m])
    assert sent > 0
    assert n_chunks == (1 if not chunk_reads else -(-share0 // chunk_reads)) and (not chunk_reads or n_chunks >= 3)
    # however many chunks travelled: ONE counting run (a run rewrites the rank's whole table), unless the memory bound asks for more
    assert runs == (1 if not count_every else -(-n_chunks // count_every))
    assert (pool_misses > 0) == (pool_slack is not None)  # (every chunk of a run lands in the one buffer unless it was made too small)
    assert (binned_runs, fine) == ((runs, OracleBackedContext.FINE) if binned is True else (0, 0))  # every run binned, or none


def test_receive_pool_holds_what_may_gather_between_counting_runs():
    """the one receive buffer was sized for ALL remaining chunks whatever MC_EXCHANGE_COUNT_EVERY / MC_EXCHANGE_KEEP_GB
    allow to gather (configs[3]: 49 GB pinned through the call beside a 137 GB table).  Five chunks: with a counting run every two
    chunks, or as soon as ~1.5 chunks' bytes are held, the buffer is well under half of the unlimited one -- same table either way."""
    from metacherchant_amd import build
    build.build_lib()
    from oracle import pyoracle as po
    from tests.helpers import synth_case
    L, n_reads, min_cov, k = 100, 300, 2, 31
    _, reads, off = synth_case(1, 3000, n_reads, L, 100)
    t = po.Table()
    t.count_reads(reads, off, k, 0)
    rows = {}
    for name, count_every, keep_gb in (("all", 0, None), ("every2", 2, None), ("keep", 0, 1.5e-5)):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, k, 0, reads, L, n_reads, min_cov, q, True, 64, 294, count_every, None, keep_gb)) for r in range(2)]
        for p in procs:
            p.start()
        total, n_solid, sk, scnt, sent, n_chunks, runs, pool_misses, pool_rows, _, _ = q.get(timeout=120)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert total == t.size() and n_chunks == 5
        rows[name] = (pool_rows, runs)
    assert rows["all"][1] == 1 and rows["every2"][1] == 3 and rows["keep"][1] >= 2, rows
    assert rows["every2"][0] < 0.55 * rows["all"][0] and rows["keep"][0] < 0.7 * rows["all"][0], rows


def test_split_reads_covers_everything():
    from metacherchant_amd.distributed import split_reads
    for n in (0, 1, 7, 100, 101):
        for w in (1, 2, 3, 8):
            spans = [split_reads(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


@pytest.mark.parametrize("fails", [False, True])
def test_attach_shards_tells_every_rank_whether_the_tables_could_be_mapped(fails):
    # (a rank 0 that cannot map the other tables must not leave the others waiting at walk_done: bench.py then gathers instead)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_attach_worker, args=(r, 2, port, fails, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r, (got, err)) for r, got, err in (q.get(timeout=120) for _ in range(2)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        assert [ok for ok, _ in res[r][0]] == [not fails, not fails]
    assert res[0][0][0][1] == (None if fails else [0, 1])  # rank 0 held both tables during the walk
    assert ("no peer access" in res[0][1]) == fails
