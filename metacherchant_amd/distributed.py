"""Multi-GPU counting for the environment-finder path: one process per GPU, RCCL over xGMI.

SURVEY.md section 8(e).  The reference has nothing distributed (one JVM, threads over one
shared map, src/io/IOUtils.java:283-315); what shards here is its data-parallel work list
(src/io/ReadsDispatcher.java:34-53): every rank takes a slice of the read set, turns it into keys
bucketed by owner rank (owner = hash bits disjoint from the slot index, mc_key_owner), the ranks
exchange the buckets with ONE all-to-all (all 7 xGMI links of a GPU busy at once -- unlike a ring),
and each rank counts only the keys it owns.  Counting is then identical to the 1-GPU path.  For
the BFS the thresholded shards (count >= coverage: everything the BFS can ever ask for, since
absent and below-threshold are indistinguishable to `occs >= minOccurences`,
src/algo/OneSequenceCalculator.java:203-204) are sent to the rank that runs the BFS (or all-gathered when every rank does)
and merged into one "solid" table there; a distributed per-level BFS is rejected because the frontier is
typically one vertex wide.

`backend` objects only need the Context methods used below, so the CPU (gloo) tests drive this
file with a stand-in built on the oracle; the product always passes metacherchant_amd.Context.
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist


class ShardedCounter:
    def __init__(self, ctx, device, group=None, bfs_rank=0):
        self.ctx = ctx
        self.device = device
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.bytes_sent = 0
        self.attach_ok = None    # attach_shards: could the walking rank map the other ranks' tables (None: not tried yet)
        self.attach_error = ""
        self.n_chunks = 0
        self.by_minimizer = False
        # Read pointers (include/mcgpu.h mc_set_read_pointers): the BFS rank walks with look-ahead read from ITS OWN
        # reads, so only its records carry pointers; the other ranks keep no read store and send zeros.
        self.bfs_rank = bfs_rank
        # ... but a walk whose look-ahead has one read in W to follow is slow (8 x configs[1]: 146 ms against 7.9 with all of them):
        # the other ranks' packed reads are brought to the walking rank's store (2 bits a base, chunk by chunk beside the
        # records: a tenth of their bytes), every rank works out the pointers of its reads as if they sat there, and every
        # record carries one (MC_EXCHANGE_GATHER_READS=0: the old way, pointers from the walking rank's reads alone).
        self.gather_reads = (self.world > 1 and bfs_rank is not None and hasattr(ctx, "read_store_seek")
                             and os.environ.get("MC_EXCHANGE_GATHER_READS", "1") != "0")
        self._store_fill = 0  # words of the shared store handed out so far (the same number on every rank)
        self._reads_pending = []  # (handle, the sender's words, what the walking rank will import) of the chunks' read transfers
        if self.world > 1 and bfs_rank is not None and hasattr(ctx, "set_read_pointers"):
            if self.gather_reads:
                ctx.set_read_pointers((1 if self.rank == bfs_rank else 2) | 0x10)
            elif self.rank != bfs_rank:
                ctx.set_read_pointers(False)
        # A rank's reads are exchanged in chunks of at most this many reads (MC_EXCHANGE_CHUNK_READS): what bounds the memory of
        # one extraction whatever the size of the rank's share (configs[3]: 125 M reads a rank; DESIGN.md section 6 has the
        # budget); see add_reads_dev for what happens to the chunks.
        self.chunk_reads = int(os.environ.get("MC_EXCHANGE_CHUNK_READS", 32 << 20))
        self.min_chunks = int(os.environ.get("MC_EXCHANGE_MIN_CHUNKS", 4))          # so that transfers and extraction overlap ...
        self.min_chunk_share = int(os.environ.get("MC_EXCHANGE_MIN_SHARE", 1 << 20))  # ... for shares of at least this many reads
        self.count_every = int(os.environ.get("MC_EXCHANGE_COUNT_EVERY", 0))          # chunks per counting run (0: one run for all ...
        # ... that fit: what is kept of the chunks received so far is counted as soon as it passes this many bytes.  One rank of
        # configs[3] receives 11 GB a chunk beside a 137 GB table: kept until the end (44 GB) the one run's scratch leaves no
        # room -- 287 of 288 GB in use, the allocations stall, 1 108 ms against 787 for a run per chunk (profiles/r04_rank_shard_*)
        self.keep_bytes = int(float(os.environ.get("MC_EXCHANGE_KEEP_GB", 12)) * 1e9)
        self.n_count_runs = 0
        self.pool_rows = 0  # rows of the one receive buffer of the last add_reads_dev call that made one (tests)
        self.pool_slack = float(os.environ.get("MC_EXCHANGE_POOL_SLACK", 1.12))  # the one receive buffer: the first chunk's records x chunks x this
        self.fine_buckets = 0   # fine buckets of the binned record exchange of the last add_reads_dev call (0: the flat form ran)
        self.n_pool_misses = 0  # chunks that did not fit it any more (they get tensors of their own; the tests look at it)
        # where a rank's time goes, as its host sees it (seconds, summed over add_reads_dev calls until reset_phases): `extract`
        # = mc_extract_*_dev (synchronous), `exchange_wait` = waiting for transfers that the next chunk's extraction did not
        # cover, `count` = the counting runs up to their enqueueing (finalize waits for them: bench.py books that wait as well)
        self.phase_s = {"extract": 0.0, "exchange_wait": 0.0, "count": 0.0}

    def clear(self):
        """empties the sharded table (every rank calls it): the context's mc_clear, and the table may then be fed in either form"""
        self._import_reads()  # (transfers still running finish first)
        self.ctx.clear()
        self._fed = False
        self._store_fill = 0

    def reset_phases(self):
        for key in self.phase_s:
            self.phase_s[key] = 0.0
        self.bytes_sent = 0

    def add_reads_dev(self, d_words, d_offsets, n_reads, n_bases, max_windows):
        """Counts this rank's reads into the sharded table: extract -> all-to-all -> count owned keys.
        max_windows bounds the number of k-mer occurrences of the local reads (n_bases is always enough).

        The reads go through the exchange in chunks (at least MC_EXCHANGE_MIN_CHUNKS = 4 for a share of a million reads or
        more, and never more than MC_EXCHANGE_CHUNK_READS reads each): the all-to-alls of chunk c are issued asynchronously
        and travel while chunk c + 1 is extracted, what arrives is kept, and the rank counts everything it received in ONE
        run of the pipeline at the end -- a run rewrites the rank's whole table, so one is what a rank wants (round 3: a run
        per chunk, and nothing overlapped) -- as long as what is kept stays under MC_EXCHANGE_KEEP_GB (12): beyond that, and
        after every MC_EXCHANGE_COUNT_EVERY chunks when that is set, a counting run takes what has arrived (configs[3]: a run per
        chunk)."""
        ctx, W = self.ctx, self.world
        if W == 1:
            ctx.add_reads_packed_dev(d_words, d_offsets, n_reads, n_bases)
            return
        # Every rank must issue the same collectives: the number of chunks is the largest any rank needs (one all-reduce, the
        # only host round trip besides one per chunk for the record counts), and a rank that has run out of reads still
        # takes part in the remaining exchanges with zero counts.
        want = max(1, -(-int(n_reads) // self.chunk_reads))
        if n_reads >= self.min_chunk_share:
            want = max(want, self.min_chunks)
        sk = hasattr(ctx, "superkmer_capacity") and ctx.superkmer_capacity(max(int(max_windows), 1), max(int(n_reads), 1)) != 0
        # The binned form of the record exchange (include/mcgpu.h mc_extract_superkmers_binned_dev): the sender puts every owner's
        # records in the order of the level-1 buckets of the owner's counting run, which then starts at its second level.  The
        # bucket count comes from the table's layout; every rank must use the same one (the counts travel as rows of that length),
        # so the ranks agree on it -- in the all-reduce they need anyway -- and take the flat form when they differ.
        f_mine = int(ctx.superkmer_fine_buckets(W)) if sk and hasattr(ctx, "superkmer_fine_buckets") else 0
        nc = torch.tensor([want, f_mine, -f_mine], dtype=torch.int64, device=self.device)
        dist.all_reduce(nc, op=dist.ReduceOp.MAX, group=self.group)
        nc = [int(x) for x in nc.cpu().tolist()]
        n_chunks = nc[0]
        fine = nc[1] if nc[1] == -nc[2] else 0
        self.fine_buckets = fine  # (of the last call; 0: flat records or keys)
        self.n_chunks = n_chunks  # (of the last call: the tests look at it)
        bounds = [n_reads * c // n_chunks for c in range(n_chunks + 1)]
        if n_chunks > 1:  # the chunk boundaries' base offsets in one copy
            idx = torch.tensor(bounds[1:-1], dtype=torch.int64, device=self.device)
            base_at = [0] + [int(x) for x in d_offsets[idx].cpu().tolist()] + [int(n_bases)]
        else:
            base_at = [0, int(n_bases)]
        # (records are dealt to the owners of their minimizers, keys to the owners of their own hashes: a table fed in both
        # forms would hold a k-mer on two ranks, and attach_shards could name only one rule -- ADVICE r4)
        if getattr(self, "_fed", False) and self.by_minimizer != sk:
            raise RuntimeError("ShardedCounter: this table was fed %s before and would now be fed %s: clear it first" % (
                "super-k-mer records" if self.by_minimizer else "keys", "super-k-mer records" if sk else "keys"))
        self.by_minimizer = sk
        self._fed = True
        if self.gather_reads:
            # every rank's stretch of the walking rank's store for this call: its words, a pad word a chunk, a spare -- one all-gather
            mine = torch.tensor([(int(n_bases) + 31) // 32 + 2 * n_chunks + 2], dtype=torch.int64, device=self.device)
            sizes = torch.empty(W, dtype=torch.int64, device=self.device)
            dist.all_gather_into_tensor(sizes, mine, group=self.group)
            sizes = [int(x) for x in sizes.cpu().tolist()]
            at = self._store_fill + sum(sizes[:self.rank])
            self._store_fill += sum(sizes)
            ctx.read_store_seek(32 * at, 32 * self._store_fill if self.rank == self.bfs_rank else 0)
        # Read pointers travel only FROM the rank that walks (they lead into its read store; the others keep none and used to
        # send an array of zeros: a fifth of the bytes of seven ranks out of eight)
        with_ptrs = self.bfs_rank is None or self.rank == self.bfs_rank or self.gather_reads
        ptr_sources = list(range(W)) if self.bfs_rank is None or self.gather_reads else [self.bfs_rank]
        pending = []  # per chunk: [handles, send buffers (kept alive while the transfers run), received payload, received pointers, n]
        self.n_count_runs = 0
        # What arrives is received straight into ONE buffer sized from the first chunk (x chunks left, + 12 %), so that the
        # one counting run needs no second copy of everything (torch.cat doubled the peak); a chunk that does not
        # fit any more gets tensors of its own and the run's input is concatenated after all.
        pool = {"recv": None, "recv_p": None, "at": 0}

        def settle(entry):
            """waits for one chunk's transfers and lets go of its send buffers (capacity-sized: round 4 kept every chunk's until the end)"""
            if entry[0]:
                t_w = time.perf_counter()
                for h in entry[0]:
                    h.wait()
                self.phase_s["exchange_wait"] += time.perf_counter() - t_w
                entry[0] = []
            entry[1] = None

        def count_pending():
            if not pending:
                return
            for entry in pending:
                settle(entry)
            t_w = time.perf_counter()
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            self.phase_s["exchange_wait"] += time.perf_counter() - t_w
            n_all = sum(x[4] for x in pending)
            binned = [x[6] for x in pending]
            in_pool = pool["recv"] is not None and all(x[5] for x in pending)
            if len(pending) == 1:
                recv, recv_p = pending[0][2], pending[0][3]
            elif in_pool:  # the chunks lie back to back in the pool, from its start
                recv, recv_p = pool["recv"], pool["recv_p"]
            else:
                recv = torch.cat([x[2][:x[4]] for x in pending]) if n_all else pending[0][2]
                recv_p = torch.cat([x[3][:x[4]] for x in pending]) if n_all else pending[0][3]
            pending.clear()
            pool["at"] = 0
            t_c = time.perf_counter()
            # (a rank that received nothing still calls: the context must know its pipeline buffers were reused, mcgpu.hip)
            if sk and fine and n_all:
                # every (chunk, source) is a part in fine-bucket order; the parts lie back to back as they were received
                lens = [n for b in binned for n in b[0]]
                part_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
                part_counts = torch.cat([b[1] for b in binned]) if len(binned) > 1 else binned[0][1]
                ctx.add_superkmers_binned_dev(recv, recv_p, n_all, sum(b[2] for b in binned), fine, part_off, part_counts)
            elif sk:
                ctx.add_superkmers_dev(recv, recv_p, n_all)
            else:
                ctx.add_keys_dev(recv, n_all, recv_p)
            self.phase_s["count"] += time.perf_counter() - t_c
            self.n_count_runs += 1

        for c in range(n_chunks):
            a, b = bounds[c], bounds[c + 1]
            nb = base_at[c + 1] - base_at[c]
            # (offsets are absolute in the rank's buffer: a chunk is the reads [a, b) with the bases up to base_at[c + 1])
            pending.append(self._exchange_chunk(sk, d_words, d_offsets[a:], b - a, base_at[c + 1], nb if sk else min(int(max_windows), nb),
                                                with_ptrs, ptr_sources, pool, n_chunks - c, fine, base_at[c]))
            if len(pending) >= 2:
                settle(pending[-2])  # (its transfers had this chunk's extraction to travel in)
            # what is held: the received payloads, and the send buffers of the chunk whose transfers are still running
            kept = sum(x[4] * ((16 if sk else 8) + 4) for x in pending) + sum(t.numel() * t.element_size() for x in pending if x[1] for t in x[1] if t is not None)
            if (self.count_every and len(pending) >= self.count_every) or kept >= self.keep_bytes:
                count_pending()
        count_pending()

    def _counts(self, send_counts, send_windows, reads_words, reads_at):
        """The one host round trip of an exchange: what every rank will send me -- records (keys); the windows they hold (binned
        form); and, to the walking rank, how many words of packed reads and where in its store they belong."""
        W = self.world
        rows = [[int(send_counts[o]), int(send_windows[o]), int(reads_words) if o == self.bfs_rank else 0, int(reads_at)] for o in range(W)]
        sc = torch.tensor(rows, dtype=torch.int64, device=self.device)
        rc = torch.empty((W, 4), dtype=torch.int64, device=self.device)
        dist.all_to_all_single(rc, sc, group=self.group)
        rc = rc.cpu().tolist()
        return [int(x[0]) for x in rc], sum(int(x[1]) for x in rc), [int(x[2]) for x in rc], [int(x[3]) for x in rc]

    def _exchange_chunk(self, sk, d_words, d_offsets, n_reads, n_bases_end, room, with_ptrs, ptr_sources, pool, chunks_left, fine=0, n_bases_start=0):
        """One chunk: extract (super-k-mer records of 16 bytes for packed keys, k >= 23 -- a seventh of the bytes of one key
        per window --, else keys), one exchange of the counts, then the payload and the read pointers as asynchronous
        all-to-alls.  Returns (handles, send buffers, received payload, received pointers, n received): the caller waits."""
        ctx, W = self.ctx, self.world
        if sk:
            cap = max(ctx.superkmer_capacity(max(int(room), 1), max(int(n_reads), 1)), 1)
            send = torch.empty((cap, 2), dtype=torch.int64, device=self.device)   # 16-byte records
        else:
            cap = max(int(room), 1)
            send = torch.empty(cap, dtype=torch.int64, device=self.device)
        send_p = torch.empty(cap, dtype=torch.int32, device=self.device)          # the read pointers (of the records' first windows)
        t_e = time.perf_counter()
        binned = sk and fine
        fc = None
        win = np.zeros(W, dtype=np.uint64)
        if binned:  # a row of `fine` counts for every owner: how its records split into fine buckets
            fc = (torch.empty if n_reads else torch.zeros)((W, fine), dtype=torch.int32, device=self.device)
        # the chunk's packed reads: words [w0, w1] of this rank's buffer (the pad word behind them included), deemed to sit at
        # reads_at in the walking rank's store -- where the extraction below works its pointers out for
        reads_words, reads_at, w0 = 0, 0, 0
        if self.gather_reads and n_reads and self.rank != self.bfs_rank:
            w0 = int(n_bases_start) // 32
            reads_words = (int(n_bases_end) + 31) // 32 - w0 + 1
            reads_at = ctx.read_store_tell()
        if n_reads:
            if binned:
                off, win = ctx.extract_superkmers_binned_dev(d_words, d_offsets, n_reads, n_bases_end, W, fine, send, send_p, cap, fc)
            elif sk:
                off = ctx.extract_superkmers_dev(d_words, d_offsets, n_reads, n_bases_end, W, send, send_p, cap)
            else:
                off = ctx.extract_keys_dev(d_words, d_offsets, n_reads, n_bases_end, W, send, cap, send_p)
        else:
            off = np.zeros(W + 1, dtype=np.uint64)
        self.phase_s["extract"] += time.perf_counter() - t_e
        send_counts = [int(off[o + 1] - off[o]) for o in range(W)]
        recv_counts, recv_windows, reads_in, reads_in_at = self._counts(send_counts, [int(x) for x in win], reads_words, reads_at)
        n_recv, n_send = sum(recv_counts), int(off[W])
        # pointers: only the ranks in ptr_sources send theirs; what comes from the others is zero (no pointer)
        all_send = len(ptr_sources) == W
        if pool["recv"] is None and chunks_left > 1:  # the first of several chunks sizes the buffer for all that are held together
            # ... which is not all of them where a counting run comes every few chunks (MC_EXCHANGE_COUNT_EVERY) or as soon as
            # MC_EXCHANGE_KEEP_GB are held (configs[3]: 11 GB a chunk beside a 137 GB table -- a pool for all four chunks pinned 49 GB
            # through the whole call and brought back the stalls the limit was made for): at most what the limit lets
            # gather, plus the chunk that crosses it
            held = chunks_left if not self.count_every else min(chunks_left, self.count_every)
            row_bytes = (16 if sk else 8) + 4
            room_all = min(n_recv * held * self.pool_slack, self.keep_bytes / row_bytes + n_recv * self.pool_slack)
            room_all = max(int(room_all) + 1024, 1)
            self.pool_rows = room_all
            pool["recv"] = torch.empty((room_all, 2) if sk else room_all, dtype=torch.int64, device=self.device)
            pool["recv_p"] = torch.zeros(room_all, dtype=torch.int32, device=self.device)
            pool["at"] = 0
        in_pool = pool["recv"] is not None and pool["at"] + n_recv <= pool["recv"].shape[0]
        if in_pool:
            at0 = pool["at"]
            recv, recv_p = pool["recv"][at0:at0 + max(n_recv, 1)], pool["recv_p"][at0:at0 + max(n_recv, 1)]
            if not all_send:
                recv_p.zero_()  # (the pool is reused by the next counting run's chunks)
            pool["at"] = at0 + n_recv
        else:
            self.n_pool_misses += pool["recv"] is not None
            recv = torch.empty((max(n_recv, 1), 2) if sk else max(n_recv, 1), dtype=torch.int64, device=self.device)
            recv_p = (torch.empty if all_send else torch.zeros)(max(n_recv, 1), dtype=torch.int32, device=self.device)
        hs = [dist.all_to_all_single(recv[:n_recv], send[:n_send], output_split_sizes=recv_counts, input_split_sizes=send_counts,
                                     group=self.group, async_op=True)]
        if all_send:
            hs.append(dist.all_to_all_single(recv_p[:n_recv], send_p[:n_send], output_split_sizes=recv_counts, input_split_sizes=send_counts,
                                             group=self.group, async_op=True))
            ptr_bytes = 4 * (n_send - send_counts[self.rank])
        else:
            # the pointer stream as an all-to-all in which only ptr_sources have anything to send: the receiver's slice for source
            # r starts where r's records start
            p_send = send_counts if with_ptrs else [0] * W
            p_recv = [recv_counts[r] if r in ptr_sources else 0 for r in range(W)]
            src = ptr_sources[0]
            at = sum(recv_counts[:src])
            hs.append(dist.all_to_all_single(recv_p[at:at + recv_counts[src]], send_p[:n_send if with_ptrs else 0], output_split_sizes=p_recv,
                                             input_split_sizes=p_send, group=self.group, async_op=True))
            ptr_bytes = 4 * (n_send - send_counts[self.rank]) if with_ptrs else 0
        self.bytes_sent += (16 if sk else 8) * (n_send - send_counts[self.rank]) + ptr_bytes
        if self.gather_reads:
            # the packed reads, to the walking rank alone: an all-to-all in which nobody else receives.  Nothing needs them before
            # the walk: the transfer is waited for in finalize(), behind the counting run (_import_reads).
            is_dst = self.rank == self.bfs_rank
            got = torch.empty(max(sum(reads_in), 1), dtype=torch.int64, device=self.device) if is_dst else torch.empty(1, dtype=torch.int64, device=self.device)
            src = d_words[w0:w0 + reads_words]
            h = dist.all_to_all_single(got[:sum(reads_in) if is_dst else 0], src, output_split_sizes=reads_in if is_dst else [0] * W,
                                       input_split_sizes=[reads_words if o == self.bfs_rank else 0 for o in range(W)], group=self.group, async_op=True)
            self.bytes_sent += 8 * reads_words
            imports, at = [], 0
            if is_dst:
                for r in range(W):
                    if reads_in[r]:
                        imports.append((got[at:at + reads_in[r]], reads_in[r], reads_in_at[r]))
                    at += reads_in[r]
            self._reads_pending.append((h, src, imports))
        info = None
        if binned:  # the counts: row r of what arrives is source r's row for me
            recv_fc = torch.empty((W, fine), dtype=torch.int32, device=self.device)
            hs.append(dist.all_to_all_single(recv_fc, fc, group=self.group, async_op=True))
            self.bytes_sent += 4 * fine * (W - 1)
            info = (recv_counts, recv_fc, recv_windows)
        return [hs, (send, send_p, fc), recv, recv_p, n_recv, in_pool, info]

    def _import_reads(self):
        """the other ranks' packed reads into the walking rank's store, where their records' pointers lead (every rank calls: the
        senders let go of their buffers)"""
        if not self._reads_pending:
            return
        t_w = time.perf_counter()
        for h, _, _ in self._reads_pending:
            h.wait()
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)  # (the context copies on its own stream)
        self.phase_s["exchange_wait"] += time.perf_counter() - t_w
        for _, _, imports in self._reads_pending:
            for words, n_words, at_bases in imports:
                self.ctx.read_store_import_dev(words, n_words, at_bases)
        self._reads_pending = []

    def finalize(self):
        """Local distinct keys; the sum over ranks is the table size (owners are disjoint)."""
        self._import_reads()
        n = self.ctx.finalize()
        if self.world == 1:
            return n
        t = torch.tensor([n], dtype=torch.int64, device=self.device)
        dist.all_reduce(t, group=self.group)
        return int(t.item())

    def attach_shards(self, dst=0):
        """The walk in place (include/mcgpu.h mc_shard_*): every rank describes its counting table in 128 bytes (geometry + an
        IPC handle of its memory), one all-gather brings the descriptions to everybody, and rank `dst` maps the other ranks'
        tables: its bfs / bfs_batch then look every k-mer up in its owner's table over xGMI.  Nothing is exported, copied or
        rebuilt (gather_solid: 0.4 + ~14 + 3.2 ms at 8 x configs[1], and a copy that does not fit at configs[3]'s size).  The
        other ranks must leave their tables alone until walk_done().  Returns False -- on every rank alike -- when `dst` cannot map
        them (attach_error says why): the caller then gathers the solid k-mers (gather_solid)."""
        ctx, W = self.ctx, self.world
        if W == 1:
            return True
        self._import_reads()
        if self.attach_ok is False:
            return False
        mine = torch.frombuffer(bytearray(ctx.shard_export()), dtype=torch.uint8).to(self.device)
        allh = torch.empty(W * mine.numel(), dtype=torch.uint8, device=self.device)
        dist.all_gather_into_tensor(allh, mine, group=self.group)
        ok = 1
        if self.rank == dst:
            raw = allh.cpu().numpy().tobytes()
            n = mine.numel()
            try:
                ctx.shard_attach([raw[i * n:(i + 1) * n] for i in range(W)], self.rank, self.by_minimizer)
            except Exception as e:  # (no peer mapping between these devices / processes: the caller gathers the solid k-mers instead)
                if self.attach_ok:  # it worked before: this is not the machine's answer but a fault
                    raise
                ok, self.attach_error = 0, str(e)
        if self.attach_ok is None:  # the first walk: every rank learns whether `dst` could map the tables (later walks take that as read)
            flag = torch.tensor([ok], dtype=torch.int64, device=self.device)
            dist.broadcast(flag, src=dst, group=self.group)
            self.attach_ok = bool(int(flag.item()))
        return self.attach_ok

    def walk_done(self, dst=0):
        """behind the walk: rank `dst` gives the mappings up, and nobody touches its table before that"""
        if self.world == 1 or not self.attach_ok:
            return
        if self.rank == dst:
            self.ctx.shard_detach()
        dist.barrier(group=self.group)

    def gather_solid(self, solid_ctx, min_cov, dst=0):
        """Brings the (key, count >= min_cov, hint) entries of every shard to rank `dst` (direct sends) and builds solid_ctx's
        BFS table from them there; dst=None: an all-gather, and every rank builds it.  Returns the number of solid k-mers."""
        ctx, W = self.ctx, self.world
        if W == 1:
            return None  # the caller BFSes on ctx itself
        n_local = ctx.export_count(min_cov)  # no sweep when the context tracked this threshold (set_coverage_hint)
        nt = torch.tensor([n_local], dtype=torch.int64, device=self.device)
        sizes_t = torch.empty(W, dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(sizes_t, nt, group=self.group)
        sizes = [int(x) for x in sizes_t.cpu().tolist()]  # (one copy, not one per rank)
        if dst is not None:
            # Only `dst` builds a table: every rank SENDS its shard straight to it (all-to-alls in which the other ranks receive
            # nothing: seven point-to-point transfers into dst over seven xGMI links at once, exact sizes) -- an all-gather
            # would put all W shards, padded to the largest, on every rank (8 x 0.7 GB each way at configs[1] x 8, through
            # whatever rings the library forms).
            keys = torch.empty(max(n_local, 1), dtype=torch.int64, device=self.device)
            cnts = torch.empty(max(n_local, 1), dtype=torch.int16, device=self.device)
            hints = torch.empty(max(n_local, 1), dtype=torch.int32, device=self.device)
            got = ctx.export_dev(min_cov, keys, cnts, max(n_local, 1), hints)
            assert got == n_local
            total = sum(sizes) if self.rank == dst else 0
            all_k = torch.empty(max(total, 1), dtype=torch.int64, device=self.device)
            all_c = torch.empty(max(total, 1), dtype=torch.int16, device=self.device)
            all_h = torch.empty(max(total, 1), dtype=torch.int32, device=self.device)
            send = [n_local if o == dst else 0 for o in range(W)]
            recv = sizes if self.rank == dst else [0] * W
            dist.all_to_all_single(all_k[:total], keys[:n_local], output_split_sizes=recv, input_split_sizes=send, group=self.group)
            # counts travel as bytes: neither RCCL nor gloo has a 16-bit integer type
            dist.all_to_all_single(all_c.view(torch.uint8)[:2 * total], cnts.view(torch.uint8)[:2 * n_local],
                                   output_split_sizes=[2 * x for x in recv], input_split_sizes=[2 * x for x in send], group=self.group)
            dist.all_to_all_single(all_h[:total], hints[:n_local], output_split_sizes=recv, input_split_sizes=send, group=self.group)
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            self.bytes_sent += 14 * (n_local if self.rank != dst else 0)
            if self.rank == dst:
                if hasattr(solid_ctx, "share_read_store"):
                    solid_ctx.share_read_store(ctx)  # the pointers refer to this rank's reads
                kept = solid_ctx.solid_from_pairs_dev(all_k, all_c, total, min_cov, all_h)
                assert kept == sum(sizes), (kept, sizes)
            return sum(sizes)
        mx = max(max(sizes), 1)
        keys = torch.zeros(mx, dtype=torch.int64, device=self.device)
        cnts = torch.full((mx,), -1, dtype=torch.int16, device=self.device)  # -1 marks the padding behind a short shard
        hints = torch.zeros(mx, dtype=torch.int32, device=self.device)
        got = ctx.export_dev(min_cov, keys, cnts, mx, hints)
        assert got == n_local
        all_k = torch.empty(W * mx, dtype=torch.int64, device=self.device)
        all_c = torch.empty(W * mx, dtype=torch.int16, device=self.device)
        all_h = torch.empty(W * mx, dtype=torch.int32, device=self.device)
        dist.all_gather_into_tensor(all_k, keys, group=self.group)
        # counts travel as bytes: neither RCCL nor gloo has a 16-bit integer type
        dist.all_gather_into_tensor(all_c.view(torch.uint8), cnts.view(torch.uint8), group=self.group)
        dist.all_gather_into_tensor(all_h, hints, group=self.group)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        if dst is None or self.rank == dst:
            # owners are disjoint, so the shards simply sit side by side: one pass over the gathered arrays puts
            # them into the BFS table (padding skipped by its count); the pointers refer to this rank's reads
            if hasattr(solid_ctx, "share_read_store"):
                solid_ctx.share_read_store(ctx)
            kept = solid_ctx.solid_from_pairs_dev(all_k, all_c, W * mx, min_cov, all_h)
            assert kept == sum(sizes), (kept, sizes)
        return sum(sizes)


def split_reads(n_reads_total, world, rank):
    """Contiguous equal ranges of the read set, the GPU analogue of ReadsDispatcher's work list."""
    base, rem = divmod(n_reads_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
