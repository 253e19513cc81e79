"""What one rank of an N-GPU job computes per step, timed on one GPU (the collectives themselves are not in it): extraction
of super-k-mer records bucketed for N owners, chunk by chunk as metacherchant_amd/distributed.py sends them; ONE counting
run over as many records as the rank receives (here: its own, all owners together -- a rank receives about what it sends);
finalize; and, on the rank that walks, the walk over the counting table(s) in place (mc_shard_attach: no export, no
gathered copy, no second table -- round 3 spent 0.4 + 3.2 ms there and could not fit configs[3]).

  python scripts/rank_phases.py [n_owners]                          bench scale: 10 M x 150 bp, E1; the one-GPU step beside it
  python scripts/rank_phases.py 8 --shard 125000000 --contigs 125   one rank of configs[3]: 125 M reads' worth of records over the
                                                                    rank's eighth of the key space (125 of the 1000 contigs at the
                                                                    same 30-fold depth); device memory in use after every phase;
                                                                    --check: sampled loci and the walk against the oracle"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metacherchant_amd as m

ap = argparse.ArgumentParser()
ap.add_argument("owners", nargs="?", type=int, default=8)
ap.add_argument("--shard", type=int, default=0, help="reads of the rank (0: bench scale, 10 M reads, 3 repetitions)")
ap.add_argument("--contigs", type=int, default=10)
ap.add_argument("--chunk", type=int, default=32 << 20, help="reads per exchange chunk at most (MC_EXCHANGE_CHUNK_READS)")
ap.add_argument("--min-chunks", type=int, default=4, help="chunks at least (MC_EXCHANGE_MIN_CHUNKS: transfers overlap the next chunk's extraction)")
ap.add_argument("--count-every", type=int, default=0, help="chunks per counting run (MC_EXCHANGE_COUNT_EVERY; 0: one run for all)")
ap.add_argument("--keep-gb", type=float, default=12.0, help="received records kept at most before a counting run takes them (MC_EXCHANGE_KEEP_GB)")
ap.add_argument("--flat", action="store_true", help="the flat form of the record exchange (mc_extract_superkmers_dev / mc_add_superkmers_dev) instead of the binned one")
ap.add_argument("--own-ptrs-only", action="store_true", help="MC_EXCHANGE_GATHER_READS=0: only the walking rank's reads are in its store -- the records of one read in W keep their pointers (default: every rank's packed reads are brought there, distributed.py, and every record carries one)")
ap.add_argument("--check", action="store_true", help="--shard: sampled loci and the walk against the oracle (replays the read generator on the host)")
args = ap.parse_args()
W = args.owners
dev = torch.device("cuda:0")
k, L, cov = 31, 150, 5
clen, err = 5_000_000, 100
GENOME_SEED, READ_SEED = 20240531, 42
R = args.shard or 10_000_000
contigs = args.contigs
n_chunks = max(args.min_chunks, -(-R // args.chunk))
bounds = [R * c // n_chunks for c in range(n_chunks + 1)]
chunk_max = max(b - a for a, b in zip(bounds, bounds[1:]))


def used():
    f, t = torch.cuda.mem_get_info()
    return (t - f) / 1e9


def timed(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return r, 1e3 * (time.perf_counter() - t0)


windows_total = R * (L - k + 1)
est_distinct = int(contigs * clen + windows_total * (1.0 - (1.0 - err / 10000.0) ** k))
ctx = m.Context(k, m.KEY_PACKED, 0, est_distinct + (1 << 20))
ctx.set_coverage_hint(cov)
if not args.own_ptrs_only:  # (as ShardedCounter sets it on the walking rank: the store holds every rank's reads, every record brings a pointer)
    ctx.set_read_pointers(1 | 0x10)
    if args.shard:  # ... and has room for them: W shares of packed reads (configs[3]: 8 x 4.7 GB), what the walking rank's memory must hold
        ctx.read_store_seek(0, 32 * W * ((R * L + 31) // 32 + 2 * n_chunks + 2))
print("table for %.0f M expected keys: %.1f GB in use; %d chunks of <= %d reads" % (est_distinct / 1e6, used(), n_chunks, chunk_max), flush=True)
d_words = torch.empty((chunk_max * L + 31) // 32 + 1, dtype=torch.int64, device=dev)
d_off = torch.empty(chunk_max + 1, dtype=torch.int64, device=dev)
seed = m.native.synth_genome(GENOME_SEED, 100000, 1000)
sv = []
for i in range(len(seed) - k + 1):
    v = 0
    for c in seed[i:i + k]:
        v = (v << 2) | int(c)
    sv.append(v)
hi, lo = np.zeros(len(sv), dtype=np.uint64), np.array(sv, dtype=np.uint64)
jobs = [(hi, lo, -1), (hi, lo, 1)]

for rep in range(2 if args.shard else 3):  # (the first repetition pays for fresh device memory: 20 GB buffers zeroed by the driver)
    ctx.clear()
    ms_ext = ms_add = 0.0
    n_rec = 0
    peak = used()
    kept, kept_p, kept_fc, kept_len, kept_win = [], [], [], [], []
    pool = {"recs": None, "ptrs": None, "at": 0}
    runs = 0
    fine = 0 if args.flat else ctx.superkmer_fine_buckets(W)  # the binned form of the exchange where the table has a second level

    def count_kept():
        global ms_add, peak, runs
        if not kept:
            return
        nr = sum(x.shape[0] for x in kept)
        if pool["recs"] is not None and pool["at"] == nr:  # (the chunks lie back to back in the one buffer: distributed.py's receive pool)
            recv, recv_p = pool["recs"], pool["ptrs"]
        else:
            recv = torch.cat(kept) if len(kept) > 1 else kept[0]
            recv_p = torch.cat(kept_p) if len(kept_p) > 1 else kept_p[0]
        kept.clear()
        kept_p.clear()
        pool["at"] = 0
        peak = max(peak, used())
        if fine:  # every (chunk, owner) piece is a part in fine-bucket order, as if W ranks had sent them
            part_off = np.concatenate([[0], np.cumsum(kept_len)]).astype(np.uint64)
            fc_all = torch.cat(kept_fc)
            n_win = sum(kept_win)
            kept_fc.clear(); kept_len.clear(); kept_win.clear()
            _, ms = timed(lambda: ctx.add_superkmers_binned_dev(recv, recv_p, nr, n_win, fine, part_off, fc_all))
        else:
            _, ms = timed(lambda: ctx.add_superkmers_dev(recv, recv_p, nr))
        ms_add += ms
        runs += 1
        peak = max(peak, used())
        if args.shard:
            print("  counting run %d: %d records, %.1f ms; %.1f GB in use" % (runs, nr, ms, used()), flush=True)

    for c in range(n_chunks):
        first, n = bounds[c], bounds[c + 1] - bounds[c]
        ctx.synth_reads_dev(GENOME_SEED, contigs, clen, READ_SEED, first, n, L, err, d_words, d_off)
        cap = ctx.superkmer_capacity(n * (L - k + 1), n)
        send = torch.empty((cap, 2), dtype=torch.int64, device=dev)
        send_b = torch.empty(cap, dtype=torch.int32, device=dev)
        if fine:
            fc = torch.empty((W, fine), dtype=torch.int32, device=dev)
            (off, win), ms = timed(lambda: ctx.extract_superkmers_binned_dev(d_words, d_off, n, n * L, W, fine, send, send_b, cap, fc))
            kept_fc.append(fc)
            kept_len.extend(int(off[o + 1] - off[o]) for o in range(W))
            kept_win.append(int(win.sum()))
        else:
            off, ms = timed(lambda: ctx.extract_superkmers_dev(d_words, d_off, n, n * L, W, send, send_b, cap))
        ms_ext += ms
        nr = int(off[W])
        n_rec += nr
        # (a rank receives about what it sends: its own records stand in for the received ones, in buffers of their own; the
        # pointers of 7 ranks out of 8 do not travel -- zeros where they arrive)
        if pool["recs"] is None and n_chunks > 1:  # what may gather between counting runs, as ShardedCounter sizes its receive buffer
            held = n_chunks if not args.count_every else min(n_chunks, args.count_every)
            rows = int(min(nr * held * 1.12, args.keep_gb * 1e9 / 20 + nr * 1.12)) + 1024
            pool["recs"] = torch.empty((rows, 2), dtype=torch.int64, device=dev)
            pool["ptrs"] = torch.empty(rows, dtype=torch.int32, device=dev)
        if pool["recs"] is not None and pool["at"] + nr <= pool["recs"].shape[0]:
            a0 = pool["at"]
            pool["recs"][a0:a0 + nr] = send[:nr]
            pool["ptrs"][a0:a0 + nr] = send_b[:nr]
            kept.append(pool["recs"][a0:a0 + nr])
            ptrs = pool["ptrs"][a0:a0 + nr]
            pool["at"] = a0 + nr
        else:
            kept.append(send[:nr].clone())
            ptrs = send_b[:nr].clone()
            pool["at"] = -1 << 60  # (this run's input is put together after all)
        if args.own_ptrs_only and nr:
            # what a rank receives when the other ranks' reads stay where they are: records of ALL ranks' reads, of which only the
            # walking rank's carry pointers -- the records of the chunk's first eighth of reads keep theirs (exact pointers:
            # position + 1), the others arrive with none
            pz = ptrs.to(torch.int64) & 0xFFFFFFFF
            base = int(pz[pz > 0].min())
            ptrs[pz > base + (n * L) // W] = 0
        kept_p.append(ptrs)
        peak = max(peak, used())
        if args.shard:
            print("  reads %d..%d: extract %.1f ms so far (%d records, send buffers %.1f GB); %.1f GB in use" % (
                first, first + n, ms_ext, nr, cap * 20 / 1e9, used()), flush=True)
        del send, send_b
        if args.shard:
            torch.cuda.empty_cache()  # (torch keeps freed blocks: "in use" below means the buffers that are held, as in distributed.py)
        if (args.count_every and len(kept) >= args.count_every) or sum(x.numel() * 8 for x in kept) + sum(x.numel() * 4 for x in kept_p) >= args.keep_gb * 1e9:
            count_kept()
    count_kept()
    nd, ms_fin = timed(lambda: ctx.finalize())
    res, ms_bfs = timed(lambda: ctx.bfs_batch(jobs, cov, 100000, -1))
    peak = max(peak, used())
    st = ctx.stats()
    reached = sum(len(r["lo"]) for r in res if r is not None)
    tot = ms_ext + ms_add + ms_fin + ms_bfs
    print("owners %d, %s exchange, %d reads in %d chunks: extract %.2f (%d records, %.2f GB to send as 16-byte records%s) | count in %d run(s) %.2f | finalize %.2f (%d distinct) | "
          "walk in place %.2f (%d reached; rank 0 only) | sum %.2f ms (every rank: %.2f) | table %.1f GB, grows %d, handed on / spilled %d | peak device memory %.1f GB" % (
              W, ("binned (%d fine buckets, %d binned runs)" % (fine, st.binned_runs)) if fine else "flat", R, n_chunks, ms_ext, n_rec, n_rec * 16 / 1e9, (" + 4-byte pointers from the walking rank" if args.own_ptrs_only else " + %.2f GB of pointers + %.2f GB of packed reads to the walking rank" % (n_rec * 4 / 1e9, R * L / 4e9)), runs, ms_add, ms_fin, nd, ms_bfs, reached, tot,
              ms_ext + ms_add + ms_fin, st.table_bytes / 1e9, st.grows, st.spill_keys, peak), flush=True)
    if not args.shard and rep == 2:
        # the same reads on one GPU, whole step (what bench.py times at N = 1), for the ratio
        d_w = torch.empty((R * L + 31) // 32 + 1, dtype=torch.int64, device=dev)
        d_o = torch.empty(R + 1, dtype=torch.int64, device=dev)
        ctx.synth_reads_dev(GENOME_SEED, contigs, clen, READ_SEED, 0, R, L, err, d_w, d_o)
        one = []
        for i in range(3):
            def step():
                ctx.clear()
                ctx.add_reads_packed_dev(d_w, d_o, R, R * L)
                ctx.finalize()
                ctx.bfs_batch(jobs, cov, 100000, -1)
            one.append(timed(step)[1])
        print("one GPU, the same reads, whole step: %.2f ms -> bound on weak scaling before any wire time: %d x %.2f / %.2f = %.2f x" % (
            min(one), W, min(one), tot, W * min(one) / tot), flush=True)

if args.shard and args.check:
    # sampled loci: exact counts from an independent replay of the generator; the walk against the oracle's on the reads
    # around the seed gene (tests/test_gpu_fullsize_config2.py does the same for configs[2])
    from oracle import pyoracle as po
    from tests.helpers import assert_bfs_equal
    from tests.test_gpu_fullsize_config2 import _reads_near
    genome = po.synth_genome(GENOME_SEED, contigs * clen)
    rng = np.random.default_rng(11)
    Wd = 200
    loci = [(int(rng.integers(0, contigs)), int(rng.integers(1000, clen - 2000))) for _ in range(8)]
    near = _reads_near(R, contigs, clen, L, [(c, p - L + 1, p + Wd + k) for c, p in loci] + [(0, 100000 - 120000 - L, 101000 + 120000)])
    checked = 0
    for (c, p), reads in zip(loci, near[:-1]):
        pieces = []
        for r, strand, s in reads:
            codes = po.synth_reads(genome, contigs, clen, READ_SEED, r, 1, L, err)
            j_lo, j_hi = (p - s, p + Wd - 1 - s) if strand == 0 else (L - k - (p + Wd - 1 - s), L - k - (p - s))
            j_lo, j_hi = max(j_lo, 0), min(j_hi, L - k)
            if j_lo <= j_hi:
                pieces.append(codes[j_lo:j_hi + k])
        o = np.zeros(len(pieces) + 1, dtype=np.uint64)
        o[1:] = np.cumsum([len(x) for x in pieces])
        local = po.Table()
        local.count_reads(np.concatenate(pieces), o, k, po.KEY_PACKED)
        keys, counts = local.dump()
        assert np.array_equal(ctx.get(keys), counts), (c, p)
        checked += len(keys)
    reads = near[-1]
    codes = np.concatenate([po.synth_reads(genome, contigs, clen, READ_SEED, r, 1, L, err) for r, _, _ in reads])
    o = np.arange(len(reads) + 1, dtype=np.uint64) * np.uint64(L)
    local = po.Table()
    local.count_reads(codes, o, k, po.KEY_PACKED)
    sd = genome[100000:101000]
    for d, got in zip((-1, 1), res):
        assert_bfs_equal(got, po.bfs(local, k, po.KEY_PACKED, [sd], d, cov, 100000, -1))
    print("parity: %d keys at %d sampled loci hold exactly the counts a replay of the read generator gives; both walks equal the oracle's" % (checked, len(loci)), flush=True)
