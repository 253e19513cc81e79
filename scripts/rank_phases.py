"""What one rank of an N-GPU job computes per step, timed on one GPU: extraction of super-k-mer records bucketed for N
owners, counting of as many records as a rank receives (here: its own, all owners together), finalize, export of the
solid shard (from the list the merge kernel leaves), BFS table from the gathered shards, the walk.  The all-to-all and
the all-gather themselves are not in it.

  python scripts/rank_phases.py [n_owners]                          bench scale: 10 M x 150 bp, E1
  python scripts/rank_phases.py 8 --shard 125000000 --contigs 125   one rank of configs[3]: 125 M reads' worth of records over the
                                                                    rank's eighth of the key space (125 of the 1000 contigs at the
                                                                    same 30-fold depth), in chunks as metacherchant_amd/distributed.py
                                                                    makes them; prints the device memory in use after every phase"""
import argparse, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metacherchant_amd as m

ap = argparse.ArgumentParser()
ap.add_argument("owners", nargs="?", type=int, default=8)
ap.add_argument("--shard", type=int, default=0, help="reads of the rank (0: bench scale, 10 M reads, 3 repetitions)")
ap.add_argument("--contigs", type=int, default=10)
ap.add_argument("--chunk", type=int, default=32 << 20, help="reads per exchange (MC_EXCHANGE_CHUNK_READS)")
args = ap.parse_args()
W = args.owners
dev = torch.device("cuda:0")
k, L, cov = 31, 150, 5
clen, err = 5_000_000, 100
GENOME_SEED, READ_SEED = 20240531, 42
R = args.shard or 10_000_000
contigs = args.contigs
chunk = min(args.chunk, R)

def used():
    f, t = torch.cuda.mem_get_info()
    return (t - f) / 1e9

def t(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return r, 1e3 * (time.perf_counter() - t0)

windows_total = R * (L - k + 1)
est_distinct = int(contigs * clen + windows_total * (1.0 - (1.0 - err / 10000.0) ** k))
ctx = m.Context(k, m.KEY_PACKED, 0, est_distinct + (1 << 20), m.native.FLAG_SOLID_LIST)
ctx.set_coverage_hint(cov)
print("table for %.0f M expected keys: %.1f GB in use" % (est_distinct / 1e6, used()), flush=True)
d_words = torch.empty((chunk * L + 31) // 32 + 1, dtype=torch.int64, device=dev)
d_off = torch.empty(chunk + 1, dtype=torch.int64, device=dev)
seed = m.native.synth_genome(GENOME_SEED, 100000, 1000)
sv = []
for i in range(len(seed) - k + 1):
    v = 0
    for c in seed[i:i + k]:
        v = (v << 2) | int(c)
    sv.append(v)
hi, lo = np.zeros(len(sv), dtype=np.uint64), np.array(sv, dtype=np.uint64)
jobs = [(hi, lo, -1), (hi, lo, 1)]
solid = m.Context(k, m.KEY_PACKED, 0, 1 << 20)

for rep in range(1 if args.shard else 3):
    ctx.clear()
    ms_ext = ms_add = 0.0
    n_rec = 0
    peak = used()
    for first in range(0, R, chunk):
        n = min(chunk, R - first)
        ctx.synth_reads_dev(GENOME_SEED, contigs, clen, READ_SEED, first, n, L, err, d_words, d_off)
        cap = ctx.superkmer_capacity(n * (L - k + 1), n)
        send = torch.empty((cap, 2), dtype=torch.int64, device=dev)
        send_b = torch.empty(cap, dtype=torch.int32, device=dev)
        off, ms = t(lambda: ctx.extract_superkmers_dev(d_words, d_off, n, n * L, W, send, send_b, cap))
        ms_ext += ms
        nr = int(off[W])
        n_rec += nr
        # (a rank receives about what it sends: its own records stand in for the received ones, in a buffer of their own)
        recv, recv_b = send[:nr].clone(), send_b[:nr].clone()
        peak = max(peak, used())
        _, ms = t(lambda: ctx.add_superkmers_dev(recv, recv_b, nr))
        ms_add += ms
        peak = max(peak, used())
        if args.shard:
            print("  reads %d..%d: extract %.1f ms (%d records, send buffer %.1f GB), count %.1f ms; %.1f GB in use" % (
                first, first + n, ms_ext, nr, cap * 20 / 1e9, ms, used()), flush=True)
        del send, send_b, recv, recv_b
    nd, ms_fin = t(lambda: ctx.finalize())
    if args.shard:
        # (the BFS table of configs[3] is a matter of its own -- 5 G solid k-mers of 8 shards on one device: DESIGN.md section 6 --
        # and this rank's eighth of them does not fit next to its counting table and the pipeline's scratch: counting phase only)
        st = ctx.stats()
        n, ms_cnt = t(lambda: ctx.export_count(cov))
        print("owners %d, %d reads: extract %.1f ms (%d records, %.1f GB sent / received) | count %.1f ms | finalize %.1f (%d distinct, %d solid) | "
              "table %.1f GB, grows %d, handed on / spilled %d | peak device memory %.1f GB" % (
                  W, R, ms_ext, n_rec, n_rec * 20 / 1e9, ms_add, ms_fin, nd, n, st.table_bytes / 1e9, st.grows, st.spill_keys, peak), flush=True)
        break
    n, ms_cnt = t(lambda: ctx.export_count(cov))
    keys = torch.zeros(n, dtype=torch.int64, device=dev)
    cnts = torch.full((n,), -1, dtype=torch.int16, device=dev)
    hints = torch.zeros(n, dtype=torch.int32, device=dev)
    _, ms_exp = t(lambda: ctx.export_dev(cov, keys, cnts, n, hints))
    solid.clear()
    _, ms_solid = t(lambda: solid.solid_from_pairs_dev(keys, cnts, n, cov, hints))
    solid.share_read_store(ctx)  # (the pointers in `hints` lead into the counting context's read store: distributed.py gather_solid)
    res, ms_bfs = t(lambda: solid.bfs_batch(jobs, cov, 100000, -1))
    peak = max(peak, used())
    tot = ms_ext + ms_add + ms_fin + ms_cnt + ms_exp + ms_solid + ms_bfs
    st = ctx.stats()
    print("  (exports from the merge kernel's list so far: %d, table sweeps for counting: %d, spilled records: %d, table %.1f GB, grows %d)" % (
        st.solid_list_builds, st.solid_sweeps, st.spill_keys, st.table_bytes / 1e9, st.grows))
    print("owners %d, %d reads: extract %.2f (%d records, %.2f GB) | add %.2f | finalize %.2f (%d distinct) | export_count %.2f export %.2f (%d solid) | "
          "solid table %.2f | walk %.2f (%d reached) | sum %.2f ms | peak device memory %.1f GB" % (
              W, R, ms_ext, n_rec, n_rec * 20 / 1e9, ms_add, ms_fin, nd, ms_cnt, ms_exp, n, ms_solid, ms_bfs, sum(len(r["lo"]) for r in res), tot, peak), flush=True)
    del keys, cnts, hints
