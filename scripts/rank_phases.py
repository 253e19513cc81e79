"""What one rank of an N-GPU job computes per step, timed on one GPU at bench scale (10 M x 150 bp, E1): extraction
of super-k-mer records bucketed for N owners, counting of as many records as a rank receives (here: its own, all
owners together), finalize, export of the solid shard, BFS table from the gathered shards, the walk.  The all-to-all
and the all-gather themselves are not in it.  Usage: python scripts/rank_phases.py [n_owners]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metacherchant_amd as m

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
k, L, R, cov = 31, 150, 10_000_000, 5
contigs, clen, err = 10, 5_000_000, 100
GENOME_SEED, READ_SEED = 20240531, 42
n_bases, windows = R * L, R * (L - k + 1)
d_words = torch.empty((n_bases + 31) // 32 + 1, dtype=torch.int64, device=dev)
d_off = torch.empty(R + 1, dtype=torch.int64, device=dev)
ctx = m.Context(k, m.KEY_PACKED, 0, 370_000_000)
ctx.set_coverage_hint(cov)
ctx.synth_reads_dev(GENOME_SEED, contigs, clen, READ_SEED, 0, R, L, err, d_words, d_off)
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

def t(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return r, 1e3 * (time.perf_counter() - t0)

for rep in range(3):
    ctx.clear()
    cap = ctx.superkmer_capacity(windows, R)
    send = torch.empty((cap, 2), dtype=torch.int64, device=dev)
    send_b = torch.empty(cap, dtype=torch.int32, device=dev)
    off, ms_ext = t(lambda: ctx.extract_superkmers_dev(d_words, d_off, R, n_bases, W, send, send_b, cap))
    n_rec = int(off[W])
    _, ms_add = t(lambda: ctx.add_superkmers_dev(send, send_b, n_rec))
    nd, ms_fin = t(lambda: ctx.finalize())
    n, ms_cnt = t(lambda: ctx.export_count(cov))
    keys = torch.zeros(n, dtype=torch.int64, device=dev)
    cnts = torch.full((n,), -1, dtype=torch.int16, device=dev)
    hints = torch.zeros(n, dtype=torch.int32, device=dev)
    _, ms_exp = t(lambda: ctx.export_dev(cov, keys, cnts, n, hints))
    solid.clear()
    _, ms_solid = t(lambda: solid.solid_from_pairs_dev(keys, cnts, n, cov, hints))
    solid.share_read_store(ctx)  # (the pointers in `hints` lead into the counting context's read store: distributed.py gather_solid)
    res, ms_bfs = t(lambda: solid.bfs_batch(jobs, cov, 100000, -1))
    tot = ms_ext + ms_add + ms_fin + ms_cnt + ms_exp + ms_solid + ms_bfs
    st = ctx.stats()
    print("  (exports from the merge kernel's list so far: %d, table sweeps for counting: %d, spilled records: %d)" % (st.solid_list_builds, st.solid_sweeps, st.spill_keys))
    print("owners %d: extract %.2f (%d records, %.2f GB) | add %.2f | finalize %.2f (%d distinct) | export_count %.2f export %.2f (%d solid) | "
          "solid table %.2f | walk %.2f (%d reached) | sum %.2f ms" % (W, ms_ext, n_rec, n_rec * 20 / 1e9, ms_add, ms_fin, nd, ms_cnt, ms_exp, n,
                                                                      ms_solid, ms_bfs, sum(len(r["lo"]) for r in res), tot))
    del send, send_b, keys, cnts, hints
