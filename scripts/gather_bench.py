"""Rank 0's side of the multi-GPU gather at bench scale, on one GPU: the solid shard of a 10 M-read count
(E1) is exported and turned into the BFS table (a) through a counting context (mc_add_pairs_dev, then the usual
solid-table build at the first BFS) and (b) directly (mc_solid_from_pairs_dev).  Usage: python scripts/gather_bench.py"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metacherchant_amd as m

dev = torch.device("cuda:0")
k, L, R, cov = 31, 150, 10_000_000, 5
contigs, clen, err = 10, 5_000_000, 100
GENOME_SEED, READ_SEED = 20240531, 42
n_bases = R * L
d_words = torch.empty((n_bases + 31) // 32 + 1, dtype=torch.int64, device=dev)
d_off = torch.empty(R + 1, dtype=torch.int64, device=dev)
ctx = m.Context(k, m.KEY_PACKED, 0, 370_000_000)
ctx.set_coverage_hint(cov)
ctx.synth_reads_dev(GENOME_SEED, contigs, clen, READ_SEED, 0, R, L, err, d_words, d_off)
ctx.add_reads_packed_dev(d_words, d_off, R, n_bases)
ctx.finalize()
seed = m.native.synth_genome(GENOME_SEED, 100000, 1000)
sv = []
for i in range(len(seed) - k + 1):
    v = 0
    for c in seed[i:i + k]:
        v = (v << 2) | int(c)
    sv.append(v)
hi = np.zeros(len(sv), dtype=np.uint64)
lo = np.array(sv, dtype=np.uint64)
jobs = [(hi, lo, -1), (hi, lo, 1)]

def t(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return r, 1e3 * (time.perf_counter() - t0)

for rep in range(2):
    n, ms_cnt = t(lambda: ctx.export_count(cov))
    keys = torch.zeros(n, dtype=torch.int64, device=dev)
    cnts = torch.full((n,), -1, dtype=torch.int16, device=dev)
    hints = torch.zeros(n, dtype=torch.int32, device=dev)
    _, ms_exp = t(lambda: ctx.export_dev(cov, keys, cnts, n, hints))
    print("export_count %.2f ms (%d solid), export_dev %.2f ms" % (ms_cnt, n, ms_exp))
    a = m.Context(k, m.KEY_PACKED, 0, contigs * clen + (1 << 20))
    _, ms_add = t(lambda: (a.add_pairs_dev(keys, cnts, n, hints), a.finalize()))
    ra, ms_bfs_a = t(lambda: a.bfs_batch(jobs, cov, 100000, -1))
    b = m.Context(k, m.KEY_PACKED, 0, 1 << 20)
    _, ms_solid = t(lambda: b.solid_from_pairs_dev(keys, cnts, n, cov, hints))
    rb, ms_bfs_b = t(lambda: b.bfs_batch(jobs, cov, 100000, -1))
    for x, y in zip(ra, rb):
        assert np.array_equal(x["lo"], y["lo"]) and np.array_equal(x["dist"], y["dist"]) and np.array_equal(x["cov"], y["cov"])
    print("(a) add_pairs+finalize %.2f ms, bfs_batch (solid build + walk) %.2f ms, total %.2f ms" % (ms_add, ms_bfs_a, ms_add + ms_bfs_a))
    print("(b) solid_from_pairs %.2f ms, bfs_batch (walk) %.2f ms, total %.2f ms" % (ms_solid, ms_bfs_b, ms_solid + ms_bfs_b))
    a.close(); b.close()
