"""The walk over W owners' tables in place (mc_shard_attach) against the walk over one table, on ONE GPU: the tables of W contexts
hold what W ranks would own of 10 M x 150 bp reads (bench scale; every record with its pointer, the reads in the walking context's
store -- what distributed.py arranges), so the difference is what looking a k-mer up in its OWNER's table costs the walk before any
xGMI hop.   python scripts/shard_walk_probe.py [owners]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metacherchant_amd as m

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
k, L, cov, clen, err, contigs, R = 31, 150, 5, 5_000_000, 100, 10, 10_000_000
GENOME_SEED, READ_SEED = 20240531, 42
n_chunks = 4
bounds = [R * c // n_chunks for c in range(n_chunks + 1)]
est = int(contigs * clen + R * (L - k + 1) * (1.0 - (1.0 - err / 10000.0) ** k))


def timed(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return r, 1e3 * (time.perf_counter() - t0)


seed = m.native.synth_genome(GENOME_SEED, 100000, 1000)
sv = []
for i in range(len(seed) - k + 1):
    v = 0
    for c in seed[i:i + k]:
        v = (v << 2) | int(c)
    sv.append(v)
hi, lo = np.zeros(len(sv), dtype=np.uint64), np.array(sv, dtype=np.uint64)
jobs = [(hi, lo, -1), (hi, lo, 1)]

ctxs = [m.Context(k, m.KEY_PACKED, 0, est // W + (1 << 22)) for _ in range(W)]
for o, c in enumerate(ctxs):
    c.set_coverage_hint(cov)
    c.set_read_pointers((1 if o == 0 else 2) | 0x10)
fine = ctxs[0].superkmer_fine_buckets(W)
assert fine
n = bounds[1]
d_words = torch.empty((n * L + 31) // 32 + 1, dtype=torch.int64, device=dev)
d_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
for c in range(n_chunks):
    first, n = bounds[c], bounds[c + 1] - bounds[c]
    ctxs[0].synth_reads_dev(GENOME_SEED, contigs, clen, READ_SEED, first, n, L, err, d_words, d_off)
    cap = ctxs[0].superkmer_capacity(n * (L - k + 1), n)
    send = torch.empty((cap, 2), dtype=torch.int64, device=dev)
    send_p = torch.empty(cap, dtype=torch.int32, device=dev)
    fc = torch.empty((W, fine), dtype=torch.int32, device=dev)
    off, win = ctxs[0].extract_superkmers_binned_dev(d_words, d_off, n, n * L, W, fine, send, send_p, cap, fc)
    for o in range(W):
        a, b = int(off[o]), int(off[o + 1])
        ctxs[o].add_superkmers_binned_dev(send[a:b], send_p[a:b], b - a, int(win[o]), fine, np.array([0, b - a], dtype=np.uint64), fc[o:o + 1])
    del send, send_p
total = sum(c.finalize() for c in ctxs)
ctxs[0].shard_attach([c.shard_export() for c in ctxs], 0, True)
for rep in range(4):
    res, ms = timed(lambda: ctxs[0].bfs_batch(jobs, cov, 100000, -1))
    print("walk over %d owners' tables in place: %.2f ms (%d reached, %d + %d rounds, %d + %d levels); %d distinct k-mers in all" % (
        W, ms, sum(len(r["lo"]) for r in res), res[0]["rounds"], res[1]["rounds"], res[0]["levels"], res[1]["levels"], total), flush=True)
# a step of a sharded run attaches anew (the tables changed): what the first walk behind an attachment costs
exports = [c.shard_export() for c in ctxs]
for rep in range(3):
    ctxs[0].shard_detach()
    _, ms_a = timed(lambda: ctxs[0].shard_attach(exports, 0, True))
    res, ms = timed(lambda: ctxs[0].bfs_batch(jobs, cov, 100000, -1))
    print("attached again in %.2f ms, the walk behind it: %.2f ms" % (ms_a, ms), flush=True)
ctxs[0].shard_detach()
reached = [r["lo"].copy() for r in res]
for c in ctxs[1:]:
    c.close()
# one table, the same reads
one = m.Context(k, m.KEY_PACKED, 0, est + (1 << 20))
one.set_coverage_hint(cov)
for c in range(n_chunks):
    first, n = bounds[c], bounds[c + 1] - bounds[c]
    one.synth_reads_dev(GENOME_SEED, contigs, clen, READ_SEED, first, n, L, err, d_words, d_off)
    one.add_reads_packed_dev(d_words, d_off, n, n * L)
assert one.finalize() == total
for rep in range(4):
    res, ms = timed(lambda: one.bfs_batch(jobs, cov, 100000, -1))
    print("walk over one table: %.2f ms (%d reached, %d + %d rounds)" % (ms, sum(len(r["lo"]) for r in res), res[0]["rounds"], res[1]["rounds"]), flush=True)
assert all(np.array_equal(a, r["lo"]) for a, r in zip(reached, res))
