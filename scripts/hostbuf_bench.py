"""Counting with the reads handed over as HOST buffers (mc_add_reads_packed: H2D copy + count), for DESIGN.md's
PCIe-inclusive figure.  Usage: python scripts/hostbuf_bench.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metacherchant_amd as m
dev = torch.device("cuda:0")
R, L, k = 10_000_000, 150, 31
n_bases = R * L
ctx = m.Context(k, m.KEY_PACKED, 0, 469_000_000)
d_words = torch.empty((n_bases + 31) // 32 + 1, dtype=torch.int64, device=dev)
d_off = torch.empty(R + 1, dtype=torch.int64, device=dev)
ctx.synth_reads_dev(20240531, 10, 5_000_000, 42, 0, R, L, 100, d_words, d_off)
words = d_words.cpu().numpy().view(np.uint64)
offs = d_off.cpu().numpy().view(np.uint64)
for rep in range(3):
    ctx.clear()
    t0 = time.perf_counter()
    ctx.add_reads_packed(words, offs)
    n = ctx.finalize()
    t1 = time.perf_counter()
print("host buffers: %d distinct, %.1f ms for %d windows = %.2f Gk-mers/s (count only, %d MB over PCIe)" % (
    n, 1e3 * (t1 - t0), R * (L - k + 1), R * (L - k + 1) / (t1 - t0) / 1e9, words.nbytes >> 20))
ctx.clear()
t0 = time.perf_counter(); ctx.add_reads_packed_dev(d_words, d_off, R, n_bases); ctx.finalize(); t1 = time.perf_counter()
print("device buffers: %.1f ms" % (1e3 * (t1 - t0)))
