"""Randomised soak of the binned record exchange (mc_extract_superkmers_binned_dev / mc_add_superkmers_binned_dev) in one process:
random k (23..31), owners (2..9), senders, chunks (some empty), read sets (fixed-length and ragged, tiny to mid-size), tables with
and without a capacity hint, sometimes senders that planned for twice or half the owner's buckets; the owners' tables together must
hold exactly the oracle's pairs.   python scripts/soak_binned.py <iterations> <seed>"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metacherchant_amd as m
from oracle import pyoracle as po
from tests.helpers import ragged_case, synth_case

n_it, seed0 = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed0)
dev = torch.device("cuda:0")
t_start = time.time()
for it in range(n_it):
    k = int(rng.choice([23, 25, 27, 28, 29, 30, 31, 31]))
    W = int(rng.integers(2, 10))
    senders, chunks = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    ragged = bool(rng.integers(0, 4) == 0)
    if ragged:
        _, codes, offs = ragged_case(rng, int(rng.integers(50, 40000)), max_len=int(rng.choice([60, 220, 400])), genome_len=int(rng.choice([3000, 60000])))
    else:
        n_reads = int(rng.choice([3, 40, 2000, 30000, 90000]))
        _, codes, offs = synth_case(int(rng.integers(1, 3)), int(rng.choice([3000, 100000, 400000])), n_reads, int(rng.choice([70, 150, 250])), int(rng.choice([0, 100, 400])))
    t = po.Table()
    t.count_reads(codes, offs, k, po.KEY_PACKED)
    ok, oc = t.dump()
    hint = int(rng.choice([0, 3_000_000, 40_000_000]))
    probe = m.Context(k, m.KEY_PACKED, 0, hint)
    fine = probe.superkmer_fine_buckets(W)
    probe.close()
    scale = float(rng.choice([1, 1, 1, 2, 0.5]))
    if fine == 0 or fine * scale * W > 16384 or fine * scale < 2:
        scale = 1
    if fine == 0:
        print("it %d: no binned form (k=%d W=%d hint=%d)" % (it, k, W, hint), flush=True)
        continue
    fine = int(fine * scale)
    desc = "k=%d W=%d senders=%d chunks=%d reads=%d ragged=%d hint=%d fine=%d (x%.1f)" % (k, W, senders, chunks, len(offs) - 1, ragged, hint, fine, scale)
    n_reads = len(offs) - 1
    parts = [[] for _ in range(W)]
    for s in range(senders):
        ctx = m.Context(k, m.KEY_PACKED, 0, hint)
        lo, hi = n_reads * s // senders, n_reads * (s + 1) // senders
        for c in range(chunks):
            a, b = lo + (hi - lo) * c // chunks, lo + (hi - lo) * (c + 1) // chunks
            sub = codes[int(offs[a]):int(offs[b])]
            o = (offs[a:b + 1] - offs[a]).astype(np.uint64)
            d_words = torch.from_numpy(po.pack(sub).view(np.int64)).to(dev)
            d_off = torch.from_numpy(o.view(np.int64)).to(dev)
            nb = int(o[-1])
            cap = max(ctx.superkmer_capacity(max(nb, 1), max(b - a, 1)), 1)
            send = torch.empty((cap, 2), dtype=torch.int64, device=dev)
            send_p = torch.empty(cap, dtype=torch.int32, device=dev)
            fc = torch.full((W, fine), 7, dtype=torch.int32, device=dev)
            off, win = ctx.extract_superkmers_binned_dev(d_words, d_off, b - a, nb, W, fine, send, send_p, cap, fc)
            assert np.array_equal(fc.sum(dim=1).cpu().numpy(), np.diff(off).astype(np.int64)), desc
            for w in range(W):
                x, y = int(off[w]), int(off[w + 1])
                parts[w].append((send[x:y].clone(), send_p[x:y].clone(), fc[w].clone(), int(win[w])))
        ctx.close()
    ks, cs, binned_runs = [], [], 0
    for w in range(W):
        recs = torch.cat([p[0] for p in parts[w]])
        ptrs = torch.cat([p[1] for p in parts[w]])
        n = recs.shape[0]
        ctx = m.Context(k, m.KEY_PACKED, 0, hint)
        ctx.set_coverage_hint(3)
        po_ = np.concatenate([[0], np.cumsum([p[0].shape[0] for p in parts[w]])]).astype(np.uint64)
        ctx.add_superkmers_binned_dev(recs, ptrs, n, sum(p[3] for p in parts[w]), fine, po_, torch.stack([p[2] for p in parts[w]]))
        ctx.finalize()
        binned_runs += ctx.stats().binned_runs
        a, b = ctx.export(0)
        ks.append(a)
        cs.append(b)
        ctx.close()
    gk, gc = np.concatenate(ks), np.concatenate(cs)
    o = np.argsort(gk, kind="stable")
    assert np.array_equal(gk[o], ok) and np.array_equal(gc[o], oc), desc
    print("it %d ok: %s distinct=%d binned_runs=%d/%d (%.0f s)" % (it, desc, len(ok), binned_runs, W, time.time() - t_start), flush=True)
