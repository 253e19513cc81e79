"""Does the second level's time (1.65-2.1 ms between otherwise identical processes) follow WHERE the driver put the pipeline's
streams?  One process, fresh contexts one after the other with the pools off (MC_SCRATCH_POOL=0 MC_TABLE_POOL=0: every context
gets memory of its own from the driver): the same reads counted by each, the kernels' times from mc_get_stats.
  MC_SCRATCH_POOL=0 MC_TABLE_POOL=0 python scripts/p2_placement_probe.py [contexts]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metacherchant_amd as m

n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
k, L, R, contigs, clen, err = 31, 150, 10_000_000, 10, 5_000_000, 100
windows = R * (L - k + 1)
est = int(contigs * clen + windows * (1.0 - (1.0 - err / 10000.0) ** k)) + (1 << 20)
d_words = torch.empty((R * L + 31) // 32 + 1, dtype=torch.int64, device=dev)
d_off = torch.empty(R + 1, dtype=torch.int64, device=dev)
keep = []
for i in range(n_ctx):
    ctx = m.Context(k, m.KEY_PACKED, 0, est)
    ctx.set_coverage_hint(5)
    if i == 0:
        ctx.synth_reads_dev(20240531, contigs, clen, 42, 0, R, L, err, d_words, d_off)
    times = []
    for rep in range(3):
        ctx.clear()
        s0 = ctx.stats()
        ctx.add_reads_packed_dev(d_words, d_off, R, R * L)
        ctx.finalize()
        s1 = ctx.stats()
        times.append((s1.p1_ms - s0.p1_ms, s1.p2_ms - s0.p2_ms, s1.p3_ms - s0.p3_ms))
    print("context %d: P1 / P2 / P3 ms of three runs: %s" % (i, "  ".join("%.2f / %.2f / %.2f" % t for t in times)), flush=True)
    keep.append(ctx)  # (kept: the next context's memory is other memory)
