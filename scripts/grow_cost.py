"""What a missing capacity hint costs: configs[1]'s 10 M reads counted into a fresh context with capacity_hint = 0 (the table
starts at 64 MB and is rebuilt as it fills, like the reference's BigLong2ShortHashMap) against a context sized by the hint.
Usage: python scripts/grow_cost.py [n_reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import metacherchant_amd as m
R = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
k, L = 31, 150
dev = torch.device("cuda:0")
d_words = torch.empty((R * L + 31) // 32 + 1, dtype=torch.int64, device=dev)
d_off = torch.empty(R + 1, dtype=torch.int64, device=dev)
warm = m.Context(k, m.KEY_PACKED, 0, 1 << 20)
warm.synth_reads_dev(20240531, 10, 5_000_000, 42, 0, R, L, 100, d_words, d_off)
warm.add_reads_packed_dev(d_words[:1 << 20], d_off[:100001], 100000, 100000 * L)  # (code objects loaded, pinned buffers made)
warm.close()
for hint in (0, int(50e6 + R * 120 * 0.27) + (1 << 20)):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx = m.Context(k, m.KEY_PACKED, 0, hint)
        ctx.set_coverage_hint(5)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ctx.add_reads_packed_dev(d_words, d_off, R, R * L)
        nd = ctx.finalize()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        st = ctx.stats()
        print("hint %d: create %.1f ms, count + finalize %.1f ms (kernels %.1f ms, %d table rebuilds, table %.1f GB, %d distinct)" % (
            hint, 1e3 * (t1 - t0), 1e3 * (t2 - t1), st.count_ms, st.grows, st.table_bytes / 1e9, nd), flush=True)
        ctx.close()
