"""Counting phase only on the bench workload (debug builds of the library whose tables are not usable):
prints the pipeline kernel times.  Usage: python scripts/count_only.py [err]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metacherchant_amd as m
err = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
R, L, k = 10_000_000, 150, 31
n_bases = R * L
ctx = m.Context(k, m.KEY_PACKED, 0, 469_000_000 if err else 51_000_000)
d_words = torch.empty((n_bases + 31) // 32 + 1, dtype=torch.int64, device=dev)
d_off = torch.empty(R + 1, dtype=torch.int64, device=dev)
ctx.synth_reads_dev(20240531, 10, 5_000_000, 42, 0, R, L, err, d_words, d_off)
for rep in range(3):
    ctx.clear(); ctx.reset_stats()
    ctx.add_reads_packed_dev(d_words, d_off, R, n_bases)
    try:
        ctx.finalize()
    except Exception as e:
        print("finalize:", e)
    st = ctx.stats()
print("p1 %.3f p2 %.3f p3 %.3f ms" % (st.p1_ms, st.p2_ms, st.p3_ms))
