"""Does the second level's time follow how far apart its leaves' streams lie?  One process, the bench's reads, contexts made one
after the other with MC_CAP2_SIGMAS = 32 (the default: a leaf's stream has room for mean x 1.15 + 32 sqrt(mean) records) and
smaller: the pipeline's scratch comes from the process's pool, so the contexts run on the same memory.  Prints the kernels' times
and the spilled records.  Usage: python scripts/cap2_probe.py [sigmas ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metacherchant_amd as m
sig = [float(x) for x in sys.argv[1:]] or [32, 12, 32, 12, 20, 32]   # values >= 100 are taken as records per leaf stream (MC_CAP2) instead
dev = torch.device("cuda:0")
R, L, k = 10_000_000, 150, 31
n_bases = R * L
d_words = torch.empty((n_bases + 31) // 32 + 1, dtype=torch.int64, device=dev)
d_off = torch.empty(R + 1, dtype=torch.int64, device=dev)
boot = m.Context(k, m.KEY_PACKED, 0, 372_000_000)
boot.synth_reads_dev(20240531, 10, 5_000_000, 42, 0, R, L, 100, d_words, d_off)
boot.close()
for s in sig:
    os.environ.pop("MC_CAP2", None)
    os.environ.pop("MC_CAP2_SIGMAS", None)
    if s >= 100:
        os.environ["MC_CAP2"] = str(int(s))
    else:
        os.environ["MC_CAP2_SIGMAS"] = str(s)
    ctx = m.Context(k, m.KEY_PACKED, 0, 372_000_000)
    for rep in range(3):
        ctx.clear(); ctx.reset_stats()
        ctx.add_reads_packed_dev(d_words, d_off, R, n_bases)
        nd = ctx.finalize()
        st = ctx.stats()
    print("sigmas %5.1f: p1 %.3f p2 %.3f p3 %.3f ms, spilled %d, distinct %d" % (s, st.p1_ms, st.p2_ms, st.p3_ms, st.spill_keys, nd), flush=True)
    ctx.close()
