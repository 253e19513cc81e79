"""Times the counting of configs[1] in fresh contexts without a capacity hint (what bench.py's no-hint leg does) and
prints every context's phase times: python scripts/no_hint_probe.py [reads]"""
import sys, time
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metacherchant_amd as m
from bench import GENOME_SEED, READ_SEED

R = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0")
L = 150
n_bases = R * L
d_words = torch.empty((n_bases + 31) // 32 + 1, dtype=torch.int64, device=dev)
d_off = torch.empty(R + 1, dtype=torch.int64, device=dev)
g = m.Context(31, 0, 0, 1 << 20)
g.synth_reads_dev(GENOME_SEED, 10, 5_000_000, READ_SEED, 0, R, L, 100, d_words, d_off)
g.close()
for i in range(int(os.environ.get('N_CTX', 10))):
    t00 = time.perf_counter()
    c = m.Context(31, 0, 0, 0)
    t01 = time.perf_counter()
    c.set_coverage_hint(5)
    torch.cuda.synchronize(dev)
    t = time.perf_counter()
    c.add_reads_packed_dev(d_words, d_off, R, n_bases)
    t1 = time.perf_counter()
    d = c.finalize()
    t2 = time.perf_counter()
    st = c.stats()
    print("ctx %d: add %.1f ms finalize %.1f ms | p1 %.2f p2 %.2f p3 %.2f count %.2f total %.2f launches %d grows %d spill %d table %.2f GB distinct %d"
          % (i, 1e3 * (t1 - t), 1e3 * (t2 - t1), st.p1_ms, st.p2_ms, st.p3_ms, st.count_ms, st.count_total_ms, st.count_launches, st.grows, st.spill_keys,
             st.table_bytes / 1e9, d), flush=True)
    t3 = time.perf_counter()
    c.close()
    print('   create %.1f ms close %.1f ms' % (1e3 * (t01 - t00), 1e3 * (time.perf_counter() - t3)), flush=True)
