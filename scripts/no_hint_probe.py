"""Where a context WITHOUT a capacity hint spends its counting time (configs[1]: 10 M x 150 bp, fresh context a step), beside a hinted one."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metacherchant_amd as m
dev = torch.device("cuda:0")
k, L, R, contigs, clen, err, cov = 31, 150, 10_000_000, 10, 5_000_000, 100, 5
d_words = torch.empty((R * L + 31) // 32 + 1, dtype=torch.int64, device=dev)
d_off = torch.empty(R + 1, dtype=torch.int64, device=dev)
c0 = m.Context(k, m.KEY_PACKED, 0, 1 << 20)
c0.synth_reads_dev(20240531, contigs, clen, 42, 0, R, L, err, d_words, d_off)
c0.close()
est = int(contigs * clen + R * (L - k + 1) * (1.0 - (1.0 - err / 10000.0) ** k))
for hint in (0, 0, 0, est + (1 << 20), est + (1 << 20)):
    c = m.Context(k, m.KEY_PACKED, 0, hint)
    c.set_coverage_hint(cov)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c.add_reads_packed_dev(d_words, d_off, R, R * L)
    t1 = time.perf_counter()
    n = c.finalize()
    t2 = time.perf_counter()
    st = c.stats()
    print("hint %d: add %.2f ms + finalize %.2f ms; kernels p1 %.2f p2 %.2f p3 %.2f (count_total %.2f), grows %d, table %.1f GB, distinct %d" % (
        hint, 1e3 * (t1 - t0), 1e3 * (t2 - t1), st.p1_ms, st.p2_ms, st.p3_ms, st.count_total_ms, st.grows, st.table_bytes / 1e9, n), flush=True)
    c.close()
