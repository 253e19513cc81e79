import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import metacherchant_amd as m
from oracle import pyoracle as po
from tests.helpers import synth_case, seed_windows
genome, reads, off = synth_case(1, 200000, 40000, 150, 0)
for path in ("direct", "partition"):
    os.environ["MC_COUNT_PATH"] = path
    ctx = m.Context(31, 0, 0, 400000)
    ctx.add_reads_packed(po.pack(reads), off)
    print(path, "distinct", ctx.finalize(), "p3_ms", ctx.stats().p3_ms)
    hi, lo = seed_windows(genome[100000:100200], 31)
    r = ctx.bfs(hi, lo, -1, 3, 50000, -1)
    print("  levels", r["levels"], "lookups", r["lookups"], "rounds", r["rounds"], "ms", r["device_ms"])
    ctx.close()
