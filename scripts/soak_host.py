"""Randomised comparison of the C++ host side (trim, subgraph map in java.util.HashMap order, compaction, writers) with the
Python restatement in oracle/host_oracle.py: random k, graphs with repeats and errors, directions, --trim, --chunklength.
CPU only (BFS passes come from the oracle).  Usage: python scripts/soak_host.py [iterations] [seed]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from metacherchant_amd import build
from oracle import host_oracle as ho
from oracle import pyoracle as po
from tests.test_host_cpp import _dump, _oracle_files

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
build.build_host()
hosttest = build.HOSTTEST
for it in range(iters):
    k = int(rng.choice([5, 9, 15, 21, 25, 31, 32, 41, 63]))
    mode = po.KEY_PACKED if k <= 31 else (po.KEY_POLY if rng.integers(0, 2) else po.KEY_FNV1A)
    glen = int(rng.integers(400, 3000))
    genome = rng.integers(0, 4, glen).astype(np.uint8)
    for _ in range(int(rng.integers(0, 4))):  # repeats: the graph branches and has cycles
        a, b, ln = int(rng.integers(0, glen - 120)), int(rng.integers(0, glen - 120)), int(rng.integers(k, 100))
        genome[b:b + ln] = genome[a:a + ln]
    L = int(rng.integers(max(k + 5, 40), 160))
    n = int(rng.integers(100, 900))
    starts = rng.integers(0, glen - L, n)
    reads = np.concatenate([genome[s:s + L] for s in starts])
    errs = rng.random(len(reads)) < float(rng.choice([0.0, 0.005, 0.02]))
    reads = np.where(errs, (reads + rng.integers(1, 4, len(reads))) % 4, reads).astype(np.uint8)
    off = np.arange(n + 1, dtype=np.uint64) * L
    t = po.Table()
    t.count_reads(reads, off, k, mode)
    s0 = int(rng.integers(0, glen - 150))
    seed = genome[s0:s0 + int(rng.integers(k, 140))]
    trim, both = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    cov, maxk, maxr = int(rng.integers(1, 4)), int(rng.choice([-1, 50, 400, 5000])), int(rng.choice([-1, 5, 40]))
    if maxk < 0 and maxr < 0:
        maxk = 1000
    chunk = int(rng.choice([1, 1, 10, 60]))
    passes = []
    for d in ([0] if both else [-1, 1]):
        r = po.bfs(t, k, mode, [seed], d, cov, maxk, maxr, trim)
        if r is None:
            passes = None
            break
        passes.append((d, r))
    if passes is None:
        continue
    with tempfile.TemporaryDirectory() as tmp:
        dump, out = os.path.join(tmp, "dump.txt"), os.path.join(tmp, "out")
        _dump(dump, k, chunk, trim, [po.decode(seed)], passes)
        subprocess.check_call([hosttest, "env", dump, out])
        want = _oracle_files(k, [po.decode(seed)], passes, trim, chunk)
        for name, text in want.items():
            with open(os.path.join(out, name)) as f:
                got = f.read()
            assert got == text, (it, name, k, trim, both, chunk)
    if it % 20 == 0:
        print("it %d ok (k=%d, %d vertices)" % (it, k, sum(len(p[1]["lo"]) for p in passes)), flush=True)
print("host soak ok: %d iterations" % iters)
