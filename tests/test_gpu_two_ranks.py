"""The multi-GPU orchestration with REAL contexts: both ranks of a world-2 job share the one GPU of the test
box (gloo stages the CUDA tensors through the host), so the whole path -- split reads, extract super-k-mer
records (or keys) by owner, all-to-all, count owned records, all-gather solid shards, BFS on rank 0 -- runs
through the C ABI and is compared with the oracle.  scripts/two_ranks_one_gpu.py does the work."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("k,records", [(31, True), (25, True), (41, False), (21, False)])
def test_two_ranks_on_one_gpu(k, records):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "two_ranks_one_gpu.py"), str(k)], capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("two ranks on one GPU")][-1]
    assert "'ok'" in line
    assert line.rstrip(")").endswith("True" if records else "False")  # which form of the exchange ran
