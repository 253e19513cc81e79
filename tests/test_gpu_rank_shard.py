"""One rank of configs[3] (8 GPUs x 125 M reads, k = 31), scaled to a few contigs at the same 30-fold depth: what the rank does
between the collectives -- super-k-mer records bucketed for 8 owners chunk by chunk (binned form and flat form), the kept records counted in several runs
(MC_EXCHANGE_COUNT_EVERY), finalize, the walk over the table in place -- through the C ABI, with the counts at sampled loci and
both walks compared with the oracle on a replay of the read generator.  scripts/rank_phases.py does the work (at full size it is the
source of profiles/r*_rank_phases*.txt); this keeps that path under the test runner.  Needs a real MI355X."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("count_every,flat", [(0, False), (2, False), (2, True)])
def test_one_rank_of_configs3_scaled(count_every, flat):
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "rank_phases.py"), "8", "--shard", "4000000", "--contigs", "4",
           "--min-chunks", "5", "--count-every", str(count_every), "--check"] + (["--flat"] if flat else [])
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = p.stdout
    assert "parity:" in out and "both walks equal the oracle's" in out
    line = [l for l in out.splitlines() if l.startswith("owners 8")][-1]
    assert "in 5 chunks" in line
    runs = int(line.split("count in ")[1].split(" run")[0])
    assert runs == (1 if count_every == 0 else 3), line
    assert ", grows 0," in line  # the table sized by the expected keys held them
    # the form of the record exchange: binned (the rank's counting runs start at their second level, every one of them) unless --flat
    # (the script's last line is its second repetition's: the context has seen every run twice)
    assert ("flat exchange" in line) if flat else ("binned (512 fine buckets, %d binned runs)" % (2 * runs) in line), line
