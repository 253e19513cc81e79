"""The walk's workgroup barriers (csrc/bfs_device.h).  Round 3's kernel returned one wrong walk in its soak
(gpurun_out/soak_r3.log:80: 19 974 vertices for the oracle's 19 972): three places let a late wave decide differently from
its workgroup about a branch that holds barriers.  A wave is that late once in ~10^7 rounds on an idle
GPU, once in ~10^4 walks beside another kernel, and every few rounds in a -DMC_BFS_FUZZ build, which pauses every wave for a
different time behind every barrier.  So:

* the fuzzed build of round 3's code (-DMC_BFS_OLD_RACE, kept for this test) must go wrong within a few hundred walks --
  seen by the device-side self-check (MC_BFS_SELFCHECK=1, MC_ECHECK) or by the comparison with the oracle;
* the fuzzed build of the kernel as it is now must walk a few thousand direction-0 end-games (components that end within 64
  vertices of --maxkmers, two and three jobs a launch) exactly like src/algo/OneSequenceCalculator.java:154-214 restated in
  oracle/mc_oracle.c.

The fuzzed build of the current kernel comes with __graft_entry__.build(); the old kernel's half is opt-in (MC_RUN_OLD_RACE=1:
it needs a library of its own, is probabilistic by nature, and a kernel whose waves fall out of step can hold the GPU)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STRESS = os.path.join(ROOT, "scripts", "bfs_endgame_stress.py")


def _variant(name):
    from metacherchant_amd import build as b
    return b.build_variant(name)  # (`fuzz` comes with __graft_entry__.build(); anything else is compiled here, a minute of hipcc)


def _stress(lib, *args):
    env = dict(os.environ, MC_LIB=lib, MC_BFS_SELFCHECK="1")
    env.pop("MC_BFS_TRACE_DUMP", None)
    return subprocess.run([sys.executable, STRESS] + list(args), capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)


@pytest.mark.skipif(os.environ.get("MC_RUN_OLD_RACE") != "1", reason="opt-in (MC_RUN_OLD_RACE=1): a deliberately racy kernel must FAIL within 900 walks -- "
                    "probabilistic, and waves out of step at barriers can hold a shared GPU for minutes (ADVICE r4); profiles/r04_bfs_hunts.txt has its runs")
def test_round3_kernel_goes_wrong_when_its_waves_are_held_up():
    try:
        p = _stress(_variant("fuzz_old"), "--walks", "900", "--jobs", "3", "--max-bad", "2")
    except subprocess.TimeoutExpired:
        pytest.skip("the racy kernel hung until the time limit: that is one of the ways it goes wrong, but nothing to gate on")
    out = p.stdout + p.stderr
    assert p.returncode == 1, out[-3000:]
    assert "self-check failed" in out or "first differences" in out, out[-3000:]


@pytest.mark.parametrize("dirs,jobs", [("0", 3), ("0,1,-1", 2)])
def test_fixed_kernel_walks_exactly_with_its_waves_held_up(dirs, jobs):
    p = _stress(_variant("fuzz"), "--walks", "3000", "--jobs", str(jobs), "--dirs", dirs)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert " 0 bad" in p.stdout.splitlines()[-1]


def test_selfcheck_passes_on_the_product_library_beside_a_counting_context():
    """the product library, a second context counting on the same GPU meanwhile (the regime of the round-3 failure)"""
    env = dict(os.environ, MC_BFS_SELFCHECK="1")
    env.pop("MC_LIB", None)
    p = subprocess.run([sys.executable, STRESS, "--walks", "6000", "--jobs", "3", "--contend", "1"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
