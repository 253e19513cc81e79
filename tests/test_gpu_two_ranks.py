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


# walk: "shards" -- rank 0 maps rank 1's counting table (hipIpc handle, mc_shard_attach) and walks over both in place;
# "gather" -- round 3's way: the k-mers at or above --coverage gathered into a BFS-only context on rank 0
@pytest.mark.parametrize("k,records,walk", [(31, True, "shards"), (25, True, "shards"), (41, False, "shards"), (21, False, "shards"),
                                           (31, True, "gather"), (41, False, "gather")])
def test_two_ranks_on_one_gpu(k, records, walk):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "two_ranks_one_gpu.py"), str(k)], capture_output=True,
                       text=True, timeout=900, cwd=ROOT, env=dict(os.environ, TWO_RANKS_WALK=walk, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("two ranks on one GPU")][-1]
    assert "'ok'" in line
    assert line.rstrip(")").endswith("True" if records else "False")  # which form of the exchange ran
    assert ", 5, " in line  # ... in five chunks, rank 1 with three reads in all (scripts/two_ranks_one_gpu.py)
    # records travel binned (the sender does the owner's first level; the script checks that every counting run started at its second)
    assert ("fine buckets: 512" if records else "fine buckets: 0") in line


def test_bench_multi_rank_path_on_one_gpu():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run), both ranks on the one GPU over gloo:
    the sharded step runs and rank 0 prints one JSON line with the whole-job numbers."""
    import json
    env = dict(os.environ, MC_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MC_EXCHANGE_CHUNK_READS="40000")  # four chunks a rank, as configs[3] needs
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--reads", "300000", "--contigs", "2", "--contig-len", "1000000", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["bfs"]["reached"] > 1000 and out["solid_kmers"] > 100000
    # the super-k-mer form of the exchange ran, binned: the counting runs had no first level of their own
    assert out["exchange_form"].startswith("binned (512 fine buckets") and "k_sk1_records" not in out["roofline"]["kernel_ms"]
    assert out["roofline"]["kernel_ms"]["k_sk2_scatter_staged<listed>"] > 0
    # the record says what ran: how many ranks the process group had, which walk, what travelled, what every rank did
    assert out["ranks_seen"] == 2 and out["backend"] == "gloo" and out["walk_mode"] == "in_place" and out["walk_fallback"] is None
    assert out["exchange_chunks"] >= 4 and out["count_runs_per_step"] == 1 and out["exchange_GB_per_step"] > 0
    ph = out["rank_phases_ms_per_step"]
    assert [p["rank"] for p in ph] == [0, 1] and all(p["reads"] == 300000 and p["extract_ms"] > 0 and p["count_kernels_ms"] > 0 for p in ph)
    assert ph[0]["walk_ms"] > 0 and ph[1]["walk_ms"] == 0


def test_bench_default_line_has_what_the_contract_names():
    """python bench.py on one GPU (a small workload): ONE JSON line with the metric, `roofline`, `cpu_baseline` and the `config2`
    object (configs[2]'s k = 63 pipeline on the same reads)."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--reads", "400000", "--contigs", "2",
           "--contig-len", "1000000", "--cpu-sample-reads", "20000", "--config2-steps", "2"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["dtype"] == "int64" and out["vs_baseline"] is None and "workload" in out["config"]
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert set(r["kernel_ms"]) == {"k_sk1w_extract", "k_sk2_scatter", "k_p3_dedup"} and all(v > 0 for v in r["kernel_ms"].values())
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    c2 = out["config2"]
    assert "error" not in c2 and c2["value"] > 0 and c2["roofline"]["frac"] > 0 and c2["distinct_kmers"] > c2["bfs"]["reached"] > 0
    # (the leg's context has a capacity hint: k = 63 polynomial keys then travel as long records, csrc/count_long.h)
    assert c2["roofline"]["form"].startswith("long records")
    assert set(c2["roofline"]["kernel_ms"]) == {"k_skl_extract", "k_sk2_scatter_compact<2,2>", "k_p3_long"}
