#!/bin/bash
# round 4, several GPUs on one: the tests of the exchange and of the walk in place, then what one rank of an 8-GPU job computes
# (bench scale and one rank of configs[3] with its parity check).  Logs under gpurun_out/.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_cli.py tests/test_gpu_parity.py -x -q -m gpu -k "two_ranks or several_devices or falls_back or walk_over or bench_multi or crowded" > gpurun_out/r4_multi_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r4_multi_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scripts/rank_phases.py 8 > gpurun_out/r4_rank_phases_8owners.txt 2>&1; echo "rank_phases rc=$?"; tail -4 gpurun_out/r4_rank_phases_8owners.txt
timeout -k 10 900 python scripts/rank_phases.py 8 --shard 125000000 --contigs 125 --check > gpurun_out/r4_rank_shard_configs3.txt 2>&1; echo "shard rc=$?"; tail -6 gpurun_out/r4_rank_shard_configs3.txt
