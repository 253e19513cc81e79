#!/bin/bash
# The round's randomised soaks on the final build, one gpurun call: scripts/soak.py on fresh seeds (tables and walks against the
# oracle, MC_BFS_SELFCHECK=1), scripts/soak_cli.py (every output file of the native CLI), and the walk's end-game stress on the
# product library beside a counting context and on the fuzzed build.  Logs under gpurun_out/; a summary line each.
# usage: SOAK_S=420 bash scripts/gpu_round_soaks.sh seed1 seed2 ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
S=${SOAK_S:-420}
for seed in "$@"; do
  timeout -k 10 $S python scripts/soak.py 100000 $seed > gpurun_out/soak_${ROUND:-r06}_$seed.log 2>&1; rc=$?
  echo "soak seed $seed rc=$rc (124 = ran until the time limit without a difference): $(grep -c ' ok' gpurun_out/soak_${ROUND:-r06}_$seed.log) iterations ok; $(grep -c 'AssertionError\|Traceback' gpurun_out/soak_${ROUND:-r06}_$seed.log) failures"
  [ $rc -eq 124 ] || [ $rc -eq 0 ] || { tail -5 gpurun_out/soak_${ROUND:-r06}_$seed.log | cut -c1-400; exit 1; }
done
timeout -k 10 $S python scripts/soak_cli.py 100000 77 > gpurun_out/soak_${ROUND:-r06}_cli.log 2>&1; rc=$?
echo "soak_cli rc=$rc: $(grep -c ' ok' gpurun_out/soak_${ROUND:-r06}_cli.log) runs ok; $(grep -c 'AssertionError\|Traceback' gpurun_out/soak_${ROUND:-r06}_cli.log) failures"
HUNT_WALKS=${HUNT_WALKS:-20000} HUNT_STEPS="product fuzz fuzz_contend" bash scripts/gpu_bfs_hunt.sh
