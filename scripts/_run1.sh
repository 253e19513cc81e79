#!/bin/bash
# scratch: the whole GPU suite, soaks of the long-record lengths under both bin rules, the round's profiles -- on the final build
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_all.log 2>&1; rc=$?; tail -4 gpurun_out/gpu_all.log
[ $rc -eq 0 ] || exit $rc
for b in 1 2; do
  MC_LONG_BINS=$b SOAK_KS=33,41,47,55,63,36,60,50 SOAK_SEEDS=2 timeout -k 10 200 python scripts/soak.py 100000 60$b > gpurun_out/soak_long_b$b.log 2>&1; rc=$?
  echo "long soak bins=$b rc=$rc: $(grep -c ' ok' gpurun_out/soak_long_b$b.log) iterations ok, $(grep -c 'long=[1-9]' gpurun_out/soak_long_b$b.log) with long runs; $(grep -c 'AssertionError\|Traceback' gpurun_out/soak_long_b$b.log) failures"
  [ $rc -eq 124 ] || [ $rc -eq 0 ] || exit 1
done
timeout -k 10 200 python scripts/soak.py 100000 603 > gpurun_out/soak_r5_603.log 2>&1; rc=$?
echo "soak rc=$rc: $(grep -c ' ok' gpurun_out/soak_r5_603.log) iterations ok; $(grep -c 'AssertionError\|Traceback' gpurun_out/soak_r5_603.log) failures"
[ $rc -eq 124 ] || [ $rc -eq 0 ] || exit 1
ROUND=r05 MC_COMMIT=$1 bash scripts/gpu_round_profiles.sh > gpurun_out/profiles_run.log 2>&1
ls gpurun_out/p | wc -l
