#!/bin/bash
# scratch: what one gpurun call of the moment runs
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CLI_K=63 NOHINT=1 timeout -k 10 400 python scripts/cli_wallclock.py 4000000 2>&1 | tail -2
CLI_K=63 NOHINT=1 MC_LONG_RECORDS=0 timeout -k 10 400 python scripts/cli_wallclock.py 4000000 2>&1 | tail -2
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_all.log 2>&1; rc=$?; tail -4 gpurun_out/gpu_all.log
[ $rc -eq 0 ] || exit $rc
SOAK_KS=33,41,47,55,63,36,60,50 SOAK_SEEDS=3 timeout -k 10 200 python scripts/soak.py 100000 606 > gpurun_out/soak_long6.log 2>&1; rc=$?
echo "long soak rc=$rc: $(grep -c ' ok' gpurun_out/soak_long6.log) iterations ok, $(grep -c 'long=[1-9]' gpurun_out/soak_long6.log) with long runs; $(grep -c 'AssertionError\|Traceback' gpurun_out/soak_long6.log) failures"
[ $rc -eq 124 ] || [ $rc -eq 0 ] || exit 1
timeout -k 10 200 python scripts/soak_cli.py 100000 80 > gpurun_out/soak_cli4.log 2>&1; rc=$?
echo "soak_cli rc=$rc: $(grep -c ' ok' gpurun_out/soak_cli4.log) runs ok; $(grep -c 'AssertionError\|Traceback' gpurun_out/soak_cli4.log) failures"
[ $rc -eq 124 ] || [ $rc -eq 0 ]
