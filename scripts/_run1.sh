#!/bin/bash
# scratch: the whole GPU suite, then a soak of the long-record lengths
set -o pipefail
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_all.log 2>&1; rc=$?; tail -4 gpurun_out/gpu_all.log
[ $rc -eq 0 ] || exit $rc
SOAK_KS=33,41,47,55,63,36,60 SOAK_SEEDS=2 timeout -k 10 900 python scripts/soak.py 40 77 > gpurun_out/soak_long.log 2>&1; rc=$?; tail -3 gpurun_out/soak_long.log; grep -c "long=[1-9]" gpurun_out/soak_long.log
exit $rc
