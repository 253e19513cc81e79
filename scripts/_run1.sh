#!/bin/bash
# scratch: the round's profiles on the final build, then the whole GPU suite
ROUND=r05 MC_COMMIT=$1 bash scripts/gpu_round_profiles.sh > gpurun_out/profiles_run.log 2>&1
ls gpurun_out/p | head -40
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_all.log 2>&1; rc=$?; tail -4 gpurun_out/gpu_all.log
exit $rc
