#!/bin/bash
# scratch: the round's profiles on the final build
ROUND=r05 MC_COMMIT=$1 bash scripts/gpu_round_profiles.sh > gpurun_out/profiles_run.log 2>&1
tail -30 gpurun_out/profiles_run.log
