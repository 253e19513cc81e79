#!/bin/bash
# scratch: what one gpurun call of the moment runs
ROUND=r05 MC_COMMIT=$1 bash scripts/gpu_round_profiles.sh > gpurun_out/profiles_run.log 2>&1
ls gpurun_out/p | wc -l
