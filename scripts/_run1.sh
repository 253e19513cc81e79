#!/bin/bash
# scratch: what one gpurun call of the moment runs (edited freely between calls; the scripts that matter are gpu_round_*.sh)
set -o pipefail
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/suite.log 2>&1
grep "^E .*Assert\|^E  *assert" gpurun_out/suite.log | cut -c1-400 | head -n 6
tail -n 4 gpurun_out/suite.log
