#!/bin/bash
# scratch: what one gpurun call of the moment runs (edited freely between calls; the scripts that matter are gpu_round_*.sh)
set -o pipefail
timeout -k 10 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
