#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_long_records.py -x -q > gpurun_out/long1.log 2>&1; rc=$?; tail -12 gpurun_out/long1.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scripts/long_probe.py | grep "^lib"
MC_LONG_BINS=2 timeout -k 10 300 python scripts/long_probe.py | grep "^lib"
