#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_long_records.py -x -q > gpurun_out/long1.log 2>&1; rc=$?; tail -12 gpurun_out/long1.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize_config2.py -x -q -s > gpurun_out/full_c2.log 2>&1; rc=$?; tail -12 gpurun_out/full_c2.log
exit $rc
