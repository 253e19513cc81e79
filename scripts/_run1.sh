#!/bin/bash
# scratch: longer soaks on the final build
timeout -k 10 300 python scripts/soak_cli.py 100000 79 > gpurun_out/soak_cli3.log 2>&1; rc=$?
echo "soak_cli rc=$rc: $(grep -c ' ok' gpurun_out/soak_cli3.log) runs ok; $(grep -c 'AssertionError\|Traceback' gpurun_out/soak_cli3.log) failures"
[ $rc -eq 124 ] || [ $rc -eq 0 ] || { tail -5 gpurun_out/soak_cli3.log | cut -c1-300; exit 1; }
MC_LONG_BINS=2 SOAK_KS=63,33,47,55,41 SOAK_SEEDS=3 timeout -k 10 300 python scripts/soak.py 100000 604 > gpurun_out/soak_long_b2b.log 2>&1; rc=$?
echo "long soak bins=2 rc=$rc: $(grep -c ' ok' gpurun_out/soak_long_b2b.log) iterations ok, $(grep -c 'long=[1-9]' gpurun_out/soak_long_b2b.log) with long runs; $(grep -c 'AssertionError\|Traceback' gpurun_out/soak_long_b2b.log) failures"
[ $rc -eq 124 ] || [ $rc -eq 0 ] || exit 1
HUNT_WALKS=6000 HUNT_STEPS="product fuzz_contend" bash scripts/gpu_bfs_hunt.sh
