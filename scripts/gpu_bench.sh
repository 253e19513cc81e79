#!/bin/bash
# run on the GPU box through gpurun: bench + rocprofv3 kernel trace of the same command
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_e1.json 2> gpurun_out/bench_e1.err; echo "rc=$?"; tail -c 3000 gpurun_out/bench_e1.json; tail -5 gpurun_out/bench_e1.err
timeout 600 python bench.py --steps 3 --warmup 1 --err 0 --no-cpu-baseline > gpurun_out/bench_e0.json 2> gpurun_out/bench_e0.err; echo "rc=$?"; tail -c 3000 gpurun_out/bench_e0.json; tail -5 gpurun_out/bench_e0.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_e1 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_e1.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/prof_e1.log
find gpurun_out/prof_e1 -name '*stats*' | head; for f in $(find gpurun_out/prof_e1 -name '*kernel_stats*.csv'); do head -20 $f; done
