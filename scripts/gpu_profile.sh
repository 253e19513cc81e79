#!/bin/bash
# run on the GPU box through gpurun: rocprofv3 kernel trace + stats of one bench invocation
# usage: bash scripts/gpu_profile.sh <tag> [bench args...]
tag=$1; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/prof_$tag.log 2>&1
echo "rc=$?"; grep '"metric"' gpurun_out/prof_$tag.log | cut -c1-200
for f in $(find gpurun_out/prof_$tag -name '*kernel_stats*.csv'); do cp $f gpurun_out/${tag}_kernel_stats.csv; cut -c1-160 $f | head -14; done
