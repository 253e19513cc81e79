#!/bin/bash
# kernel statistics of scripts/rank_phases.py (one rank of an 8-GPU job at bench scale) -> gpurun_out/prof_rank_<tag>/ ; $2...: its flags
set -e
TAG=${1:-a}
shift || true
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_rank_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/rank_phases.py 8 "$@" > $OUT/run.log 2>&1
for f in $(find $OUT -name "*kernel_stats*.csv"); do cp $f $OUT/kernel_stats.csv; done
head -n 22 $OUT/kernel_stats.csv | cut -c1-220
