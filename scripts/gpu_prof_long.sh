#!/bin/bash
# kernel statistics of scripts/long_probe.py (configs[2] scaled: long records, the key join, the walk) -> gpurun_out/prof_long_<tag>/
set -e
TAG=${1:-a}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_long_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/long_probe.py > $OUT/run.log 2>&1
for f in $(find $OUT -name "*kernel_stats*.csv"); do cp $f $OUT/kernel_stats.csv; done
head -25 $OUT/kernel_stats.csv | cut -c1-200
