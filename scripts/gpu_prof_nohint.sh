#!/bin/bash
# kernel statistics of scripts/no_hint_probe.py -> gpurun_out/prof_nohint/
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_nohint
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/no_hint_probe.py > $OUT/run.log 2>&1
for f in $(find $OUT -name "*kernel_stats*.csv"); do cp $f $OUT/kernel_stats.csv; done
for f in $(find $OUT -name "*kernel_trace*.csv"); do cp $f $OUT/kernel_trace.csv; done
head -n 24 $OUT/kernel_stats.csv | cut -c1-90,250-400
