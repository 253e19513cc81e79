#!/bin/bash
# Round 5: what bounds the merge kernel.  Tuning builds (scripts/variants.py build t2:-DMC_P3_TIMING nowb:-DMC_P3_NOWB nob:-DMC_P3_NOB
# nobwb:-DMC_P3_NOB,-DMC_P3_NOWB) run the counting phase of configs[1] E1; their tables are not usable.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "" t2 nowb nob nobwb; do
  for v2 in 1 0; do
    [ -n "$v" ] && [ "$v2" = 0 ] && [ "$v" != t2 ] && continue
    echo "== lib=${v:-product} MC_P3_V2=$v2"
    if [ -n "$v" ]; then export MC_LIB=metacherchant_amd/lib/libmcgpu_$v.so; else unset MC_LIB; fi
    MC_P3_V2=$v2 timeout -k 10 200 python scripts/count_only.py 100 2>&1 | tail -4
  done
done
