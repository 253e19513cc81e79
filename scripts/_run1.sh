#!/bin/bash
set -o pipefail
SOAK_KS=33,41,47,55,63,36,60,50 SOAK_SEEDS=2 timeout -k 10 300 python scripts/soak.py 100000 505 > gpurun_out/soak_long4.log 2>&1; rc=$?
echo "long soak rc=$rc: $(grep -c ' ok' gpurun_out/soak_long4.log) iterations ok, $(grep -c 'long=[1-9]' gpurun_out/soak_long4.log) with long runs; $(grep -c 'AssertionError\|Traceback' gpurun_out/soak_long4.log) failures"
[ $rc -eq 124 ] || [ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --skip-no-hint > gpurun_out/b_final.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/b_final.json').read().strip().splitlines()[-1])
print(d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['traffic'])
c=d['config2']; print(c['value']/1e9, c['ms_per_step'], c['bfs'], c['roofline']['kernel_ms'])
PY
