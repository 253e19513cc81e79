#!/bin/bash
# scratch: long-record tests, then the default bench line
set -o pipefail
timeout -k 10 800 python -m pytest tests/test_gpu_long_records.py -x -q > gpurun_out/long1.log 2>&1; rc=$?; tail -3 gpurun_out/long1.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --steps 3 --warmup 1 > gpurun_out/b_long1.json 2> gpurun_out/b_long1.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/b_long1.json').read().strip().splitlines()[-1])
print(d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_ms'])
c=d['config2']; print(c['value']/1e9, c['ms_per_step'], c['bfs'], c['roofline']['kernel_ms'])
PY
