#!/bin/bash
# round 4, the walk: parity tests of the BFS, the end-game stress on the fuzzed build, then the bench (BFS phase) and the
# scout's per-hop split (-DMC_SCOUT_TIMING build).  Logs under gpurun_out/.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -c "from metacherchant_amd import build as b; b.build_variants(names=('fuzz', 'sctime'))" || exit 1   # (only `fuzz` comes with build())
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bfs_race.py -x -q -m gpu -k "bfs or walk or race or fixed or selfcheck or round3" > gpurun_out/r4_bfs_tests.log 2>&1; rc=$?; echo "bfs tests rc=$rc"; tail -3 gpurun_out/r4_bfs_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --skip-no-hint > gpurun_out/r4_bench_bfs.json 2> gpurun_out/r4_bench_bfs.err; echo "bench rc=$?"; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r4_bench_bfs.json') if l.startswith('{')][-1])
print("value %.2f Gk-mers/s, %.2f ms/step, count %.2f ms, bfs %s" % (j['value']/1e9, j['ms_per_step'], j['roofline'].get('count_ms_per_step', -1), json.dumps(j.get('bfs'))))
PY
MC_BFS_STATS=1 MC_LIB=metacherchant_amd/lib/libmcgpu_sctime.so timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-no-hint > gpurun_out/r4_bench_sctime.json 2> gpurun_out/r4_bench_sctime.err; echo "sctime rc=$?"; grep -h "scout companion\|bfs job" gpurun_out/r4_bench_sctime.err gpurun_out/r4_bench_sctime.json | tail -8
