#!/bin/bash
# Benchmarks tuning builds side by side in ONE gpurun call: every argument is  name[:lib-variant[:ENV=VAL,ENV=VAL...]]  -- the
# bench (configs[1] E1, 8 steps) runs with MC_LIB=metacherchant_amd/lib/libmcgpu_<lib-variant>.so (empty: the product library)
# and the given environment; one summary line per argument, full JSON lines under gpurun_out/variants/.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/variants
for spec in "$@"; do
  IFS=: read -r name lib envs <<< "$spec"
  (
    [ -n "$lib" ] && export MC_LIB=metacherchant_amd/lib/libmcgpu_$lib.so
    IFS=, read -ra kv <<< "$envs"; for e in "${kv[@]}"; do [ -n "$e" ] && export "$e"; done
    timeout -k 10 240 python bench.py --steps ${VAR_STEPS:-8} --warmup 2 --no-cpu-baseline --skip-no-hint ${VAR_ARGS:-} > gpurun_out/variants/$name.json 2> gpurun_out/variants/$name.err
    echo -n "$name rc=$? "
  )
  python - "$name" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open('gpurun_out/variants/%s.json' % sys.argv[1]) if l.startswith('{')][-1])
    r = j['roofline']; km = r.get('kernel_ms', {})
    print("%.2f Gk/s  %.2f ms/step  count %.2f (%s)  bfs %.2f ms  table %.1f GB" % (j['value'] / 1e9, j['ms_per_step'], r.get('count_ms_per_step', -1),
          " ".join("%s %.2f" % (k.replace('k_', ''), v) for k, v in km.items() if v), j['bfs']['ms_per_step'], j.get('table_bytes', 0) / 1e9))
except Exception as e:
    print("no result:", e)
PY
done
