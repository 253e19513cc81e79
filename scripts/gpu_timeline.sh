#!/bin/bash
# kernel timeline of the last bench step (start, duration, gap to the previous kernel): where the time between the kernels goes
# usage: bash scripts/gpu_timeline.sh [bench args...]
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/trace
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/trace.log 2>&1
echo "rc=$?"
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]))
for f in glob.glob("gpurun_out/trace/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[:30]))
rows.sort()
# the last step: from the last k_tile_first_read to the end
idx = max(i for i, r in enumerate(rows) if "k_tile_first_read" in r[2])
start = max(0, idx - 12)
t0 = rows[start][0]
prev_end = rows[start][0]
with open("gpurun_out/timeline.txt", "w") as out:
    for s, e, n in rows[start:]:
        line = "%10.1f us  dur %9.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n)
        out.write(line + "\n")
        prev_end = max(prev_end, e)
print(open("gpurun_out/timeline.txt").read()[-6000:])
PY
