#!/bin/bash
# HBM traffic of the bench kernels from PMC counters (separate passes, kernel-trace only), MI355X_MICROARCH.md "HBM"
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_$ctr -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > gpurun_out/pmc_$ctr.log 2>&1
  echo "$ctr rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
out = collections.OrderedDict()
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv" % ctr):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if row["Counter_Name"] != ctr:
                continue
            d = out.setdefault(name, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0})
            d[ctr] += float(row["Counter_Value"])
            if ctr == "FETCH_SIZE":
                d["n"] += 1
with open("gpurun_out/pmc_summary.csv", "w") as f:
    f.write("kernel,dispatches,FETCH_SIZE_KB_sum,WRITE_SIZE_KB_sum,fetch_GB_corrected_x2,write_GB\n")
    for k, d in out.items():
        f.write("\"%s\",%d,%.0f,%.0f,%.3f,%.3f\n" % (k, d["n"], d["FETCH_SIZE"], d["WRITE_SIZE"], 2 * d["FETCH_SIZE"] * 1024 / 1e9, d["WRITE_SIZE"] * 1024 / 1e9))
print(open("gpurun_out/pmc_summary.csv").read())
PY
