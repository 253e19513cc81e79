#!/bin/bash
# SQ counters of the bench kernels (separate passes of up to 8 counters, kernel-trace only).
# usage: bash scripts/gpu_pmc_sq.sh <tag> [bench args...]
tag=$1; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
p=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
  p=$((p+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/sq_${tag}_$p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > gpurun_out/sq_${tag}_$p.log 2>&1
  echo "pass $p rc=$?"
done
python3 - "$tag" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
acc = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/sq_%s_*/*/*counter_collection.csv" % tag)):
    first = {}
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        # first dispatch of each kernel only
        did = int(r["Dispatch_Id"])
        if name not in first: first[name] = did
        if first[name] != did: continue
        acc.setdefault(name, collections.OrderedDict()).setdefault(r["Counter_Name"], 0.0)
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
with open("gpurun_out/sq_%s_summary.csv" % tag, "w") as out:
    for k, d in acc.items():
        if not any(x in k for x in ("k_p3", "k_p1", "k_p2", "k_sk1", "k_sk2", "k_skl", "k_solid", "k_bfs")): continue
        line = k + "," + ",".join("%s=%.4g" % kv for kv in d.items())
        out.write(line + "\n"); print(line)
PY
