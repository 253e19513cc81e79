cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_ATOMIC_RETURN SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d gpurun_out/sq3 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/sq3.log 2>&1
echo rc=$?
python3 - <<'PY'
import csv, glob, collections
acc = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/sq3/*/*counter_collection.csv")):
    first = {}
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        did = int(r["Dispatch_Id"])
        if name not in first: first[name] = did
        if first[name] != did: continue
        acc.setdefault(name, collections.OrderedDict()).setdefault(r["Counter_Name"], 0.0)
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    if any(x in k for x in ("k_p3", "k_sk1", "k_sk2")): print(k, dict((a, "%.3g" % b) for a, b in d.items()))
PY
