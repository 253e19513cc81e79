#!/bin/bash
# HBM traffic of the bench kernels from PMC counters (separate passes, kernel-trace only), MI355X_MICROARCH.md "HBM".
# Two steps without warm-up: dispatch 1 of every kernel is the COLD step (a table never touched before), dispatch 2 the
# WARM one (mc_clear, the same table memory again -- the step the bench times)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_$ctr -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline "$@" > gpurun_out/pmc_$ctr.log 2>&1
  echo "$ctr rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
# one row per kernel and ordinal of its dispatch in the run (a kernel used by two phases stays separable)
out = collections.OrderedDict()
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    seen = collections.Counter()
    rows = []
    for f in glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv" % ctr):
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == ctr]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    per_dispatch = collections.OrderedDict()
    for r in rows:  # counters of one dispatch may come in several rows (one per XCD / dimension): sum them
        per_dispatch.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"]), 0.0)
        per_dispatch[(int(r["Dispatch_Id"]), r["Kernel_Name"])] += float(r["Counter_Value"])
    for (did, kname), v in per_dispatch.items():
        name = kname.split("(")[0].replace("void ", "")
        seen[name] += 1
        d = out.setdefault("%s#%d" % (name, seen[name]), {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
        d[ctr] += v
import os
with open("gpurun_out/pmc_summary.csv", "w") as f:
    # bench.py reports these bytes as roofline.traffic only while the counting kernels are unchanged since this commit
    f.write("# commit %s\n" % os.environ.get("MC_COMMIT", "unknown"))
    import hashlib
    hh = hashlib.sha256()
    for src in ("count_pipeline.h", "kmer_device.h"):
        hh.update(open(os.path.join("metacherchant_amd", "csrc", src), "rb").read())
    f.write("# sources %s  (sha256[:16] of csrc/count_pipeline.h + csrc/kmer_device.h as measured: bench.py reports these bytes only while they are unchanged)\n" % hh.hexdigest()[:16])
    f.write("kernel,dispatch,FETCH_SIZE_KB,WRITE_SIZE_KB,fetch_GB_corrected_x2,write_GB\n")
    for k, d in out.items():
        name, n = k.rsplit("#", 1)
        f.write("\"%s\",%s,%.0f,%.0f,%.3f,%.3f\n" % (name, n, d["FETCH_SIZE"], d["WRITE_SIZE"], 2 * d["FETCH_SIZE"] * 1024 / 1e9, d["WRITE_SIZE"] * 1024 / 1e9))
print("".join(l for l in open("gpurun_out/pmc_summary.csv") if "rocclr" not in l))
PY
