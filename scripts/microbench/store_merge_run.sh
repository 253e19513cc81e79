cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for gap in 100 400 1200; do
  rm -rf gpurun_out/sm_$gap
  timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/sm_$gap -- scripts/microbench/store_merge $gap > gpurun_out/sm_$gap.log 2>&1
  echo "gap $gap rc=$?"; grep "k_store" gpurun_out/sm_$gap.log
  python3 - $gap <<'PY'
import csv, glob, sys, collections
tot = collections.OrderedDict()
for f in glob.glob("gpurun_out/sm_%s/*/*counter_collection.csv" % sys.argv[1]):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "WRITE_SIZE" and "k_store" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0]
            tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
for k, v in tot.items(): print("   %s WRITE_SIZE %.1f MB" % (k, v * 1024 / 1e6))
PY
done
