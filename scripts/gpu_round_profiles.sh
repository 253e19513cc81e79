#!/bin/bash
# Everything under profiles/<round>_* in ONE gpurun call: bench lines (E1 with the CPU baseline, E0, 20 / 50 / 100 M reads,
# configs[2] scaled and full), rocprofv3 kernel stats, PMC traffic (cold and warm step), SQ counters, the step's timeline.
# usage: ROUND=r04 MC_COMMIT=<short hash the kernels were built from> bash scripts/gpu_round_profiles.sh
R=${ROUND:-r06}
set -x
mkdir -p gpurun_out/p
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python bench.py > gpurun_out/p/${R}_bench_e1.json 2> gpurun_out/p/bench_e1.err
timeout -k 10 300 python bench.py --err 0 --no-cpu-baseline --skip-config2 > gpurun_out/p/${R}_bench_e0.json 2>/dev/null
timeout -k 10 300 python bench.py --no-cpu-baseline --skip-no-hint --reads 20000000 --steps 4 --warmup 1 --skip-config2 > gpurun_out/p/${R}_bench_e1_20Mreads.json 2>/dev/null
timeout -k 10 400 python bench.py --no-cpu-baseline --skip-no-hint --reads 50000000 --steps 3 --warmup 1 --skip-config2 > gpurun_out/p/${R}_bench_e1_50Mreads.json 2>/dev/null
timeout -k 10 500 python bench.py --no-cpu-baseline --skip-no-hint --reads 100000000 --steps 2 --warmup 1 --skip-config2 > gpurun_out/p/${R}_bench_e1_100Mreads.json 2>/dev/null
timeout -k 10 300 python bench.py --no-cpu-baseline --config 2 --reads 10000000 --contigs 10 --steps 4 --warmup 1 > gpurun_out/p/${R}_bench_config2_scaled_10Mreads.json 2>/dev/null
timeout -k 10 600 python bench.py --no-cpu-baseline --config 2 --steps 2 --warmup 1 > gpurun_out/p/${R}_bench_config2_full.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p/prof_e1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-no-hint --config2-steps 2 > gpurun_out/p/prof_e1.log 2>&1
for f in $(find gpurun_out/p/prof_e1 -name '*kernel_stats*.csv'); do cp $f gpurun_out/p/${R}_e1_kernel_stats.csv; done
bash scripts/gpu_pmc.sh --skip-no-hint --skip-config2 > gpurun_out/p/pmc.log 2>&1; cp gpurun_out/pmc_summary.csv gpurun_out/p/${R}_pmc_hbm_traffic_e1.csv
bash scripts/gpu_pmc.sh --skip-no-hint --config 2 --reads 10000000 --contigs 10 > gpurun_out/p/pmc_c2.log 2>&1; cp gpurun_out/pmc_summary.csv gpurun_out/p/${R}_pmc_hbm_traffic_config2_scaled.csv
bash scripts/gpu_pmc_sq.sh ${R} --skip-no-hint --skip-config2 > gpurun_out/p/sq.log 2>&1; cp gpurun_out/sq_${R}_summary.csv gpurun_out/p/${R}_sq_counters_e1.csv
bash scripts/gpu_timeline.sh --skip-no-hint --skip-config2 > /dev/null 2>&1; cp gpurun_out/timeline.txt gpurun_out/p/${R}_timeline_e1.txt
timeout -k 10 300 python scripts/rank_phases.py 8 > gpurun_out/p/${R}_rank_phases_8owners.txt 2>&1
# the CPU port on the WHOLE workload, once a round (the bench line's cpu_baseline times a 1 M-read sample)
timeout -k 10 900 python bench.py --steps 2 --warmup 1 --skip-no-hint --skip-config2 --cpu-baseline-full > gpurun_out/p/${R}_bench_e1_cpu_full.json 2>/dev/null
ls -la gpurun_out/p
