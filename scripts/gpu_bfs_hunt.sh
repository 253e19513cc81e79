#!/bin/bash
# One gpurun call of the hunt for the wrong walk of round 3: scripts/bfs_endgame_stress.py on the product library (with a
# second context counting beside it), on the fuzzed build of the fixed kernel and -- last, each under a time limit of its
# own -- on the kernel as round 3 shipped it (-DMC_BFS_OLD_RACE), plain under contention and fuzzed.  HUNT_STEPS picks the
# steps (default: all), HUNT_WALKS the walks per step.  Logs under gpurun_out/hunt_*.log.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=metacherchant_amd/lib
W=${HUNT_WALKS:-20000}
STEPS=${HUNT_STEPS:-product fuzz old fuzz_old}
# the libraries the chosen steps load (only `fuzz` comes with build(); a minute of hipcc each for the others)
NEED="'fuzz',"; case " $STEPS " in *" old "*) NEED="$NEED 'trace_old',";; esac; case " $STEPS " in *" fuzz_old "*) NEED="$NEED 'fuzz_old',";; esac
python -c "from metacherchant_amd import build as b; b.build_variants(names=($NEED))" || exit 1
for s in $STEPS; do
  case $s in
    product) timeout -k 10 900 python scripts/bfs_endgame_stress.py --walks $W --jobs 3 --contend 1 --dirs 0,0,1,-1 > gpurun_out/hunt_product_contend.log 2>&1; echo "product+contend rc=$?"; tail -1 gpurun_out/hunt_product_contend.log;;
    fuzz) MC_LIB=$L/libmcgpu_fuzz.so timeout -k 10 600 python scripts/bfs_endgame_stress.py --walks $W --jobs 3 --dirs 0,0,1,-1 > gpurun_out/hunt_fuzz_fixed.log 2>&1; echo "fuzz fixed rc=$?"; grep -v "^[0-9]* \(fast\|slow\|companion\|root_bad\)" gpurun_out/hunt_fuzz_fixed.log | cut -c1-400 | tail -4;;
    fuzz_contend) MC_LIB=$L/libmcgpu_fuzz.so timeout -k 10 600 python scripts/bfs_endgame_stress.py --walks $W --jobs 3 --contend 1 --dirs 0,0,1,-1 > gpurun_out/hunt_fuzz_fixed_contend.log 2>&1; echo "fuzz fixed + contend rc=$?"; grep -v "^[0-9]* \(fast\|slow\|companion\|root_bad\)" gpurun_out/hunt_fuzz_fixed_contend.log | cut -c1-400 | tail -4;;
    old) MC_LIB=$L/libmcgpu_trace_old.so timeout -k 10 300 python scripts/bfs_endgame_stress.py --walks $W --jobs 3 --contend 1 > gpurun_out/hunt_old_contend.log 2>&1; echo "old+contend rc=$?"; grep -v "^[0-9]* \(fast\|slow\|companion\|root_bad\)" gpurun_out/hunt_old_contend.log | cut -c1-400 | tail -4;;
    fuzz_old) MC_LIB=$L/libmcgpu_fuzz_old.so timeout -k 10 240 python scripts/bfs_endgame_stress.py --walks 600 --jobs 3 > gpurun_out/hunt_fuzz_old.log 2>&1; echo "fuzz old rc=$?"; grep -v "^[0-9]* \(fast\|slow\|companion\|root_bad\)" gpurun_out/hunt_fuzz_old.log | cut -c1-300 | tail -3;;
  esac
done
