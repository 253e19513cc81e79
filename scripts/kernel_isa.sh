#!/bin/bash
# Disassembles the gfx950 code object inside a built library and prints one kernel's ISA (or the list of kernels).
#   scripts/kernel_isa.sh                      -> kernel names with their register / LDS use
#   scripts/kernel_isa.sh k_p3_dedupILb1       -> the instructions of the first kernel whose mangled name contains that
#   MC_LIB=metacherchant_amd/lib/libmcgpu_x.so scripts/kernel_isa.sh ...   -> another build
set -e
LLVM=/opt/rocm/lib/llvm/bin
LIB=${MC_LIB:-$(dirname "$0")/../metacherchant_amd/lib/libmcgpu.so}
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$T/fat "$LIB" $T/stripped.so
$LLVM/clang-offload-bundler --unbundle --type=o --input=$T/fat --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
if [ -z "$1" ]; then
    $LLVM/llvm-readelf --notes $T/k.co | grep -E "^\s+\.(name|vgpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size):" | paste - - - - - | sed 's/  */ /g'
    exit 0
fi
$LLVM/llvm-objdump -d --no-show-raw-insn $T/k.co > $T/all.s
awk -v pat="$1" '
    /^[0-9a-f]+ <.*>:$/ { on = index($0, pat) > 0 ? (seen ? 0 : 1) : 0; if (on) seen = 1 }
    on { print }' $T/all.s
