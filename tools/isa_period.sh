#!/bin/bash
# Compile ONE instantiation of the periodic-phase kernel (k_fused_period<P, Q, 5, PQ table, P01x, integer dither>; default 4:3),
# print its register / spill figures and the instruction mix of its hot loop.   tools/isa_period.sh [4_3|3_2|2_3|1_2] [hipcc flags...]
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
OUT=${OUT:-/tmp/isa}
PQ=${1:-4_3}; shift || true
mkdir -p "$OUT" && cd "$OUT" && rm -f vp_fused_period_${PQ}-*
/opt/rocm/bin/hipcc -x hip -c "$HERE/videorenderer_amd/csrc/vp_fused_period_${PQ}.hip" -DMPCVR_PERIOD_DEV_ONLY -O3 -std=c++17 -fPIC \
    --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -save-temps=obj -o "$OUT/period.o" -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
    grep -E "error|SGPRs:|VGPRs:|Occupancy|Spill|ScratchSize" | sed 's/.*:0: *//; s/ \[-Rpass.*//' | tr '\n' ';'; echo
/opt/rocm/lib/llvm/bin/llvm-objdump -d "$OUT/vp_fused_period_${PQ}-hip-amdgcn-amd-amdhsa-gfx950.out" > "$OUT/period.lst"
python3 "$HERE/tools/isa_mix.py" "$OUT/period.lst" k_fused_period
