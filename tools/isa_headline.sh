#!/bin/bash
# Compile ONLY the headline instantiation k_fused_up2x<5, PQ table, P01x, integer dither> (seconds instead of minutes), print its
# register / spill figures and the instruction mix of its hot loop (tools/isa_mix.py), and leave the listing in $OUT.
#   tools/isa_headline.sh [extra hipcc flags...]        OUT=/tmp/isa by default
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
OUT=${OUT:-/tmp/isa}
mkdir -p "$OUT" && cd "$OUT" && rm -f vp_fused_up2x_nt5-*
/opt/rocm/bin/hipcc -x hip -c "$HERE/videorenderer_amd/csrc/vp_fused_up2x_nt5.hip" -DMPCVR_UP2X_HEADLINE_ONLY -O3 -std=c++17 -fPIC \
    --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -save-temps=obj -o "$OUT/nt5.o" -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
    grep -E "SGPRs:|VGPRs:|Occupancy|Spill|ScratchSize" | sed 's/.*:0: *//; s/ \[-Rpass.*//' | tr '\n' ';'; echo
/opt/rocm/lib/llvm/bin/llvm-objdump -d "$OUT/vp_fused_up2x_nt5-hip-amdgcn-amd-amdhsa-gfx950.out" > "$OUT/headline.lst"
python3 "$HERE/tools/isa_mix.py" "$OUT/headline.lst" k_fused_up2x
