#!/bin/bash
# tools/build_nopk.sh — the round-3 experiment behind DESIGN.md §4.2 "matrix pipe beside the non-packed remainder": relink libmpcvr.so
# with the matrix-core 2x kernel (vp_fused_mx.hip, MPCVR_FLAG_FUSED_MFMA) compiled WITHOUT packed fp32 — every v_pk_fma_f32 / v_pk_mul_f32
# of its convert stage and epilogue as two plain FMAs (-DMPCVR_NO_PK for the inline-asm helpers, -target-feature -packed-fp32-ops for the
# compiler's own) — because tools/ubench/mfma_overlap.hip shows the matrix pipe overlapping plain VALU (0.8-0.9) but not packed fp32 (< 0).
# The arithmetic is the same FMA for FMA (parity suite green on it).  Restore the product build with
#   touch videorenderer_amd/csrc/vp_fused_mx.hip && python -m videorenderer_amd.build
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
B="$HERE/videorenderer_amd/_build"
/opt/rocm/bin/hipcc -x hip -c "$HERE/videorenderer_amd/csrc/vp_fused_mx.hip" -o "$B/vp_fused_mx.hip.o" -DMPCVR_NO_PK -Xclang -target-feature -Xclang -packed-fp32-ops \
    -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$HERE/videorenderer_amd/libmpcvr.so" "$B"/*.o
echo "libmpcvr.so relinked with the non-packed matrix-core kernel: python bench.py --flags 32"
