#!/bin/bash
# tools/build_variant_multi.sh <name> "<src1.hip src2.hip ...>" <hipcc flags...> — like build_variant.sh for SEVERAL translation units that have to
# agree on a macro (e.g. the periodic kernel's per-ratio TUs + its launcher): gpurun_in/libmpcvr_<name>.so, selected with MPCVR_LIB.
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRCS=$2; shift 2
B=$HERE/videorenderer_amd/_build; mkdir -p $HERE/gpurun_in /tmp/variant_$NAME
OBJS=$(ls $B/*.o)
for SRC in $SRCS; do
  ( /opt/rocm/bin/hipcc -x hip -c $HERE/videorenderer_amd/csrc/$SRC -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -o /tmp/variant_$NAME/$SRC.o ) &
  OBJS=$(echo "$OBJS" | grep -v "/$SRC.o")
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $HERE/gpurun_in/libmpcvr_$NAME.so $OBJS /tmp/variant_$NAME/*.o
ls -la $HERE/gpurun_in/libmpcvr_$NAME.so
