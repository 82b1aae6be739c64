#!/bin/bash
# tools/build_variant.sh <name> <source.hip> <hipcc flags...> — an experiment build of ONE translation unit linked with the
# objects of the regular build into gpurun_in/libmpcvr_<name>.so (MPCVR_LIB=... selects it in bench.py / the tests' tools).
#   tools/build_variant.sh w5 vp_fused.hip -DMPCVR_STREAM_WAVES_PER_EU=5
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift 2
B=$HERE/videorenderer_amd/_build; mkdir -p $HERE/gpurun_in /tmp/variant_$NAME
/opt/rocm/bin/hipcc -x hip -c $HERE/videorenderer_amd/csrc/$SRC -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -o /tmp/variant_$NAME/$SRC.o
OBJS=$(ls $B/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $HERE/gpurun_in/libmpcvr_$NAME.so $OBJS /tmp/variant_$NAME/$SRC.o
ls -la $HERE/gpurun_in/libmpcvr_$NAME.so
