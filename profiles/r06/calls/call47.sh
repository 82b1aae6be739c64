mkdir -p gpurun_out/r06
{ timeout 300 python tools/debug/case1428.py; MPCVR_LIB=$PWD/gpurun_in/libmpcvr_pqenc.so timeout 300 python tools/debug/case1428.py; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/case1428_variants.txt; cat gpurun_out/r06/case1428_variants.txt
