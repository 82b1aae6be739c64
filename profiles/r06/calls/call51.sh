O=gpurun_out/r06; mkdir -p $O
for v in dvexp1 dvexp3; do MPCVR_LIB=$PWD/gpurun_in/libmpcvr_$v.so timeout 300 python tools/debug/case1428.py 2>&1 | grep -v amdgpu.ids | head -3; done | tee $O/case1428_dvexp_variants.txt
