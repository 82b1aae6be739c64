O=gpurun_out/r06; mkdir -p $O
{ timeout 300 python tools/debug/case4319.py; MPCVR_LIB=$PWD/gpurun_in/libmpcvr_dvexp1.so timeout 300 python tools/debug/case4319.py; } 2>&1 | grep -v amdgpu.ids | tee $O/case4319.txt
