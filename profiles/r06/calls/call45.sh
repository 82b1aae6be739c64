mkdir -p gpurun_out/r06; timeout 900 python tools/debug/case5624b.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/case5624b.txt; cat gpurun_out/r06/case5624b.txt
