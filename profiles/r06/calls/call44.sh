mkdir -p gpurun_out/r06; timeout 900 python tools/debug/case5624.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/cases_5624_1428.txt; cat gpurun_out/r06/cases_5624_1428.txt
