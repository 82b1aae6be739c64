mkdir -p gpurun_out/r06; timeout 600 python tools/debug/case2367.py > gpurun_out/r06/case2367.txt 2>&1; echo "rc=$?" >> gpurun_out/r06/case2367.txt; cat gpurun_out/r06/case2367.txt
