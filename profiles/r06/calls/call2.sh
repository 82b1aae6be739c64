mkdir -p gpurun_out/r06
python tools/debug/case1820.py > gpurun_out/r06/case1820_after.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r06/gpu_suite_crmath.txt
