mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r06/gpu_suite_call11.txt
