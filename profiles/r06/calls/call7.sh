mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "adapter or displayed or lds or bench" 2>&1 | tail -15 > gpurun_out/r06/gpu_new_tests_call7.txt
