mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "adapter or displayed or lds or transcendentals or 1820 or bench" 2>&1 | tail -15 > gpurun_out/r06/gpu_new_tests_call6.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_with_up2x_probe.json 2> gpurun_out/r06/bench_with_up2x_probe.err
python bench.py --workload c5 --steps 20 --warmup 5 --no-cpu-baseline --no-host-path > gpurun_out/r06/bench_c5_probe.json 2>/dev/null
