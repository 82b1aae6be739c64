set -x
mkdir -p gpurun_out/r06
python tools/debug/case1820.py > gpurun_out/r06/case1820_before.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_start.json 2> gpurun_out/r06/bench_start.err
for w in up1440 down1440 down1080 up2160 c5 jinc1080 c4ed; do python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-host-path >> gpurun_out/r06/bench_workloads_start.jsonl 2>/dev/null; done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06/gpu_suite_start.txt
