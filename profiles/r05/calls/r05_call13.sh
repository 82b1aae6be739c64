#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05c13; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests_gpu.txt 2>&1; echo "rc=$?" >> $O/tests_gpu.txt
tail -6 $O/tests_gpu.txt
B="--steps 20 --warmup 3 --no-cpu-baseline --no-host-path"
for wl in jinc1080 dovi4k up1440 down1440 down1080 up1080 up1440_nv12 c4ed; do timeout 300 python bench.py --workload $wl $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'workload': '$wl', 'frames_s': d['value'], 'frac': d['roofline']['frac'], 'shape_peak': d['roofline'].get('empirical_shape_peak_GBps'), 'copy_peak': d['roofline'].get('empirical_copy_peak_GBps')}))" >> $O/bench_workloads.jsonl; done
cat $O/bench_workloads.jsonl
timeout 300 python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/bench_default.json; cut -c1-1500 $O/bench_default.json
