#!/bin/bash
# error-diffusion pass after the instruction-count work (98 -> 76 VALU per step): tests, the c4ed line (with the per-frame leg), kernel time
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_errdiff.py -m gpu -q 2>&1 | tail -3
timeout 200 python bench.py --workload c4ed --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 > $O/bench_c4ed_final.json
python -c "
import json; r=json.loads(open('gpurun_out/bench_c4ed_final.json').read()); print(r['value'], r['ms_per_step'], r['roofline']['frac'], (r.get('process_per_frame') or {}).get('frames_per_s'))"
( cd /tmp; cd "$GRAFT_REPO_ROOT"; timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4ed_kt -o c4ed -- python bench.py --workload c4ed --steps 10 --warmup 3 --no-cpu-baseline --no-host-path > /dev/null 2>&1 )
f=$(find /tmp/c4ed_kt -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/errdiff_c4ed_final_kernel_stats.csv && head -3 $O/errdiff_c4ed_final_kernel_stats.csv | cut -c1-200
