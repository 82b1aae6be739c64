#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_errdiff.py -m gpu -q 2>&1 | tail -3
timeout 200 python bench.py --workload c4ed --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 > $O/bench_c4ed_final.json
python -c "
import json; r=json.loads(open('gpurun_out/bench_c4ed_final.json').read()); print(r['value'], r['ms_per_step'], r['roofline']['frac'], (r.get('process_per_frame') or {}).get('frames_per_s'))"
