#!/bin/bash
# round 5, call 16: the exact convert form as a compile-time capability (exact_capable) — suite + the workloads the branch had slowed
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 5 1500 python -m pytest tests -m gpu -q -x > $O/call16_suite.txt 2>&1; tail -3 $O/call16_suite.txt
for wl in c1 c3 c2 c3hdr up1080 up1440_nv12 hdrpass_1440 hdrpass_2x hdr4k up1440; do
  python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1
done > $O/call16_bench.jsonl
MPCVR_EXACT8=0 python bench.py --workload up1440_nv12 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 > $O/call16_up1440_nv12_fast.json
python - <<'PY'
import json
for l in open('gpurun_out/call16_bench.jsonl'):
    try: d=json.loads(l); print(d['config']['workload'], round(d['value']), d['roofline']['frac'])
    except Exception as e: print('bad', l[:80])
d=json.loads(open('gpurun_out/call16_up1440_nv12_fast.json').read()); print('up1440_nv12 fast', round(d['value']))
PY
