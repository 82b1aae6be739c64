#!/bin/bash
# the per-frame path against the number of hardware queues the HIP runtime spreads its streams over (GPU_MAX_HW_QUEUES, default 4)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
run() { python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d.get('process_per_frame') or {}
print(json.dumps({'tag':'$1','batch':round(d['value']),'per_frame':p.get('frames_per_s'),'one_after_the_other':p.get('frames_per_s_one_after_the_other')}))"; }
( run default
  GPU_MAX_HW_QUEUES=8 run q8
  GPU_MAX_HW_QUEUES=8 MPCVR_FRAME_LANES=6 run q8_lanes6
  GPU_MAX_HW_QUEUES=8 MPCVR_FRAME_LANES=8 run q8_lanes8
  GPU_MAX_HW_QUEUES=8 MPCVR_FRAME_LANES=6 MPCVR_FUSED_SEG=108 run q8_lanes6_seg108
  GPU_MAX_HW_QUEUES=8 MPCVR_FRAME_LANES=8 MPCVR_FUSED_SEG=144 run q8_lanes8_seg144
  GPU_MAX_HW_QUEUES=16 MPCVR_FRAME_LANES=8 run q16_lanes8
  GPU_MAX_HW_QUEUES=2 run q2
  run default_again ) | tee $O/per_frame_hw_queues.jsonl
