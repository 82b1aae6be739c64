#!/bin/bash
# the per-frame path (frame lanes) against the segment height and the lane count: same box, one bench line each
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
run() { python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d.get('process_per_frame') or {}
print(json.dumps({'tag':'$1','batch':round(d['value']),'per_frame':p.get('frames_per_s'),'one_after_the_other':p.get('frames_per_s_one_after_the_other')}))"; }
( run default
  for seg in 60 90 108 120 144; do MPCVR_FUSED_SEG=$seg run seg$seg; done
  for l in 3 6 8; do MPCVR_FRAME_LANES=$l run lanes$l; done
  MPCVR_FRAME_LANES=6 MPCVR_FUSED_SEG=108 run lanes6_seg108
  MPCVR_FRAME_LANES=8 MPCVR_FUSED_SEG=144 run lanes8_seg144
  run default_again ) | tee $O/per_frame_sweep.jsonl
