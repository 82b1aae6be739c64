O=gpurun_out/r06; mkdir -p $O
for wlk in up1440 down1440; do for seg in 0 48 64 96 120 160 192; do
  MPCVR_BATCH_LANES_ALL=1 MPCVR_PERIOD_SEG=$seg python bench.py --workload $wlk --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); b = d.get('process_batch_on_lanes') or {}
print('$wlk seg $seg', 'value', d['value'], '| on lanes', b.get('frames_per_s'), b.get('lanes'))
"
done; done > $O/period_lanes_seg_sweep_call31.txt 2>&1; cat $O/period_lanes_seg_sweep_call31.txt
