O=gpurun_out/r06; mkdir -p $O
for wlk in c3hdr up1440 down1440 c1 c2 up1080; do for cnt in 2 3 4; do
  MPCVR_BATCH_LANE_COUNT=$cnt python bench.py --workload $wlk --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); b = d.get('process_batch_on_lanes') or {}
print('$wlk lanes=$cnt', 'value', d['value'], '| on lanes', b.get('frames_per_s'), b.get('hbm_frac'))
"
done; done > $O/batch_lane_count_call34.txt 2>&1; cat $O/batch_lane_count_call34.txt
