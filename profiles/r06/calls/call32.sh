O=gpurun_out/r06; mkdir -p $O
for rep in 1 2; do for wlk in up1440 down1440 down1080 up2160 up1080 up1440_nv12 hdrpass_1440 jinc1080; do
  MPCVR_BATCH_LANES_ALL=1 python bench.py --workload $wlk --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); b = d.get('process_batch_on_lanes') or {}
print('$wlk', 'value', d['value'], d['roofline']['frac'], '| on lanes', b.get('frames_per_s'), b.get('hbm_frac'), b.get('lanes'), b.get('steps'))
"
done; done > $O/bench_batch_lanes_all_routes_call32.txt 2>&1; cat $O/bench_batch_lanes_all_routes_call32.txt
