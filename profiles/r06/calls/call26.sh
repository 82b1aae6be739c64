O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x -k "lanes or behind_a_batch or batch or give_up or frames" > $O/gpu_batch_lane_tests_call26.txt 2>&1; tail -5 $O/gpu_batch_lane_tests_call26.txt
for wlk in c3hdr up1440 down1440 down1080 up2160 c1 c2 c5 up1440_nv12 jinc1080; do
  python bench.py --workload $wlk --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); b = d.get('process_batch_on_lanes') or {}
print('$wlk', 'value', d['value'], 'frac', d['roofline']['frac'], '| on lanes', b.get('frames_per_s'), b.get('hbm_frac'), b.get('lanes'))
"
done > $O/bench_batch_lanes_call26.txt 2>&1; cat $O/bench_batch_lanes_call26.txt
