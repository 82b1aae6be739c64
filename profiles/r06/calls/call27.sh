O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x -k "lanes or behind_a_batch or batch or give_up or frames" > $O/gpu_batch_lane_tests_call27.txt 2>&1; tail -5 $O/gpu_batch_lane_tests_call27.txt
for rep in 1 2; do for wlk in c3hdr c1 c2 c3 c4 c5 hdr4k c3hdr_1080p hdrpass_2x up1440; do
  python bench.py --workload $wlk --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); b = d.get('process_batch_on_lanes') or {}
print('$wlk', 'value', d['value'], 'frac', d['roofline']['frac'], '| on lanes', b.get('frames_per_s'), b.get('hbm_frac'), b.get('lanes'))
"
done; done > $O/bench_batch_lanes_call27.txt 2>&1; cat $O/bench_batch_lanes_call27.txt
