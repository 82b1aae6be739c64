O=gpurun_out/r06; mkdir -p $O
for rep in 1 2; do for seg in 0 12 14 16 18 20 22 24 34; do
  MPCVR_FUSED_SEG=$seg python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = d['process_per_frame']
print('seg', $seg, 'batch', d['value'], 'lanes', p['frames_per_s_native_loop'], 'in order', p['frames_per_s_one_after_the_other_native_loop'], 'kernel ms', p['last_process_ms_one_after_the_other'], p['last_process_ms'])
"
done; done > $O/per_frame_seg_sweep2.txt 2>&1
cat $O/per_frame_seg_sweep2.txt
