O=gpurun_out/r06; mkdir -p $O
for rep in 1 2; do for t in off 2:90 4:90 4:60 8:90 8:60 8:120 16:90; do
  if [ $t = off ]; then unset MPCVR_FUSED_TAIL; else export MPCVR_FUSED_TAIL=$t; fi
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tail $t', 'value', d['value'], 'frac', d['roofline']['frac'], 'kernel ms', d['roofline']['kernel_ms_per_launch'])
"
done; done > $O/fused_tail_segments_ab_call35.txt 2>&1; cat $O/fused_tail_segments_ab_call35.txt
