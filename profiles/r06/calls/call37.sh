O=gpurun_out/r06; mkdir -p $O
for rep in 1 2 3; do for v in base stagger20 stagger39 stagger78; do
  if [ $v = base ]; then unset MPCVR_LIB; else export MPCVR_LIB=$PWD/gpurun_in/libmpcvr_$v.so; fi
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'value', d['value'], 'frac', d['roofline']['frac'], 'kernel ms', d['roofline']['kernel_ms_per_launch'])
"
done; done > $O/up2x_start_stagger_ab_call37.txt 2>&1; cat $O/up2x_start_stagger_ab_call37.txt
