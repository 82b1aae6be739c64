O=gpurun_out/r06; mkdir -p $O
for rep in 1 2; do for v in table literal; do
  if [ $v = table ]; then unset MPCVR_LIB; else export MPCVR_LIB=$PWD/gpurun_in/libmpcvr_dvexp1.so; fi
  python bench.py --workload dovi4k --steps 40 --warmup 8 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dovi4k EOTF $v', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch'])
"
done; done | tee $O/dovi4k_literal_eotf_ab.txt
