O=gpurun_out/r06; mkdir -p $O/soak4
timeout 300 python tools/debug/case1428.py 2>&1 | grep -v amdgpu.ids | tee $O/case1428_after_trim_fix.txt
for rep in 1 2; do for v in before new; do
  if [ $v = new ]; then unset MPCVR_LIB; else export MPCVR_LIB=$PWD/gpurun_in/libmpcvr_before.so; fi
  python bench.py --workload dovi4k --steps 40 --warmup 8 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dovi4k $v', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch'])
"
done; done | tee $O/dovi4k_trim_fix_ab.txt
unset MPCVR_LIB
timeout 1200 python -m pytest tests -q -m gpu -k "dovi or soak_case or fuzz_case" 2>&1 | tail -4
( time MPCVR_FUZZ_JINC=1 timeout 1200 python tests/tools/fuzz_strip.py 5000 5102 ) > $O/soak4/jinc_5000_seed5102_rerun2.txt 2>&1; echo "rc=$?" >> $O/soak4/jinc_5000_seed5102_rerun2.txt; tail -2 $O/soak4/jinc_5000_seed5102_rerun2.txt
