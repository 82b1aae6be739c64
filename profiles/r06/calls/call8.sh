mkdir -p gpurun_out/r06
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path"
one() { # label lib workload
  r=$(MPCVR_LIB=$2 $B --workload $3 2>/dev/null | tail -n 1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['frac'], r['roofline']['kernel_ms_per_launch'], r['config']['path'][:60])")
  echo "$3 [$1] -> $r"
}
BEFORE=$PWD/gpurun_in/libmpcvr_before_gamma.so; AFTER=$PWD/videorenderer_amd/libmpcvr.so; PG=$PWD/gpurun_in/libmpcvr_period_gamma.so
{
for rep in 1 2; do
for w in c3hdr c4 c5 c3hdr_1080p; do one before $BEFORE $w; one gamma_table $AFTER $w; done
done
for w in up1440 down1440 down1080 up2160; do one before $BEFORE $w; one period_alu $AFTER $w; one period_gamma_spills $PG $w; done
echo "== strip kernel (MPCVR_NO_PERIOD=1)"
for w in up1440 down1440; do MPCVR_NO_PERIOD=1 one before $BEFORE $w; MPCVR_NO_PERIOD=1 one gamma_table $AFTER $w; done
} > gpurun_out/r06/gamma_table_ab_call8.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r06/gpu_suite_call8.txt
