mkdir -p gpurun_out/r06
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path"
one() { r=$(MPCVR_LIB=$2 $B --workload $3 2>/dev/null | tail -n 1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['frac'], r['roofline']['kernel_ms_per_launch'])"); echo "$3 [$1] -> $r"; }
BEFORE=$PWD/gpurun_in/libmpcvr_before_gamma.so; AFTER=$PWD/videorenderer_amd/libmpcvr.so
{
for rep in 1 2; do for w in up1440 down1440 down1080 up2160; do one before $BEFORE $w; one unseen_prefetch $AFTER $w; done; done
} > gpurun_out/r06/period_unseen_prefetch_ab_call9.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r06/gpu_suite_call9.txt
