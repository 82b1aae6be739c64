mkdir -p gpurun_out/r06
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path"
one() { # workload, env...
  w=$1; shift
  r=$(env "$@" $B --workload $w 2>/dev/null | tail -n 1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['frac'], r['roofline']['kernel_ms_per_launch'])")
  echo "$w $* -> $r"
}
{
echo "== prefetch build, default segs"
for w in up1440 down1440 down1080 up2160; do one $w X=1; done
echo "== up1440 seg sweep (PB=8)"
for s in 48 64 72 80 96 104 120 128 144 160 184 240; do one up1440 MPCVR_PERIOD_SEG=$s; done
echo "== down1440 seg sweep (PB=4)"
for s in 24 32 40 48 60 72 80 96 120 144 180; do one down1440 MPCVR_PERIOD_SEG=$s; done
echo "== down1080 seg sweep (PB=3)"
for s in 18 24 30 36 45 54 60 72 90 108 135; do one down1080 MPCVR_PERIOD_SEG=$s; done
echo "== up2160 seg sweep (PB=18)"
for s in 36 54 72 90 108 144 180 216 270; do one up2160 MPCVR_PERIOD_SEG=$s; done
echo "== waves per workgroup, up1440 / down1440 at default seg"
for wv in 4 6 8 12 16; do one up1440 MPCVR_PERIOD_WAVES=$wv; one down1440 MPCVR_PERIOD_WAVES=$wv; done
} > gpurun_out/r06/period_sweep_call5.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r06/gpu_suite_call5.txt
