O=gpurun_out/r06/final4; mkdir -p $O
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/gpu_suite.txt 2>&1; tail -4 $O/gpu_suite.txt
for wl in c3hdr c3 c4 c4ext c4ed c5 c2 c1 hdr4k up1440 down1440 up1080 down1080 up2160 up1440_nv12 hdrpass_2x hdrpass_1440 c3hdr_1080p jinc1080 dovi4k; do
  python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1
done > $O/bench_workloads.jsonl
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err
python - <<'PY'
import json
for f in ("bench_default.json", "bench_driver_shape.json"):
    d = json.loads([l for l in open("gpurun_out/r06/final4/" + f) if l.startswith("{")][-1]); b = d["process_batch_on_lanes"]
    print(f, d["value"], d["roofline"]["frac"], "| lanes", b["frames_per_s"], b["hbm_frac"])
for l in open("gpurun_out/r06/final4/bench_workloads.jsonl"):
    if l.startswith("{"):
        d = json.loads(l); b = d.get("process_batch_on_lanes") or {}
        print(d["config"]["workload"][:12].ljust(12), d["value"], d["roofline"]["frac"], "| lanes", b.get("frames_per_s"), b.get("hbm_frac"), b.get("lanes"))
PY
