O=gpurun_out/r06; mkdir -p $O
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/gpu_suite_last.txt 2>&1; tail -4 $O/gpu_suite_last.txt | head -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > $O/bench_driver_shape_last.json 2> $O/bench_driver_shape_last.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06/bench_driver_shape_last.json") if l.startswith("{")][-1])
r = d["roofline"]; print(d["value"], r["frac"], "traffic", r["traffic"], "co_limit", (r.get("co_limit") or {}).get("frac"), "| lanes", d["process_batch_on_lanes"]["frames_per_s"], "| cpu", d["cpu_baseline"]["value"])
PY
