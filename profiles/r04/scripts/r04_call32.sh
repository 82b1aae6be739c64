#!/bin/bash
# error-diffusion pass at other sizes: 1080p -> 4K, 4K same size (no resize), 1080p same size
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; rm -f $O/bench_c4ed_sizes.jsonl
run() { tag=$1; shift; timeout 200 python bench.py --workload c4ed --steps 8 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | tail -n 1 | sed "s/^{/{\"case\": \"$tag\", /" >> $O/bench_c4ed_sizes.jsonl; }
run 1080p_to_4k --src 1920x1080
run 4k_same_size --scale 1
run 1080p_same_size --src 1920x1080 --scale 1 --batch 128
python - <<'PY'
import json
for l in open("gpurun_out/bench_c4ed_sizes.jsonl"):
    r = json.loads(l); pf = r.get("process_per_frame") or {}
    print(r["case"], r["value"], r["ms_per_step"], r["config"].get("path"), pf.get("frames_per_s"))
PY
