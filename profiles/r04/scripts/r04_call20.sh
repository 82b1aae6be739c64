#!/bin/bash
# error-diffusion pass: resident wavefronts per SIMD (LDS claimed per workgroup as the occupancy knob) against throughput
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; rm -f $O/ab20.jsonl
for lds in 0 13000 20000 40000 64000; do
  for b in 32 96; do
    MPCVR_ERRDIFF_LDS=$lds timeout 300 python bench.py --workload c4ed --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-host-path 2>/dev/null | tail -n 1 | sed "s/^{/{\"lds\": $lds, \"batch\": $b, /" >> $O/ab20.jsonl
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/ab20.jsonl"):
    r = json.loads(l); print(r["lds"], r["batch"], r["value"], r["ms_per_step"])
PY
