#!/bin/bash
# error-diffusion pass, band-major order: does it matter that a band's producer sits on the same XCD (frames per launch a multiple of 8)?
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; rm -f $O/ab25.jsonl
for ord in 1 0; do
  for b in 32 33 36 40 28; do
    MPCVR_ERRDIFF_ORDER=$ord timeout 300 python bench.py --workload c4ed --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-host-path 2>/dev/null | tail -n 1 | sed "s/^{/{\"order\": $ord, \"batch\": $b, /" >> $O/ab25.jsonl
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/ab25.jsonl"):
    r = json.loads(l); print(r["order"], r["batch"], r["value"], r["ms_per_step"])
PY
