#!/bin/bash
# error-diffusion pass: workgroup order (frame-major / band-major) against throughput, single frame to 96 frames
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; rm -f $O/ab24.jsonl
python -m pytest tests/test_errdiff.py -m gpu -q -x -k "kernel_equals and dpp" 2>&1 | tail -2
MPCVR_ERRDIFF_ORDER=1 python -m pytest tests/test_errdiff.py -m gpu -q -x 2>&1 | tail -2
for ord in 0 1; do
  for b in 1 8 32 96; do
    MPCVR_ERRDIFF_ORDER=$ord timeout 300 python bench.py --workload c4ed --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-host-path 2>/dev/null | tail -n 1 | sed "s/^{/{\"order\": $ord, \"batch\": $b, /" >> $O/ab24.jsonl
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/ab24.jsonl"):
    r = json.loads(l); print(r["order"], r["batch"], r["value"], r["ms_per_step"])
PY
