#!/bin/bash
# tools/ablate_mx.sh — ablation of the matrix-core fused kernel on the headline workload (GPU box): MPCVR_MX_DBG bits switch stages off
#   1 no X stage   2 no Y MFMAs   4 no epilogue and no stores   8 no convert arithmetic   16 epilogue computed, stores skipped   32 stores kept, epilogue arithmetic skipped
cd "$GRAFT_REPO_ROOT"
for d in ${@:-0 16 32 4 8 3}; do
  r=$(MPCVR_MX_DBG=$d python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-host-path --flags 32 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['roofline']['kernel_ms_per_launch'])")
  echo "MX dbg=$d  fps,ms/launch: $r"
done
python - <<'PY'
import torch
n = 1 << 32
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty(n, dtype=torch.uint8, device="cuda")
def t(f, reps=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
print(f"write-only (fill 4 GiB)   {n / t(lambda: a.zero_()) / 1e9:8.1f} GB/s")
print(f"read-only  (sum 4 GiB i32){n / t(lambda: a.view(torch.int32).sum()) / 1e9:8.1f} GB/s")
print(f"copy 4 GiB (read + write) {2 * n / t(lambda: b.copy_(a)) / 1e9:8.1f} GB/s")
PY
