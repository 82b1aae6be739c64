#!/bin/bash
# per-frame path: segment height x lane count
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
pf() { tag=$1; wl=$2; shift 2; env "$@" python bench.py --workload $wl --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -n 1 | sed "s/^{/{\"ab\": \"$tag\", /" >> $O/ab9.jsonl; }
for sg in 60 72 90 108 120 144 180; do for ln in 3 4 5; do pf seg${sg}_lanes$ln c3hdr MPCVR_FUSED_SEG=$sg MPCVR_FRAME_LANES=$ln; done; done
python - <<'PY'
import json
for l in open("gpurun_out/ab9.jsonl"):
    if not l.startswith("{"): continue
    r = json.loads(l); pf = r.get("process_per_frame") or {}
    print(f"{r['ab']:16s} batch {r['value']:>9.1f}  per-frame: lanes {pf.get('frames_per_s')} serial {pf.get('frames_per_s_one_after_the_other')} (process_ms {pf.get('last_process_ms')})")
PY
timeout 600 python -m pytest tests -m gpu -q -x -k "sweep" 2>&1 | tail -3
