#!/bin/bash
# tools/r04_call4.sh — round 4, fourth GPU-box call: frame-lane count sweep, register-cap variants of the streaming convert, which
# step of the Dolby Vision block convert moves the ill-conditioned channels.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
pf() { tag=$1; wl=$2; shift 2; env "$@" python bench.py --workload $wl --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -n 1 | sed "s/^{/{\"ab\": \"$tag\", /" >> $O/ab4.jsonl; }
for n in 1 2 3 4 6 8; do pf lanes$n c3hdr MPCVR_FRAME_LANES=$n; done
for n in 4 8; do pf lanes$n hdr4k MPCVR_FRAME_LANES=$n; pf lanes$n up1440 MPCVR_FRAME_LANES=$n; done
for v in w5 w6; do pf $v hdr4k MPCVR_LIB=$GRAFT_REPO_ROOT/gpurun_in/libmpcvr_$v.so; pf $v c1 MPCVR_LIB=$GRAFT_REPO_ROOT/gpurun_in/libmpcvr_$v.so; done
pf base hdr4k A=1; pf base c1 A=1
python - <<'PY'
import json
for l in open("gpurun_out/ab4.jsonl"):
    if not l.startswith("{"): continue
    r = json.loads(l)
    pf = r.get("process_per_frame") or {}
    print(f"{r['ab']:10s} {r['config']['workload'].split(':')[0]:10s} {r['value']:>10.1f} frames/s  kernel {r['roofline']['kernel_ms_per_launch']:.4f} ms  frac {r['roofline']['frac']:.4f}"
          + (f"  per-frame: lanes {pf['frames_per_s']} serial {pf['frames_per_s_one_after_the_other']} (process_ms {pf['last_process_ms']} / {pf['last_process_ms_one_after_the_other']})" if pf else ""))
PY
for v in "" dv1 dv3; do
  if [ -z "$v" ]; then python tests/tools/diag_dovi_tiers.py; else MPCVR_LIB=$GRAFT_REPO_ROOT/gpurun_in/libmpcvr_$v.so python tests/tools/diag_dovi_tiers.py; fi
done 2>/dev/null | grep "^{" > $O/dovi_tiers.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/dovi_tiers.jsonl"):
    r = json.loads(l); print(r["lib"], r["case"], r["flags"], r["beyond_1lsb"], r["max"], round(r["identical"], 5), r["path"][:40])
PY
