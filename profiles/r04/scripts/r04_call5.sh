#!/bin/bash
# tools/r04_call5.sh — knob sweeps of the periodic kernel (waves per workgroup, segment height), Dolby Vision tiers with the finer EOTF table
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
B="--no-cpu-baseline --no-host-path --steps 30 --warmup 5"
ab() { tag=$1; wl=$2; shift 2; env "$@" python bench.py --workload $wl $B 2>/dev/null | tail -n 1 | sed "s/^{/{\"ab\": \"$tag\", /" >> $O/ab5.jsonl; }
for wl in up1440 down1440; do
  ab base $wl A=1
  for w in 4 6 12 16; do ab waves$w $wl MPCVR_PERIOD_WAVES=$w; done
  for sg in 24 48 96 192; do ab seg$sg $wl MPCVR_PERIOD_SEG=$sg; done
done
ab base hdr4k A=1; ab base c1 A=1; ab base c3hdr A=1
python - <<'PY'
import json
for l in open("gpurun_out/ab5.jsonl"):
    if not l.startswith("{"): continue
    r = json.loads(l)
    print(f"{r['ab']:10s} {r['config']['workload'].split(':')[0]:10s} {r['value']:>10.1f} frames/s  kernel {r['roofline']['kernel_ms_per_launch']:.4f} ms  frac {r['roofline']['frac']:.4f}")
PY
python tests/tools/diag_dovi_tiers.py 2>/dev/null | grep "^{" > $O/dovi_tiers2.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/dovi_tiers2.jsonl"):
    r = json.loads(l); print(r["lib"], r["case"], r["flags"], r["beyond_1lsb"], r["max"], round(r["identical"], 5), r["path"][:40])
PY
timeout 600 python -m pytest tests -m gpu -q -x -k "dovi or Dovi or dolby" 2>&1 | tail -3
