#!/bin/bash
# tools/r04_call6.sh — suite (batched HDR10 tone-mapping step, finer Dolby Vision EOTF table) + the periodic kernel's new segment rule
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout -k 5 900 python -m pytest tests -m gpu -q > $O/suite.txt 2>&1; grep -E "passed|failed|Error" $O/suite.txt | tail -8
B="--no-cpu-baseline --no-host-path --steps 30 --warmup 5"
ab() { tag=$1; wl=$2; shift 2; env "$@" python bench.py --workload $wl $B 2>/dev/null | tail -n 1 | sed "s/^{/{\"ab\": \"$tag\", /" >> $O/ab6.jsonl; }
for wl in up1440 down1440 up2160 down1080 up1080 up1440_nv12 hdrpass_1440; do ab new $wl A=1; ab r03 $wl MPCVR_LIB=$GRAFT_REPO_ROOT/gpurun_in/libmpcvr_r03.so; done
python - <<'PY'
import json
for l in open("gpurun_out/ab6.jsonl"):
    if not l.startswith("{"): continue
    r = json.loads(l)
    print(f"{r['ab']:10s} {r['config']['workload'].split(':')[0]:12s} {r['value']:>10.1f} frames/s  kernel {r['roofline']['kernel_ms_per_launch']:.4f} ms  frac {r['roofline']['frac']:.4f}")
PY
python tools/bench_general.py "Dolby" 2>/dev/null | grep "^{" > $O/bench_general_dovi.jsonl
python tools/bench_general.py "Jinc" 2>/dev/null | grep "^{" >> $O/bench_general_dovi.jsonl
cat $O/bench_general_dovi.jsonl | cut -c1-260
