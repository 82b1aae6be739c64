#!/bin/bash
# XCD-contiguous workgroup mapping in k_fused_period / k_fused_strip: same-box A/B against the library built before the change
# (gpurun_in/libmpcvr_pre_xcd.so), HBM traffic of the periodic workloads with the new mapping, parity of both kernels
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; rm -f $O/ab13.jsonl
ab() { tag=$1; wl=$2; shift 2; env "$@" python bench.py --workload $wl --no-cpu-baseline --no-host-path --steps 30 --warmup 8 2>/dev/null | tail -n 1 | sed "s/^{/{\"ab\": \"$tag\", /" >> $O/ab13.jsonl; }
for rep in 1 2; do
  for wl in up1440 down1440 up2160 up1080 down1080 up1440_nv12 hdrpass_1440 c4ext; do
    ab pre_xcd $wl MPCVR_LIB=$GRAFT_REPO_ROOT/gpurun_in/libmpcvr_pre_xcd.so
    ab xcd $wl X=1
  done
done
python - <<'PY'
import json, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/ab13.jsonl"):
    if l.startswith("{"):
        r = json.loads(l); d[(r["config"]["workload"].split(":")[0], r["ab"])].append(r["value"])
for (w, t), v in sorted(d.items()): print(f"{w:14s} {t:8s} " + " ".join(f"{x:9.0f}" for x in v))
PY
for w in up1440 down1440; do bash tools/pmc_traffic.sh $w 2>/dev/null | tail -1 | cut -c1-260; done
timeout 900 python -m pytest tests -m gpu -q -x -k "period or strip or general_ratio or sweep or batch" 2>&1 | tail -3
