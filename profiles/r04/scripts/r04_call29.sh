#!/bin/bash
# per-frame path: the timing event pair on every frame against every 8th / 64th frame of the lanes
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; rm -f $O/ab29.jsonl
for rep in 1 2; do
  for ev in 1 8 64; do
    for wl in c3hdr c1; do
      MPCVR_LANE_TIMING_EVERY=$ev python bench.py --workload $wl --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -n 1 | sed "s/^{/{\"every\": $ev, /" >> $O/ab29.jsonl
    done
  done
done
python - <<'PY'
import json, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/ab29.jsonl"):
    r = json.loads(l); pf = r.get("process_per_frame") or {}
    d[(r["config"]["workload"].split(":")[0], r["every"])].append((round(r["value"]), round(pf.get("frames_per_s", 0)), round(pf.get("frames_per_s_one_after_the_other", 0))))
for k, v in sorted(d.items()): print(k, v)
PY
