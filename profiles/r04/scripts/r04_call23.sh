#!/bin/bash
# same-box A/B: the library at the start of this session (commit b976b1c, gpurun_in/libmpcvr_b976b1c.so) against the final one — the
# error-diffusion pass and the Jinc texcoord tables must not have moved anything else (C1's per-frame figure differed between two boxes)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; rm -f $O/ab23.jsonl
ab() { tag=$1; wl=$2; shift 2; env "$@" python bench.py --workload $wl --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | tail -n 1 | sed "s/^{/{\"ab\": \"$tag\", /" >> $O/ab23.jsonl; }
for rep in 1 2; do
  for wl in c1 c3hdr hdr4k up1440; do
    ab b976b1c $wl MPCVR_LIB=$GRAFT_REPO_ROOT/gpurun_in/libmpcvr_b976b1c.so
    ab final $wl X=1
  done
done
python - <<'PY'
import json, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/ab23.jsonl"):
    if l.startswith("{"):
        r = json.loads(l); pf = r.get("process_per_frame") or {}
        d[(r["config"]["workload"].split(":")[0], r["ab"])].append((round(r["value"]), round(pf.get("frames_per_s", 0)), round(pf.get("frames_per_s_one_after_the_other", 0))))
for (w, t), v in sorted(d.items()): print(f"{w:8s} {t:8s} ", v)
PY
