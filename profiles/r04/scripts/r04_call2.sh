#!/bin/bash
# tools/r04_call2.sh — round 4, second GPU-box call: the suite again (frame lanes are now the default of every context that owns its
# stream, the RCCL example's test parses RCCL's banner), the default bench line with its per-frame figures, c1 at its new batch,
# the 3:1 ratio back on pair stores.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
B="--no-cpu-baseline --no-host-path --steps 30 --warmup 5"
timeout -k 5 900 python -m pytest tests -m gpu -q -x > $O/suite.txt 2>&1; tail -4 $O/suite.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 2500 $O/bench_default.json; tail -3 $O/bench_default.err
ab() { tag=$1; wl=$2; shift 2; env "$@" python bench.py --workload $wl $B 2>/dev/null | tail -n 1 | sed "s/^{/{\"ab\": \"$tag\", /" >> $O/ab2.jsonl; }
R03=MPCVR_LIB=$GRAFT_REPO_ROOT/gpurun_in/libmpcvr_r03.so
for wl in c1 up2160 hdr4k; do ab new $wl A=1; ab r03 $wl $R03; done
for wl in c3hdr c1 hdr4k up1440; do python bench.py --workload $wl --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -n 1 | sed "s/^{/{\"ab\": \"per_frame\", /" >> $O/ab2.jsonl; done
python - <<'PY'
import json
for l in open("gpurun_out/ab2.jsonl"):
    if not l.startswith("{"): continue
    r = json.loads(l)
    pf = r.get("process_per_frame") or {}
    print(f"{r['ab']:10s} {r['config']['workload'].split(':')[0]:10s} {r['value']:>10.1f} frames/s  kernel {r['roofline']['kernel_ms_per_launch']:.4f} ms  frac {r['roofline']['frac']:.4f} batch {r['config']['frames_per_step_per_gpu']}"
          + (f"  per-frame: lanes {pf['frames_per_s']} serial {pf['frames_per_s_one_after_the_other']} (process_ms {pf['last_process_ms']} / {pf['last_process_ms_one_after_the_other']})" if pf else ""))
PY
