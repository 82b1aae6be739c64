#!/bin/bash
# tools/r04_call3.sh — round 4, third GPU-box call: suite under a kernel trace (coverage table for the pruned build), the default
# bench line (four frame lanes + segment sizing for overlapping frames), per-frame figures of the other workloads.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( cd /tmp; cd "$GRAFT_REPO_ROOT"; MPCVR_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_identical_channels.jsonl timeout -k 5 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/suite_kt -o suite -- python -m pytest tests -m gpu -q > $O/suite_under_kernel_trace.txt 2>&1 )
f=$(find /tmp/suite_kt -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/gpu_suite_kernel_stats.csv
grep -E "passed|failed" $O/suite_under_kernel_trace.txt | tail -3
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/bench_default.json") if l.startswith("{")][-1])
print(r["value"], r["roofline"]["frac"], r["roofline"]["kernel_ms_per_launch"], json.dumps(r.get("process_per_frame")))
PY
for wl in c3hdr c1 hdr4k up1440 c5; do python bench.py --workload $wl --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -n 1 | sed "s/^{/{\"ab\": \"per_frame\", /" >> $O/ab3.jsonl; done
python - <<'PY'
import json
for l in open("gpurun_out/ab3.jsonl"):
    if not l.startswith("{"): continue
    r = json.loads(l)
    pf = r.get("process_per_frame") or {}
    print(f"{r['ab']:10s} {r['config']['workload'].split(':')[0]:10s} {r['value']:>10.1f} frames/s  kernel {r['roofline']['kernel_ms_per_launch']:.4f} ms  frac {r['roofline']['frac']:.4f} batch {r['config']['frames_per_step_per_gpu']}"
          + (f"  per-frame: lanes {pf['frames_per_s']} serial {pf['frames_per_s_one_after_the_other']} (process_ms {pf['last_process_ms']} / {pf['last_process_ms_one_after_the_other']})" if pf else ""))
PY
