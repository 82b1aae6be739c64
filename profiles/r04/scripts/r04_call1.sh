#!/bin/bash
# tools/r04_call1.sh — round 4, first GPU-box call: the GPU suite under a kernel trace (coverage table + parity log), same-box A/B
# benches (this build against the round-3 library kept as gpurun_in/libmpcvr_r03.so; the new kernels against their own knobs),
# issue-slot / traffic counters of the headline, counter digests of the streaming convert and the periodic kernel.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
B="--no-cpu-baseline --no-host-path --steps 30 --warmup 5"
# 1. suite (no -x: every failure is wanted), parity log on
( cd /tmp; cd "$GRAFT_REPO_ROOT"; MPCVR_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_identical_channels.jsonl timeout -k 5 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/suite_kt -o suite -- python -m pytest tests -m gpu -q > $O/suite_under_kernel_trace.txt 2>&1 )
f=$(find /tmp/suite_kt -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/gpu_suite_kernel_stats.csv
tail -5 $O/suite_under_kernel_trace.txt
# 2. A/B benches, interleaved so that box drift hits both sides alike
ab() { tag=$1; wl=$2; shift 2; env "$@" python bench.py --workload $wl $B 2>/dev/null | tail -n 1 | sed "s/^{/{\"ab\": \"$tag\", /" >> $O/ab.jsonl; }
R03=MPCVR_LIB=$GRAFT_REPO_ROOT/gpurun_in/libmpcvr_r03.so
for rep in 1 2; do ab new c3hdr A=1; [ -f gpurun_in/libmpcvr_r03.so ] && ab r03 c3hdr $R03; done
for wl in c5 c1 hdr4k up1440 down1440 up2160; do
  ab new $wl A=1
  [ -f gpurun_in/libmpcvr_r03.so ] && ab r03 $wl $R03
done
for wl in c1 hdr4k; do ab new_wide $wl MPCVR_NO_STREAM_CONVERT=1; done
for wl in up1440 down1440 up2160; do ab new_own0 $wl MPCVR_PERIOD_OWN=0; done
# c1 with longer launches (the stream kernel's runs grow with the batch) and its knobs
for bt in 64 128; do python bench.py --workload c1 --batch $bt --ring $((bt + 32)) $B 2>/dev/null | tail -n 1 | sed "s/^{/{\"ab\": \"new_batch$bt\", /" >> $O/ab.jsonl; done
for k in 2 8 16; do MPCVR_STREAM_MIN_PAIRS=$k python bench.py --workload c1 $B 2>/dev/null | tail -n 1 | sed "s/^{/{\"ab\": \"new_minpairs$k\", /" >> $O/ab.jsonl; done
for k in 4 8; do MPCVR_STREAM_WG_WAVES=$k python bench.py --workload hdr4k $B 2>/dev/null | tail -n 1 | sed "s/^{/{\"ab\": \"new_wg$k\", /" >> $O/ab.jsonl; done
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/ab.jsonl") if l.startswith("{")]
for r in rows:
    print(f"{r['ab']:14s} {r['config']['workload'].split(':')[0]:10s} {r['value']:>10.1f} frames/s  kernel {r['roofline']['kernel_ms_per_launch']:.4f} ms  frac {r['roofline']['frac']:.4f}  {r['config']['path'][:60]}")
PY
# 3. counters: headline traffic + issue slots; digests of the new kernels
bash tools/pmc_traffic.sh c3hdr
bash tools/pmc_traffic.sh c1
KFILTER=k_convert_stream bash tools/prof_headline.sh stream_c1 --workload c1 > /dev/null 2>&1
KFILTER=k_fused_period bash tools/prof_headline.sh period_up1440 --workload up1440 > /dev/null 2>&1
cat $O/stream_c1_summary.txt $O/period_up1440_summary.txt 2>/dev/null | grep -v "^  void at::\|__amd_rocclr" | head -90
# keep the tables, drop the raw traces (gpurun merges at most 64 MiB back)
K=/tmp/keep1; rm -rf $K; mkdir -p $K
for d in stream_c1 stream_hdr4k period_up1440 period_down1440; do f=$(find $O/$d/kt -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $K/${d}_kernel_stats.csv; done
cp $O/*_summary.txt $O/traffic_*.json $O/ab.jsonl $O/suite_under_kernel_trace.txt $O/gpu_suite_kernel_stats.csv $O/parity_identical_channels.jsonl $K/ 2>/dev/null
rm -rf $O/*; cp $K/* $O/; du -sh $O
