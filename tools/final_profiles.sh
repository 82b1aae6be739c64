#!/bin/bash
# tools/final_profiles.sh — everything the committed evidence of a round is made of, in one GPU-box call (run it LAST: the traffic
# figures are tagged with the hash of the kernel sources and bench.py drops them when the sources change afterwards):
#   HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of the five profiled workloads, kernel trace + PMC passes of the headline kernel
#   and of the arbitrary-ratio kernel, one bench line per workload, the general-path table.  Results land in gpurun_out/;
#   tools/update_traffic.py and a copy into profiles/<round>/ follow on the development machine.
cd "$GRAFT_REPO_ROOT"
for w in c3hdr c1 hdr4k up1440 down1440 up2160; do bash tools/pmc_traffic.sh $w > /dev/null 2>&1; done
KFILTER=k_fused_up2x bash tools/prof_headline.sh headline_final > /dev/null 2>&1
KFILTER=k_fused_period bash tools/prof_headline.sh period_up1440_final --workload up1440 > /dev/null 2>&1
KFILTER=k_fused_period bash tools/prof_headline.sh period_down1440_final --workload down1440 > /dev/null 2>&1
KFILTER=k_fused_period bash tools/prof_headline.sh period_up2160_final --workload up2160 > /dev/null 2>&1
# which kernel instantiations the GPU suite launches (tests/test_kernel_coverage.py reads the stats table)
( cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/suite_kt -o suite -- python -m pytest tests -m gpu -x -q > gpurun_out/suite_under_kernel_trace.txt 2>&1 )
for wl in c3hdr c3 c4 c4ext c5 c2 c1 hdr4k up1440 down1440 up1080 down1080 up2160 up1440_nv12 hdrpass_2x hdrpass_1440 c3hdr_1080p; do
  python bench.py --workload $wl --no-host-path --steps 30 --warmup 5 $( [ $wl = c3hdr ] || echo --no-cpu-baseline ) 2>/dev/null | tail -n 1
done > gpurun_out/bench_workloads.jsonl
python tools/bench_general.py 2>/dev/null | grep "^{" > gpurun_out/bench_general.jsonl
for wl in up1440 down1440 down1080 up2160 hdrpass_1440; do python bench.py --workload $wl --flags 128 --no-host-path --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -n 1; done > gpurun_out/bench_workloads_strip_kernel.jsonl
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_shape.json 2> gpurun_out/bench_driver_shape.err
# gpurun merges at most 64 MiB back: keep the tables anyone reads (summaries, stats, traffic, bench lines), drop the raw traces
du -a gpurun_out 2>/dev/null | sort -n | tail -12 > /tmp/du_before.txt
K=/tmp/keep_final; rm -rf $K; mkdir -p $K
cp gpurun_out/*_summary.txt gpurun_out/traffic_*.json gpurun_out/bench_*.json* gpurun_out/suite_under_kernel_trace.txt /tmp/du_before.txt $K/ 2>/dev/null
for d in headline_final period_up1440_final period_down1440_final period_up2160_final; do f=$(find gpurun_out/$d/kt -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $K/${d}_kernel_stats.csv; done
f=$(find gpurun_out/suite_kt -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $K/gpu_suite_kernel_stats.csv
rm -rf gpurun_out/*; cp $K/* gpurun_out/; du -sh gpurun_out; ls gpurun_out
