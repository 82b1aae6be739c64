#!/bin/bash
# tools/final_profiles.sh — everything the committed evidence of a round is made of, in one GPU-box call (run it LAST: the traffic
# figures are tagged with the hash of the kernel sources and bench.py drops them when the sources change afterwards):
#   HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of the five profiled workloads, kernel trace + PMC passes of the headline kernel
#   and of the arbitrary-ratio kernel, one bench line per workload, the general-path table.  Results land in gpurun_out/;
#   tools/update_traffic.py and a copy into profiles/<round>/ follow on the development machine.
cd "$GRAFT_REPO_ROOT"
for w in c3hdr c1 hdr4k up1440 down1440; do bash tools/pmc_traffic.sh $w > /dev/null 2>&1; done
KFILTER=k_fused_up2x bash tools/prof_headline.sh headline_final > /dev/null 2>&1
KFILTER=k_fused_strip bash tools/prof_headline.sh strip_up1440_final --workload up1440 > /dev/null 2>&1
KFILTER=k_fused_strip bash tools/prof_headline.sh strip_down1440_final --workload down1440 > /dev/null 2>&1
for wl in c3hdr c3 c4 c4ext c5 c2 c1 hdr4k up1440 down1440 up1440_nv12 hdrpass_2x hdrpass_1440 c3hdr_1080p; do
  python bench.py --workload $wl --no-host-path --steps 30 --warmup 5 $( [ $wl = c3hdr ] || echo --no-cpu-baseline ) 2>/dev/null | tail -n 1
done > gpurun_out/bench_workloads.jsonl
python tools/bench_general.py 2>/dev/null | grep "^{" > gpurun_out/bench_general.jsonl
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
ls gpurun_out | head -40
