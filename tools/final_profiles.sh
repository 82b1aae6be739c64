#!/bin/bash
# tools/final_profiles.sh — everything the committed evidence of a round is made of, in one GPU-box call (run it LAST: the traffic and
# issue-slot figures are tagged with the hash of the kernel sources and bench.py drops them when the sources change afterwards):
#   HBM traffic + VALU issue-slot occupancy (FETCH_SIZE / WRITE_SIZE / SQ passes) of the profiled workloads, kernel trace + PMC passes of
#   the headline kernel, the streaming convert and the periodic kernel, the GPU suite under a kernel trace (coverage table + parity
#   log), one bench line per workload, the general-path table, the default and the driver-shaped bench line.
# Results land in gpurun_out/; tools/update_traffic.py and a copy into profiles/<round>/ follow on the development machine
# (tools/collect_profiles.py <round> does both and regenerates the README rows from the files).
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for w in c3hdr c1 hdr4k up1440 down1440 up2160 c5 c4ed jinc1080 dovi4k; do bash tools/pmc_traffic.sh $w > /dev/null 2>&1; done
KFILTER=k_fused_up2x bash tools/prof_headline.sh headline_final > /dev/null 2>&1
KFILTER=k_fused_period bash tools/prof_headline.sh period_up1440_final --workload up1440 > /dev/null 2>&1
KFILTER=k_error_diffusion bash tools/prof_headline.sh errdiff_c4ed_final --workload c4ed > /dev/null 2>&1
KFILTER=k_fused_jinc2x bash tools/prof_headline.sh jinc1080_final --workload jinc1080 > /dev/null 2>&1
KFILTER=k_convert_blocks bash tools/prof_headline.sh dovi4k_final --workload dovi4k > /dev/null 2>&1
# which kernel instantiations the GPU suite launches (tests/test_kernel_coverage.py reads the stats table) + the parity log
rm -f /tmp/test_times.jsonl
( cd /tmp; cd "$GRAFT_REPO_ROOT"; MPCVR_TEST_TIMES=/tmp/test_times.jsonl MPCVR_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_identical_channels.jsonl timeout -k 5 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/suite_kt -o suite -- python -m pytest tests -m gpu -q > $O/suite_under_kernel_trace.txt 2>&1 )
f=$(find /tmp/suite_kt -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/gpu_suite_kernel_stats.csv
t=$(find /tmp/suite_kt -name "*kernel_trace.csv" 2>/dev/null | head -1); [ -n "$t" ] && python tests/tools/kernel_witnesses.py "$t" /tmp/test_times.jsonl $O/kernels_by_test.json   # which test launched which instantiation
grep -E "passed|failed" $O/suite_under_kernel_trace.txt | grep -v rocprofv3 | tail -2
for wl in c3hdr c3 c4 c4ext c4ed c5 c2 c1 hdr4k up1440 down1440 up1080 down1080 up2160 up1440_nv12 hdrpass_2x hdrpass_1440 c3hdr_1080p jinc1080 dovi4k; do
  python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1
done > $O/bench_workloads.jsonl
python tools/bench_general.py 2>/dev/null | grep "^{" > $O/bench_general.jsonl
python bench.py > $O/bench_default.json 2> $O/bench_default.err
# the fuzz evidence on the final code: random fused-path cases against the oracle (the tool FAILS on a channel beyond the bar where no
# transcendental decides the last code), the error-diffusion pass on random shapes against its serial model
( time timeout 900 python tests/tools/fuzz_strip.py 8000 ) > $O/fuzz_8000.txt 2>&1; echo "rc=$?" >> $O/fuzz_8000.txt
( time timeout 600 python tests/tools/fuzz_strip.py 3000 777 ) > $O/fuzz_3000_seed777.txt 2>&1; echo "rc=$?" >> $O/fuzz_3000_seed777.txt
( time MPCVR_FUZZ_JINC=1 timeout 600 python tests/tools/fuzz_strip.py 2000 5 ) > $O/fuzz_2000_jinc.txt 2>&1; echo "rc=$?" >> $O/fuzz_2000_jinc.txt
# the four combined modes of round 5's last call, same seeds (402 is the run that ended rc = 1 there: case 1820), on the final code: the plain
# tier is held to the oracle bit for bit in every oracle-compared case
run() { n=$1; shift; ( time env "$@" timeout 900 python tests/tools/fuzz_strip.py 2500 $SEED ) > $O/$n.txt 2>&1; echo "rc=$?" >> $O/$n.txt; }
SEED=401 run fuzz_2500_jinc_flags8 MPCVR_FUZZ_JINC=1 MPCVR_FUZZ_FLAGS=8
SEED=402 run fuzz_2500_jinc_flags64 MPCVR_FUZZ_JINC=1 MPCVR_FUZZ_FLAGS=64
SEED=403 run fuzz_2500_scalers_unaligned_flags72 MPCVR_FUZZ_SCALERS=1 MPCVR_FUZZ_UNALIGNED=1 MPCVR_FUZZ_FLAGS=72
SEED=404 run fuzz_2500_host_unaligned_flags4 MPCVR_FUZZ_HOST=1 MPCVR_FUZZ_UNALIGNED=1 MPCVR_FUZZ_FLAGS=4
SEED=31 run fuzz_2500_periodic MPCVR_FUZZ_PERIODIC=1
# the soak run that ended on the count cap (case 7172: 24 witnessed channels on a rotated Dolby Vision frame), with the cap at 1.5 x that rate
( time timeout 900 python tests/tools/fuzz_strip.py 8000 1017 ) > $O/fuzz_8000_seed1017.txt 2>&1; echo "rc=$?" >> $O/fuzz_8000_seed1017.txt
( timeout 400 python tests/tools/fuzz_errdiff.py 250 1 2>&1 | tail -8; timeout 400 python tests/tools/fuzz_errdiff.py 250 11 2>&1 | tail -8 ) > $O/fuzz_errdiff.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err
# gpurun merges at most 64 MiB back: keep the tables anyone reads (summaries, stats, traffic, bench lines), drop the raw traces
K=/tmp/keep_final; rm -rf $K; mkdir -p $K
cp $O/*_summary.txt $O/traffic_*.json $O/bench_*.json* $O/fuzz_*.txt $O/suite_under_kernel_trace.txt $O/gpu_suite_kernel_stats.csv $O/kernels_by_test.json $O/parity_identical_channels.jsonl $K/ 2>/dev/null
for d in headline_final stream_c1_final stream_hdr4k_final period_up1440_final period_down1440_final errdiff_c4ed_final jinc1080_final dovi4k_final; do f=$(find $O/$d/kt -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $K/${d}_kernel_stats.csv; done
rm -rf $O/*; cp $K/* $O/; du -sh $O; ls $O
