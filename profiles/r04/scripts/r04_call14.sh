#!/bin/bash
# the error-diffusion pass (bUseDither = 2, vp_errdiff.hip) and the plain Jinc2m kernel's texcoord tables on hardware: the whole GPU suite
# under a kernel trace (coverage table, witnesses, parity log: the round's suite evidence if nothing changes afterwards), the config-4
# workload with the pass (c4ed) beside its ordered-dither twin (c4ext), the pass's own kernel time, and the headline's traffic /
# issue-slot pass on the current sources
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
# the new tests first, on their own: their verdict must not hang on the rest of the suite
timeout -k 5 600 python -m pytest tests/test_errdiff.py -m gpu -q -x 2>&1 | tail -15 > $O/errdiff_tests.txt; cat $O/errdiff_tests.txt
rm -f /tmp/test_times.jsonl
( cd /tmp; cd "$GRAFT_REPO_ROOT"; MPCVR_TEST_TIMES=/tmp/test_times.jsonl MPCVR_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_identical_channels.jsonl timeout -k 5 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/suite_kt -o suite -- python -m pytest tests -m gpu -q > $O/suite_under_kernel_trace.txt 2>&1 )
f=$(find /tmp/suite_kt -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/gpu_suite_kernel_stats.csv
t=$(find /tmp/suite_kt -name "*kernel_trace.csv" 2>/dev/null | head -1)
[ -n "$t" ] && python tests/tools/kernel_witnesses.py "$t" /tmp/test_times.jsonl $O/kernels_by_test.json
grep -E "^FAILED|^ERROR|passed|failed" $O/suite_under_kernel_trace.txt | grep -v rocprofv3 | tail -20
grep -i "error_diffusion" $O/gpu_suite_kernel_stats.csv | cut -c1-200
for wl in c4ext c4ed; do
  python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-host-path 2>/dev/null | tail -n 1
done > $O/bench_c4ed.jsonl
MPCVR_ERRDIFF_SHIFT=bpermute python bench.py --workload c4ed --steps 10 --warmup 3 --no-cpu-baseline --no-host-path 2>/dev/null | tail -n 1 | sed 's/^{/{"ab": "bpermute", /' >> $O/bench_c4ed.jsonl
cut -c1-420 $O/bench_c4ed.jsonl
( cd /tmp; cd "$GRAFT_REPO_ROOT"; timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4ed_kt -o c4ed -- python bench.py --workload c4ed --steps 6 --warmup 2 --no-cpu-baseline --no-host-path > /dev/null 2>&1 )
f=$(find /tmp/c4ed_kt -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/errdiff_c4ed_kernel_stats.csv && head -4 $O/errdiff_c4ed_kernel_stats.csv | cut -c1-220
bash tools/pmc_traffic.sh c3hdr 2>/dev/null | tail -1 | cut -c1-400
rm -rf $O/traffic_c3hdr
du -sh $O; ls $O
