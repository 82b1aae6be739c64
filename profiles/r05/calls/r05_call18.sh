#!/bin/bash
# round 5, call 18: the exact-form twins as kernels of their own — suite under a kernel trace (which instantiations still lack a test) + the
# 8-bit-internal-format workloads
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
rm -f /tmp/test_times.jsonl
( cd /tmp; cd "$GRAFT_REPO_ROOT"; MPCVR_TEST_TIMES=/tmp/test_times.jsonl timeout -k 5 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/suite_kt -o suite -- python -m pytest tests -m gpu -q > $O/suite_under_kernel_trace.txt 2>&1 )
f=$(find /tmp/suite_kt -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/gpu_suite_kernel_stats.csv
t=$(find /tmp/suite_kt -name "*kernel_trace.csv" 2>/dev/null | head -1); [ -n "$t" ] && python tests/tools/kernel_witnesses.py "$t" /tmp/test_times.jsonl $O/kernels_by_test.json
grep -E "passed|failed|^FAILED|^ERROR" $O/suite_under_kernel_trace.txt | grep -v rocprofv3 | tail -8
for wl in up1440_nv12 up1080 c1 c3hdr; do
  python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', round(d['value']))"
done | tee $O/call18_bench.txt
