#!/bin/bash
# suite under a kernel trace after the pruning of (8-bit loader x tail) / one-pixel strips and with the convert / column / Jinc sweep
# blocks: coverage table, parity log, and which test launched which instantiation (tests/tools/kernel_witnesses.py)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
rm -f /tmp/test_times.jsonl
( cd /tmp; cd "$GRAFT_REPO_ROOT"; MPCVR_TEST_TIMES=/tmp/test_times.jsonl MPCVR_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_identical_channels.jsonl timeout -k 5 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/suite_kt -o suite -- python -m pytest tests -m gpu -q --durations=30 > $O/suite_under_kernel_trace.txt 2>&1 )
f=$(find /tmp/suite_kt -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/gpu_suite_kernel_stats.csv
t=$(find /tmp/suite_kt -name "*kernel_trace.csv" 2>/dev/null | head -1)
[ -n "$t" ] && python tests/tools/kernel_witnesses.py "$t" /tmp/test_times.jsonl $O/kernels_by_test.json $O/kernels_of_each_test.json
grep -E "^FAILED|passed|failed" $O/suite_under_kernel_trace.txt | grep -v rocprofv3 | tail -40; grep -A34 "slowest 30" $O/suite_under_kernel_trace.txt | head -36; nproc; python -c "import sys; sys.path.insert(0, \".\"); from oracle import oracle as O; print(O.host_cpus())"
