#!/bin/bash
# suite under a kernel trace with the kernel-family sweep in it (coverage table), parity log
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( cd /tmp; cd "$GRAFT_REPO_ROOT"; MPCVR_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_identical_channels.jsonl timeout -k 5 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/suite_kt -o suite -- python -m pytest tests -m gpu -q > $O/suite_under_kernel_trace.txt 2>&1 )
f=$(find /tmp/suite_kt -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/gpu_suite_kernel_stats.csv
grep -E "^FAILED|passed|failed" $O/suite_under_kernel_trace.txt | grep -v rocprofv3 | tail -40
