#!/bin/bash
# round 5, call 19+: the fused Jinc2m kernel — the suite under a kernel trace (which instantiations lack a test), the jinc workload
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
rm -f /tmp/test_times.jsonl
( cd /tmp; cd "$GRAFT_REPO_ROOT"; MPCVR_TEST_TIMES=/tmp/test_times.jsonl timeout -k 5 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/suite_kt -o suite -- python -m pytest tests -m gpu -q > $O/suite_under_kernel_trace.txt 2>&1 )
f=$(find /tmp/suite_kt -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/gpu_suite_kernel_stats.csv
t=$(find /tmp/suite_kt -name "*kernel_trace.csv" 2>/dev/null | head -1); [ -n "$t" ] && python tests/tools/kernel_witnesses.py "$t" /tmp/test_times.jsonl $O/kernels_by_test.json
grep -E "passed|failed|^FAILED|^ERROR" $O/suite_under_kernel_trace.txt | grep -v rocprofv3 | cut -c1-250 | tail -12
for wl in jinc1080 c3hdr; do python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', round(d['value']), d['roofline']['frac'], d['config']['path'])"; done | tee $O/call19_bench.txt
python tools/bench_general.py 2>/dev/null | grep "^{" | grep -i jinc | cut -c1-400
