cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
rm -f /tmp/test_times.jsonl $O/parity_log_final.jsonl
( MPCVR_TEST_TIMES=/tmp/test_times.jsonl MPCVR_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_log_final.jsonl timeout -k 5 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/suite_kt -o suite -- python -m pytest tests -m gpu -q > $O/suite_under_kernel_trace.txt 2>&1 )
f=$(find /tmp/suite_kt -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/gpu_suite_kernel_stats.csv
t=$(find /tmp/suite_kt -name "*kernel_trace.csv" 2>/dev/null | head -1); [ -n "$t" ] && python tests/tools/kernel_witnesses.py "$t" /tmp/test_times.jsonl $O/kernels_by_test.json
grep -E "passed|failed" $O/suite_under_kernel_trace.txt | grep -v rocprofv3 | tail -2
( time timeout 900 python tests/tools/fuzz_strip.py 8000 ) > $O/fuzz_8000.txt 2>&1; echo "rc=$?" >> $O/fuzz_8000.txt
for seed in 123 999 20261001 4242; do ( time timeout 900 python tests/tools/fuzz_strip.py 8000 $seed ) > $O/fuzz_8000_seed$seed.txt 2>&1; echo "rc=$?" >> $O/fuzz_8000_seed$seed.txt; done
grep -H "^rc=" $O/fuzz_8000*.txt
