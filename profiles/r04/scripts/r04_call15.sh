#!/bin/bash
# error-diffusion pass, second version: one free-running wavefront per band (workgroups = frames x bands), tagged hand-off words in
# device memory instead of workgroup barriers + LDS rows: its tests, the config-4 workload at 32 and 128 frames per step, kernel time
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_errdiff.py -m gpu -q -x 2>&1 | tail -15 > $O/errdiff_tests.txt; cat $O/errdiff_tests.txt
for b in 32 8 1; do
  timeout 300 python bench.py --workload c4ed --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-host-path 2>/dev/null | tail -n 1 | sed "s/^{/{\"batch\": $b, /"
done > $O/bench_c4ed.jsonl
MPCVR_ERRDIFF_SHIFT=bpermute timeout 300 python bench.py --workload c4ed --steps 10 --warmup 3 --no-cpu-baseline --no-host-path 2>/dev/null | tail -n 1 | sed 's/^{/{"ab": "bpermute", /' >> $O/bench_c4ed.jsonl
cut -c1-330 $O/bench_c4ed.jsonl
( cd /tmp; cd "$GRAFT_REPO_ROOT"; timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4ed_kt -o c4ed -- python bench.py --workload c4ed --steps 6 --warmup 2 --no-cpu-baseline --no-host-path > /dev/null 2>&1 )
f=$(find /tmp/c4ed_kt -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/errdiff_c4ed_kernel_stats.csv && head -4 $O/errdiff_c4ed_kernel_stats.csv | cut -c1-220
