#!/bin/bash
# round 5, call 1: the exact convert stage (8-bit internal formats) — new tests, the same tests with the fast form (A/B: they must fail), rates
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05c1; mkdir -p $O
K='exact_convert_stage or amplified_convert or hunt_8bit or up1440_nv12 or behind_a_batch or misaligned_device or frame_lanes_equal'
MPCVR_PARITY_LOG=$O/parity.jsonl timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "$K" > $O/tests_exact.txt 2>&1; echo "exact rc=$?" >> $O/tests_exact.txt
MPCVR_EXACT8=0 timeout 600 python -m pytest tests/test_parity_gpu.py -q -k "exact_convert_stage or amplified_convert or hunt_8bit" > $O/tests_fast_form.txt 2>&1; echo "fast rc=$?" >> $O/tests_fast_form.txt
for wl in up1440_nv12 up1080; do
  for e in 0 1; do
    MPCVR_EXACT8=$e timeout 300 python bench.py --workload $wl --steps 30 --warmup 5 2>/dev/null | tail -1 | sed "s/^/{\"exact8\": $e, \"line\": /; s/$/}/" >> $O/bench_exact8.jsonl
  done
done
timeout 300 python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/bench_headline.json
tail -3 $O/tests_exact.txt; tail -3 $O/tests_fast_form.txt; cat $O/bench_exact8.jsonl | cut -c1-400
