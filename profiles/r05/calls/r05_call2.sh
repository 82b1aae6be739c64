#!/bin/bash
# round 5, call 2: error diffusion v3 (one channel per lane, tickets, tiles in LDS) — tests, rates, A/B of resident wavefronts
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05c2; mkdir -p $O
timeout 900 python -m pytest tests/test_errdiff.py -x -q -m gpu > $O/tests_errdiff.txt 2>&1; echo "rc=$?" >> $O/tests_errdiff.txt
tail -5 $O/tests_errdiff.txt
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "exact_convert_stage or amplified_convert or up1440_nv12 or behind_a_batch" > $O/tests_exact.txt 2>&1; tail -2 $O/tests_exact.txt
timeout 300 python bench.py --workload c4ed --steps 10 --warmup 3 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 > $O/bench_c4ed.json; cat $O/bench_c4ed.json | cut -c1-300
for w in 2048 4096 6144 8192; do MPCVR_ERRDIFF_WAVES=$w timeout 300 python bench.py --workload c4ed --steps 10 --warmup 3 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'waves': $w, 'frames_s': d['value'], 'ms_per_step': d['ms_per_step']}))" >> $O/ab_waves.jsonl; done
MPCVR_ERRDIFF_ORDER=0 timeout 300 python bench.py --workload c4ed --steps 10 --warmup 3 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'order': 0, 'frames_s': d['value'], 'ms_per_step': d['ms_per_step']}))" >> $O/ab_waves.jsonl
timeout 300 python bench.py --workload c4ed --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'batch': 1, 'frames_s': d['value'], 'ms_per_step': d['ms_per_step']}))" >> $O/ab_waves.jsonl
cat $O/ab_waves.jsonl
for wl in up1440_nv12 up1080; do for e in 0 1; do MPCVR_EXACT8=$e timeout 300 python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'workload': '$wl', 'exact8': $e, 'frames_s': d['value']}))" >> $O/exact8_ab.jsonl; done; done; cat $O/exact8_ab.jsonl
