#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05c10; mkdir -p $O
timeout 900 python -m pytest tests/test_errdiff.py -x -q -m gpu > $O/tests_errdiff.txt 2>&1; echo "rc=$?" >> $O/tests_errdiff.txt
tail -3 $O/tests_errdiff.txt
MPCVR_ERRDIFF_NW=4 timeout 900 python -m pytest tests/test_errdiff.py -x -q -m gpu > $O/tests_errdiff_nw4.txt 2>&1; echo "rc=$?" >> $O/tests_errdiff_nw4.txt
tail -3 $O/tests_errdiff_nw4.txt
B="--steps 10 --warmup 3 --no-cpu-baseline --no-host-path"
run() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'tag': '$1', 'frames_s': d['value'], 'ms_per_step': d['ms_per_step']}))"; }
timeout 300 python bench.py --workload c4ed $B 2>/dev/null | tail -1 | run nw8 >> $O/ab.jsonl
MPCVR_ERRDIFF_NW=4 timeout 300 python bench.py --workload c4ed $B 2>/dev/null | tail -1 | run nw4 >> $O/ab.jsonl
MPCVR_ERRDIFF_GROUPS=256 timeout 300 python bench.py --workload c4ed $B 2>/dev/null | tail -1 | run nw8_groups256 >> $O/ab.jsonl
MPCVR_ERRDIFF_ORDER=0 timeout 300 python bench.py --workload c4ed $B 2>/dev/null | tail -1 | run nw8_framemajor >> $O/ab.jsonl
timeout 300 python bench.py --workload c4ed --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | run nw8_batch1 >> $O/ab.jsonl
MPCVR_ERRDIFF_NW=4 timeout 300 python bench.py --workload c4ed --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | run nw4_batch1 >> $O/ab.jsonl
timeout 300 python bench.py --workload c4ed --batch 96 --steps 5 --warmup 2 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | run nw8_batch96 >> $O/ab.jsonl
cat $O/ab.jsonl
MPCVR_NO_FRAME_LANES=1 timeout 600 python tools/ed_probe.py 3840 > $O/probe.jsonl 2>&1; cat $O/probe.jsonl
