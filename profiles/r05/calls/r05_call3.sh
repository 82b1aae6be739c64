#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05c4; mkdir -p $O
timeout 900 python -m pytest tests/test_errdiff.py -x -q -m gpu > $O/tests_errdiff.txt 2>&1; echo "rc=$?" >> $O/tests_errdiff.txt
tail -3 $O/tests_errdiff.txt
B="--steps 10 --warmup 3 --no-cpu-baseline --no-host-path"
run() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'tag': '$1', 'frames_s': d['value'], 'ms_per_step': d['ms_per_step']}))"; }
timeout 300 python bench.py --workload c4ed $B 2>/dev/null | tail -1 | run default >> $O/ab.jsonl
for w in 1024 2048 3072 4096; do MPCVR_ERRDIFF_WAVES=$w timeout 300 python bench.py --workload c4ed $B 2>/dev/null | tail -1 | run waves$w >> $O/ab.jsonl; done
for w in 1024 2048 3072; do MPCVR_ERRDIFF_ORDER=0 MPCVR_ERRDIFF_WAVES=$w timeout 300 python bench.py --workload c4ed $B 2>/dev/null | tail -1 | run framemajor_waves$w >> $O/ab.jsonl; done
timeout 300 python bench.py --workload c4ed --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | run batch1 >> $O/ab.jsonl
timeout 300 python bench.py --workload c4ed --batch 96 --steps 5 --warmup 2 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | run batch96 >> $O/ab.jsonl
MPCVR_ERRDIFF_ORDER=0 MPCVR_ERRDIFF_WAVES=2048 timeout 300 python bench.py --workload c4ed --batch 96 --steps 5 --warmup 2 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | run batch96_framemajor_2048 >> $O/ab.jsonl
cat $O/ab.jsonl
