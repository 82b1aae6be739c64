#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05c11; mkdir -p $O
for nw in 1 2; do MPCVR_ERRDIFF_NW=$nw timeout 900 python -m pytest tests/test_errdiff.py -x -q -m gpu > $O/tests_errdiff_nw$nw.txt 2>&1; tail -1 $O/tests_errdiff_nw$nw.txt; done
B="--steps 10 --warmup 3 --no-cpu-baseline --no-host-path"
run() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'tag': '$1', 'frames_s': d['value'], 'ms_per_step': d['ms_per_step']}))"; }
for nw in 1 2 4 8; do
MPCVR_ERRDIFF_NW=$nw timeout 300 python bench.py --workload c4ed $B 2>/dev/null | tail -1 | run nw$nw >> $O/ab.jsonl
MPCVR_ERRDIFF_NW=$nw timeout 300 python bench.py --workload c4ed --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | run nw${nw}_batch1 >> $O/ab.jsonl
done
cat $O/ab.jsonl
MPCVR_ERRDIFF_NW=1 MPCVR_NO_FRAME_LANES=1 MPCVR_ERRDIFF_DEBUG=1 timeout 300 python tools/ed_debug.py 336 2> $O/dbg_nw1.txt >/dev/null; tail -33 $O/dbg_nw1.txt | head -14
