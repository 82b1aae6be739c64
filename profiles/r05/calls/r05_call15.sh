#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05c15; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "dovi or Dovi" > $O/tests_dovi.txt 2>&1; tail -3 $O/tests_dovi.txt
B="--steps 20 --warmup 3 --no-cpu-baseline --no-host-path"
run() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'tag': '$1', 'frames_s': d['value'], 'ms_per_step': d['ms_per_step']}))"; }
for i in 1 2; do
timeout 300 python bench.py --workload dovi4k $B 2>/dev/null | tail -1 | run table >> $O/ab_dovi.jsonl
MPCVR_LIB=$GRAFT_REPO_ROOT/gpurun_in/libmpcvr_pqenc_alu.so timeout 300 python bench.py --workload dovi4k $B 2>/dev/null | tail -1 | run alu_chain >> $O/ab_dovi.jsonl
done
cat $O/ab_dovi.jsonl
python tests/tools/diag_dovi_tiers.py > $O/dovi_tiers.jsonl 2>&1; tail -5 $O/dovi_tiers.jsonl | cut -c1-400
