#!/bin/bash
# tools/pmc_quick.sh [bench args] — one PMC pass (instruction mix) + plain bench; prints VALU instr per wave-iteration
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcq; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/p -o p -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-path "$@" > $OUT/log 2>&1
python - <<'PY'
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/pmcq/p/p_counter_collection.csv')):
    if 'fused' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
g = {k: sum(v)/len(v) for k, v in acc.items()}
print({k: '%.4g' % v for k, v in g.items()})
print('VALU instr per wave: %.0f   SALU per wave: %.0f  LDS per wave: %.0f' % (g['SQ_INSTS_VALU']/g['SQ_WAVES'], g['SQ_INSTS_SALU']/g['SQ_WAVES'], g['SQ_INSTS_LDS']/g['SQ_WAVES']))
PY
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-path "$@" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['frac'], r['roofline']['kernel_ms_per_launch'])"
