#!/bin/bash
# tools/pmc_wait.sh — where do the fused kernel's waves wait?  (one PMC pass, kernel-trace only)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcw; rm -rf $OUT; mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" "SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAVES SQ_LEVEL_WAVES SQ_IFETCH SQ_IFETCH_LEVEL"; do
  n=$(echo $set | md5sum | cut -c1-6)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$n -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $OUT/log_$n 2>&1
  python - <<PY
import csv, collections, glob
acc = collections.defaultdict(list)
for f in glob.glob('$OUT/$n/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'fused' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
g = {k: sum(v)/len(v) for k, v in acc.items()}
wc = g.get('SQ_WAVE_CYCLES', 1)
for k, v in sorted(g.items()): print('%-24s %.4g  (%.1f%% of wave cycles)' % (k, v, 100*v/wc))
PY
done
