#!/bin/bash
# tools/prof_general.sh [case substring ...] — rocprofv3 kernel-trace stats of tools/bench_general.py cases (general path)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/gen
i=0
for c in "$@"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gen -o case$i -- python tools/bench_general.py "$c" > gpurun_out/gen/case$i.log 2>&1
  grep '"case"' gpurun_out/gen/case$i.log | cut -c1-200
  python - "$i" <<'PY'
import csv, sys
i = sys.argv[1]
for r in csv.DictReader(open(f"gpurun_out/gen/case{i}_kernel_stats.csv")):
    if "mpcvr" in r["Name"]:
        print(f'  {r["Name"][:90]:90s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:8.2f}')
PY
done
