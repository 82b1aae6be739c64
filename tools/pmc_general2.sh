#!/bin/bash
# tools/pmc_general2.sh "<case substring>" — memory-side PMC passes over one tools/bench_general.py case
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcg2; rm -rf $OUT; mkdir -p $OUT
i=0
# one counter family per pass and a hard timeout around every pass: "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" in ONE pass
# aborted rocprofv3 on this pool and then sat in its signal handler until gpurun's limit (20 GPU-minutes lost)
for set in "MemUnitBusy MemUnitStalled WriteUnitStalled VALUBusy" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python tools/bench_general.py "$1" > $OUT/log$i 2>&1 || tail -3 $OUT/log$i
done
python - <<'PY'
import csv, collections, glob
for d in sorted(glob.glob("gpurun_out/pmcg2/p*")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "mpcvr" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        print(k, {n: "%.4g" % (sum(v) / len(v)) for n, v in sorted(c.items())})
PY
