#!/bin/bash
# tools/pmc_general2.sh "<case substring>" — memory-side PMC passes over one tools/bench_general.py case
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcg2; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "MemUnitBusy MemUnitStalled WriteUnitStalled VALUBusy" "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" "TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_64B_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python tools/bench_general.py "$1" > $OUT/log$i 2>&1 || tail -3 $OUT/log$i
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
