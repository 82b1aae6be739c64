#!/bin/bash
# tools/pmc_general.sh "<case substring>" — two PMC passes (kernel-trace only) over one tools/bench_general.py case;
# prints per-kernel averages of the counters for the mpcvr kernels
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcg; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/p1 -o p -- python tools/bench_general.py "$1" > $OUT/log1 2>&1
timeout -k 5 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/p2 -o p -- python tools/bench_general.py "$1" > $OUT/log2 2>&1
python - <<'PY'
import csv, collections, glob
for d in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmcg/{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "mpcvr" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        print(k)
        print("   ", {n: "%.4g" % (sum(v) / len(v)) for n, v in sorted(c.items())})
PY
