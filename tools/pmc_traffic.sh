#!/bin/bash
# tools/pmc_traffic.sh <workload> [steps] — HBM traffic of one bench.py workload: FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 --pmc passes (kernel-trace only, hard timeouts), summed over the mpcvr kernels and divided by the number of steps;
# a third pass collects the issue-slot counters (SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY): the
# VALU pipe's busy share of the launch is the roof that binds the fused kernels (bench.py: roofline.valu_issue)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
W=${1:-c1}; STEPS=${2:-6}; WARM=2
OUT=gpurun_out/traffic_$W; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  d=${c%% *}
  timeout -k 5 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$d -o p -- python bench.py --workload $W --steps $STEPS --warmup $WARM --settle 0 --no-cpu-baseline --no-host-path > $OUT/log_$d 2>&1 || tail -2 $OUT/log_$d
done
python - "$W" "$STEPS" "$WARM" <<'PY'
import csv, glob, sys, json
w, steps, warm = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
# the library's kernels of the step — not bench.py's own bandwidth probes (mpcvr::k_probe_shape / k_probe_up2x: they move several steps' worth of bytes)
ours = lambda name: "mpcvr" in name and "k_probe" not in name
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    s = 0.0; n = 0
    for f in glob.glob(f"gpurun_out/traffic_{w}/{c}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if ours(r["Kernel_Name"]) and r["Counter_Name"] == c:
                s += float(r["Counter_Value"]); n += 1
    tot[c] = (s, n)
launches = steps + warm
sq = {}
for f in glob.glob(f"gpurun_out/traffic_{w}/SQ_ACTIVE_INST_VALU/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if ours(r["Kernel_Name"]):
            sq[r["Counter_Name"]] = sq.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
# SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs, SQ_BUSY_CYCLES cycles summed over the 32 shader engines' SQs
# (MI355X_MICROARCH.md, rocprofv3 units): VALU-busy cycles per SIMD / cycles of the launch
issue = (sq["SQ_ACTIVE_INST_VALU"] * 4 / 1024) / (sq["SQ_BUSY_CYCLES"] / 32) if sq.get("SQ_BUSY_CYCLES") else None
# the clock the kernels sustained in that pass: busy cycles per SQ over the kernels' own duration (kernel trace of the same run)
dur_ns = 0.0
for f in glob.glob(f"gpurun_out/traffic_{w}/SQ_ACTIVE_INST_VALU/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if ours(r.get("Kernel_Name", "")):
            dur_ns += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
mhz = (sq["SQ_BUSY_CYCLES"] / 32) / dur_ns * 1e3 if dur_ns and sq.get("SQ_BUSY_CYCLES") else None
line = json.dumps({"workload": w, "steps_profiled": launches, "fetch_kb_per_step": tot["FETCH_SIZE"][0] / launches,
                   "write_kb_per_step": tot["WRITE_SIZE"][0] / launches, "dispatches": tot["FETCH_SIZE"][1],
                   "valu_issue_frac": issue, "sustained_mhz": mhz, "sq": {k: v / launches for k, v in sq.items()},
                   "wait_inst_any_share": (sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"]) if sq.get("SQ_WAVE_CYCLES") else None})
print(line)
open(f"gpurun_out/traffic_{w}.json", "w").write(line + "\n")      # tools/update_traffic.py folds it into profiles/hbm_traffic.json
PY
