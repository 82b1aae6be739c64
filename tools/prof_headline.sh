#!/bin/bash
# tools/prof_headline.sh <tag> <bench flags...> — rocprofv3 evidence for the headline workload (runs on the GPU box):
#   kernel trace (+ --stats) of `bench.py <flags>` and separate --pmc passes (kernel-trace only, hard timeouts; gpurun refuses
#   pmc + sys/hip traces), summarised into gpurun_out/<tag>_summary.txt; the csv files stay next to it.
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
# the kernel trace runs bench.py with its DEFAULT --steps / --warmup (what the driver runs: the average launch duration in the stats
# table is then directly comparable with the bench line's kernel_ms_per_launch; a 12-step run reads ~7 % slower, the clocks are still
# ramping); the counter passes below only need a few dispatches
FULL="python bench.py --no-cpu-baseline --no-host-path $*"
BENCH="python bench.py --steps 12 --warmup 4 --settle 0 --no-cpu-baseline --no-host-path $*"
$FULL > $OUT/bench_plain.json 2> $OUT/bench_plain.err
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $FULL > $OUT/bench_under_kernel_trace.json 2> $OUT/kt.err
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- $BENCH > $OUT/pmc$i.log 2>&1 || tail -3 $OUT/pmc$i.log
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections, json, os
out = sys.argv[1]
lines = []
for f in glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        lines.append(f"  {r['Name'][:70]:70s} calls={r['Calls']} avg_ns={float(r['AverageNs']):.0f} pct={r['Percentage']}")
acc = collections.defaultdict(list); meta = {}
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ.get("KFILTER", "k_fused_up2x") in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r.get(k) for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size")}
avg = {k: sum(v) / len(v) for k, v in acc.items()}
with open(out + "_summary.txt", "w") as o:
    o.write("== kernel stats (rocprofv3 --kernel-trace --stats)\n" + "\n".join(lines) + "\n== dispatch: " + json.dumps(meta) + "\n")
    for k in sorted(avg):
        o.write(f"  {k:28s} avg per dispatch = {avg[k]:.4g}   (n={len(acc[k])})\n")
    wc = avg.get("SQ_WAVE_CYCLES")
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
            if k in avg: o.write(f"  {k:28s} {100 * avg[k] / wc:.1f}% of wave cycles\n")
    if "SQ_BUSY_CYCLES" in avg and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
        o.write(f"  MFMA busy / SQ busy cycles   {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / avg['SQ_BUSY_CYCLES']:.4f}\n")
    if "SQ_INSTS_VALU" in avg and "SQ_WAVES" in avg:
        o.write(f"  VALU instructions per wave   {avg['SQ_INSTS_VALU'] / avg['SQ_WAVES']:.0f}   MFMA per wave {avg.get('SQ_INSTS_MFMA', 0) / avg['SQ_WAVES']:.0f}\n")
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        o.write(f"  HBM traffic per dispatch     {(2 * avg['FETCH_SIZE'] + avg['WRITE_SIZE']) * 1024:.4g} B  (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, KB -> B)\n")
    for nm in ("bench_plain.json", "bench_under_kernel_trace.json"):
        try:
            j = json.loads([l for l in open(os.path.join(out, nm)) if l.startswith("{")][0])
            o.write(f"  {nm}: {j['value']} frames/s, kernel {j['roofline']['kernel_ms_per_launch']} ms/launch, frac {j['roofline']['frac']}\n")
        except Exception as e:
            o.write(f"  {nm}: unreadable ({e})\n")
print(open(out + "_summary.txt").read())
PY
