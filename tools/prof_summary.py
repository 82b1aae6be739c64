#!/usr/bin/env python3
"""Summarise a tools/prof.sh output directory: per-kernel time (kernel-trace --stats) and PMC averages per dispatch."""
import csv, glob, os, sys, collections

d = sys.argv[1]
kern_filter = sys.argv[2] if len(sys.argv) > 2 else "fused"
for f in glob.glob(os.path.join(d, "kt", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", f)
    for row in list(csv.DictReader(open(f)))[:6]:
        print("  %-60.60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (row.get("Name"), row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
acc = collections.defaultdict(list)
meta = {}
for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if kern_filter not in row["Kernel_Name"]:
            continue
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        meta = {k: row[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size") if k in row}
print("== dispatch:", meta)
for k in sorted(acc):
    v = acc[k]
    print("  %-24s avg=%.4g  n=%d" % (k, sum(v) / len(v), len(v)))
g = lambda k: (sum(acc[k]) / len(acc[k])) if acc.get(k) else None
if g("SQ_WAVE_CYCLES"):
    wc = g("SQ_WAVE_CYCLES")
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
        if g(k): print("  %-24s %.1f%% of wave cycles" % (k, 100 * g(k) / wc))
if g("FETCH_SIZE"): print("  FETCH_SIZE KB/dispatch %.0f  (x2 on gfx950 for wide streaming reads, see MI355X_MICROARCH.md)" % g("FETCH_SIZE"))
if g("WRITE_SIZE"): print("  WRITE_SIZE KB/dispatch %.0f" % g("WRITE_SIZE"))
