#!/usr/bin/env python3
"""tests/tools/kernel_witnesses.py <kernel_trace.csv> <test_times.jsonl> <out.json> — which test launched which kernel instantiation.

The GPU suite runs under `rocprofv3 --kernel-trace` with MPCVR_TEST_TIMES set (tests/conftest.py logs every test's start / end on three
clocks); a dispatch belongs to the test whose interval holds its start timestamp.  The clock the trace is stamped in is found by trying
each: the one under which (nearly) every dispatch falls inside some test wins.  Output: {kernel instantiation: {"launches": n, "tests":
[first three test ids]}} — the coverage table of tests/test_kernel_coverage.py with a witness per row."""
import bisect
import csv
import json
import re
import sys

trace, times, out = sys.argv[1:4]
by_test_out = sys.argv[4] if len(sys.argv) > 4 else None          # optional: the inverse table {test: [kernels]} (large; for exploration)
tests = [json.loads(l) for l in open(times) if l.startswith("{")]
rows = []
for r in csv.DictReader(open(trace)):
    name = r.get("Kernel_Name") or r.get("Name") or ""
    if "mpcvr" not in name:
        continue
    m = re.search(r"(k_[a-z0-9_]+)(<[^(]*>)?", name)
    if not m:
        continue
    targs = re.sub(r"\((?:int|bool|unsigned int)\)", "", (m.group(2) or "").replace(" ", "")).replace("true", "1").replace("false", "0")
    rows.append((int(r["Start_Timestamp"]), m.group(1) + targs))
best = None
for clock in ("mono", "boot", "real"):
    iv = sorted((t["t0"][clock], t["t1"][clock], t["test"]) for t in tests)
    starts = [a for a, _, _ in iv]
    hit, table, inv = 0, {}, {}
    for ts, k in rows:
        i = bisect.bisect_right(starts, ts) - 1
        if i >= 0 and ts <= iv[i][1]:
            hit += 1
            e = table.setdefault(k, {"launches": 0, "tests": []})
            e["launches"] += 1
            tid = iv[i][2].split("::", 1)[-1]
            if k not in inv.setdefault(tid, []):
                inv[tid].append(k)
            if tid not in e["tests"] and len(e["tests"]) < 3:
                e["tests"].append(tid)
    if best is None or hit > best[0]:
        best = (hit, clock, table, inv)
hit, clock, table, inv = best
print(f"{len(rows)} dispatches, {hit} inside a test interval on clock '{clock}', {len(table)} kernel instantiations", file=sys.stderr)
json.dump({"clock": clock, "dispatches": len(rows), "attributed": hit, "kernels": dict(sorted(table.items()))}, open(out, "w"), indent=0)
if by_test_out:
    json.dump(inv, open(by_test_out, "w"), indent=0)
