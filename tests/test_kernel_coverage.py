"""Which kernel instantiations does the GPU suite actually launch?  All of them.

libmpcvr.so holds ~860 kernel instantiations (template arguments = taps x tail x source layout x epilogue x ...); a planner typo can
select one no test ever ran.  The GPU suite runs under `rocprofv3 --kernel-trace --stats` (tools/final_profiles.sh) with every test's
start / end logged (MPCVR_TEST_TIMES, tests/conftest.py); the stats table and the join of the two — which test launched which
instantiation, tests/tools/kernel_witnesses.py — are committed (profiles/<round>/gpu_suite_kernel_stats.csv, kernels_by_test.json).
This CPU test compares them with the kernels of the CURRENT build (the host-side __device_stub__ symbols of the library):

  * every instantiation in the library was launched by the suite (round 4: what no plan can select was pruned — one-pixel strips
    below 9 taps, the 8-bit loaders behind a PQ / HLG / BT.2020 tail, resize (texture, epilogue) pairs no plan produces — and the
    kernel-family sweep of tests/test_parity_gpu.py drives the rest, each case against the oracle), or is named in
    tests/golden/kernels_not_launched.json with a reason (empty since round 4);
  * every instantiation has a witness: a test id in kernels_by_test.json whose interval held one of its dispatches.
A kernel that exists in the build but not in the profile's era (added since) fails until the profile is refreshed.
"""
import csv
import glob
import json
import os
import re
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(ROOT, "videorenderer_amd", "libmpcvr.so")
def norm(name):
    """'void mpcvr::(anonymous namespace)::k_x<5, 1, (int)1>(args)' / '...__device_stub__k_x<5, 1, 1>(args)' -> 'k_x<5,1,1>'"""
    m = re.search(r"(k_[a-z0-9_]+)(<[^(]*>)?\s*\(", name)
    if not m:
        return None
    targs = (m.group(2) or "").replace(" ", "")
    targs = re.sub(r"\((?:int|bool|unsigned int)\)", "", targs).replace("true", "1").replace("false", "0")
    return m.group(1) + targs


def built_kernels():
    out = subprocess.run(["nm", "-C", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    ks = set()
    for line in out.splitlines():
        if "__device_stub__" in line:
            k = norm(line.split("__device_stub__", 1)[1])
            if k:
                ks.add(k)
    return ks


def launched_kernels():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "gpu_suite_kernel_stats.csv")))
    if not files:
        pytest.skip("no profiles/*/gpu_suite_kernel_stats.csv yet (tools/final_profiles.sh on the GPU box)")
    ks = set()
    for r in csv.DictReader(open(files[-1])):
        k = norm(r["Name"] + "(") if "(" not in r["Name"] else norm(r["Name"])
        if k and "mpcvr" in r["Name"]:
            ks.add(k)
    return ks, files[-1]


def family(k):
    return k.split("<")[0]


def test_gpu_suite_launches_every_kernel_family_and_every_headline_instantiation():
    if not os.path.exists(LIB):
        pytest.skip("libmpcvr.so not built")
    built = built_kernels()
    launched, src = launched_kernels()
    with open(os.path.join(HERE, "golden", "kernels_not_launched.json")) as f:
        excused = json.load(f)
    fams = sorted({family(k) for k in built})
    missing_fam = [f for f in fams if not any(family(k) == f for k in launched) and not all(k in excused for k in built if family(k) == f)]
    assert not missing_fam, f"kernel families the GPU suite never launches ({src}): {missing_fam}"
    report = []
    for fam in fams:
        b = sorted(k for k in built if family(k) == fam)
        got = [k for k in b if k in launched]
        miss = [k for k in b if k not in launched and k not in excused]
        report.append(f"{fam}: {len(got)}/{len(b)}")
        assert not miss, f"{fam}: {len(miss)} instantiations are never launched by the GPU suite and carry no reason in kernels_not_launched.json: {miss[:12]}"
    print("kernel coverage of the GPU suite:", "; ".join(report))
    # the witness table: a test for every instantiation
    wfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "kernels_by_test.json")))
    assert wfiles, "no profiles/*/kernels_by_test.json (tests/tools/kernel_witnesses.py on the GPU box)"
    with open(wfiles[-1]) as f:
        wit = json.load(f)
    assert wit["attributed"] >= 0.99 * wit["dispatches"], f"{wfiles[-1]}: only {wit['attributed']} of {wit['dispatches']} dispatches fall inside a test interval"
    nowit = sorted(k for k in built if k not in excused and not wit["kernels"].get(k, {}).get("tests"))
    assert not nowit, f"{len(nowit)} instantiations without a witness test in {wfiles[-1]}: {nowit[:12]}"
    stale = sorted(k for k in excused if k not in built)
    assert not stale, f"kernels_not_launched.json names kernels that are not in the build any more: {stale[:8]}"
