"""Which kernel instantiations does the GPU suite actually launch?

libmpcvr.so holds ~1,200 kernel instantiations (template arguments = taps x tail x source layout x epilogue x ...); a planner typo can
select one no test ever ran.  tools/final_profiles.sh runs the whole GPU suite under `rocprofv3 --kernel-trace --stats` and the stats
table is committed (profiles/<round>/gpu_suite_kernel_stats.csv).  This CPU test compares it with the kernels of the CURRENT build
(the host-side __device_stub__ symbols of the library):

  * every kernel FAMILY of the library is launched by the suite;
  * every instantiation of the families the BASELINE configurations and the bench workloads run on by default (the exact-2x kernel and
    its matrix-core twin, the periodic-phase kernel, the streaming same-size convert) and of the folded row / tiled two-draw resize
    kernels is launched — or named in tests/golden/kernels_not_launched.json with a reason;
  * for the remaining table-driven families (k_fused_strip, k_convert_420, k_convert_blocks, k_resize_cols, k_jinc2_quad) the share of
    launched instantiations is reported and held to the recorded figure, so coverage cannot silently fall; the unlaunched ones are
    listed in the assertion message.
A kernel that exists in the build but not in the profile's era (added since) fails the second rule until the profile is refreshed.
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
STRICT_FAMILIES = ("k_fused_up2x", "k_fused_period", "k_convert_stream", "k_fused_up2x_mx", "k_resize_rows", "k_resize_2d")
# share of a family's instantiations the suite launched when the profile was taken (round 4, after the kernel-family sweep of
# tests/test_parity_gpu.py and the pruning of combinations no plan can produce: k_resize_cols 46 / 51, k_jinc2_quad 7 / 9, k_convert_420 71 / 160,
# k_fused_strip 137 / 441, k_convert_blocks 34 / 92; round 3: 16 / 54, 1 / 9, 30 / 160, 97 / 378, 30 / 92): the floors sit just under those
# figures — coverage of the table-driven families is partial and must not fall; the strict families are complete
FLOORS = {"k_fused_strip": 0.30, "k_convert_blocks": 0.36, "k_convert_420": 0.43, "k_resize_cols": 0.88, "k_jinc2_quad": 0.75}


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
    missing_fam = [f for f in fams if not any(family(k) == f for k in launched)]
    assert not missing_fam, f"kernel families the GPU suite never launches ({src}): {missing_fam}"
    report = []
    for fam in fams:
        b = sorted(k for k in built if family(k) == fam)
        got = [k for k in b if k in launched]
        miss = [k for k in b if k not in launched and k not in excused]
        share = len(got) / len(b)
        report.append(f"{fam}: {len(got)}/{len(b)}")
        if fam in STRICT_FAMILIES:
            assert not miss, f"{fam}: {len(miss)} instantiations are never launched by the GPU suite and carry no reason in kernels_not_launched.json: {miss[:12]}"
        elif fam in FLOORS:
            assert share >= FLOORS[fam], f"{fam}: only {len(got)} of {len(b)} instantiations launched (floor {FLOORS[fam]}); unlaunched e.g. {miss[:8]}"
    print("kernel coverage of the GPU suite:", "; ".join(report))
    stale = sorted(k for k in excused if k not in built)
    assert not stale, f"kernels_not_launched.json names kernels that are not in the build any more: {stale[:8]}"
