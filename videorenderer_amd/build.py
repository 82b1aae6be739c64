"""Build libmpcvr.so (HIP kernels + C-ABI) in-tree with hipcc for gfx950.

    python -m videorenderer_amd.build        # or: from videorenderer_amd.build import build; build()

The shared library lands next to this file (videorenderer_amd/libmpcvr.so); it is git-ignored but
travels with gpurun snapshots.  No torch involvement: the library links only libamdhip64.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmpcvr.so")
BUILD = os.path.join(HERE, "_build")
ARCH = "gfx950"

# (source, extra flags).  vp_kernels.hip and the host planner are built with -ffp-contract=off so the
# pass-per-kernel path rounds exactly like the CPU oracle; the fused path may contract.
SOURCES = [
    ("vp_plan.cpp", ["-ffp-contract=off"]),
    ("vp_dovi.cpp", ["-ffp-contract=off", "-std=c++20"]),
    ("hip_video_processor.cpp", []),
    ("mpcvr_capi.cpp", []),
    ("vp_kernels.hip", ["-ffp-contract=off", "-DMPCVR_EXACT_FP"]),
    ("vp_fused.hip", []),
    ("vp_fused_up2x_nt4.hip", []),
    ("vp_fused_up2x_nt5.hip", []),
    ("vp_fused_up2x_nt6.hip", []),
    ("vp_fused_strip.hip", []),
    ("vp_fused_strip_nt4.hip", []),
    ("vp_fused_strip_nt6.hip", []),
    ("vp_fused_strip_nt8.hip", []),
    ("vp_fused_strip_nt16.hip", []),
    ("vp_fused_period.hip", []),
    ("vp_fused_period_4_3.hip", []),
    ("vp_fused_period_3_2.hip", []),
    ("vp_fused_period_2_3.hip", []),
    ("vp_fused_period_1_2.hip", []),
    ("vp_fused_period_3_1.hip", []),
    ("vp_jinc.hip", []),
    ("vp_fused_jinc.hip", []),
    ("vp_errdiff.hip", []),
    ("vp_probe.hip", []),
]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(src, dst):
    return not os.path.exists(dst) or os.path.getmtime(src) > os.path.getmtime(dst)


def _project_deps(path, seen=None):
    """The project headers a file includes, transitively (#include "..." resolved beside the file): a translation unit is rebuilt
    when one of ITS headers changed, not when any header did (a host-only header costs seconds, not the four-minute kernel build)."""
    import re
    seen = set() if seen is None else seen
    try:
        text = open(path, encoding="utf-8", errors="replace").read()
    except OSError:
        return seen
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, flags=re.M):
        dep = os.path.normpath(os.path.join(os.path.dirname(path), inc))
        if os.path.exists(dep) and dep not in seen:
            seen.add(dep)
            _project_deps(dep, seen)
    return seen


def build(force=False, verbose=False):
    hipcc = _hipcc()
    os.makedirs(BUILD, exist_ok=True)
    objs = []
    common = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
    jobs = []
    for name, extra in SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(BUILD, name + ".o")
        objs.append(obj)
        if force or _newer(src, obj) or any(_newer(h, obj) for h in _project_deps(src)):
            # host files include hip_runtime.h for launch types: everything goes through -x hip
            cmd = [hipcc, "-x", "hip", "-c", src, "-o", obj] + common + extra
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append(cmd)
    if jobs:        # the translation units are independent: compile them side by side (the two kernel files dominate)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
            for rc in pool.map(lambda c: subprocess.run(c).returncode, jobs):
                if rc != 0:
                    raise subprocess.CalledProcessError(rc, "hipcc")
    if force or any(_newer(o, OUT) for o in objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
