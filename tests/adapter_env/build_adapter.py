#!/usr/bin/env python3
"""tests/adapter_env/build_adapter.py — TEST INFRASTRUCTURE: link the adapter class and its driver into oracle/_ref/adapter_driver.

examples/hip_video_processor_adapter.cpp + adapter_driver.cpp are compiled with g++ against the reference's own CVideoProcessor method list
(interface.py cuts it out of /root/reference/Source/VideoProcessor.h into oracle/_ref/gen/adapter/VideoProcessor.h) and linked with
videorenderer_amd/libmpcvr.so.  Runs only where /root/reference is mounted; the binary is git-ignored (oracle/_ref/) and travels to the GPU box
with the snapshot, like the other artefacts built from the reference tree.  __graft_entry__.build() calls build()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import interface  # noqa: E402

OUT = os.path.join(ROOT, "oracle", "_ref", "adapter_driver")
GEN = os.path.join(ROOT, "oracle", "_ref", "gen", "adapter")


def build(quiet=True):
    if not interface.available():
        return None
    lib = os.path.join(ROOT, "videorenderer_amd", "libmpcvr.so")
    if not os.path.exists(lib):
        raise RuntimeError("build libmpcvr.so first (python -m videorenderer_amd.build)")
    os.makedirs(GEN, exist_ok=True)
    with open(os.path.join(GEN, "VideoProcessor.h"), "w") as f:
        f.write(interface.header_text())
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror=overloaded-virtual", "-Wsuggest-override", "-Werror=suggest-override"] + \
        interface.include_dirs(GEN, ROOT) + \
        [os.path.join(ROOT, "examples", "hip_video_processor_adapter.cpp"), os.path.join(HERE, "adapter_driver.cpp"),
         "-o", OUT, "-L", os.path.dirname(lib), "-l:libmpcvr.so", "-Wl,-rpath,$ORIGIN/../../videorenderer_amd", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("adapter link failed:\n" + r.stderr[-6000:])
    if not quiet:
        print("built", OUT)
    return OUT


if __name__ == "__main__":
    build(quiet=False)
