"""Generate tests/golden/full_size_pins.json: the REFERENCE's own shader text run over the BASELINE configurations at their
real sizes (tests/golden/cases.py FULL_SIZE_CASES), through oracle/ref_hlsl (the real Shaders/*.hlsl and the convert shader the
real Source/Shaders.cpp emits, compiled for the CPU).  Runs only where /root/reference is mounted (this container); about
15 s per 8K frame on 8 threads.  Recorded per case: sha256 of the B, G, R channels of the reference-text render target and how
the oracle compares with it (it must be bit-identical: tests/test_ref_hlsl.py re-runs the oracle against the hash everywhere,
and against the live shader text wherever oracle/_ref/libref_hlsl.so exists).

    python tests/golden/make_full_size_pins.py
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_hlsl"))
from oracle import oracle as O  # noqa: E402
import ref_pipeline as RP  # noqa: E402
from tests.golden import cases  # noqa: E402
from tests.golden.make_ref_hlsl_golden import rgb_channels  # noqa: E402


def rgb_sha(img):
    return hashlib.sha256(rgb_channels(img).astype(np.uint16).tobytes()).hexdigest()


def main():
    RP.build_all()          # the convert shaders of these sizes are part of libref_hlsl.so (so the GPU box can run them)
    pins = {}
    for name, c in cases.FULL_SIZE_CASES.items():
        frame, pitch = cases.case_frame(c)
        p = cases.oracle_params(O, c)
        t = time.time()
        ref = RP.process(p, frame, pitch)
        dt = time.time() - t
        got = O.process(p, frame, pitch)
        d = np.abs(rgb_channels(got) - rgb_channels(ref))
        pins[name] = dict(rgb_sha256=rgb_sha(ref), oracle_max=int(d.max()), oracle_differing=float((d > 0).mean()),
                          shape=list(ref.shape))
        print(f"{name:24s} reference text {dt:5.1f} s   oracle vs text: max {d.max()} differing {100 * (d > 0).mean():.5f} %")
    with open(os.path.join(HERE, "full_size_pins.json"), "w") as f:
        json.dump(dict(source="reference HLSL text executed by oracle/ref_hlsl over tests/golden/cases.py FULL_SIZE_CASES", cases=pins),
                  f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
