"""Generate tests/golden/ref_hlsl_pins.json (+ ref_hlsl_outputs.npz) by executing the REFERENCE's own shader text.

Runs only where /root/reference is mounted (this container).  For every golden / pinning case the whole Process() is run
through oracle/ref_hlsl (the real HLSL of /root/reference/Shaders and the text the real Source/Shaders.cpp generates, compiled
for the CPU) and the result is recorded:
  * sha256 of the B,G,R bytes of the render target (every case);
  * how the oracle compares (max |difference| per 8/10-bit channel, share of differing channels) — the oracle must reproduce
    these figures, see tests/test_ref_hlsl.py;
  * the full reference output for the cases where the oracle is not bit-identical, so that the <= 1 LSB check also runs where
    the reference tree does not exist.
Alpha: the render target's A is whatever the shader leaves there (not 1 after the float4-wide HLG / Dolby Vision tails); the
swap chain ignores it (B8G8R8X8 semantics) and so do the pins.

    python tests/golden/make_ref_hlsl_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_hlsl"))
from oracle import oracle as O  # noqa: E402
import ref_pipeline as RP  # noqa: E402
from tests.golden import cases  # noqa: E402


def rgb_channels(img):
    """(h, w, 3) integer channels of a BGRA8 (h,w,4) uint8 or RGB10A2 (h,w) uint32 render target."""
    if img.ndim == 2:
        return np.stack([(img >> (10 * k)) & 0x3ff for k in range(3)], -1).astype(np.int32)
    return img[..., :3].astype(np.int32)


def all_cases():
    d = dict(cases.GOLDEN_CASES)
    d.update(cases.PINNING_CASES)
    return d


def comparable(c):
    # interleaved RGB is not decoded by ref_pipeline; flags=1 is OUR Lanczos3 tap-layout fix (no reference counterpart)
    return RP.supported(c["cformat"]) and not c.get("flags", 0)


def run_pair(name, c):
    frame, pitch = cases.case_frame(c)
    p = cases.oracle_params(O, c)
    a = cases.run_case(O, name)
    b = RP.process(p, frame, pitch)
    if b.ndim == 2:
        a = a.view(np.uint32).reshape(b.shape)
    return a, b


def main():
    pins, outs = {}, {}
    for name, c in all_cases().items():
        if not comparable(c):
            continue
        a, b = run_pair(name, c)
        ca, cb = rgb_channels(a), rgb_channels(b)
        d = np.abs(ca - cb)
        pins[name] = dict(rgb_sha256=hashlib.sha256(cb.astype(np.uint16).tobytes()).hexdigest(),
                          oracle_max=int(d.max()), oracle_differing=float((d > 0).mean()))
        if d.max() > 0:
            outs[name] = b
        print(f"{name:45s} max {d.max()} differing {100 * (d > 0).mean():.4f}%")
    with open(os.path.join(HERE, "ref_hlsl_pins.json"), "w") as f:
        json.dump(dict(source="reference HLSL text executed by oracle/ref_hlsl (real Shaders/*.hlsl + real Source/Shaders.cpp output)",
                       cases=pins), f, indent=0, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "ref_hlsl_outputs.npz"), **outs)
    print(f"{len(pins)} cases, {len(outs)} with stored outputs")


if __name__ == "__main__":
    main()
