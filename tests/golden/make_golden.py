"""Generate tests/golden/*.json from the REAL reference parameter code.

Runs only where /root/reference is mounted (this container): builds oracle/_ref/libref_csputils.so from
/root/reference/Source/csputils.cpp (oracle/Makefile `ref`) and records the matrices that
mp_get_csp_matrix / GetColorspaceGamutConversionMatrix produce, as exact fp32 bit patterns.
The fixtures travel to the GPU box; the reference tree does not.

    python tests/golden/make_golden.py
"""
import ctypes as C
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def main():
    O.build(ref=True)
    R = O.ref()
    assert R is not None, "oracle/_ref not built: /root/reference missing?"
    cases = []
    # (space, levels, bits) x procamp settings; mp_csp / mp_csp_levels numeric values
    procamps = [(0.0, 1.0, 0.0, 1.0), (10 / 255, 1.1, 0.3, 0.9), (-25 / 255, 0.8, -1.2, 1.5), (0.0, 1.0, 3.1, 0.0)]
    for space in (0, 1, 2, 3, 4, 8):
        for levels in (0, 1, 2):
            for nbits in (8, 10, 16):
                for (b, c, h, s) in procamps:
                    m = (C.c_float * 9)()
                    cc = (C.c_float * 3)()
                    for gray in (0, 1):      # csp_params.gray: CS_GRAY sources (DX11VideoProcessor.cpp:843)
                        if gray and (b, c, h, s) not in procamps[:2]:
                            continue
                        R.ref_csp_matrix(space, levels, nbits, b, c, h, s, gray, m, cc)
                        cases.append(dict(space=space, levels=levels, bits=nbits, gray=gray,
                                          brightness=bits(b), contrast=bits(c), hue=bits(h), saturation=bits(s),
                                          m=[bits(v) for v in m], c=[bits(v) for v in cc]))
    g = (C.c_float * 9)()
    R.ref_gamut_matrix(4, 3, g)          # MP_CSP_PRIM_BT_2020 -> MP_CSP_PRIM_BT_709
    out = dict(source="/root/reference/Source/csputils.cpp via oracle/_ref (real reference code)",
               csp_matrix=cases, gamut_2020_to_709=[bits(v) for v in g])
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "csputils_ref.json"), "w") as f:
        json.dump(out, f)
    print(f"wrote {len(cases)} csp cases")


if __name__ == "__main__":
    main()
