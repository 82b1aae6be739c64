"""Generate tests/golden/hostmath_ref.json from the REFERENCE's own host-side parameter maths (oracle/ref_hlsl/ref_hostmath.py:
SetShaderDoviCurves[Poly], SetHDR10ShaderParams, the level 1 / 3 / 2 block of CopySample + SetDolbyVisionDynamicParams,
SpecifyExtendedFormat, CopyFrameV210 compiled from /root/reference).  Runs only where the reference is mounted; the recorded
outputs let tests/test_oracle_pins.py hold the oracle AND the product's host code (vp_dovi.cpp, vp_plan.cpp) to the reference's
results everywhere.

    python tests/golden/make_hostmath_golden.py
"""
import ctypes as C
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
import ref_hostmath as RH  # noqa: E402
from videorenderer_amd import synth  # noqa: E402

DOVI_CASES = {
    "poly": dict(kind="poly"),
    "poly_l1": dict(kind="poly", l1=True),
    "poly_l1_l3": dict(kind="poly", l1=True, l3=True),
    "mmr": dict(kind="mmr"),
    "mmr_l2_between": dict(kind="mmr", l2=(100, 600, 1000)),
    "mixed_l2_brighter": dict(kind="mixed", l2=(100, 600)),
    "mixed_l2_dimmer": dict(kind="mixed", l2=(2000, 4000)),
    "identity_l2_exact": dict(kind="identity", l2=(1000,)),
}
DISPLAYS = (100, 400, 800, 1000, 4000)
HDR10_CASES = [(0.0, 0.0, 0.0, 0.0, 0.0, 0), (0.005, 1000.0, 800.0, 400.0, 600.0, 3), (-1.0, 10.0, 10.0, 1.0, 99.0, 7),
               (0.01, 4000.0, 0.0, 0.0, 10000.0, 6), (0.5, 11.0, 11.0, 2.0, 10001.0, 1), (50.0, 1200.0, 1100.0, 0.5, 100.0, 5)]
EXTFMT_CASES = [(ex, cf, w, h) for ex in (0, 0x0288ca500 & 0xffffffff, (5 << 8) | (1 << 12), (4 << 15) | (9 << 22) | (15 << 27), (7 << 8) | (2 << 12) | (2 << 15))
                for cf, w, h in ((1, 1920, 1080), (2, 720, 576), (2, 1024, 576), (6, 1280, 720), (16, 640, 360), (30, 1920, 1080), (37, 1920, 1080), (20, 1025, 576))]
# (ref_copy_frame_rgb kind = the oracle's RPK_* code, sample bytes per pixel, texture bytes per pixel, width, lines, bottom-up)
RGB_COPY_CASES = [(1, 3, 4, 46, 5, 0), (1, 3, 4, 47, 4, 0), (1, 3, 4, 48, 3, 1), (0, 4, 4, 33, 4, 0), (0, 4, 4, 32, 4, 1), (2, 4, 4, 31, 3, 0),
                  (3, 6, 8, 48, 3, 0), (3, 6, 8, 46, 3, 0), (4, 6, 8, 45, 3, 0), (4, 6, 8, 46, 3, 1), (4, 6, 8, 47, 2, 0), (4, 6, 8, 48, 2, 0),
                  (5, 8, 8, 19, 3, 0), (6, 8, 8, 20, 3, 1)]
V210_CASES = [(48, 4), (46, 3), (6, 1), (1280, 2), (1921, 2)]          # (width, lines)


def dovi_struct(kw):
    return O.fill_dovi(O.OrcDovi(), synth.dovi_metadata(**kw))


def ref_curves(L, st, poly):
    buf = (C.c_uint8 * (3 * 1024))()
    n = L.ref_dovi_curves(C.byref(st), 1 if poly else 0, buf)
    return np.frombuffer(bytes(buf)[:n], np.uint32).copy()


def ref_levels(L, st, disp):
    k5 = (C.c_float * 5)(); en = C.c_int(0); l1 = (C.c_uint * 3)()
    present = L.ref_dovi_levels(C.byref(st), disp, k5, C.byref(en), l1)
    return dict(k5=[float(np.float32(x)) for x in k5], k5_bits=[int(np.float32(x).view(np.uint32)) for x in k5], enabled=int(en.value), l1=[int(x) for x in l1], l1_present=int(present))


def fmt_row(cf):
    """CSType / Subsampling of the format's s_FmtConvMapping row (Helper.cpp:309-359) as the stand-in enum orders them: CS_YUV 0, CS_RGB 1, CS_GRAY 2"""
    if cf in (26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36):
        return 1, 444
    if cf in (37, 38, 39):
        return 2, 400
    sub = {1: 420, 2: 420, 3: 420, 4: 422, 5: 422, 6: 422, 7: 422, 8: 422, 9: 422, 10: 422, 11: 444, 12: 444, 13: 444, 14: 420, 15: 422, 16: 444,
           17: 420, 18: 422, 19: 444, 20: 420, 21: 420, 22: 422, 23: 422, 24: 444, 25: 444}[cf]
    return 0, sub


def v210_sample(width, lines):
    pitch = ((width + 47) // 48) * 128
    rng = np.random.default_rng(width * 131 + lines)
    return rng.integers(0, 2 ** 32, size=pitch * lines // 4, dtype=np.uint32).view(np.uint8), pitch


def rgb_sample(kind, pack, width, lines):
    pitch = (width * pack + 3) & ~3
    rng = np.random.default_rng(kind * 1009 + width * 31 + lines)
    return rng.integers(0, 256, size=pitch * lines + 16, dtype=np.uint8), pitch


def rgb_copy(fn, kind, pack, tbpp, width, lines, bottom_up):
    """The upload as MemCopyToTexSrcVideo drives it (:1243-1248): a bottom-up DIB is walked from its last row with a negative pitch;
    the texture row is wide enough for every pixel the copy loop writes (the oracle's rule, mpcvr_oracle.c setup_convert)."""
    src, pitch = rgb_sample(kind, pack, width, lines)
    row_px = max(width, pitch // pack + 1)
    tp = row_px * tbpp
    dst = np.zeros(tp * lines + 16, np.uint8)
    base = src.ctypes.data + (pitch * (lines - 1) if bottom_up else 0)
    fn(kind, lines, dst.ctypes.data, tp, base, -pitch if bottom_up else pitch)
    rows = np.lib.stride_tricks.as_strided(dst, shape=(lines, tp), strides=(tp, 1))[:, :width * tbpp]
    return hashlib.sha256(np.ascontiguousarray(rows).tobytes()).hexdigest()


def main():
    L = RH.lib()
    assert L is not None, "the reference tree is not mounted"
    doc = dict(source="reference host code compiled by oracle/ref_hlsl/ref_hostmath.py", dovi={}, hdr10=[], extfmt=[], v210=[])
    for name, kw in DOVI_CASES.items():
        st = dovi_struct(kw)
        full, poly = ref_curves(L, st, False), ref_curves(L, st, True)
        doc["dovi"][name] = dict(kw=kw, curves_sha256=hashlib.sha256(full.tobytes()).hexdigest(), curves_poly_sha256=hashlib.sha256(poly.tobytes()).hexdigest(),
                                 curves_words=[int(x) for x in full], curves_poly_words=[int(x) for x in poly],
                                 levels={str(d): ref_levels(L, st, d) for d in DISPLAYS})
    for c in HDR10_CASES:
        out = (C.c_uint32 * 6)()
        L.ref_hdr10_params(*c, out)
        doc["hdr10"].append(dict(args=list(c), words=[int(x) for x in out]))
    for ex, cf, w, h in EXTFMT_CASES:
        cs, sub = fmt_row(cf)
        doc["extfmt"].append(dict(exfmt=ex, cformat=cf, w=w, h=h, out=int(L.ref_specify_extfmt(ex, cs, sub, w, h))))
    for width, lines in V210_CASES:
        src, pitch = v210_sample(width, lines)
        tp = O.lib().orc_v210_tex_pitch(width)
        dst = np.zeros(tp * lines + 16, np.uint8)
        L.ref_copy_frame_v210(lines, dst.ctypes.data, tp, src.ctypes.data, pitch)
        doc["v210"].append(dict(width=width, lines=lines, pitch=pitch, tex_pitch=int(tp), sha256=hashlib.sha256(dst[:tp * lines].tobytes()).hexdigest()))
    doc["rgbcopy"] = []
    for kind, pack, tbpp, width, lines, bu in RGB_COPY_CASES:
        ref_fn = lambda k, n, d, dp, s_, sp: L.ref_copy_frame_rgb(k, n, d, dp, s_, sp)
        doc["rgbcopy"].append(dict(kind=kind, pack=pack, tbpp=tbpp, width=width, lines=lines, bottom_up=bu,
                                   sha256=rgb_copy(ref_fn, kind, pack, tbpp, width, lines, bu)))
    with open(os.path.join(HERE, "hostmath_ref.json"), "w") as f:
        json.dump(doc, f, indent=0, sort_keys=True)
    print("recorded", {k: len(v) for k, v in doc.items() if k != "source"})


if __name__ == "__main__":
    main()
