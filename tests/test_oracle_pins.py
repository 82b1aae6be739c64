"""Pins for the CPU oracle (oracle/mpcvr_oracle.c).

The reference holds no tests or golden vectors for this path, so the pins are
manufactured: (1) csputils matrices from the REAL reference code, live via oracle/_ref
and as committed fixtures (tests/golden/csputils_ref.json); (2) dither sha256 + permutation property;
(3) resize phase weights; (4) grey-ramp PQ/HLG->SDR values; (5) chroma-siting weight tables;
(6) golden hashes of whole-frame oracle outputs on the seeded synthetic inputs.
"""
import ctypes as C
import hashlib
import json
import os
import re
import struct

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def f32(bits):
    return struct.unpack("<f", struct.pack("<I", bits))[0]


def csp(lib, fn, *args):
    m = (C.c_float * 9)()
    c = (C.c_float * 3)()
    getattr(lib, fn)(*args, m, c)
    return np.array(m, dtype=np.float32), np.array(c, dtype=np.float32)


# ---------------------------------------------------------------- (1) csputils
def test_csp_matrix_matches_reference_fixtures(oracle):
    with open(os.path.join(HERE, "golden", "csputils_ref.json")) as f:
        g = json.load(f)
    assert len(g["csp_matrix"]) >= 200
    for case in g["csp_matrix"]:
        m, c = csp(oracle.lib(), "orc_csp_matrix", case["space"], case["levels"], case["bits"],
                   f32(case["brightness"]), f32(case["contrast"]), f32(case["hue"]), f32(case["saturation"]), case.get("gray", 0))
        assert [int(v) for v in m.view(np.uint32)] == case["m"], case
        assert [int(v) for v in c.view(np.uint32)] == case["c"], case
    gm = (C.c_float * 9)()
    oracle.lib().orc_gamut_2020_to_709(gm)
    assert [int(v) for v in np.array(gm, dtype=np.float32).view(np.uint32)] == g["gamut_2020_to_709"]


def test_csp_matrix_matches_live_reference(oracle):
    R = oracle.ref()
    if R is None:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(7)
    for space in (0, 1, 2, 3, 4, 8):
        for levels in (0, 1, 2):
            for bits in (8, 10, 16):
                for _ in range(8):
                    b, ct = rng.uniform(-100, 100) / 255, rng.uniform(0, 2)
                    h, s = rng.uniform(-np.pi, np.pi), rng.uniform(0, 2)
                    for gray in (0, 1):
                        a = csp(oracle.lib(), "orc_csp_matrix", space, levels, bits, b, ct, h, s, gray)
                        r = csp(R, "ref_csp_matrix", space, levels, bits, b, ct, h, s, gray)
                        assert np.array_equal(a[0], r[0]) and np.array_equal(a[1], r[1])


def test_known_answer_matrices(oracle):
    # SURVEY.md §8a-3, computed there with the real code
    m, c = csp(oracle.lib(), "orc_csp_matrix", 2, 1, 8, 0, 1, 0, 1, 0)
    assert np.allclose(m[:3], [1.16438353, 0, 1.79274106], rtol=0, atol=1e-7) and abs(c[0] + 0.972945094) < 1e-7
    m, c = csp(oracle.lib(), "orc_csp_matrix", 2, 1, 10, 0, 1, 0, 1, 0)
    assert np.allclose(m[:3], [1.16780818, 0, 1.79801381], rtol=0, atol=1e-7) and abs(c[0] + 0.972945035) < 1e-7
    m, c = csp(oracle.lib(), "orc_csp_matrix", 4, 1, 16, 0, 1, 0, 1, 0)
    assert np.allclose(m, [1.16893196, 0, 1.68523157, 1.16893196, -0.188057855, -0.652965128,
                           1.16893196, 2.15013862, 0], rtol=0, atol=2e-7)
    assert np.allclose(c, [-0.915687978, 0.347458541, -1.14814508], rtol=0, atol=2e-7)
    g = (C.c_float * 9)()
    oracle.lib().orc_gamut_2020_to_709(g)
    assert np.allclose(np.array(g), [1.66049695, -0.587656736, -0.0728399456, -0.124547064, 1.13289523,
                                     -0.0083479844, -0.0181536824, -0.100597292, 1.11875105], rtol=0, atol=1e-7)


def test_color_matrix_from_extfmt(oracle):
    # P010 BT.2020 / PQ / TV range -> the 16-bit BT.2020NC matrix; NV12 with nothing declared -> BT.709 TV (HD)
    ex = oracle.make_extfmt(chroma=5, nominal_range=2, matrix=4, primaries=9, transfer=15)
    p = oracle.default_params(cformat=oracle.CF["P010"], width=3840, height=2160, exfmt=ex)
    cm = oracle.color_matrix(p)
    assert abs(cm[0] - 1.16893196) < 2e-7 and abs(cm[2] - 1.68523157) < 2e-7
    p = oracle.default_params(cformat=oracle.CF["NV12"], width=1920, height=1080, exfmt=0)
    cm = oracle.color_matrix(p)
    assert abs(cm[0] - 1.16438353) < 1e-7 and abs(cm[2] - 1.79274106) < 1e-7
    p = oracle.default_params(cformat=oracle.CF["NV12"], width=720, height=576, exfmt=0)      # SD -> BT.601
    cm = oracle.color_matrix(p)
    assert abs(cm[2] - 1.59602678) < 2e-7


def test_specify_extended_format_defaults(oracle):
    L = oracle.lib()
    v = L.orc_specify_extfmt(0, oracle.CF["NV12"], 1920, 1080)
    f = lambda sh, m: (v >> sh) & m
    assert (f(8, 0xf), f(12, 7), f(15, 7), f(18, 0xf), f(22, 0x1f), f(27, 0x1f)) == (5, 2, 1, 3, 2, 5)
    v = L.orc_specify_extfmt(0, oracle.CF["NV12"], 720, 480)
    assert (v >> 15) & 7 == 2
    v = L.orc_specify_extfmt(oracle.make_extfmt(chroma=7), oracle.CF["YUV444P10"], 1920, 1080)
    assert (v >> 8) & 0xf == 0          # non-4:2:0 => chroma siting cleared (Helper.cpp:1176-1178)


# ---------------------------------------------------------------- (2) dither
def test_dither_table_bit_exact(oracle):
    raw = open(oracle.DITHER_PATH, "rb").read()
    assert hashlib.sha256(raw).hexdigest() == "24b5048f1390879d3435072f6e48b62b042fd4787b33bfac5e6d9bc84ec28e7c"
    d = np.frombuffer(raw, dtype=np.float16).astype(np.float64)
    assert sorted((d * 1024).round().astype(int).tolist()) == list(range(1024))
    ref = "/root/reference/Source/res/dither32x32float16.bin"
    if os.path.exists(ref):
        assert open(ref, "rb").read() == raw
    # the table compiled into the product library
    inc = open(os.path.join(ROOT, "videorenderer_amd", "csrc", "dither_table.inc")).read()
    vals = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]{4}", inc.split("\n", 3)[3])]
    assert np.array(vals, dtype=np.uint16).tobytes() == raw


# ---------------------------------------------------------------- (3) weights
PINS_T025 = {   # SURVEY.md §8a-6, fp32 as written, t = 0.25
    4: [-0.01083153, -0.08472481, 0.89105344, 0.23991315, -0.01790517, -0.01750513],
    3: [-0.08472481, 0.86980093, 0.23282897, -0.01790517],
    2: [-0.0703125, 0.8671875, 0.2265625, -0.0234375],
    1: [-0.0234375, 0.78211808, 0.25607640, -0.01475694],
}


def up_weights(oracle, method, t):
    w = (C.c_float * 6)()
    n = oracle.lib().orc_upscale_weights(method, t, w)
    return np.array(w[:n], dtype=np.float32)


@pytest.mark.parametrize("method", [1, 2, 3, 4])
def test_upscale_weight_pins(oracle, method):
    w = up_weights(oracle, method, 0.25)
    assert np.allclose(w, PINS_T025[method], rtol=0, atol=2e-8 + 6e-8)
    # t = 0.75 is the mirror image
    assert np.allclose(up_weights(oracle, method, 0.75), w[::-1], rtol=0, atol=2e-7)
    for t in np.linspace(0.01, 0.99, 50, dtype=np.float32):
        assert abs(float(np.sum(up_weights(oracle, method, float(t)).astype(np.float64))) - 1.0) < 4e-7
    if method >= 3:     # "case t == 0 is required to return the centre sample"
        w0 = up_weights(oracle, method, 0.0)
        assert w0.sum() == 1.0 and np.count_nonzero(w0) == 1


def test_downscale_filters(oracle):
    L = oracle.lib()
    sup = C.c_float()
    expected_support = {0: 0.5, 1: 1.0, 2: 1.0, 3: 2.0, 4: 2.0, 5: 3.0}
    for m, s in expected_support.items():
        assert L.orc_downscale_filter(m, 0.0, C.byref(sup)) == 1.0
        assert sup.value == s
        assert L.orc_downscale_filter(m, s + 0.01, None) == 0.0
    assert L.orc_downscale_filter(1, 0.25, None) == 0.75
    assert abs(L.orc_downscale_filter(2, 0.5, None) - (np.sin(np.pi / 2) / (np.pi / 2) * 0.54)) < 1e-6
    assert abs(L.orc_downscale_filter(3, 1.5, None) - (-0.0625)) < 1e-7       # bicubic a=-0.5
    assert abs(L.orc_downscale_filter(4, 1.5, None) - (-0.1875)) < 1e-7       # a=-1.5


def test_lanczos3_quirk_taps(oracle):
    """Q1: the D3D11 shader samples Q1 at Q0's coordinate; the fixed flag restores base-1."""
    idx = (C.c_int32 * 128)()
    w = (C.c_float * 128)()
    n = oracle.lib().orc_axis_taps(1, 4, 0, 100, 200, 100, 0, 51, idx, w, None)   # odd output 51: base = 25
    assert n == 6 and list(idx[:6]) == [23, 23, 25, 26, 27, 28]
    n = oracle.lib().orc_axis_taps(1, 4, 0, 100, 200, 100, 1, 51, idx, w, None)
    assert list(idx[:6]) == [23, 24, 25, 26, 27, 28]
    n = oracle.lib().orc_axis_taps(1, 4, 0, 100, 200, 100, 0, 50, idx, w, None)   # even output 50: base = 24, t = .75
    assert list(idx[:6]) == [22, 22, 24, 25, 26, 27]
    assert np.allclose(np.array(w[:6]), PINS_T025[4][::-1], atol=2e-7)
    # clamp-to-edge of the whole texture
    n = oracle.lib().orc_axis_taps(1, 2, 0, 100, 200, 100, 0, 0, idx, w, None)    # Catmull, output 0: base = -1
    assert list(idx[:4]) == [0, 0, 0, 1]


# ---------------------------------------------------------------- (4) HDR tails
def test_pq_hlg_grey_pins(oracle):
    L = oracle.lib()
    tail = lambda v, trc: (lambda a: (L.orc_hdr_tail(a, trc, 9, 1, 80.0), a[0])[1])((C.c_float * 3)(v, v, v))
    assert abs(L.orc_st2084_to_linear(0.5, 80.0) - 0.737949) < 2e-6
    assert abs(tail(0.5, 15) * 255 - 149.8) < 0.05
    assert abs(tail(0.58, 15) * 255 - 195.45) < 0.05
    assert tail(0.75, 15) == 1.0 and tail(1.0, 15) == 1.0
    # hable(0) = D*E/(D*F) - E/F leaves an fp32 residual that pow(1/2.2) lifts to ~2.6e-4 (< half a 10-bit LSB)
    assert 0.0 <= tail(0.0, 15) < 4.9e-4
    assert abs(tail(0.5, 16) * 255 - 113.65) < 0.05
    assert abs(tail(0.75, 16) * 255 - 189.76) < 0.05
    assert abs(L.orc_hable(4.8) - 0.55875593) < 1e-6
    assert L.orc_luminance_scale(125) == 80.0
    # PQ round trip
    for x in (0.1, 0.3, 0.6, 0.9):
        assert abs(L.orc_linear_to_st2084(L.orc_st2084_to_linear(x, 1000.0), 1000.0) - x) < 2e-5
    # SDR BT.2020 gamma path: grey stays grey (gamut rows sum to 1), pow 2.2 then pow 1/2.2
    a = (C.c_float * 3)(0.5, 0.5, 0.5)
    L.orc_hdr_tail(a, 5, 9, 1, 80.0)
    assert max(abs(v - 0.5) for v in a) < 2e-6
    # no tail for BT.709 SDR
    a = (C.c_float * 3)(0.25, 1.5, -0.5)
    L.orc_hdr_tail(a, 5, 2, 1, 80.0)
    assert list(a) == [0.25, 1.5, -0.5]


def test_hlsl_constants_against_reference_text(oracle):
    base = "/root/reference/Shaders/convert"
    if not os.path.isdir(base):
        pytest.skip("reference not mounted")
    src = open(os.path.join(base, "st2084.hlsl")).read()
    consts = {}
    for name, expr in re.findall(r"static const float (ST2084_\w+)\s*=\s*([^;]+);", src):
        consts[name] = eval(expr.replace("f", ""))
    x = np.float64(0.6)
    p = x ** (1.0 / consts["ST2084_m2"])
    p = max(p - consts["ST2084_c1"], 0) / (consts["ST2084_c2"] - consts["ST2084_c3"] * p)
    ref = p ** (1.0 / consts["ST2084_m1"]) * 80.0
    assert abs(oracle.lib().orc_st2084_to_linear(0.6, 80.0) - ref) < 1e-4 * ref
    hl = open(os.path.join(base, "hlg.hlsl")).read()
    a, b, c = (float(re.search(r"B67_%s = ([0-9.]+)" % k, hl).group(1)) for k in "abc")
    v = (C.c_float * 3)(0.8, 0.8, 0.8)
    oracle.lib().orc_hlg_to_linear(v)
    lin = np.exp((0.8 - c) / a) + b
    assert abs(v[0] - lin * (2000.0 * lin) ** 0.2) < 1e-4 * v[0]


# ---------------------------------------------------------------- half rounding
def test_half_round_matches_ieee(oracle):
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(-2, 2, 4000), rng.uniform(-1e-4, 1e-4, 1000), [0.0, 1.0, 65504.0, 1e-8, 0.99975586]])
    for x in xs.astype(np.float32):
        assert oracle.lib().orc_half_round(float(x)) == float(np.float32(np.float16(x)))


# ---------------------------------------------------------------- (5) chroma siting
def _nv12_with_chroma_impulse(w, h, cx, cy):
    buf = np.full(w * h * 3 // 2, 128, dtype=np.uint8)
    buf[: w * h] = 126
    uv = buf[w * h:].reshape(h // 2, w)
    uv[cy, 2 * cx] = 228          # U impulse (+100)
    return buf


@pytest.mark.parametrize("chroma_loc,hw,vw", [
    (5, {8: 1.0, 9: 0.5, 7: 0.5}, {8: 0.75, 9: 0.75, 7: 0.25, 10: 0.25}),          # MPEG-2
    (7, {8: 1.0, 9: 0.5, 7: 0.5}, {8: 1.0, 9: 0.5, 7: 0.5}),                       # co-sited
    (1, {8: 0.75, 9: 0.75, 7: 0.25, 10: 0.25}, {8: 0.75, 9: 0.75, 7: 0.25, 10: 0.25}),   # MPEG-1
])
def test_bilinear_chroma_siting_weights(oracle, chroma_loc, hw, vw):
    """Impulse at chroma texel (4,4): the blue channel carries weight_x * weight_y of the impulse."""
    w, h = 32, 32
    frame = _nv12_with_chroma_impulse(w, h, 4, 4)
    p = oracle.default_params(cformat=oracle.CF["NV12"], width=w, height=h,
                              exfmt=oracle.make_extfmt(chroma=chroma_loc, nominal_range=1, matrix=1),
                              iTexFormat=16)
    out, fmt = oracle.convert_only(p, frame, w)
    assert fmt == 16
    base = out[0, 0, 2]
    cm = oracle.color_matrix(p)
    gain = cm[7] * (100.0 / 255.0)           # cm_b[1] * dU
    resp = (out[:, :, 2] - base) / gain
    for y in range(4, 14):
        for x in range(4, 14):
            expect = hw.get(x, 0.0) * vw.get(y, 0.0)
            assert abs(resp[y, x] - expect) < 2e-3, (x, y, resp[y, x], expect)


def test_nearest_and_catmull_chroma(oracle):
    w, h = 32, 32
    frame = _nv12_with_chroma_impulse(w, h, 4, 4)
    ex = oracle.make_extfmt(chroma=5, nominal_range=1, matrix=1)
    p = oracle.default_params(cformat=oracle.CF["NV12"], width=w, height=h, exfmt=ex, iTexFormat=16, iChromaScaling=0)
    out, _ = oracle.convert_only(p, frame, w)
    d = out[:, :, 2] - out[0, 0, 2]
    assert np.count_nonzero(np.abs(d) > 1e-3) == 4 and abs(d[8, 8] - d[9, 9]) < 1e-6      # 2x2 block
    p = oracle.default_params(cformat=oracle.CF["NV12"], width=w, height=h, exfmt=ex, iTexFormat=16, iChromaScaling=2)
    out, _ = oracle.convert_only(p, frame, w)
    d = (out[:, :, 2] - out[0, 0, 2])
    # Catmull-Rom 4:2:0: 4x4 chroma taps => support of 8x8 luma px around the impulse, weights sum to 1 per phase
    nz = np.argwhere(np.abs(d) > 1e-4)
    assert nz[:, 0].min() >= 4 and nz[:, 0].max() <= 13 and nz[:, 1].min() >= 4 and nz[:, 1].max() <= 13
    tot = d.sum() / d[8:10, 8:10].sum()
    assert 0.5 < tot < 2.0


# ---------------------------------------------------------------- whole path behaviour
def _grey_p010(w, h, code10):
    n, pitch = w * h * 3, w * 2
    buf = np.zeros(n // 2, dtype=np.uint16)
    buf[: w * h] = code10 << 6
    buf[w * h:] = 512 << 6
    return buf.view(np.uint8), pitch


def test_constant_frame_through_whole_path(oracle):
    """A flat grey frame stays flat through convert, Lanczos3 2x and dither; alpha is 255."""
    w, h = 48, 32
    frame, pitch = _grey_p010(w, h, 502)
    p = oracle.default_params(cformat=oracle.CF["P010"], width=w, height=h, iUpscaling=4,
                              window_w=2 * w, window_h=2 * h, video_rect=(0, 0, 2 * w, 2 * h))
    out = oracle.process(p, frame, pitch)
    assert out.shape == (2 * h, 2 * w, 4) and (out[..., 3] == 255).all()
    # (502-64)/876 = 0.5 -> 10-bit 511.5 -> rounds to 512/1023 -> *255 = 127.62: dither gives 127 or 128
    assert set(np.unique(out[..., :3]).tolist()) <= {127, 128}
    frac128 = (out[..., 1] == 128).mean()
    assert abs(frac128 - 0.62) < 0.05
    # identical thresholds for B, G, R of a pixel
    assert (out[..., 0] == out[..., 1]).all() and (out[..., 1] == out[..., 2]).all()
    # dither pattern has period 32 in window coordinates
    assert np.array_equal(out[:32, :32], out[32:64, 32:64])


def test_8bit_source_never_dithers_and_passthrough(oracle):
    """C1: NV12 -> BGRA8, no resize, internal BGRA8 => no final pass (DX11VideoProcessor.cpp:2896-2900)."""
    from videorenderer_amd import synth
    w, h = 64, 48
    frame, pitch = synth.make_frame(1, w, h, "structure", seed=1)
    p = oracle.default_params(cformat=1, width=w, height=h, window_w=w, window_h=h, video_rect=(0, 0, w, h))
    a = oracle.process(p, frame, pitch)
    p.bUseDither = 0
    b = oracle.process(p, frame, pitch)
    assert np.array_equal(a, b)
    conv, fmt = oracle.convert_only(p, frame, pitch)
    assert fmt == 8
    assert np.array_equal(a[..., 2], np.round(conv[..., 0] * 255).astype(np.uint8))     # R
    assert np.array_equal(a[..., 0], np.round(conv[..., 2] * 255).astype(np.uint8))     # B


def test_window_placement_and_clipping(oracle):
    from videorenderer_amd import synth
    w, h = 32, 16
    frame, pitch = synth.make_frame(2, w, h, "structure", seed=2)
    full = oracle.process(oracle.default_params(cformat=2, width=w, height=h, window_w=2 * w, window_h=2 * h,
                                                video_rect=(0, 0, 2 * w, 2 * h)), frame, pitch)
    # letterboxed inside a bigger window at an offset that is a multiple of 32 (dither phase preserved)
    p = oracle.default_params(cformat=2, width=w, height=h, window_w=160, window_h=128, video_rect=(32, 64, 32 + 2 * w, 64 + 2 * h))
    dst = np.full((128, 160, 4), 7, dtype=np.uint8)
    out = oracle.process(p, frame, pitch, dst=dst)
    assert np.array_equal(out[64:64 + 2 * h, 32:32 + 2 * w], full)
    mask = np.ones((128, 160), bool)
    mask[64:64 + 2 * h, 32:32 + 2 * w] = False
    assert (out[mask] == 7).all()
    # partially outside the window: visible part identical modulo the dither phase -> compare undithered
    p0 = oracle.default_params(cformat=2, width=w, height=h, window_w=2 * w, window_h=2 * h, video_rect=(0, 0, 2 * w, 2 * h), bUseDither=0)
    ref = oracle.process(p0, frame, pitch)
    p1 = oracle.default_params(cformat=2, width=w, height=h, window_w=40, window_h=20, video_rect=(-10, -6, -10 + 2 * w, -6 + 2 * h), bUseDither=0)
    clip = oracle.process(p1, frame, pitch)
    assert np.array_equal(clip, ref[6:26, 10:50])


def test_src_rect_crop(oracle):
    from videorenderer_amd import synth
    w, h = 64, 32
    frame, pitch = synth.make_frame(1, w, h, "structure", seed=3)
    whole = oracle.process(oracle.default_params(cformat=1, width=w, height=h, window_w=w, window_h=h, video_rect=(0, 0, w, h)), frame, pitch)
    p = oracle.default_params(cformat=1, width=w, height=h, src_rect=(16, 8, 48, 24), exfmt=oracle.make_extfmt(matrix=1),
                              window_w=32, window_h=16, video_rect=(0, 0, 32, 16))
    crop = oracle.process(p, frame, pitch)
    # (full frame at 64x32 defaults to BT.601 (SD); pin both to BT.709 for the comparison)
    whole709 = oracle.process(oracle.default_params(cformat=1, width=w, height=h, exfmt=oracle.make_extfmt(matrix=1),
                                                    window_w=w, window_h=h, video_rect=(0, 0, w, h)), frame, pitch)
    assert np.array_equal(crop, whole709[8:24, 16:48])
    assert not np.array_equal(whole, whole709)


def test_resizer_selection_rules(oracle):
    """ResizeShaderPass :3108-3126: downscale shader only below 50 % when bInterpolateAt50pct."""
    from videorenderer_amd import synth
    w, h = 64, 64
    frame, pitch = synth.make_frame(1, w, h, "noise", seed=4)
    def run(dw, dh, **kw):
        p = oracle.default_params(cformat=1, width=w, height=h, window_w=dw, window_h=dh, video_rect=(0, 0, dw, dh), **kw)
        return oracle.process(p, frame, pitch)
    # 64 -> 40 is within 50 %: the *upscale* shader is used, so changing iDownscaling must not matter
    assert np.array_equal(run(40, 40, iDownscaling=0), run(40, 40, iDownscaling=5))
    assert not np.array_equal(run(40, 40, iUpscaling=1), run(40, 40, iUpscaling=4))
    # 64 -> 24 is beyond: convolution shader, iUpscaling must not matter
    assert np.array_equal(run(24, 24, iUpscaling=1), run(24, 24, iUpscaling=4))
    assert not np.array_equal(run(24, 24, iDownscaling=0), run(24, 24, iDownscaling=5))
    # with bInterpolateAt50pct off every shrink uses the convolution
    assert not np.array_equal(run(40, 40, iDownscaling=0, bInterpolateAt50pct=0), run(40, 40, iDownscaling=5, bInterpolateAt50pct=0))
    # nearest upscale = pixel replication
    big = run(128, 128, iUpscaling=0)
    small = run(64, 64)
    assert np.array_equal(big[::2, ::2], small) and np.array_equal(big[1::2, 1::2], small)
    # one-axis resize
    assert run(128, 64, iUpscaling=2).shape == (64, 128, 4)
    # Jinc2m: a flat field stays flat (weights are normalised, anti-ringing clamps to the local range) and the result
    # differs from the separable Lanczos3
    jn = run(128, 128, iUpscaling=5)
    assert jn.shape == (128, 128, 4) and not np.array_equal(jn, run(128, 128, iUpscaling=4))
    flat, fp = synth.make_frame(1, w, h, "noise", seed=4)
    flat[:] = 128
    p = oracle.default_params(cformat=1, width=w, height=h, window_w=128, window_h=128, video_rect=(0, 0, 128, 128), iUpscaling=5)
    out = oracle.process(p, flat, fp)
    assert len(np.unique(out[..., :3].reshape(-1, 3), axis=0)) == 1


def test_rgb10a2_output(oracle):
    from videorenderer_amd import synth
    w, h = 32, 16
    frame, pitch = synth.make_frame(2, w, h, "structure", seed=5)
    p = oracle.default_params(cformat=2, width=w, height=h, window_w=w, window_h=h, video_rect=(0, 0, w, h), output_format=1)
    dst = np.zeros((h, w, 4), dtype=np.uint8)
    out = oracle.process(p, frame, pitch, dst=dst).view(np.uint32)[..., 0]
    assert ((out >> 30) == 3).all()
    conv, fmt = oracle.convert_only(p, frame, pitch)
    assert fmt == 10
    assert np.array_equal(out & 1023, np.round(conv[..., 0] * 1023).astype(np.uint32))


# ---------------------------------------------------------------- (6) golden hashes
def test_oracle_golden_hashes(oracle):
    from tests.golden.cases import GOLDEN_CASES, run_case
    with open(os.path.join(HERE, "golden", "oracle_hashes.json")) as f:
        want = json.load(f)
    for name in GOLDEN_CASES:
        out = run_case(oracle, name)
        assert hashlib.sha256(out.tobytes()).hexdigest() == want[name], name


# ---------------------------------------------------------------- cross-checks between independent oracle paths
def _conv(oracle, cf, w, h, seed, **kw):
    from videorenderer_amd import synth
    fr, pitch = synth.make_frame(cf, w, h, "noise", seed=seed)
    p = oracle.default_params(cformat=cf, width=w, height=h, window_w=w, window_h=h, video_rect=(0, 0, w, h), **kw)
    out = oracle.convert_only(p, fr, pitch)
    return np.asarray(out[0] if isinstance(out, tuple) else out)


def _proc(oracle, cf, w, h, dst, seed=5, kind="noise", **kw):
    from videorenderer_amd import synth
    fr, pitch = synth.make_frame(cf, w, h, kind, seed=seed)
    p = oracle.default_params(cformat=cf, width=w, height=h, window_w=dst[0], window_h=dst[1], video_rect=(0, 0, dst[0], dst[1]), **kw)
    return oracle.process(p, fr, pitch, dst=np.zeros((dst[1], dst[0], 4), np.uint8))


def test_packed_formats_equal_their_planar_twins(oracle):
    """The same samples in a packed container and in planes go through different oracle code (fetch_pixel branches, the
    v210 unpack) and must convert identically: the reference's packed 4:2:2 'linear' chroma is the planar bilinear one."""
    pairs = [(4, 18), (5, 18), (8, 22), (9, 23), (10, 22), (11, 19), (13, 25)]     # YUY2, UYVY, Y210, Y216, v210, AYUV, Y416
    for a, b in pairs:
        for cs in (1, 2):
            for (w, h) in ((48, 16), (46, 10)):
                assert np.array_equal(_conv(oracle, a, w, h, 7, iChromaScaling=cs), _conv(oracle, b, w, h, 7, iChromaScaling=cs)), (a, b, cs, w)
    # Y410 carries true 10-bit UNORM; the planar 10-bit twin is <<6 in 16 bits with a 10-bit matrix (quirk Q5): close, not equal
    d = np.abs(_conv(oracle, 12, 48, 16, 7).astype(np.float64) - _conv(oracle, 24, 48, 16, 7).astype(np.float64))
    assert 0 < d.max() < 3e-3


def test_interleaved_rgb_equals_planar_rgb(oracle):
    """XRGB32 (no convert draw: the source texture feeds the resize) vs GBRP8 (identity matrix through the convert draw)."""
    perm = lambda g: np.stack([g[..., 2], g[..., 0], g[..., 1], g[..., 3]], -1)    # synth puts (y,u,v) into (G,B,R) resp. (R,G,B)
    w, h = 48, 20
    for kw, dst in (({}, (w, h)), (dict(iUpscaling=2), (2 * w, 2 * h)), (dict(brightness=10.0, contrast=1.1), (w, h)),
                    (dict(src_rect=(8, 2, 40, 18)), (32, 16)), (dict(iDownscaling=2), (16, 8))):
        assert np.array_equal(_proc(oracle, 30, w, h, dst, **kw), perm(_proc(oracle, 26, w, h, dst, **kw))), kw
    a = _proc(oracle, 30, w, h, (w, h))
    for cf in (29, 31):
        assert np.array_equal(a, _proc(oracle, cf, w, h, (w, h)))                  # RGB24 / ARGB32 containers
    e = _proc(oracle, 33, w, h, (w, h))
    for cf in (34, 35, 36):
        assert np.array_equal(e, _proc(oracle, cf, w, h, (w, h)))                  # BGR48 / BGRA64 / b64a vs RGB48


def test_rotation_and_flip_are_pure_permutations_without_scaling(oracle):
    base = _proc(oracle, 1, 64, 40, (64, 40), kind="structure", seed=3)
    r = lambda **kw: _proc(oracle, 1, 64, 40, kw.pop("dst"), kind="structure", seed=3, **kw)
    assert np.array_equal(r(dst=(40, 64), rotation=90), np.rot90(base, -1))       # clockwise
    assert np.array_equal(r(dst=(64, 40), rotation=180), np.rot90(base, 2))
    assert np.array_equal(r(dst=(40, 64), rotation=270), np.rot90(base, 1))
    assert np.array_equal(r(dst=(64, 40), flip=1), base[:, ::-1])
    assert np.array_equal(r(dst=(40, 64), rotation=90, flip=1), np.rot90(base[:, ::-1], -1))
    # 180 degrees commutes with a symmetric separable scaler; 90 degrees with the same scaler on both axes is ONE draw
    # (resizerX == resizerY, DX11VideoProcessor.cpp:3131-3137), so only one axis is filtered — as written
    up = r(dst=(128, 80), iUpscaling=2)
    assert np.array_equal(r(dst=(128, 80), rotation=180, iUpscaling=2), np.rot90(up, 2))
    assert not np.array_equal(r(dst=(80, 128), rotation=90, iUpscaling=2), np.rot90(up, -1))


# ---------------------------------------------------------------- Dolby Vision
def _dovi_cb(oracle, md):
    od = oracle.fill_dovi(oracle.OrcDovi(), md)
    cb = (oracle.OrcDoviCb * 3)()
    has_mmr = C.c_int(0)
    oracle.lib().orc_dovi_pack_curves(C.byref(od), cb, C.byref(has_mmr))
    return od, cb, has_mmr.value


def _dovi_reference_float64(md, yuv):
    """Independent float64 evaluation of the Dolby Vision reshaping as the RPU defines it (piecewise polynomial /
    multivariate multiple regression), written from the metadata dict, not from the packed cbuffer."""
    den = float(1 << md["coef_log2_denom"])
    maxv = float((1 << md["bl_bit_depth"]) - 1)
    sig = np.clip(np.asarray(yuv, np.float64), 0, 1)
    out = np.zeros(3)
    for c, cv in enumerate(md["curves"]):
        piv = [p / maxv for p in cv["pivots"]]
        s = sig[c]
        k = 0
        while k + 1 < len(cv["pieces"]) and s >= np.float32(np.float32(1.0 / maxv) * np.float32(cv["pivots"][k + 1])):
            k += 1
        piece = cv["pieces"][k]
        if "poly" in piece:
            co = [x / den for x in piece["poly"]] + [0, 0]
            order = piece["order"]
            v = co[0] + (co[1] * s if order >= 1 else 0) + (co[2] * s * s if order >= 2 else 0)
        else:
            v = piece["constant"] / den
            x = np.array([sig[0], sig[1], sig[2], sig[0] * sig[1], sig[0] * sig[2], sig[1] * sig[2], sig[0] * sig[1] * sig[2]])
            for o, row in enumerate(piece["mmr"]):
                v += float(np.dot(np.array(row) / den, x ** (o + 1)))
        out[c] = min(max(v, 0.0), 1.0)
        del piv
    return out


@pytest.mark.parametrize("kind", ["poly", "mmr", "mixed"])
def test_dovi_reshape_against_float64_definition(oracle, kind):
    """ShaderDoviReshape[Poly] as restated (packed cbuffer, float4 tricks, early returns by order) must agree with the
    plain definition of the curves evaluated in float64, to fp32 rounding."""
    from videorenderer_amd import synth
    md = synth.dovi_metadata(kind)
    od, cb, has_mmr = _dovi_cb(oracle, md)
    rng = np.random.default_rng(7)
    pts = np.concatenate([rng.random((400, 3)), [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5]],
                          [[p / 1023.0, 0.3, 0.7] for p in (64, 160, 300, 460, 620, 780, 900)]]).astype(np.float32)
    worst = 0.0
    for p in pts:
        v = (C.c_float * 3)(*p)
        oracle.lib().orc_dovi_reshape(cb, has_mmr, v)
        want = _dovi_reference_float64(md, p)
        worst = max(worst, float(np.abs(np.array(v) - want).max()))
    assert worst < 2e-6, worst


def test_dovi_degenerates_to_hdr10(oracle):
    """A Dolby Vision stream whose curves are the identity, whose ycc_to_rgb is the BT.2020 limited-range matrix and
    whose rgb_to_lms undoes dovi_lms2rgb must give the HDR10 result of the same samples (the extra PQ -> linear -> PQ
    round trip costs at most 1 LSB): pins the order and direction of every Dolby Vision step against the HDR10 path."""
    from videorenderer_amd import synth
    w, h = 64, 32
    exf = (5 << 8) | (2 << 12) | (4 << 15) | (9 << 22) | (15 << 27)
    fr, pitch = synth.make_frame(20, w, h, "hdr", seed=3)
    base = oracle.default_params(cformat=20, width=w, height=h, exfmt=exf, window_w=w, window_h=h, video_rect=(0, 0, w, h))
    ref = oracle.process(base, fr, pitch, dst=np.zeros((h, w, 4), np.uint8))
    cm = oracle.color_matrix(base).astype(np.float64)
    m = cm[:9].reshape(3, 3)
    # c = -m . offset  =>  offset = -m^-1 . c
    off = -np.linalg.solve(m, cm[9:])
    lms2rgb = np.array([[3.06441879, -2.16597676, 0.10155818], [-0.65612108, 1.78554118, -0.12943749],
                        [0.01736321, -0.04725154, 1.03004253]])
    md = synth.dovi_metadata("identity")
    md.update(ycc_to_rgb_matrix=m.reshape(-1).tolist(), ycc_to_rgb_offset=off.tolist(),
              rgb_to_lms_matrix=np.linalg.inv(lms2rgb).reshape(-1).tolist())
    dv = oracle.default_params(cformat=20, width=w, height=h, exfmt=0, window_w=w, window_h=h, video_rect=(0, 0, w, h), dovi=md)
    got = oracle.process(dv, fr, pitch, dst=np.zeros((h, w, 4), np.uint8))
    d = np.abs(got.astype(int) - ref.astype(int))
    # super-white samples (a channel > 1 before the tail) are clamped per channel by the HDR10 shader but linearised
    # unclamped by the Dolby Vision one, where fp32 error of the (mathematically identity) matrix times their huge
    # linear value leaks into the other channels: compare where the converted colour is inside [0, 1]
    conv = oracle.convert_only(oracle.default_params(cformat=20, width=w, height=h, exfmt=exf & ~(0x1f << 27) & ~(0x1f << 22),
                                                     window_w=w, window_h=h, video_rect=(0, 0, w, h)), fr, pitch)
    conv = np.asarray(conv[0] if isinstance(conv, tuple) else conv)[..., :3]
    inside = np.all(conv < 1.0, axis=-1)       # (stored UNORM: 1.0 means clamped)
    assert inside.mean() > 0.5
    assert d[inside].max() <= 1 and (d[inside] == 0).mean() > 0.97, (d[inside].max(), (d[inside] == 0).mean())


def test_dovi_l1_nits_and_trims_known_answers(oracle):
    """PqToLinearNits on the 12-bit codes (ST 2084: code 2081/4095 ~ 100 nits, 3079/4095 ~ 1000 nits) and the
    level-3 offsets (-2048 bias)."""
    from videorenderer_amd import synth
    md = synth.dovi_metadata("identity")
    md.update(l1_present=1, l1_min_pq=0, l1_max_pq=3079, l1_avg_pq=2081)
    od = oracle.fill_dovi(oracle.OrcDovi(), md)
    out = (C.c_uint32 * 3)()
    assert oracle.lib().orc_dovi_l1_nits(C.byref(od), out) == 1
    assert out[0] == 0 and 995 <= out[1] <= 1005 and 98 <= out[2] <= 102
    md.update(l3_present=1, l3_min_pq_offset=2048, l3_max_pq_offset=2048 + (3696 - 3079), l3_avg_pq_offset=2048)
    od = oracle.fill_dovi(oracle.OrcDovi(), md)
    assert oracle.lib().orc_dovi_l1_nits(C.byref(od), out) == 1
    assert 3950 <= out[1] <= 4050 and 98 <= out[2] <= 102        # 3696/4095 ~ 4000 nits


def test_oracle_under_address_and_ub_sanitizers():
    """`make -C oracle sanitize`: the restatement compiled with -fsanitize=address,undefined (no OpenMP) and run over whole frames
    of awkward shapes — all 39 ColorFormat_t values, odd sizes, source rects, clipped windows, rotations, HDR tails.  Any
    report (the recipe compiles with -fno-sanitize-recover) fails the target."""
    import os, shutil, subprocess
    if not shutil.which("gcc") and not shutil.which("cc"):
        pytest.skip("no C compiler")
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    out = subprocess.run(["make", "-C", here, "sanitize"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "no sanitizer report" in out.stdout


# ---- host-side parameter maths pinned to the REFERENCE's own code (oracle/ref_hlsl/ref_hostmath.py -> tests/golden/hostmath_ref.json) ----
# SetShaderDoviCurves / SetShaderDoviCurvesPoly (DX11VideoProcessor.cpp:990-1141), SetHDR10ShaderParams (:907-923), the level 1 / 3 / 2
# block of CopySample + SetDolbyVisionDynamicParams (:2326-2469, :953-960), SpecifyExtendedFormat (Helper.cpp:1169-1211) and
# CopyFrameV210 (:709-748), extracted mechanically from the reference tree and compiled; their outputs are recorded so that the
# oracle's restatements AND the product's host code are held to them everywhere, and live wherever the library exists.
with open(os.path.join(HERE, "golden", "hostmath_ref.json")) as _f:
    HOSTMATH = json.load(_f)


def _orc_curve_words(oracle, st):
    """orc_dovi_pack_curves' result laid out as the reference's PS_DOVI_CURVE[3] (pivots_data[7] as float4.x, coeffs_data[8],
    mmr_data[48], params) and as PS_DOVI_POLY_CURVE[3] (pivots + coefficients, MMR pieces replaced by the identity {0, 1, 0, 0})."""
    cb3 = (oracle.OrcDoviCb * 3)()
    has_mmr = C.c_int(0)
    oracle.lib().orc_dovi_pack_curves(C.byref(st), cb3, C.byref(has_mmr))
    full, poly = [], []
    for c in range(3):
        cb = cb3[c]
        piv = np.zeros((7, 4), np.float32); piv[:, 0] = np.array(cb.pivots, np.float32)
        co = np.array(cb.coeffs, np.float32).reshape(8, 4)
        mm = np.array(cb.mmr, np.float32).reshape(48, 4)
        par = np.array([cb.methods, cb.mmr_single, cb.min_order, cb.max_order], np.uint32)
        full.append(np.concatenate([piv.view(np.uint32).ravel(), co.view(np.uint32).ravel(), mm.view(np.uint32).ravel(), par]))
        cop = co.copy()
        for i in range(8):
            if cop[i, 3] != 0:                      # an MMR piece: "not supported, leave as is" (:1019-1025)
                cop[i] = (0.0, 1.0, 0.0, 0.0)
        poly.append(np.concatenate([piv.view(np.uint32).ravel(), cop.view(np.uint32).ravel()]))
    return np.concatenate(full), np.concatenate(poly), has_mmr.value


@pytest.mark.parametrize("name", sorted(HOSTMATH["dovi"]))
def test_dovi_host_maths_against_reference_code(oracle, mpcvr, name):
    """Curve packing, level-2 trim selection / interpolation, level-1 (+3) nits: oracle restatement and product (vp_dovi.cpp through
    mpcvr_plan_dovi) against what the reference's own functions return for the same metadata — bit for bit."""
    from videorenderer_amd import api, synth
    rec = HOSTMATH["dovi"][name]
    md = synth.dovi_metadata(**rec["kw"])
    st = oracle.fill_dovi(oracle.OrcDovi(), md)
    full, poly, has_mmr = _orc_curve_words(oracle, st)
    want_full, want_poly = np.array(rec["curves_words"], np.uint32), np.array(rec["curves_poly_words"], np.uint32)
    # coefficient slots behind the curve's last piece are never read (the pivots stop the search): compare what the shader can reach
    assert full.shape == want_full.shape and poly.shape == want_poly.shape
    assert np.array_equal(full, want_full), f"{name}: orc_dovi_pack_curves differs from SetShaderDoviCurves in {int((full != want_full).sum())} words"
    assert np.array_equal(poly, want_poly), f"{name}: polynomial packing differs from SetShaderDoviCurvesPoly"
    for disp, lv in rec["levels"].items():
        k5 = (C.c_float * 5)()
        en = oracle.lib().orc_dovi_l2_constants(C.byref(st), int(disp), k5)
        l1 = (C.c_uint32 * 3)()
        l1p = oracle.lib().orc_dovi_l1_nits(C.byref(st), l1)
        assert en == lv["enabled"] and l1p == lv["l1_present"], (name, disp)
        if en:
            assert [int(np.float32(x).view(np.uint32)) for x in k5] == lv["k5_bits"], (name, disp, list(k5), lv["k5"])
        if l1p:
            assert [int(x) for x in l1] == lv["l1"], (name, disp)
        pd = api.plan_dovi(md, int(disp))
        assert pd["l2_enabled"] == lv["enabled"] and pd["l1_present"] == lv["l1_present"]
        if lv["enabled"]:
            assert [int(x) for x in pd["l2k"].view(np.uint32)] == lv["k5_bits"], (name, disp, "product")
        if lv["l1_present"]:
            assert [int(x) for x in pd["l1_nits"]] == lv["l1"], (name, disp, "product")
    # the product's cbuffer: 3 x (pivots[7], coeffs[8][4], mmr[48][4], params as floats)
    pd = api.plan_dovi(md, 1000)
    for c in range(3):
        row = pd["cb"][c]
        ref = want_full[c * 256:(c + 1) * 256]
        assert np.array_equal(row[:7].view(np.uint32), ref[0:28:4])
        assert np.array_equal(row[7:39].view(np.uint32), ref[28:60]) and np.array_equal(row[39:231].view(np.uint32), ref[60:252])
        assert [int(v) for v in row[231:235]] == [int(v) for v in ref[252:256]]


def test_hdr10_params_against_reference_code(oracle, mpcvr):
    from videorenderer_amd import api
    for rec in HOSTMATH["hdr10"]:
        a = rec["args"]
        out = (C.c_uint32 * 6)()
        oracle.lib().orc_hdr10_params(*a[:5], int(a[5]), out)
        assert list(out) == rec["words"], (a, list(out))
        out2 = (C.c_uint32 * 6)()
        assert api.load_library().mpcvr_plan_hdr10_params(*a[:5], int(a[5]), out2) == 0
        assert list(out2) == rec["words"], (a, "product")


def test_specify_extended_format_against_reference_code(oracle, mpcvr):
    from videorenderer_amd import api
    for rec in HOSTMATH["extfmt"]:
        got = oracle.lib().orc_specify_extfmt(rec["exfmt"], rec["cformat"], rec["w"], rec["h"])
        assert got == rec["out"], (rec, hex(got))
        _, ex = api.plan_color_matrix(rec["cformat"], rec["w"], rec["h"], extfmt=rec["exfmt"])
        assert ex == rec["out"], (rec, hex(ex), "product")


def test_copy_frame_v210_against_reference_code(oracle):
    import hashlib
    from tests.golden.make_hostmath_golden import v210_sample
    for rec in HOSTMATH["v210"]:
        src, pitch = v210_sample(rec["width"], rec["lines"])
        assert pitch == rec["pitch"]
        tp = oracle.lib().orc_v210_tex_pitch(rec["width"])
        assert tp == rec["tex_pitch"]
        dst = np.zeros(tp * rec["lines"] + 16, np.uint8)
        oracle.lib().orc_repack_v210(C.c_int(rec["lines"]), C.c_void_p(dst.ctypes.data), C.c_int(tp), C.c_void_p(src.ctypes.data), C.c_int(pitch))
        assert hashlib.sha256(dst[:tp * rec["lines"]].tobytes()).hexdigest() == rec["sha256"], rec


def test_rgb_upload_copies_against_reference_code(oracle):
    """CopyPlaneAsIs / CopyFrameRGB24 / R210 / RGB48 / BGR48 / BGRA64 / B64A (Helper.cpp:414-787, the uploads of the interleaved RGB
    formats): the oracle's orc_repack_rgb against the reference functions' recorded output — ragged widths (the 4-pixel loops'
    remainders), bottom-up DIBs."""
    from tests.golden.make_hostmath_golden import rgb_copy
    L = oracle.lib()
    L.orc_repack_rgb.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    fn = lambda k, n, d, dp, s_, sp: L.orc_repack_rgb(k, n, d, dp, s_, sp)
    for rec in HOSTMATH["rgbcopy"]:
        got = rgb_copy(fn, rec["kind"], rec["pack"], rec["tbpp"], rec["width"], rec["lines"], rec["bottom_up"])
        assert got == rec["sha256"], rec


def test_hostmath_fixture_is_what_the_reference_code_returns_live():
    """Where oracle/_ref/libref_hostmath.so exists (or can be built from the mounted reference): the recorded fixture is regenerated
    in memory and must be identical — so the pins above are the reference code's, not a stale file."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle", "ref_hlsl"))
    import ref_hostmath
    if ref_hostmath.lib() is None:
        pytest.skip("oracle/_ref/libref_hostmath.so not built (no /root/reference here)")
    from tests.golden import make_hostmath_golden as G
    L = ref_hostmath.lib()
    for name, kw in G.DOVI_CASES.items():
        st = G.dovi_struct(kw)
        assert [int(x) for x in G.ref_curves(L, st, False)] == HOSTMATH["dovi"][name]["curves_words"]
        assert [int(x) for x in G.ref_curves(L, st, True)] == HOSTMATH["dovi"][name]["curves_poly_words"]
        for d in G.DISPLAYS:
            assert G.ref_levels(L, st, d) == HOSTMATH["dovi"][name]["levels"][str(d)]
    for rec in HOSTMATH["hdr10"]:
        out = (C.c_uint32 * 6)()
        L.ref_hdr10_params(*rec["args"][:5], int(rec["args"][5]), out)
        assert list(out) == rec["words"]
    ref_fn = lambda k, n, d, dp, s_, sp: L.ref_copy_frame_rgb(k, n, d, dp, s_, sp)
    for rec in HOSTMATH["rgbcopy"]:
        assert G.rgb_copy(ref_fn, rec["kind"], rec["pack"], rec["tbpp"], rec["width"], rec["lines"], rec["bottom_up"]) == rec["sha256"]


# ------------------------------------------------------------------------------------------------------------------------------
# "Bit-identical to the reference's shader text" is a statement UNDER THE MODELLED INTERPOLATOR (oracle/ref_hlsl/ref_draw.h, TexCenter,
# axis_center: TEXCOORD interpolated exactly and rounded once to fp32).  A real rasteriser works in fixed-point barycentrics and lands
# within an ulp or so of that.  This test moves every draw's interpolated coordinate by one ulp either way and shows what hangs on it:
# nothing but a rare single code for the interpolation and convolution filters; whole texel rows at 3:1, where a third of the outputs sit
# EXACTLY on a texel centre and floor(pos) decides which rows the taps read; and nearly every output of the box filter at integer
# ratios, whose `x < 0.5` support edge is hit exactly.  Those cases are pinned to the model, not to hardware.
# ------------------------------------------------------------------------------------------------------------------------------
def _tex_probe(oracle, c, bias):
    from tests.golden.cases import case_frame, oracle_params
    frame, pitch = case_frame(c)
    p = oracle_params(oracle, c)
    base = oracle.process(p, frame, pitch).astype(np.int16)[..., :3]
    moved = oracle.process_with_tex_bias(p, frame, pitch, bias).astype(np.int16)[..., :3]
    return np.abs(moved - base)


def test_texcoord_one_ulp_off_moves_only_the_ill_conditioned_channels(oracle):
    from tests.golden.cases import GOLDEN_CASES, M709, ext
    well = {
        "lanczos3_2x": dict(cformat=2, w=248, h=40, kind="noise", seed=7, dst=(496, 80), exfmt=ext(matrix=M709), iUpscaling=4),
        "lanczos3_1p5x": dict(GOLDEN_CASES["up_1p5x_lanczos3"], kind="noise"),
        "mitchell_4_3": dict(cformat=2, w=96, h=72, kind="noise", seed=21, dst=(128, 96), iUpscaling=1),
        "hamming_down_3x": dict(GOLDEN_CASES["down_hamming_3x"], kind="noise"),
        "nearest_2x": dict(GOLDEN_CASES["nearest_2x"]),
    }
    for name, c in well.items():
        for bias in (-1, 1):
            d = _tex_probe(oracle, c, bias)
            assert d.max() <= 1 and (d > 0).mean() < 0.005, (name, bias, int(d.max()), float((d > 0).mean()))
    # 3:1: outputs 3 n + 1 sit on texel centre n; one ulp below, floor(Tex * wh - .5) is n - 1 and the whole tap row moves by a texel
    c = dict(cformat=2, w=40, h=24, kind="noise", seed=11, dst=(120, 72), iUpscaling=4)
    seen = 0
    for bias in (-1, 1):
        d = _tex_probe(oracle, c, bias)
        ys, xs = np.nonzero((d > 1).any(axis=2))
        assert ((ys % 3 == 1) | (xs % 3 == 1)).all(), (bias, sorted(set(zip(ys % 3, xs % 3))))
        seen += len(ys)
    assert seen > 50          # ... and they do move (noise content: by whole texels' worth)
    # the box filter at an integer ratio: every tap distance is exactly k + 0.5, the support test `x < 0.5` flips with the last ulp
    c = dict(cformat=1, w=128, h=96, kind="noise", seed=15, dst=(32, 24), iDownscaling=0, bInterpolateAt50pct=0)
    shares = sorted(float((_tex_probe(oracle, c, bias) > 0).mean()) for bias in (-1, 1))
    assert shares[0] < 0.05 and shares[1] > 0.5, shares


def test_dovi_colour_matrix_with_procamp_rounds_every_step(oracle):
    """SetShaderConvertColorParams, Dolby Vision branch (DX11VideoProcessor.cpp:817-834): `cmatrix.c[i] -= cmatrix.m[i][j] * offset[j]` is
    float -= float * double — every step rounded to float.  gcc 11 -O3 vectorised the oracle's plain loop and dropped the intermediate
    roundings of rows 0 and 1 (one ulp; found by the round-6 fuzz, case 6375, as a plain-tier mismatch on 2 channels of 54 k pixels): the
    oracle's constants against an independent numpy evaluation of the expression as written."""
    from videorenderer_amd import synth
    from tests.golden.cases import oracle_params
    f32 = np.float32
    rng = np.random.default_rng(6375)
    for k in range(40):
        pa = (float(rng.uniform(-20, 20)), float(rng.uniform(0.8, 1.2)), float(rng.uniform(-30, 30)), float(rng.uniform(0.5, 1.5)))
        kind = ("mmr", "poly", "mixed")[k % 3]
        c = dict(cformat=2, w=64, h=32, kind="noise", seed=1, dst=(64, 32), exfmt=0x7a420100, procamp=pa, dovi=dict(kind=kind, l2=()))
        md = synth.dovi_metadata(kind)
        got = oracle.color_matrix(oracle_params(oracle, c))
        b, ct = f32(pa[0]) / f32(255), f32(pa[1])
        for i in range(3):
            m = [f32(f32(md["ycc_to_rgb_matrix"][3 * i + j]) * ct) for j in range(3)]
            acc = b
            for j in range(3):
                acc = f32(float(acc) - float(m[j]) * float(md["ycc_to_rgb_offset"][j]))
            assert [float(v) for v in got[3 * i:3 * i + 3]] == [float(v) for v in m], (k, i)
            assert float(got[9 + i]) == float(acc), (k, i, float(got[9 + i]).hex(), float(acc).hex())


def test_tonemap_input_probe_is_off_by_default_and_moves_only_what_it_names(oracle):
    """orc_set_tonemap_input_bias (the witness for plans with an HDR10 tone-mapping operator, compare_behind_tail operator_input): off = the
    pinned output; one code up on the blue channel of the operator's input moves the oracle's answer (operator 6 near black: by up to 7
    ten-bit codes on soak case 2367's frame) and the probe resets; a plan without an operator is not touched by it."""
    from tests.golden.cases import case_frame, oracle_params
    from tests.test_parity_gpu import FUZZ_2367, _codes10, BG
    c = dict(FUZZ_2367, w=96, h=64, dst=(192, 128))
    frame, pitch = case_frame(c)
    p = oracle_params(oracle, c)
    bg = lambda: np.full((p.window_h, p.window_w, 4), BG, dtype=np.uint8)
    want = _codes10(oracle.process(p, frame, pitch, dst=bg()))
    up_b = _codes10(oracle.process_with_tonemap_input_bias(p, frame, pitch, 1, channel=2, dst=bg()))
    up_all = _codes10(oracle.process_with_tonemap_input_bias(p, frame, pitch, 1, dst=bg()))
    noise = _codes10(oracle.process_with_tonemap_input_bias(p, frame, pitch, 1, seed=5, dst=bg()))
    assert np.array_equal(_codes10(oracle.process(p, frame, pitch, dst=bg())), want), "the probe did not reset"
    assert (up_b != want).any() and (up_all != want).any() and (noise != want).any()
    assert int(np.abs(up_all - want).max()) >= 2, "a local operator maps a code of its input with its own slope"
    assert not np.array_equal(up_b, up_all) and not np.array_equal(noise, up_all)
    c0 = {k: v for k, v in c.items() if k not in ("hdr_tonemap", "hdr_display", "hdr_meta")}
    p0 = oracle_params(oracle, c0)
    a = oracle.process(p0, frame, pitch, dst=bg())
    b = oracle.process_with_tonemap_input_bias(p0, frame, pitch, 1, dst=bg())
    assert np.array_equal(a, b)


def test_convert_output_and_log2_probes_are_off_by_default(oracle):
    """orc_set_convert_output_bias / orc_set_pow_log2_noise (witnesses of the fused tiers' bars): off = the pinned output, on = something else,
    and both reset; the convert-output probe leaves an fp16 internal format alone."""
    from tests.golden.cases import GOLDEN_CASES, case_frame, oracle_params
    from tests.test_parity_gpu import BG
    c = dict(GOLDEN_CASES["c3hdr_p010_pq_lanczos3_2x"])
    frame, pitch = case_frame(c)
    p = oracle_params(oracle, c)
    bg = lambda: np.full((p.window_h, p.window_w, 4), BG, dtype=np.uint8)
    want = oracle.process(p, frame, pitch, dst=bg())
    a = oracle.process_with_convert_output_bias(p, frame, pitch, 1, dst=bg())
    b = oracle.process_with_convert_output_bias(p, frame, pitch, 1, seed=3, dst=bg())
    l = oracle.process_with_log2_noise(p, frame, pitch, 4, seed=2, dst=bg())
    assert np.array_equal(oracle.process(p, frame, pitch, dst=bg()), want), "a probe did not reset"
    assert not np.array_equal(a, want) and not np.array_equal(b, want) and not np.array_equal(a, b)
    assert int(np.abs(l[..., :3].astype(int) - want[..., :3].astype(int)).max()) <= 2        # a smooth tail: an ulp of log2 is no code
    c16 = dict(c, iTexFormat=16)
    p16 = oracle_params(oracle, c16)
    assert np.array_equal(oracle.process(p16, frame, pitch, dst=bg()), oracle.process_with_convert_output_bias(p16, frame, pitch, 1, dst=bg()))
