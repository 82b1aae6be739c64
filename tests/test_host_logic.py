"""Host-side logic of the product (videorenderer_amd/csrc/vp_plan.cpp) against the oracle and the real
reference fixtures — no GPU involved: these are the parameter computations the reference also does on
the CPU (SetShaderConvertColorParams, SpecifyExtendedFormat, resize constants, pass selection)."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

from tests.golden.cases import GOLDEN_CASES, SETTING_KEYS, case_geometry, oracle_params

HERE = os.path.dirname(os.path.abspath(__file__))


def bits(v):
    return struct.unpack("<I", struct.pack("<f", v))[0]


def test_color_matrix_matches_oracle_bit_exact(mpcvr, oracle):
    rng = np.random.default_rng(11)
    from videorenderer_amd import api
    for cf in (1, 2, 3, 6, 14, 16, 17, 19, 20, 21, 22, 24, 25, 4, 5, 8, 9, 10, 11, 12, 13, 26, 27, 28, 37, 38, 39,
               29, 30, 32, 33, 36):
        for it in range(24):
            ex = oracle.make_extfmt(chroma=int(rng.choice([0, 1, 5, 7])), nominal_range=int(rng.choice([0, 1, 2])),
                                    matrix=int(rng.choice([0, 1, 2, 3, 4, 7])), primaries=int(rng.choice([0, 2, 9])),
                                    transfer=int(rng.choice([0, 4, 5, 15, 16])))
            w, h = int(rng.choice([640, 1024, 1920, 3840])), int(rng.choice([480, 576, 1080, 2160]))
            b, ct, hu, s = rng.uniform(-100, 100), rng.uniform(0, 2), rng.uniform(-180, 180), rng.uniform(0, 2)
            if rng.random() < 0.3:
                b, ct, hu, s = 0.0, 1.0, 0.0, 1.0
            if cf >= 37 and it % 2:
                ex = 0               # gray sources usually arrive without an extended format => MP_CSP_RGB path
            got, ex_out = api.plan_color_matrix(cf, w, h, ex, b, ct, hu, s)
            p = oracle.default_params(cformat=cf, width=w, height=h, exfmt=ex, brightness=b, contrast=ct, hue=hu, saturation=s)
            want = oracle.color_matrix(p)
            assert [bits(v) for v in got] == [bits(float(v)) for v in want], (cf, hex(ex), w, h)
            assert ex_out == oracle.lib().orc_specify_extfmt(ex, cf, w, h)


def test_color_matrix_matches_reference_fixtures(mpcvr):
    """The committed outputs of the REAL csputils.cpp (tests/golden/csputils_ref.json) through the product path."""
    from videorenderer_amd import api
    with open(os.path.join(HERE, "golden", "csputils_ref.json")) as f:
        g = json.load(f)
    f32 = lambda b: struct.unpack("<f", struct.pack("<I", b))[0]
    # product inputs are DXVA2 codes: matrix code -> mp_csp, range -> levels, CDepth via the format
    space_to_matrix = {1: 2, 2: 1, 3: 3, 4: 4, 8: 7}
    fmt_for_bits = {8: 1, 10: 20, 16: 2}
    checked = 0
    for case in g["csp_matrix"]:
        if case["space"] not in space_to_matrix or case["levels"] == 0:
            continue
        gray = case.get("gray", 0)
        # only the default ProcAmp can be expressed exactly in DXVA2 units (brightness*255 etc. round-trips differ)
        if (f32(case["brightness"]), f32(case["contrast"]), f32(case["hue"]), f32(case["saturation"])) != (0.0, 1.0, 0.0, 1.0):
            continue
        ex = api.make_extfmt(nominal_range=2 if case["levels"] == 1 else 1, matrix=space_to_matrix[case["space"]])
        if gray:
            # CS_GRAY: csp_params.gray, then cm_g.x <- cm_g.y, cm_b.x <- cm_b.z (DX11VideoProcessor.cpp:868-873)
            got, _ = api.plan_color_matrix({8: 37, 10: 38, 16: 39}[case["bits"]], 1920, 1080, ex)
            want = list(case["m"])
            want[3], want[4] = want[4], bits(0.0)
            want[6], want[8] = want[8], bits(0.0)
            assert [bits(v) for v in got[:9]] == want and [bits(v) for v in got[9:]] == case["c"], case
        else:
            got, _ = api.plan_color_matrix(fmt_for_bits[case["bits"]], 1920, 1080, ex)
            assert [bits(v) for v in got[:9]] == case["m"] and [bits(v) for v in got[9:]] == case["c"], case
        checked += 1
    assert checked >= 40
    assert [bits(v) for v in api.plan_gamut_2020_to_709()] == g["gamut_2020_to_709"]


def test_upscale_weights_match_oracle(mpcvr, oracle):
    from videorenderer_amd import api
    w = (C.c_float * 6)()
    for method in (1, 2, 3, 4):
        for t in list(np.linspace(0, 1, 41, dtype=np.float32)[:-1]) + [0.25, 0.75]:
            n = oracle.lib().orc_upscale_weights(method, float(t), w)
            got = api.plan_upscale_weights(method, float(t))
            assert len(got) == n and [bits(v) for v in got] == [bits(v) for v in w[:n]]


@pytest.mark.parametrize("kind,method,src_l,src_len,n_out,tex_len,flags", [
    (1, 4, 0, 128, 256, 128, 0), (1, 4, 0, 128, 256, 128, 1), (1, 2, 0, 100, 150, 100, 0), (1, 1, 0, 64, 50, 64, 0),
    (1, 3, 0, 33, 100, 33, 0), (2, 2, 0, 192, 64, 192, 0), (2, 5, 0, 160, 64, 160, 0), (2, 0, 0, 128, 32, 128, 0),
    (2, 1, 0, 128, 40, 128, 0), (2, 3, 0, 200, 31, 200, 0), (2, 4, 0, 96, 36, 96, 0), (0, 0, 0, 32, 64, 32, 0),
    (0, 0, 0, 96, 30, 96, 0), (1, 4, 0, 3840, 7680, 3840, 0), (2, 5, 0, 3840, 1280, 3840, 0),
])
def test_axis_taps_match_oracle(mpcvr, oracle, kind, method, src_l, src_len, n_out, tex_len, flags):
    from videorenderer_amd import api
    I, W, WS = api.plan_axis_taps(kind, method, src_l, src_len, n_out, tex_len, flags)
    idx = (C.c_int32 * 128)()
    w = (C.c_float * 128)()
    ws = C.c_float()
    step = max(1, n_out // 257)
    for i in list(range(0, n_out, step)) + [n_out - 1]:
        n = oracle.lib().orc_axis_taps(kind, method, src_l, src_len, n_out, tex_len, flags, i, idx, w, C.byref(ws))
        assert n > 0
        assert I[i][:n] == list(idx[:n]), (i, I[i], list(idx[:n]))
        assert [bits(v) for v in W[i][:n]] == [bits(v) for v in w[:n]], i
        assert all(v == 0.0 for v in W[i][n:])          # padding taps carry zero weight
        if WS is not None:
            assert bits(WS[i]) == bits(ws.value)


def test_frame_layout(mpcvr, oracle):
    from videorenderer_amd import api
    for cf in oracle.CF.values():
        for (w, h) in ((1920, 1080), (62, 32), (3840, 2160)):
            if cf in (1, 2, 3, 14, 17, 20, 21) and (w % 2 or h % 2):
                continue
            if cf == 10 and w == 62:
                continue
            assert api.plan_frame_layout(cf, w, h) == oracle.frame_bytes(cf, w, h)
    assert api.plan_frame_layout(1, 1920, 1080) == (3110400, 1920)
    assert api.plan_frame_layout(2, 3840, 2160) == (24883200, 7680)
    assert api.plan_frame_layout(20, 1920, 1080) == (6220800, 3840)
    assert api.plan_frame_layout(4, 1920, 1080) == (1920 * 2 * 1080, 3840)       # YUY2
    assert api.plan_frame_layout(10, 1920, 1080) == (5120 * 1080, 5120)          # v210: ALIGN((W+5)/6*16, 128)
    assert api.plan_frame_layout(10, 1280, 720) == (3456 * 720, 3456)
    assert api.plan_frame_layout(37, 62, 32) == (64 * 32, 64)                    # Y8: ALIGN(W, 4)
    assert api.plan_frame_layout(26, 64, 32) == (64 * 32 * 3, 64)                # GBRP8: three full planes
    assert api.plan_frame_layout(29, 46, 20) == (140 * 20, 140)                  # RGB24: ALIGN(3W, 4)
    assert api.plan_frame_layout(34, 45, 8) == (272 * 8, 272)                    # BGR48: ALIGN(6W, 4)
    with pytest.raises(api.MpcvrError):
        api.plan_frame_layout(40, 64, 64)         # no such ColorFormat_t


def test_pq_lut(mpcvr, oracle):
    from videorenderer_amd import api
    lut = np.array(api.plan_pq_lut(80.0), dtype=np.float32)
    L = oracle.lib()
    div = L.orc_hable(4.8)
    assert lut.size == 4096
    for i in (0, 1, 100, 2047, 3071, 4094, 4095):
        want = L.orc_hable(L.orc_st2084_to_linear(np.float32(i) / np.float32(4095), 80.0)) / div
        assert abs(lut[i] - want) <= 2e-6 * max(1.0, abs(want))
    assert np.all(np.diff(lut) >= 0)


def test_pass_plan_matches_reference_rules(mpcvr):
    from videorenderer_amd import api
    s = api.default_settings()
    d = lambda cf, w, h, vr, ww, wh, st=s: api.plan_describe(st, cf, w, h, vr, ww, wh)
    # C1: 8-bit source, no resize -> BGRA8 internal, never dithers (DX11VideoProcessor.cpp:2896-2900)
    assert d(1, 1920, 1080, (0, 0, 1920, 1080), 1920, 1080) == "direct:convert+copy;internal=8;swap=8;final=0"
    assert d(1, 1920, 1080, (0, 0, 1920, 1080), 1920, 1080, s.copy(flags=api.FLAG_NO_FUSED)) == "passes:convert,copy;internal=8;swap=8;final=0"
    # same size, 10-bit source: the final pass (dither) rides in the convert kernel's epilogue too
    assert d(2, 3840, 2160, (0, 0, 3840, 2160), 3840, 2160) == "direct:convert+final;internal=10;swap=8;final=1"
    # C2/C3: exact 2x of a 10/16-bit 4:2:0 source -> fused kernel, RGB10A2 internal, final pass on
    assert d(20, 1920, 1080, (0, 0, 3840, 2160), 3840, 2160).startswith("fused_up2x;internal=10;swap=8;final=1")
    assert d(2, 3840, 2160, (0, 0, 7680, 4320), 7680, 4320, s.copy(iUpscaling=4)).startswith("fused_up2x")
    assert d(2, 3840, 2160, (0, 0, 7680, 4320), 7680, 4320, s.copy(iUpscaling=4, flags=api.FLAG_NO_FUSED)) == \
        "passes:convert,resizeX,resizeY+final;internal=10;swap=8;final=1"
    # not exactly 2x / partly outside the window / Catmull chroma -> pass-per-kernel path
    assert d(2, 1920, 1080, (0, 0, 2880, 1620), 2880, 1620).startswith("passes:convert,resizeX,resizeY+final")
    assert d(2, 1920, 1080, (-10, 0, 3830, 2160), 3840, 2160).startswith("passes:")
    assert d(2, 1920, 1080, (0, 0, 3840, 2160), 3840, 2160, s.copy(iChromaScaling=2)).startswith("passes:")
    # one-axis and nearest
    assert d(2, 1920, 1080, (0, 0, 3840, 1080), 3840, 1080).startswith("passes:convert,resizeX+final")
    assert d(2, 1920, 1080, (0, 0, 1920, 540), 1920, 540).startswith("passes:convert,resizeY+final")
    assert d(1, 640, 360, (0, 0, 1280, 720), 1280, 720, s.copy(iUpscaling=0)).startswith("passes:convert,resizeX;")
    # 10-bit swap chain: no final pass with RGB10A2 internal, final pass with fp16 internal
    assert d(2, 64, 64, (0, 0, 64, 64), 64, 64, s.copy(output_format=1)).endswith("final=0")
    assert d(2, 64, 64, (0, 0, 64, 64), 64, 64, s.copy(output_format=1, iTexFormat=16)).endswith("internal=16;swap=10;final=1")
    assert d(2, 64, 64, (0, 0, 64, 64), 64, 64, s.copy(bUseDither=0)).endswith("final=0")
    # Jinc2m: one 2-D shader for both axes => a single draw (m_pShaderUpscaleY = m_pShaderUpscaleX, :2921,3131-3137)
    # (at exactly 2x the fused Jinc2m kernel replaces convert + that draw + the final pass since round 5; the passes show with the fused tier off)
    assert d(2, 64, 64, (0, 0, 128, 128), 128, 128, s.copy(iUpscaling=5)).startswith("fused_jinc2x")
    assert d(2, 64, 64, (0, 0, 128, 128), 128, 128, s.copy(iUpscaling=5, flags=api.FLAG_NO_FUSED)).startswith("passes:convert,resizeX+final")
    assert d(2, 64, 64, (0, 0, 192, 192), 192, 192, s.copy(iUpscaling=5)).startswith("passes:convert,resizeX+final")
    assert d(2, 64, 64, (0, 0, 128, 20), 128, 20, s.copy(iUpscaling=5)).startswith("passes:convert,resizeX,resizeY+final")


def test_exact_2x_candidates_by_source_layout(mpcvr):
    """DecidePlan's fused-2x candidate rule = the layouts and chroma settings the block convert inside the fused kernels serves
    (BlockConvertLayout): every YUV / gray / three-plane RGB layout, nearest or bilinear chroma where the layout has a filter at all;
    Catmull-Rom chroma and interleaved RGB (no convert draw to fuse) stay on the per-draw plan."""
    from videorenderer_amd import api
    s = api.default_settings(iUpscaling=4)
    up2x = lambda cf, **kw: api.plan_describe(s.copy(**kw), cf, 640, 360, (0, 0, 1280, 720), 1280, 720).startswith("fused_up2x")
    yuv_gray_gbrp = [cf for cf in range(1, 40) if cf not in range(29, 37)]
    for cf in yuv_gray_gbrp:
        assert up2x(cf), cf
        assert up2x(cf, iChromaScaling=0), cf                      # nearest: a rule of the block code, or no chroma filter to choose
    for cf in range(29, 37):                                       # RGB24 ... b64a
        assert not up2x(cf), cf
    subsampled = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 14, 15, 17, 18, 20, 21, 22, 23)       # 4:2:0 and 4:2:2, planar and packed
    for cf in yuv_gray_gbrp:
        assert up2x(cf, iChromaScaling=2) == (cf not in subsampled), cf              # Catmull-Rom only matters where chroma is subsampled


def test_spline36_extension_weights(mpcvr, oracle):
    """MPCVR_UPSCALE_Spline36_EXT (not a reference setting; BASELINE config 4's optional run): six taps, normalised, mirror
    symmetric in t, the centre tap alone at t = 0 — and the planner's table equals the oracle's bit for bit.  Its taps sit at
    base-2 .. base+3 without Lanczos3's Q1 quirk (that is a property of that shader's text)."""
    import ctypes as C
    from videorenderer_amd import api
    L = oracle.lib()
    for t in (0.0, 0.25, 0.5, 0.75, 0.1234, 0.999):
        w = np.array(api.plan_upscale_weights(api.UPSCALE_Spline36_EXT, t), dtype=np.float32)
        wo = (C.c_float * 6)()
        assert L.orc_upscale_weights(6, C.c_float(t), wo) == 6
        assert np.array_equal(w, np.array(list(wo), dtype=np.float32)), t
        assert abs(float(w.astype(np.float64).sum()) - 1.0) < 3e-7
        if t:
            m = np.array(api.plan_upscale_weights(api.UPSCALE_Spline36_EXT, 1.0 - t), dtype=np.float32)
            assert np.allclose(w, m[::-1], atol=2e-7)
    assert list(api.plan_upscale_weights(api.UPSCALE_Spline36_EXT, 0.0)) == [0, 0, 1, 0, 0, 0]
    s = api.default_settings(iUpscaling=api.UPSCALE_Spline36_EXT)
    assert api.plan_describe(s, 2, 1920, 1080, (0, 0, 3840, 2160), 3840, 2160).startswith("fused_up2x")
    assert api.plan_describe(s, 2, 1920, 1080, (0, 0, 2560, 1440), 2560, 1440).startswith("passes:convert,resizeX,resizeY+final")


def test_settings_default_matches_reference(mpcvr):
    from videorenderer_amd import api
    s = api.default_settings()
    # Settings_t::SetDefault — IVideoRenderer.h:140-185
    assert (s.iTexFormat, s.iChromaScaling, s.iUpscaling, s.iDownscaling) == (0, 1, 2, 2)
    assert (s.bInterpolateAt50pct, s.bUseDither, s.bDeintBlend, s.bConvertToSdr, s.iSDRDisplayNits) == (1, 1, 0, 1, 125)


def test_every_case_has_a_plan(mpcvr):
    """The planner accepts every named parity case (so the GPU tests exercise real passes)."""
    from videorenderer_amd import api
    kinds = set()
    for name, c in GOLDEN_CASES.items():
        s = api.default_settings(**{k: c[k] for k in SETTING_KEYS if k in c})
        (ww, wh), vr = case_geometry(c)
        r = c.get("src_rect", (0, 0, c["w"], c["h"]))
        kinds.add(api.plan_describe(s, c["cformat"], r[2] - r[0], r[3] - r[1], vr, ww, wh).split(";")[0])
    assert {"fused_up2x", "direct:convert+copy", "direct:convert+final", "passes:convert,resizeX,resizeY+final",
            "passes:convert,resizeX,resizeY"} <= kinds


def test_unorm_division_shortcut_is_exact(oracle):
    """vp_device.h unorm_div (reciprocal + one FMA Newton step) vs the IEEE division, for every UNORM8/10/16 code."""
    L = oracle.lib()
    for maxv in (255, 1023, 65535):
        assert L.orc_check_unorm_div(maxv) == 0, maxv


def test_final_pass_integer_form_is_exhaustively_exact(mpcvr):
    """The fused kernel's final pass, (k*M + (j << 14)) >> 24, against ps_final_pass.hlsl:29 for every (k, j).

    k = UNORM10 code of m_TexsPostScale, j/1024 = dither texel (dither32x32float16.bin holds exactly these values).
    It must equal the exact rational floor(k*255/1023 + j/1024) everywhere, and the fp32 shader arithmetic
    floor(fl(fl(k/1023)*255) + j/1024) in all but the handful of pairs where fp32 rounds the sum up onto an integer.
    """
    api = mpcvr
    M = api.plan_final_pass_multiplier(255, 1023)
    assert M == -(-255 * 2 ** 24 // 1023) == 4182004
    k = np.arange(1024, dtype=np.uint64)[:, None]
    j = np.arange(1024, dtype=np.uint64)[None, :]
    got = (k * np.uint64(M) + (j << np.uint64(14))) >> np.uint64(24)
    assert int((k * np.uint64(M) + (j << np.uint64(14))).max()) < 2 ** 32
    exact = (k * np.uint64(255 * 1024) + j * np.uint64(1023)) // np.uint64(1023 * 1024)
    assert np.array_equal(got, exact)
    kf = k.astype(np.float32)
    p = (kf / np.float32(1023)).astype(np.float32)
    d = (j.astype(np.float32) / np.float32(1024)).astype(np.float32)
    shader = np.floor(((p * np.float32(255)).astype(np.float32) + d).astype(np.float32))
    assert int((got.astype(np.float32) != shader).sum()) <= 8
    assert int(got.max()) == 255 and int(got.min()) == 0
    # not representable combinations fall back to the float epilogue
    assert api.plan_final_pass_multiplier(255, 255) == 0
    assert api.plan_final_pass_multiplier(1023, 1023) == 0


# ---------------------------------------------------------------- Dolby Vision host maths
def _orc_dovi_cb(oracle, md):
    od = oracle.fill_dovi(oracle.OrcDovi(), md)
    cb = (oracle.OrcDoviCb * 3)()
    has_mmr = C.c_int(0)
    oracle.lib().orc_dovi_pack_curves(C.byref(od), cb, C.byref(has_mmr))
    flat = np.zeros((3, 235), np.float32)
    for c in range(3):
        flat[c, :7] = np.array(cb[c].pivots)
        flat[c, 7:39] = np.array(cb[c].coeffs).reshape(-1)
        flat[c, 39:231] = np.array(cb[c].mmr).reshape(-1)
        flat[c, 231:] = [cb[c].methods, cb[c].mmr_single, cb[c].min_order, cb[c].max_order]
    return od, flat, has_mmr.value


@pytest.mark.parametrize("kind", ["poly", "mmr", "mixed", "identity"])
def test_dovi_constants_match_oracle_bit_exact(mpcvr, oracle, kind):
    """SetShaderDoviCurves packing, the LMS matrix, the level-2 selection for several display peaks and the level-1 nits:
    product host code (vp_dovi.cpp, through mpcvr_plan_dovi) vs the oracle's restatement, as fp32 bit patterns."""
    from videorenderer_amd import api, synth
    md = synth.dovi_metadata(kind, l1=True, l3=(kind == "mmr"), l2=(100, 600, 1000))
    od, want_cb, want_mmr = _orc_dovi_cb(oracle, md)
    L = oracle.lib()
    lms = (C.c_float * 9)()
    L.orc_dovi_lms_matrix(C.byref(od), lms)
    l1 = (C.c_uint32 * 3)()
    l1on = L.orc_dovi_l1_nits(C.byref(od), l1)
    for nits in (50, 100, 350, 600, 800, 1000, 2500, 4000, 9000):
        got = api.plan_dovi(md, nits)
        k = (C.c_float * 5)()
        on = L.orc_dovi_l2_constants(C.byref(od), nits, k)
        assert np.array_equal(got["cb"].view(np.uint32), want_cb.view(np.uint32))
        assert got["has_mmr"] == want_mmr == (kind in ("mmr", "mixed"))
        assert np.array_equal(got["lms"].view(np.uint32), np.array(lms, np.float32).view(np.uint32))
        assert np.array_equal(got["l2k"].view(np.uint32), np.array(k, np.float32).view(np.uint32)), nits
        assert got["l2_enabled"] == on == 1
        assert list(got["l1_nits"]) == list(l1) and got["l1_present"] == l1on == 1


def test_dovi_level2_selection_rules(mpcvr):
    """The three scenarios of CopySample's level-2 block (DX11VideoProcessor.cpp:2383-2469) on hand-made targets."""
    from videorenderer_amd import api, synth
    md = synth.dovi_metadata("identity", l2=(100, 1000))
    lo, hi = md["l2"]

    def raw(e):      # cbuffer layout: chroma_weight-0.5, saturation_gain-0.5, slope+0.5, offset-0.5, power+0.5 (:956-958)
        return np.array([e["trim_chroma_weight"] / 4096 - 0.5, e["trim_saturation_gain"] / 4096 - 0.5,
                         e["trim_slope"] / 4096 + 0.5, e["trim_offset"] / 4096 - 0.5, e["trim_power"] / 4096 + 0.5], np.float32)
    # dimmer than every target: the lowest target as is
    assert np.allclose(api.plan_dovi(md, 50)["l2k"], raw(lo), atol=1e-7)
    # exactly on a target
    assert np.allclose(api.plan_dovi(md, 100)["l2k"], raw(lo), atol=2e-4)
    # between: a PQ-domain blend, strictly inside the two
    mid = api.plan_dovi(md, 400)["l2k"]
    assert np.all((mid >= np.minimum(raw(lo), raw(hi)) - 1e-7) & (mid <= np.maximum(raw(lo), raw(hi)) + 1e-7))
    assert not np.allclose(mid, raw(lo)) and not np.allclose(mid, raw(hi))
    # brighter than the mastering display: neutral trims (2048/4096 = 0.5 -> 0, 0, 1, 0, 1)
    assert np.allclose(api.plan_dovi(md, 9000)["l2k"], [0, 0, 1, 0, 1], atol=1e-7)
    # no level-2 block: L2Enabled = 0 and the cbuffer of a zeroed L2
    none = api.plan_dovi(synth.dovi_metadata("identity"), 400)
    assert none["l2_enabled"] == 0 and np.allclose(none["l2k"], [-0.5, -0.5, 0.5, -0.5, 0.5])
    assert none["l1_present"] == 0


def test_dovi_metadata_validation(mpcvr):
    """CheckDoviMetadata's curve rules (VideoProcessor.cpp:283-292): num_pivots in [2, 9], mapping_idc <= 1."""
    from videorenderer_amd import api, synth
    good = api.DoviMetadata.from_dict(synth.dovi_metadata("mixed"))
    assert api.load_library().mpcvr_plan_dovi(C.byref(good), 1000, None, None, None, None, None, None, None) == 0
    for mutate in (lambda m: setattr(m.curves[1], "num_pivots", 1), lambda m: setattr(m.curves[2], "num_pivots", 10),
                   lambda m: m.curves[0].mapping_idc.__setitem__(2, 2), lambda m: setattr(m, "n_l2", 33)):
        bad = api.DoviMetadata.from_dict(synth.dovi_metadata("mixed"))
        mutate(bad)
        hr = api.load_library().mpcvr_plan_dovi(C.byref(bad), 1000, None, None, None, None, None, None, None)
        assert hr & 0xffffffff == 0x80070057        # E_INVALIDARG
    assert api.load_library().mpcvr_plan_dovi(None, 1000, None, None, None, None, None, None, None) & 0xffffffff == 0x80004003
    # the ctypes mirrors used by the tests have the C layout of include/mpcvr.h / oracle/mpcvr_oracle.h
    from oracle import oracle as O
    assert C.sizeof(api.DoviMetadata) == C.sizeof(O.OrcDovi)
    assert C.sizeof(api.DoviCurve) == 1 + 24 + 1 + 18 + 4 + 8 * (24 + 8 + 168)


# ---------------------------------------------------------------- correction shaders (m_pPSCorrection)
def test_correction_matrices(mpcvr, oracle):
    """fix_bt2020_matrix / fix_ycgco_matrix / convert_matrix_2020_to_709 as the correction shaders fold them: product host code
    vs the oracle bit for bit, and known answers — the HLSL (zimg) derivation of the primaries matrix agrees with csputils'
    GetColorspaceGamutConversionMatrix to ~2e-5, the fix-up matrices map grey to grey, a YCgCo mis-decode is undone."""
    from videorenderer_amd import api
    a, b, g = api.plan_correction_matrices()
    oa, ob, og = oracle.correction_matrices()
    assert np.array_equal(np.array(a, np.float32).view(np.uint32), oa.view(np.uint32))
    assert np.array_equal(np.array(b, np.float32).view(np.uint32), ob.view(np.uint32))
    assert np.array_equal(np.array(g, np.float32).view(np.uint32), og.view(np.uint32))
    assert np.allclose(og, api.plan_gamut_2020_to_709(), atol=5e-5)     # (different white-point digits: ~1.6e-5 apart)
    grey = np.array([0.5, 0.5, 0.5, 1.0], np.float32)
    for m in (oa, ob):
        assert np.allclose((m.reshape(4, 4) @ grey)[:3], 0.5, atol=1e-6)
    # what the D3D11 VP did wrong and the shader undoes: it decoded YCgCo data with the BT.709 matrix
    ycgco_rgb = np.array([[1, -1, 1], [1, 1, 0], [1, -1, -1]], np.float64)
    rgb709_from_ycc = np.linalg.inv(np.array([[0.2126, 0.7152, 0.0722], [-0.114572, -0.385428, 0.5], [0.5, -0.454153, -0.045847]]))
    ycc = np.array([0.4, 0.1, -0.05])                      # some (Y, Cg, Co)
    wrong = rgb709_from_ycc @ ycc
    fixed = ob.reshape(4, 4)[:3, :3].astype(np.float64) @ wrong
    assert np.allclose(fixed, ycgco_rgb @ ycc, atol=1e-5)


# ---------------------------------------------------------------- the C-ABI from plain C
def build_c_demo(tmpdir, name="c_abi_demo", rccl=False):
    """gcc -std=c99 on examples/<name>.c against include/mpcvr.h and libmpcvr.so; rccl: the host also talks to RCCL itself
    (rccl.h pulls in the HIP runtime API header, which is not -Werror clean as C)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(str(tmpdir), name)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    extra_c = ["-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include")] if rccl else ["-Werror"]
    extra_l = ["-L" + os.path.join(rocm, "lib"), "-lrccl", "-lamdhip64", "-Wl,-rpath," + os.path.join(rocm, "lib")] if rccl else []
    subprocess.check_call(["gcc", "-O2", "-Wall", "-std=c99"] + extra_c + ["-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", name + ".c"), "-o", exe,
                           "-L" + os.path.join(root, "videorenderer_amd"), "-lmpcvr",
                           "-Wl,-rpath," + os.path.join(root, "videorenderer_amd")] + extra_l)
    return exe


def test_c_abi_links_from_plain_c(mpcvr, tmp_path):
    """include/mpcvr.h is a C header and libmpcvr.so resolves everything a C program needs: examples/c_abi_demo.c builds
    with gcc -std=c99 -Werror against them (it runs in the GPU suite)."""
    exe = build_c_demo(tmp_path)
    assert os.path.exists(exe)
    assert os.path.exists(build_c_demo(tmp_path, "c_multi_gpu"))      # one context per device, parameter blob shared, frames by index
    # the same with the blob broadcast by the library over RCCL (mpcvr_broadcast_param_blob_begin / _end inside the host's nccl group)
    assert os.path.exists(build_c_demo(tmp_path, "c_multi_gpu_rccl", rccl=True))



@pytest.mark.parametrize("kx,mx,ky,my,sw,sh,dw,dh", [
    (1, 4, 1, 4, 1920, 1080, 2560, 1440),        # up1440: Lanczos3 1.33x
    (2, 2, 2, 2, 3840, 2160, 2560, 1440),        # down1440: Hamming 1.5x
    (1, 2, 1, 2, 1280, 720, 1919, 1079),         # odd output size, Catmull-Rom
    (1, 3, 1, 3, 1280, 1440, 1920, 800),         # up along x, down along y through the interpolation shader (50 % rule)
    (2, 2, 2, 2, 1920, 1080, 738, 416),          # 2.6x Hamming: 8 taps
    (2, 3, 2, 3, 1920, 1080, 834, 470),          # bicubic 2.3x down: 10 taps -> the 16-tap variant, a 32-row ring, one pixel per lane
    (2, 5, 2, 5, 3840, 2160, 1920, 1080),        # Lanczos down at exactly 2x without the 50 % rule: 13 taps
    (2, 3, 2, 3, 3840, 2160, 1280, 720),         # 4K -> 720p bicubic: 13 taps
    (1, 1, 1, 1, 3840, 2160, 1920, 1080),        # the interpolation shader at 50 %
    (1, 4, 1, 4, 64, 48, 100, 70),               # small frame: one strip
])
def test_strip_plan_tables(mpcvr, kx, mx, ky, my, sw, sh, dw, dh):
    """PlanFusedStrip (host side of the arbitrary-ratio fused kernel), without a device: the kernel's copies of the tap tables equal
    the draws' own tables (normalisation folded in, zero-weight padding), every tap lies inside the windows the kernel stages
    (per-strip source columns, per-row source rows), the row windows are monotonic and fit the LDS ring, and a CU's LDS holds at
    least four waves."""
    from videorenderer_amd import api
    import numpy as np
    sp = api.plan_strip(kx, mx, ky, my, sw, sh, dw, dh)
    assert sp is not None
    nt, pxl, strip_w = sp["taps"], sp["px_per_lane"], sp["strip_w"]
    assert nt in (4, 6, 8, 16) and pxl in (1, 2) and sp["ring"] in (8, 16, 32)
    assert nt != 16 or pxl == 1
    assert strip_w % pxl == 0 and pxl <= strip_w <= 64 * pxl and sp["strips"] == -(-dw // strip_w)
    for axis, (kind, method, src, n_out) in enumerate(((kx, mx, sw, dw), (ky, my, sh, dh))):
        I, W, WS = api.plan_axis_taps(kind, method, 0, src, n_out, src)
        I, W = np.array(I, np.int32), np.array(W, np.float32)
        n = I.shape[1]
        assert n <= nt
        ti = sp["xi_t"].T if axis == 0 else sp["yi"]
        tw = sp["xw_t"].T if axis == 0 else sp["yw"]
        assert np.array_equal(ti[:, :n], I)
        assert np.array_equal(ti[:, n:], np.repeat(I[:, :1], nt - n, axis=1))           # padding taps re-read the first texel ...
        assert not tw[:, n:].any()                                                       # ... with weight 0
        want = W / np.array(WS, np.float32)[:, None] if WS is not None else W
        assert np.array_equal(tw[:, :n], want.astype(np.float32))
        if WS is not None:
            assert np.allclose(tw.sum(axis=1), 1.0, atol=2e-6)
    # per-row windows: exact min / max of the row's taps, monotonic, and narrow enough for the ring (rows arrive in pairs)
    yi = sp["yi"]
    nty = len(api.plan_axis_taps(ky, my, 0, sh, dh, sh)[0][0])
    assert np.array_equal(sp["yrange"][:, 0], yi[:, :nty].min(axis=1)) and np.array_equal(sp["yrange"][:, 1], yi[:, :nty].max(axis=1))
    assert (np.diff(sp["yrange"][:, 0]) >= 0).all() and (np.diff(sp["yrange"][:, 1]) >= 0).all()
    assert int((sp["yrange"][:, 1] - sp["yrange"][:, 0]).max()) + 2 <= sp["ring"]
    # per-strip windows cover every tap of the strip and fit the converted-row slice
    xi = sp["xi_t"]
    for s_ in range(sp["strips"]):
        cols = xi[:, s_ * strip_w:(s_ + 1) * strip_w]
        lo, hi = sp["xstrip"][s_]
        assert lo == cols.min() and hi == cols.max()
        assert hi - (lo & ~1) + 1 <= sp["acols"]
    assert 4 * sp["lds_per_wave"] + 4096 + 32768 <= 160 * 1024          # at least four waves per CU beside the tables


def test_strip_plan_refuses_what_the_kernel_cannot_run(mpcvr):
    from videorenderer_amd import api
    assert api.plan_strip(2, 5, 2, 5, 3840, 2160, 1280, 720) is None          # Lanczos 3x down: 19 taps
    assert api.plan_strip(2, 3, 2, 3, 3840, 2160, 900, 506) is None           # bicubic 4.3x down: 19 taps
    assert api.plan_strip(2, 3, 2, 3, 1920, 1080, 834, 470)["taps"] == 16     # bicubic 2.3x down: 10 taps -> the 16-tap kernels
    assert api.plan_strip(1, 4, 1, 4, 1920, 1080, 3840, 2160) is not None     # exact 2x fits too (the 2x kernel is preferred)


def test_pq_eotf_table(mpcvr):
    """The Dolby Vision block convert's PQ EOTF table: log2 of ST2084ToLinear at x = (i/8192)^2, i = 0 .. 8192, floored at -150 where the
    EOTF is exactly 0 — exact to an fp32 rounding of the logarithm (built in double since round 4), and fine enough that linear
    interpolation in sqrt(x) decodes to 2e-6 relative: the accuracy of the shader's own fp32 pow chain, which is what the plain
    kernels and the oracle evaluate."""
    from videorenderer_amd import api
    import numpy as np
    N = 8192
    t = api.plan_pq_eotf_lut().astype(np.float64)
    assert t.size == N + 1
    x = (np.arange(N + 1) / N) ** 2
    m1, m2, c1, c2, c3 = 2610 / 16384, 2523 / 4096 * 128, 3424 / 4096, 2413 / 4096 * 32, 2392 / 4096 * 32
    z = x ** (1 / m2)
    lin = (np.maximum(z - c1, 0) / (c2 - c3 * z)) ** (1 / m1)
    live = lin > 1e-30
    assert np.abs(t[live] - np.log2(lin[live])).max() < 8e-6                      # half an fp32 ulp of a logarithm of magnitude <= 100
    assert (t[~live] == -150.0).all() and t[-1] == 0.0 and (np.diff(t) >= 0).all()
    # linear interpolation of the table in sqrt(x) (midpoints): within 3e-6 relative wherever the EOTF exceeds 1e-7 (0.001 nits),
    # and within 1e-9 absolute (1e-5 nits) below that, where the power law is steep but nothing visible depends on it
    xm = ((np.arange(16, N) + 0.5) / N) ** 2
    zm = xm ** (1 / m2)
    lm = (np.maximum(zm - c1, 0) / (c2 - c3 * zm)) ** (1 / m1)
    got = 2.0 ** (0.5 * (t[16:N] + t[17:N + 1]))
    vis = lm > 1e-7
    assert np.abs(got[vis] / lm[vis] - 1).max() < 3e-6
    assert np.abs(got[~vis] - lm[~vis]).max() < 1e-9


def test_deprecated_pq_eotf_lut_shim(mpcvr):
    """mpcvr_plan_pq_eotf_lut(float[4096]) — the round-3 entry point, kept one more release for callers built against the old header: exactly
    4096 floats are written (the guard words behind them stay), the same function as the table's on the 4096-point grid."""
    import ctypes as C
    import numpy as np
    from videorenderer_amd import api
    L = api.load_library()
    buf = np.full(4096 + 64, np.float32(-777.0), dtype=np.float32)
    assert L.mpcvr_plan_pq_eotf_lut(buf.ctypes.data_as(C.POINTER(C.c_float))) == 0
    assert (buf[4096:] == np.float32(-777.0)).all()
    t = buf[:4096].astype(np.float64)
    x = (np.arange(4096) / 4095) ** 2
    m1, m2, c1, c2, c3 = 2610 / 16384, 2523 / 4096 * 128, 3424 / 4096, 2413 / 4096 * 32, 2392 / 4096 * 32
    z = x ** (1 / m2)
    lin = (np.maximum(z - c1, 0) / (c2 - c3 * z)) ** (1 / m1)
    live = lin > 1e-30
    assert np.abs(t[live] - np.log2(lin[live])).max() < 8e-6 and (t[~live] == -150.0).all() and t[-1] == 0.0
    assert L.mpcvr_plan_pq_eotf_lut(None) < 0


# ---- periodic-phase fused kernel (vp_fused_period.h): the planner's side of its compile-time tap rows ----
def _period_base(P, Q, r):
    return ((2 * r + 1) * Q - P) // (2 * P)            # floor: pos = (r + .5) Q / P - .5


@pytest.mark.parametrize("method,sw,sh,dw,dh,want", [
    (4, 1920, 1080, 2560, 1440, (4, 3, 5)),       # up1440: Lanczos3 as Direct3D 11 draws it -> 5 taps
    (2, 1280, 720, 1920, 1080, (3, 2, 4)),        # 720p -> 1080p Catmull-Rom
    (4, 3840, 2160, 2560, 1440, (2, 3, 5)),       # down1440: the interpolation shader below 2x
    (1, 3840, 2160, 1920, 1080, (1, 2, 4)),       # 4K -> 1080p Mitchell at exactly 50 %
    (3, 2560, 1440, 3840, 2160, (3, 2, 4)),       # 1440p -> 4K Lanczos2
    (6, 1920, 1080, 2560, 1440, (4, 3, 6)),       # Spline36 extension: six distinct texels
    (4, 1280, 720, 3840, 2160, (3, 1, 5)),        # 720p -> 2160p: every third row exactly on a texel centre (period_centre)
    (2, 640, 360, 1920, 1080, (3, 1, 4)),
    (4, 640, 480, 1920, 1440, (3, 1, 5)),
    (2, 1919, 1080, 2559, 1440, (4, 3, 4)),       # the rows decide: any width rides along
])
def test_period_plan_matches_the_tap_tables(mpcvr, method, sw, sh, dw, dh, want):
    """PlanFusedPeriod: (P, Q, taps) per geometry, and its tables against BuildAxisTaps' — every tap row the kernel hard-codes
    (6m + base(r) + offset, clamped) is the row the reference's shader arithmetic picks, and the weights are the table's own."""
    from videorenderer_amd import api
    pp = api.plan_period(method, sw, sh, dw, dh)
    assert pp is not None and (pp["P"], pp["Q"], pp["taps"]) == want
    P, Q, nt = want
    PB = 6 * P // Q
    assert pp["rows_per_body"] == PB
    I, W, _ = api.plan_axis_taps(1, method, 0, sh, dh, sh)
    n = len(I[0])
    off = {4: [-1, 0, 1, 2], 5: [-2, -2, 0, 1, 2, 3], 6: [-2, -1, 0, 1, 2, 3]}[nt]
    below = pp["yw"][::PB, 7].copy().view(np.uint32)      # per body: bit r = row PB*m + r reads one source row lower (centre rows, 3:1)
    n_below = 0
    for y in range(dh):
        r = y % PB
        centre = ((2 * r + 1) * Q - P) % (2 * P) == 0
        low = int(below[y // PB] >> r) & 1
        assert centre or not low
        n_below += low
        base = 6 * (y // PB) + _period_base(P, Q, r) - low
        assert [min(max(base + o, 0), sh - 1) for o in off] == list(I[y])
        w = list(W[y])
        folded = [np.float32(w[0]) + np.float32(w[1])] + w[2:] if nt == 5 else w
        assert np.array_equal(np.asarray(folded, np.float32), pp["yw"][y, :nt]) and not pp["yw"][y, nt:7].any()
        assert r == 0 or pp["yw"][y, 7] == 0
    assert (P, Q) == (3, 1) or n_below == 0
    IX, WX, _ = api.plan_axis_taps(1, method, 0, sw, dw, sw)
    for x in (0, 1, dw // 2, dw - 1):
        ix = list(IX[x]); wx = list(WX[x])
        if nt == 5:
            ix = [ix[0]] + ix[2:]; wx = [np.float32(wx[0]) + np.float32(wx[1])] + wx[2:]
        assert ix == list(pp["xi_t"][:, x]) and np.array_equal(np.asarray(wx, np.float32), pp["xw_t"][:, x])
    for s_ in range(pp["strips"]):
        cols = IX[pp["strip_w"] * s_: pp["strip_w"] * (s_ + 1)]
        assert pp["xstrip"][s_, 0] == min(min(c) for c in cols) and pp["xstrip"][s_, 1] == max(max(c) for c in cols)
    assert pp["strip_w"] % 2 == 0 and 48 <= pp["strip_w"] <= 128 and pp["strips"] == -(-dw // pp["strip_w"])
    assert pp["acols"] % 2 == 0 and pp["acols"] >= max(pp["xstrip"][:, 1] - (pp["xstrip"][:, 0] & ~1)) + 1


def test_period_plan_refuses_other_ratios(mpcvr):
    from videorenderer_amd import api
    assert api.plan_period(4, 1920, 1080, 3840, 2160) is None        # exact 2x: the 2x kernel's
    assert api.plan_period(4, 1920, 1080, 2400, 1350) is None        # 5:4
    assert api.plan_period(2, 1920, 1080, 2560, 1439) is None        # not exactly 4:3 down the rows
    assert api.plan_period(4, 1920, 1080, 2560, 1440, flags=api.FLAG_LANCZOS3_FIXED)["taps"] == 6


def test_two_step_quotient_of_8bit_codes():
    """vp_fused_dev.h: xnorm2_u8 — code / 255 as fma(code, hi, fl(code * lo)) with hi + lo = 1/255 to 2^-56.  The exact form of the fused
    convert stage reads 8-bit texels that way (two packed operations instead of unorm_div's three); it must be the correctly rounded
    quotient — what a UNORM texture fetch returns — for every code."""
    import re
    from fractions import Fraction
    import numpy as np
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "videorenderer_amd", "csrc", "vp_fused_dev.h")).read()
    m = re.search(r"kInv255Hi = (\S+)f, kInv255Lo = (\S+)f;", src)
    hi, lo = float.fromhex(m.group(1)), float.fromhex(m.group(2))
    assert np.float32(hi) == hi and np.float32(lo) == lo          # both are floats
    def rn32(fr):           # the float nearest to a rational
        x = np.float32(float(fr))
        return min((np.nextafter(x, np.float32(-1)), x, np.nextafter(x, np.float32(2))), key=lambda c: abs(Fraction(float(c)) - fr))
    for code in range(256):
        t = np.float32(code) * np.float32(lo)                                     # rounded product
        got = rn32(Fraction(code) * Fraction(hi) + Fraction(float(t)))            # the FMA: one rounding of the exact sum
        assert got == rn32(Fraction(code, 255)), code


def test_committed_traffic_figures_belong_to_the_committed_kernels():
    """profiles/hbm_traffic.json ties every entry to the sha256 of the kernel sources it was measured on, and bench.py DROPS an entry whose
    sources changed (roofline.traffic = null, no co_limit): a commit that touches a kernel file has to re-run tools/pmc_traffic.sh for the
    workloads that file serves (round 6: an experiment's guarded block was committed with a profile set and removed after it — the
    headline's entry went stale without a byte of the binary changing)."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    t = json.load(open(os.path.join(root, "profiles", "hbm_traffic.json")))
    stale = [k for k, v in t.items() if isinstance(v, dict) and "csrc_sha256" in v and v["csrc_sha256"] != bench.csrc_digest(v.get("sources"))]
    assert not stale, f"stale entries (re-run tools/pmc_traffic.sh + tools/update_traffic.py for them): {stale}"
