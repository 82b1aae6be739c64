"""EXTENSION: the error-diffusion final pass (bUseDither = 2, MPCVR_DITHER_ErrorDiffusion_EXT) — BASELINE.json config 4's optional half.

PARITY UNPINNED BY CONSTRUCTION: the reference has no error diffusion (its final pass is the ordered dither of ps_final_pass.hlsl and
`grep -ri diffusion /root/reference` is empty), so there is nothing to pin against.  The definition is the serial integer model
oracle/mpcvr_oracle.c:orc_error_diffusion, and what these tests establish is that the product equals THAT, bit for bit:
  * CPU: the kernel's schedule — free-running bands behind tagged hand-off words — (tests/tools/errdiff_emulate.cpp, built from the
    product's own vp_errdiff_core.h) against the serial model; the quantiser's multiply-high division over its whole range; the planner's rule for when the pass runs;
  * GPU: k_error_diffusion against the serial model on the product's own 10-bit frames (every route: fused 2x, strip / periodic,
    same-size convert, plain kernels, batches, clipped and offset video rects, both ways of handing errors down), end to end against
    the oracle where the tier in front is bit-exact, and at BASELINE's 4K -> 8K size.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests.golden.cases import FULL_SIZE_CASES, GOLDEN_CASES, case_frame, oracle_params

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
U = 16 * 1023


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    """The schedule emulator: a test tool compiled from the product's header, never part of libmpcvr.so."""
    out = str(tmp_path_factory.mktemp("errdiff") / "liberrdiff_emu.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", out, os.path.join(HERE, "tools", "errdiff_emulate.cpp")])
    L = C.CDLL(out)
    L.ed_emulate.restype = C.c_int
    L.ed_emulate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_int] * 4 + [C.c_uint32]
    L.ed_quant_range.argtypes = [C.c_int32, C.c_int32, C.c_void_p]
    return L


def synth10(kind, h, w, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 2 ** 30, size=(h, w), dtype=np.uint32)
    if kind == "flat":
        return np.full((h, w), 513 | (2 << 10) | (1021 << 20), dtype=np.uint32)
    if kind == "dark":      # codes 0..3: the clamp at 0 carries whole errors forward
        return (rng.integers(0, 4, size=(h, w), dtype=np.uint32) | (rng.integers(1020, 1024, size=(h, w), dtype=np.uint32) << 10)
                | (rng.integers(0, 2, size=(h, w), dtype=np.uint32) << 20)).astype(np.uint32)
    x = np.arange(w, dtype=np.uint32)[None, :] * 1023 // max(w - 1, 1)
    y = np.arange(h, dtype=np.uint32)[:, None] * 1023 // max(h - 1, 1)
    return (x | (y << 10) | (((x + y) // 2) << 20)).astype(np.uint32)


def test_quantiser_division_is_exact_over_its_range(emu):
    """ed_quant's multiply-high (vp_errdiff_core.h) against clamp(floor((T + U/2) / U), 0, 255) for every T a frame can produce and far beyond."""
    lo, hi = -15 * U, 4080 * 1023 + 40 * U
    got = np.zeros(hi - lo, dtype=np.int32)
    emu.ed_quant_range(lo, hi, got.ctypes.data)
    t = np.arange(lo, hi, dtype=np.int64)
    want = np.clip((t + U // 2) // U, 0, 255)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("h,w,rect", [(5, 7, None), (64, 128, None), (65, 129, (1, 0, 129, 65)), (130, 300, (3, 2, 297, 129)),
                                      (200, 37, None), (1100, 260, (5, 7, 255, 1090)), (70, 1, None), (1, 50, None), (2100, 150, (1, 1, 149, 2100)),
                                      (66, 700, (0, 0, 700, 66))])
def test_wavefront_schedule_equals_the_serial_model(emu, oracle, h, w, rect):
    """One free-running wavefront per band of 21 rows (lane = 21 * channel + row), two columns of skew from row to row, tagged hand-off words between bands: every
    dependency of the kernel's schedule, executed on the host with the bands taking turns top-down, as-late-as-possible and at random —
    no band may read a word before it is written, and none may wait for a word that is never written (the emulator returns -1)."""
    rect = rect or (0, 0, w, h)
    for kind in ("noise", "flat", "ramp", "dark"):
        img = synth10(kind, h, w, seed=h * 1000 + w)
        want = oracle.error_diffusion(img, rect)
        for seed in (0, 1, 7, 12345):
            got = np.zeros((h, w, 4), np.uint8)
            waits = emu.ed_emulate(img.ctypes.data, w * 4, got.ctypes.data, w * 4, *rect, seed)
            assert waits >= 0, (kind, seed, "a band that can never proceed")
            assert np.array_equal(got, want), (kind, seed)


def test_serial_model_properties(oracle):
    """What makes it error diffusion: exactly representable flats stay flat, and the region's sum is kept up to what leaves at its edges."""
    h, w = 96, 160
    for k, q in ((0, 0), (1023, 255), (341, 85), (682, 170)):          # 255 k / 1023 is an integer
        out = oracle.error_diffusion(np.full((h, w), k | (k << 10) | (k << 20), dtype=np.uint32))
        assert (out[..., :3] == q).all() and (out[..., 3] == 255).all()
    img = synth10("noise", h, w, seed=5)
    out = oracle.error_diffusion(img)
    for c, byte in ((0, 2), (1, 1), (2, 0)):
        k = ((img >> (10 * c)) & 0x3ff).astype(np.int64)
        lost = abs(int((4080 * k).sum()) - int(out[..., byte].astype(np.int64).sum()) * U)
        assert lost <= (2 * h + w) * U          # shares dropped at the left / right / bottom edges, at most about a code each
    # any 8x8 block of a mid-grey flat averages to the input within a fraction of a code
    k = 500
    out = oracle.error_diffusion(np.full((h, w), k | (k << 10) | (k << 20), dtype=np.uint32))[16:, 16:, 0].astype(np.float64)
    blocks = out[:80, :144].reshape(10, 8, 18, 8).mean(axis=(1, 3))
    assert np.abs(blocks - k * 255 / 1023).max() < 0.25
    # a region inside a larger image leaves the rest alone
    out = oracle.error_diffusion(img, (3, 5, 150, 90))
    mask = np.zeros((h, w), bool); mask[5:90, 3:150] = True
    assert (out[~mask] == 0).all() and (out[mask][:, 3] == 255).all()


def test_planner_runs_the_pass_only_where_the_reference_would_dither_to_8_bits(mpcvr):
    from videorenderer_amd import api
    ed = api.default_settings(bUseDither=api.DITHER_ErrorDiffusion_EXT, iUpscaling=4)
    # P010 (10-bit internal) into an 8-bit target: the 10-bit plan + the pass
    d = api.plan_describe(ed, api.CF_P010, 128, 72, (0, 0, 256, 144), 256, 144)
    assert "errdiff" in d and "final=0" in d and "swap=10" in d, d          # (swap = the plan's target: the 10-bit intermediate)
    assert "errdiff" in api.plan_describe(ed, api.CF_P010, 128, 72, (0, 0, 128, 72), 128, 72)
    # NV12 (8-bit internal) or a 10-bit target: nothing to quantise, as with the ordered dither
    assert "errdiff" not in api.plan_describe(ed, api.CF_NV12, 128, 72, (0, 0, 256, 144), 256, 144)
    assert "errdiff" not in api.plan_describe(ed.copy(output_format=api.OUT_RGB10A2), api.CF_P010, 128, 72, (0, 0, 256, 144), 256, 144)
    # fp16 internal: the 10-bit plan keeps its ordered dither to 10 bits, then the pass
    d = api.plan_describe(ed.copy(iTexFormat=api.TEXFMT_16FLOAT), api.CF_P010, 128, 72, (0, 0, 192, 108), 192, 108)
    assert "errdiff" in d and "final=1" in d, d
    with pytest.raises(api.MpcvrError):
        api.VideoProcessor(api.default_settings(bUseDither=3))


# ------------------------------------------------------------------------------------------------------------------------------
# GPU
# ------------------------------------------------------------------------------------------------------------------------------
BG = 7


def region(c):
    from tests.golden.cases import case_geometry
    (ww, wh), (l, t, r, b) = case_geometry(c)
    return (max(l, 0), max(t, 0), min(r, ww), min(b, wh)), (ww, wh)


def product_10bit_and_diffused(mpcvr, torch, c, flags=0):
    """The product twice on the same sample: as the 10-bit swap chain (what the pass reads) and with bUseDither = 2."""
    from tests.test_parity_gpu import run_product
    ten, info10 = run_product(mpcvr, torch, dict(c, bUseDither=2, output_format=1), flags)
    out, info = run_product(mpcvr, torch, dict(c, bUseDither=2), flags)
    return ten.view(np.uint32)[:, :, 0], out, info10, info


ED_CASES = {
    "fused_2x": dict(GOLDEN_CASES["c3hdr_p010_pq_lanczos3_2x"]),
    "fused_2x_window_offset": dict(GOLDEN_CASES["x2_p010_pq_mitchell_offset"]),                  # x0 = 3: an odd first column
    "strip_1p5x": dict(GOLDEN_CASES["up_1p5x_lanczos3"]),
    "fused_jinc_2x": dict(GOLDEN_CASES["jinc2_p010_2x_dither"]),                                   # the fused Jinc2m kernel's 10-bit frame (generic epilogue) in front of the pass
    "same_size_pq": dict(cformat=2, w=192, h=80, kind="hdr", seed=901, dst=(192, 80), exfmt=GOLDEN_CASES["c3hdr_p010_pq_lanczos3_2x"]["exfmt"]),
    "tall_1100_rows": dict(cformat=2, w=72, h=1100, kind="noise", seed=902, dst=(72, 1100)),          # 18 bands, each ~20 steps long: they wait for each other all the time
    "wide_700": dict(cformat=2, w=700, h=70, kind="structure", seed=903, dst=(700, 70)),               # two bands, 104 groups each
    "clipped_left_top": dict(cformat=2, w=96, h=64, kind="structure", seed=904, dst=(144, 96), iUpscaling=2, window=(120, 80), offset=(-13, -9)),
    "clipped_right_bottom": dict(cformat=2, w=96, h=64, kind="noise", seed=905, dst=(144, 96), iUpscaling=2, window=(120, 80), offset=(31, 22)),
    "fp16_internal": dict(cformat=2, w=64, h=48, kind="structure", seed=906, dst=(96, 72), iUpscaling=4, iTexFormat=16),
    "down_hamming": dict(GOLDEN_CASES["down_hamming_3x"]),
    "rot90": dict(cformat=2, w=64, h=48, kind="structure", seed=907, dst=(72, 96), iUpscaling=2, rotation=90),
    "dolby_vision_poly": dict(GOLDEN_CASES["dovi_poly_sdr"]),                                          # reshaping in the convert stage, the pass behind it
    "rgb48_no_convert_draw": dict(cformat=33, w=70, h=44, kind="structure", seed=908, dst=(105, 66), iUpscaling=4),   # the source texture feeds the resize
    "v210_repacked": dict(cformat=10, w=96, h=40, kind="structure", seed=909, dst=(144, 60), iUpscaling=2),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(ED_CASES))
def test_kernel_equals_serial_model_on_the_products_10_bit_frame(mpcvr, oracle, name):
    import torch
    c = ED_CASES[name]
    ten, out, info10, info = product_10bit_and_diffused(mpcvr, torch, c)
    assert "errdiff" in info and "errdiff" not in info10, (info10, info)
    rect, (ww, wh) = region(c)
    want = oracle.error_diffusion(ten, rect, dst=np.full((wh, ww, 4), BG, dtype=np.uint8))
    assert np.array_equal(out, want), f"{name} [{info}]: {(out != want).any(axis=2).sum()} pixels differ"


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c3_p010_lanczos3_2x", "up_1p5x_lanczos3", "down_hamming_3x", "x_only_resize"])
def test_end_to_end_equals_the_oracle_on_the_bit_exact_tier(mpcvr, oracle, name):
    """Plain kernels, no transcendental tail: the 10-bit frame is the oracle's bit for bit, so the diffused one must be too."""
    import torch
    from tests.test_parity_gpu import run_product
    from videorenderer_amd import api
    c = dict(GOLDEN_CASES[name], bUseDither=2)
    out, info = run_product(mpcvr, torch, c, api.FLAG_NO_FUSED)
    frame, pitch = case_frame(c)
    want = oracle.process_errdiff(oracle_params(oracle, c), frame, pitch)
    rect, _ = region(c)
    x0, y0, x1, y1 = rect
    assert np.array_equal(out[y0:y1, x0:x1], want[y0:y1, x0:x1]), info
    # default tier (fused kernels in front): its 10-bit frame is within one 10-bit code of the oracle's, the diffused one within one 8-bit code
    # of the oracle's on all but the pixels where a quarter-code difference tips a threshold and the pattern shifts: compare block means
    out2, info2 = run_product(mpcvr, torch, c)
    a = out2[y0:y1, x0:x1, :3].astype(np.float64); b = want[y0:y1, x0:x1, :3].astype(np.float64)
    hh, ww = (a.shape[0] // 8) * 8, (a.shape[1] // 8) * 8
    ma = a[:hh, :ww].reshape(hh // 8, 8, ww // 8, 8, 3).mean(axis=(1, 3)); mb = b[:hh, :ww].reshape(hh // 8, 8, ww // 8, 8, 3).mean(axis=(1, 3))
    assert np.abs(ma - mb).max() <= 0.5, (info2, float(np.abs(ma - mb).max()))


@pytest.mark.gpu
def test_batch_equals_single_frames_and_is_one_pass_launch(mpcvr, oracle):
    import torch
    from tests.test_parity_gpu import make_vp
    from videorenderer_amd import synth
    for name in ("fused_2x", "strip_1p5x", "same_size_pq", "clipped_right_bottom"):
        c = dict(ED_CASES[name], bUseDither=2)
        (x0, y0, x1, y1), (ww, wh) = region(c)
        frames = [torch.from_numpy(synth.make_frame(c["cformat"], c["w"], c["h"], c["kind"], seed=c["seed"] + i)[0]).cuda() for i in range(5)]
        pitch = case_frame(c)[1]
        vp, _ = make_vp(mpcvr, c)
        singles = []
        for f in frames:
            d = torch.full((wh, ww, 4), BG, dtype=torch.uint8, device="cuda")
            vp.CopySample(f, pitch); vp.Process(d, ww * 4); vp.Synchronize()
            singles.append(d.cpu().numpy())
        dsts = [torch.full((wh, ww, 4), BG, dtype=torch.uint8, device="cuda") for _ in frames]
        vp.ProcessBatch(frames, dsts, ww * 4); vp.Synchronize()
        binfo = vp.GetLastBatchInfo()
        for i, d in enumerate(dsts):
            assert np.array_equal(d.cpu().numpy(), singles[i]), (name, i)
        assert binfo["frames"] == 5 and binfo["launches"] <= 4, (name, binfo)        # the 10-bit plan's whole-batch launches + ONE pass launch
        vp.close()


@pytest.mark.gpu
def test_full_size_4k_to_8k_equals_serial_model(mpcvr, oracle):
    """BASELINE config 4's shape at full size: 4K P010 PQ -> 2x -> PQ->SDR -> error diffusion into 8K B8G8R8A8 (Mitchell; the Spline36
    extension rides the same kernel): 68 bands of 976 groups, one workgroup each."""
    import torch
    c = dict(FULL_SIZE_CASES["c4_mitchell"])
    ten, out, info10, info = product_10bit_and_diffused(mpcvr, torch, c)
    want = oracle.error_diffusion(ten, (0, 0, 7680, 4320))
    assert "errdiff" in info
    assert np.array_equal(out, want), f"{(out != want).any(axis=2).sum()} pixels differ [{info}]"
    k = ((ten >> 0) & 0x3ff).astype(np.float64).mean() * 255 / 1023
    assert abs(out[..., 2].astype(np.float64).mean() - k) < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("label", ["same_size_sdr", "resized_sdr"])
def test_dolby_vision_batch_with_one_rpu_per_frame_through_chunked_passes(mpcvr, oracle, label):
    """mpcvr_process_batch_dovi in front of the pass: seven frames, seven RPUs, runs cut where the kernel variant changes, each run's
    pass cut into chunks of two frames (MPCVR_ERRDIFF_CHUNK — set for the whole GPU session of this module by the subprocess below):
    every chunk must see ITS frames' RPUs.  Equal, bit for bit, to SetDoviMetadata + Process frame after frame."""
    import subprocess, sys, textwrap
    code = textwrap.dedent(f"""
        import sys, torch
        sys.path.insert(0, {ROOT!r})
        from tests.test_parity_gpu import DOVI_BATCH_CASES, make_vp, BG
        from tests.golden.cases import case_frame
        from tests.golden import cases as G
        from videorenderer_amd import api, synth
        _, c, _ = next(x for x in DOVI_BATCH_CASES if x[0] == {label!r})
        c = dict(c); c.pop("exfmt_name"); c["exfmt"] = G.ext(G.MPEG2, G.TV); c["bUseDither"] = 2
        kinds = [dict(kind="poly"), dict(kind="mmr"), dict(kind="mixed"), dict(kind="mmr", l2=(100, 600, 1000)), dict(kind="poly"),
                 dict(kind="identity", l2=(600,)), dict(kind="mixed")]
        rpus = []
        for i, k in enumerate(kinds):
            md = api.DoviMetadata.from_dict(synth.dovi_metadata(**k))
            md.ycc_to_rgb_matrix[0] *= 1.0 - 0.01 * i
            md.ycc_to_rgb_offset[1] += 0.001 * i
            rpus.append(md)
        frames = [torch.from_numpy(case_frame(dict(c, seed=c["seed"] + 31 * i))[0]).cuda() for i in range(len(rpus))]
        one, (ww, wh) = make_vp(None, c)
        pitch = one.GetFrameBytes()[1]
        singles = []
        for f, md in zip(frames, rpus):
            d = torch.full((wh, ww, 4), BG, dtype=torch.uint8, device="cuda")
            one.SetDoviMetadata(md); one.CopySample(f, pitch); one.Process(d, ww * 4); singles.append(d)
        one.Synchronize()
        assert "errdiff" in one.GetVPInfo(), one.GetVPInfo()
        vp, _ = make_vp(None, c)
        dsts = [torch.full((wh, ww, 4), BG, dtype=torch.uint8, device="cuda") for _ in frames]
        vp.ProcessBatchDovi(frames, dsts, ww * 4, rpus); vp.Synchronize()
        bad = [i for i in range(len(frames)) if not torch.equal(singles[i], dsts[i])]
        assert not bad, (bad, vp.GetLastBatchInfo())
        assert not torch.equal(dsts[0], dsts[1])
        print("ok", vp.GetLastBatchInfo())
    """)
    env = dict(os.environ, MPCVR_ERRDIFF_CHUNK="2")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.gpu
def test_give_up_path_ends_in_an_error_not_in_a_hang(mpcvr, oracle, monkeypatch):
    """A band whose producer never publishes (MPCVR_ERRDIFF_TEST_STALL: frame 0's first band keeps its hand-off words to itself) waits a bounded
    number of polls, flags the launch in the pinned status word and carries on: mpcvr_synchronize (and the snapshot's read-back, and the next pass)
    answer MPCVR_E_FAIL instead of hanging or handing a bad frame back with S_OK.  The context is usable afterwards."""
    import torch
    from tests.test_parity_gpu import make_vp
    from videorenderer_amd import api
    c = dict(ED_CASES["strip_1p5x"], bUseDither=2)
    (x0, y0, x1, y1), (ww, wh) = region(c)
    assert y1 - y0 > 21          # at least two bands
    frame, pitch = case_frame(c)
    vp, _ = make_vp(mpcvr, c)
    dev = torch.from_numpy(frame).cuda()
    good = torch.full((wh, ww, 4), BG, dtype=torch.uint8, device="cuda")
    vp.CopySample(dev, pitch); vp.Process(good, ww * 4); vp.Synchronize()
    monkeypatch.setenv("MPCVR_ERRDIFF_TEST_STALL", "1")
    monkeypatch.setenv("MPCVR_ERRDIFF_SPIN", "16")
    bad = torch.full((wh, ww, 4), BG, dtype=torch.uint8, device="cuda")
    vp.CopySample(dev, pitch); vp.Process(bad, ww * 4)
    with pytest.raises(api.MpcvrError) as e:
        vp.Synchronize()
    assert "gave up" in str(e.value), str(e.value)
    monkeypatch.delenv("MPCVR_ERRDIFF_TEST_STALL"); monkeypatch.delenv("MPCVR_ERRDIFF_SPIN")
    again = torch.full((wh, ww, 4), BG, dtype=torch.uint8, device="cuda")
    vp.CopySample(dev, pitch); vp.Process(again, ww * 4); vp.Synchronize()
    assert torch.equal(again, good)
    # the same deadline through the API (mpcvr_set_error_diffusion_patience: what a host sets instead of the test's environment variable)
    monkeypatch.setenv("MPCVR_ERRDIFF_TEST_STALL", "1")
    vp.SetErrorDiffusionPatience(16)
    vp.CopySample(dev, pitch); vp.Process(bad, ww * 4)
    with pytest.raises(api.MpcvrError) as e:
        vp.Synchronize()
    assert "gave up" in str(e.value), str(e.value)
    monkeypatch.delenv("MPCVR_ERRDIFF_TEST_STALL")
    vp.SetErrorDiffusionPatience(0)                  # back to the default
    vp.CopySample(dev, pitch); vp.Process(again, ww * 4); vp.Synchronize()
    assert torch.equal(again, good)
    vp.close()
