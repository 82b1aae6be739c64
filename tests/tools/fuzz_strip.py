#!/usr/bin/env python3
"""tests/tools/fuzz_strip.py [n] [seed] — a wider net than the test suite's geometries for the fused kernels (arbitrary-ratio strip kernel,
exact-2x kernel, one-kernel same-size convert): n random (source format out of all layouts, chroma setting, size, source rect, ratio
per axis, scaler, window offset / clipping, internal format, output format, HDR tagging)
combinations, default planner against the plain kernels (MPCVR_FLAG_NO_FUSED): every channel within 1 LSB (8-bit targets) /
the 10-bit bars of tests/test_parity_gpu.py; every fifth case against the CPU oracle as well — the plain tier BIT FOR BIT on every case
(round 6: behind PQ / HLG / gamma / Dolby Vision tails and through the per-pixel Jinc2m kernel too), the default planner at the fused
tiers' bar.  Prints which kernels the cases went through."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from videorenderer_amd import api
from tests.golden.cases import GOLDEN_CASES, case_frame, oracle_params, HDR10, HLG
from tests.test_parity_gpu import BG
from oracle import oracle
from tests.test_parity_gpu import run_product, compare, compare_rgb10, internal_is_8bit, has_tail, make_vp, POW_ULPS_EOTF_TABLE

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260925)
sdr = GOLDEN_CASES["c1_nv12_bt709_passthrough"]["exfmt"]
# channels beyond the bar behind a PQ / HLG / Dolby Vision tail: each needs its witness (compare_behind_tail); uniform noise in Y, U, V is
# mostly out-of-gamut saturated colour — the worst case for the cancelling 2020 -> 709 row — so the count allowed per frame is the
# suite's Dolby Vision rate (16 per M pixels, twice what round 3 measured on its hardest frames), at least 3
# Round 6 (the plain tier now carries the oracle's bits, so these counts are the fused tiers' own v_log_f32 / v_exp_f32 error and nothing else;
# 9 x 8,000 cases of the default mode + the soak: 49 runs in every mode, 259,000 cases, profiles/r06/fuzz_8000*.txt, profiles/r06/soak/): the largest counts
# seen were 5 on a 0.023 M-pixel frame (HLG, 10-bit target), 79 per M pixels behind an 8-bit internal format (PQ, case 6005 of seed 2718: the
# convert's texel is rounded to 8 bits, then each resize pass rounds again — a flipped texel reaches several outputs) and 70 per M pixels on a
# rotated Dolby Vision frame with level-2 trims (case 7172 of seed 1017; uniform noise in Y, U, V is mostly saturated out-of-gamut colour, the
# worst case for the cancelling 2020 -> 709 row).  Every such channel has its witness; the COUNT guards against a systematic offset hiding
# behind the witness, and is capped at 1.5 x the largest rate seen: 120 per M pixels (4e-5 of the channels), at least 8 per frame.  The
# largest count of a run is printed at its end.
def FUZZ_CAP(img, c=None):
    return max(8, int(np.ceil(120 * img.shape[0] * img.shape[1] / 1e6)))
witnessed = []      # (channels beyond the bar, per M pixels, 8-bit internal format, case index)
plain_stats = collections.Counter(); paths = collections.Counter(); worst = 0.0; refused = 0; outliers = 0; batches = 0; oracle_cases = 0
for i in range(n):
    # every source layout: 4:2:0 weighted up, then planar / packed 4:2:2 and 4:4:4, gray, GBRP, one interleaved RGB
    if rng.random() < 0.55:
        cf = int(rng.choice([1, 2, 20, 17, 14, 21, 3]))
    else:
        cf = int(rng.choice([4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16, 18, 19, 22, 23, 24, 25, 26, 27, 28, 30, 30, 32, 37, 38, 39]))
    w, h = int(rng.integers(12, 330)) * 2, int(rng.integers(10, 230)) * 2
    c = dict(cformat=cf, w=w, h=h, kind="noise", seed=int(rng.integers(1, 1 << 30)),
             exfmt=int(rng.choice([sdr, sdr, HDR10, HLG])) if cf in (2, 3, 20, 21, 6, 7, 8, 9, 10, 12, 13, 22, 23, 24, 25) else sdr,
             iChromaScaling=int(rng.choice([0, 1, 1, 1, 2])),
             iUpscaling=int(rng.choice([1, 2, 3, 4])), iDownscaling=int(rng.integers(0, 6)), bInterpolateAt50pct=int(rng.integers(0, 2)))
    rw, rh = w, h
    if cf in (27, 28, 26, 30, 32): c.pop("exfmt")
    if rng.random() < (0.9 if os.environ.get("MPCVR_FUZZ_UNALIGNED") else 0.4) and cf not in (30, 32):
        l = int(rng.integers(0, w // 8)) * 4; t = int(rng.integers(0, h // 8)) * 2
        r = min(w, l + max(16, int(rng.integers(w // 2, w)) // 2 * 2)); b = min(h, t + max(16, int(rng.integers(h // 2, h)) // 2 * 2))
        if os.environ.get("MPCVR_FUZZ_UNALIGNED"):      # source rects on ANY pixel (the default mode keeps them on the 4 x 2 grid the vectorised loaders want)
            jr = np.random.default_rng(c["seed"] ^ 0x5a5a)
            l = min(l + int(jr.integers(0, 4)), w - 18); t = min(t + int(jr.integers(0, 2)), h - 18)
            r = max(l + 16, min(w, r - int(jr.integers(0, 4)))); b = max(t + 16, min(h, b - int(jr.integers(0, 2))))
        c["src_rect"] = (l, t, r, b); rw, rh = r - l, b - t
    fx, fy = float(rng.uniform(0.4, 2.7)), float(rng.uniform(0.4, 2.7))
    if rng.random() < 0.2: fy = fx
    if os.environ.get("MPCVR_FUZZ_SCALERS"):    # the scalers the default mode leaves out: nearest (0), Jinc2m (5), the Spline36 extension (6)
        c["iUpscaling"] = int(np.random.default_rng(c["seed"]).choice([0, 5, 6, 6]))
    if os.environ.get("MPCVR_FUZZ_JINC"):       # every case with the one-draw 2-D scaler, most of them at exactly 2x (the fused Jinc2m kernel)
        c["iUpscaling"] = 5
    mode = rng.random()              # 12 % same size (block convert), 12 % exactly 2x (fused_up2x), 20 % a periodic row ratio, else any ratio (strip kernel)
    if os.environ.get("MPCVR_FUZZ_JINC") and mode < 0.75: mode = 0.2
    if mode < 0.12: fx = fy = 1.0
    elif mode < 0.24: fx = fy = 2.0
    dw, dh = max(8, int(round(rw * fx))), max(8, int(round(rh * fy)))
    periodic_only = bool(os.environ.get("MPCVR_FUZZ_PERIODIC"))     # every case at a periodic row ratio (rows cropped to a multiple of Q)
    if periodic_only: mode = 0.3
    if 0.24 <= mode < 0.44:          # k_fused_period: output rows : source rows exactly 4:3 / 3:2 / 2:3 / 1:2 / 3:1, any ratio along the rows
        P_, Q_ = [(4, 3), (3, 2), (2, 3), (1, 2), (3, 1)][int(rng.integers(0, 5))]
        if periodic_only and rh % Q_ and rh - rh % (2 * Q_) >= 16:
            rh -= rh % (2 * Q_)
            l_, t_, r_, b_ = c.get("src_rect", (0, 0, w, h))
            c["src_rect"] = (l_, t_, r_, t_ + rh)
        if rh % Q_ == 0:
            dh = rh * P_ // Q_
            if rng.random() < 0.7: dw = max(8, int(round(rw * P_ / Q_)))
            c["bInterpolateAt50pct"] = 1
    if mode >= 0.24:
        if dw == rw: dw += 1
        if dh == rh: dh += 1
    c["dst"] = (dw, dh)
    if rng.random() < 0.35:
        c["window"] = (max(8, dw + int(rng.integers(-20, 40))), max(8, dh + int(rng.integers(-20, 40)))); c["offset"] = (int(rng.integers(-15, 25)), int(rng.integers(-15, 25)))
    if rng.random() < 0.2: c["output_format"] = 1
    if rng.random() < 0.2: c["iTexFormat"] = int(rng.choice([8, 10, 16]))
    # (Jinc2m behind a PQ / HLG tail with a FORCED 8-bit internal format — a setting AUTO never picks for such sources: the filter's weights sum
    # to |w| = 1.9, so the one 8-bit code by which two tiers' pow() runs differ comes out as two on every tier, the oracle's own +-4 ulp runs
    # included; tests/test_parity_gpu.py does not sweep that corner either)
    if c.get("iUpscaling") == 5 and c.get("iTexFormat") == 8 and c.get("exfmt") in (HDR10, HLG): c["iTexFormat"] = 10
    if rng.random() < 0.1: c["bUseDither"] = 0
    # the rarer switches of the sequencer: rotation / flip (first draw), ProcAmp, blend deinterlace, HDR output modes, Dolby Vision
    if rng.random() < 0.10: c["rotation"] = int(rng.choice([90, 180, 180, 270]))
    if rng.random() < 0.10: c["flip"] = 1
    if rng.random() < 0.08: c["procamp"] = (float(rng.uniform(-20, 20)), float(rng.uniform(0.8, 1.2)), float(rng.uniform(-30, 30)), float(rng.uniform(0.5, 1.5)))
    if rng.random() < 0.05: c["bDeintBlend"] = 1; c["sample_format"] = int(rng.choice([1, 2]))
    hdr_src = c.get("exfmt") in (HDR10, HLG)
    if hdr_src and rng.random() < 0.2:
        c["hdr_output"] = 1; c["output_format"] = 1; c["hdr_tonemap"] = int(rng.integers(0, 7)); c["hdr_display"] = float(rng.choice([400.0, 1000.0]))
        # (the tone-mapping step exists only with valid HDR10 metadata — SourceIsHDR() / m_lastHdr10.bValid; the oracle is told the type
        # directly, so a case that asks for an operator always carries the metadata)
        if c["hdr_tonemap"] or rng.random() < 0.5:
            c["hdr_meta"] = (0.005, float(rng.choice([1000.0, 4000.0])), float(rng.choice([0.0, 800.0, 2000.0])), float(rng.choice([0.0, 200.0])))
    if cf in (2, 3) and c.get("exfmt") == HDR10 and "hdr_output" not in c and rng.random() < 0.25:
        c["dovi"] = dict(kind=str(rng.choice(["poly", "mmr", "mixed"])), l2=(100, 600, 1000) if rng.random() < 0.5 else ())
    if c.get("rotation") in (90, 270):
        c["dst"] = (c["dst"][1], c["dst"][0])       # (keeps the ratios of the two axes in the range drawn above)
        c.pop("window", None); c.pop("offset", None)
    try:
        plain, _ = run_product(api, torch, c, extra_flags=api.FLAG_NO_FUSED)
        got, info = run_product(api, torch, c, extra_flags=int(os.environ.get("MPCVR_FUZZ_FLAGS", "0")), host_upload=bool(os.environ.get("MPCVR_FUZZ_HOST")))      # (MPCVR_FUZZ_HOST: the sample through mpcvr_copy_sample's host path — pinned ring, copy stream, device repacks)
    except api.MpcvrError:
        refused += 1; continue
    parts = info.split(";")
    kern = [q for q in parts if q.startswith("kernel=")]
    paths[kern[0].split("(")[0] if kern else parts[0].split("+")[0] + "".join(";" + q for q in parts[1:] if q.startswith("rot"))] += 1
    if i % 4 == 0:      # every fourth case also as a batch: mpcvr_process_batch of three distinct frames == three mpcvr_process calls, bit for bit
        vp, (ww, wh) = make_vp(api, c, int(os.environ.get("MPCVR_FUZZ_FLAGS", "0")))
        frames = [torch.from_numpy(case_frame(dict(c, seed=c["seed"] + 7 * k))[0]).cuda() for k in range(3)]
        pitch = vp.GetFrameBytes()[1]
        singles = []
        for f in frames:
            dst = torch.full((wh, ww, 4), 77, dtype=torch.uint8, device="cuda")
            vp.CopySample(f, pitch); vp.Process(dst, ww * 4); singles.append(dst)
        dsts = [torch.full((wh, ww, 4), 77, dtype=torch.uint8, device="cuda") for _ in frames]
        vp.ProcessBatch(frames, dsts, ww * 4); vp.Synchronize()
        for k in range(3):
            assert torch.equal(singles[k], dsts[k]), f"batch != single, frame {k}: fuzz {i} [{info}] {c}"
        vp.close(); batches += 1
    name = f"fuzz {i} [{info}] {c}"
    # own statistics instead of the tests' asserts: behind a PQ / HLG / gamma tail a saturated dark colour can sit where
    # pow(x, 1/2.2) has a slope of thousands (Dolby Vision parity note); such a channel is counted, not fatal
    if c.get("output_format", 0) == 1:
        g, p_ = got.view(np.uint32)[..., 0], plain.view(np.uint32)[..., 0]
        d = np.stack([np.abs(((g >> sh) & 1023).astype(np.int32) - ((p_ >> sh) & 1023).astype(np.int32)) for sh in (0, 10, 20)], -1)
        # (Dolby Vision on a 10-bit target: the block convert decodes PQ from a table where the plain kernel runs the literal pow chain —
        # one 8-bit code = 4 ten-bit codes is the bar the whole-frame DoVi test holds it to)
        # (an 8-bit internal format in front of an HDR10 tone-mapping operator: one 8-bit code of the intermediate goes through a curve of
        # slope > 1 before it is rounded to ten bits)
        # (a 10-bit internal format in front of an operator: one code of the intermediate — the fused tiers' own bar — times a slope of ~3)
        lim = (12 if c.get("hdr_tonemap") else 5) if internal_is_8bit(c) else 4 if ("dovi" in c or c.get("hdr_tonemap")) else 2 if has_tail(c) else 1
        if c.get("iUpscaling") == 5: lim = max(lim, 2)      # (Jinc2m's weights sum to |w| = 1.9: one code of the 10-bit texture comes out as up to two)
    else:
        d = np.abs(got[..., :3].astype(np.int32) - plain[..., :3].astype(np.int32)); lim = 1
    if i % 5 == 0:      # every fifth case against the CPU oracle as well: the plain tier bit-exact without a transcendental tail,
        # within one code of the target format behind one; the default planner's distance to the oracle at the fused tiers' bar
        want = oracle.process(oracle_params(oracle, c), *case_frame(c), dst=np.full((got.shape[0], got.shape[1], 4), BG, dtype=np.uint8))
        def dist(a, b):
            if c.get("output_format", 0) == 1:
                a32, b32 = a.view(np.uint32)[..., 0], b.view(np.uint32)[..., 0]
                return np.stack([np.abs(((a32 >> sh) & 1023).astype(np.int32) - ((b32 >> sh) & 1023).astype(np.int32)) for sh in (0, 10, 20)], -1)
            return np.abs(a[..., :3].astype(np.int32) - b[..., :3].astype(np.int32))
        dp, dg = dist(plain, want), dist(got, want)
        oracle_cases += 1
        # round 6: the plain tier evaluates the shader transcendentals as the oracle defines them (csrc/vp_crmath.h), so it is held to the
        # oracle BIT FOR BIT on every case, behind a PQ / HLG / gamma / Dolby Vision tail and through the per-pixel Jinc2m kernel too
        if os.environ.get("MPCVR_FUZZ_PLAIN_STATS"):
            plain_stats[int(dp.max())] += 1
            if dp.max(): print(f"  plain tier != oracle: max {int(dp.max())}, {int((dp > 0).sum())} channels: {name}")
        else:
            assert dp.max() == 0, f"plain tier vs oracle: max {int(dp.max())}, {int((dp > 0).sum())} channels: {name}"
        if not has_tail(c):
            # (default planner, no tail: within the bar on every channel.  Until round 4 a convert texel one code off the oracle's — the fused tiers
            # contracted a*b + c, 1e-4 of the texels of an 8-bit internal format — could leave a Catmull-Rom / Lanczos tap sum two codes off, 1 - 3
            # channels per ~1e6, and this tool counted them; round 5 gave 8-bit internal formats the exact form of the convert stage
            # (convert_block_exact).  A 10-bit target behind a 10-bit internal format keeps the fast form: lim there is in ten-bit codes.)
            n_over = int((dg > lim).sum())
            over_ok = 0 if (c.get("output_format", 0) != 1 or internal_is_8bit(c)) else 4        # (10-bit codes of a 10-bit target: a quarter of the 8-bit bar each)
            assert dg.max() <= lim + (1 if over_ok else 0) and n_over <= over_ok and float((dg == 0).mean()) >= 0.97, f"default planner vs oracle: max {int(dg.max())}, {n_over} channels beyond {lim}: {name}"
            if n_over:
                print(f"  ten-bit target, 10-bit internal format: {n_over} channel(s) at {int(dg.max())} ten-bit codes vs the oracle in {name}")
        elif c.get("output_format", 0) == 1:        # 10-bit targets behind a tail: the suite's compare_rgb10 bar (<= 2 ten-bit codes, 5 with 8-bit
            # intermediates), or — per channel — inside the oracle's own +-4 ulp pow() interval
            from tests.test_parity_gpu import compare_behind_tail
            po = oracle_params(oracle, c)
            fr, pit = case_frame(c)
            _, nb = compare_behind_tail(oracle, po, fr, pit, got, want, f"default planner vs oracle (tail, 10-bit): {name}", min_same=0.97, ten_bit=True, lim=lim, cap=FUZZ_CAP(want, c), operator_input=bool(c.get('hdr_tonemap')), convert_output=not internal_is_8bit(c), pow_ulps=(POW_ULPS_EOTF_TABLE if 'dovi' in c else None))
            if nb: witnessed.append((nb, nb / (want.shape[0] * want.shape[1] / 1e6), internal_is_8bit(c), i))
        else:                                       # 8-bit targets: <= 1 LSB, or the per-channel witness (the oracle's own +-4 ulp pow() interval)
            from tests.test_parity_gpu import compare_behind_tail
            po = oracle_params(oracle, c)
            fr, pit = case_frame(c)
            _, nb = compare_behind_tail(oracle, po, fr, pit, got, want, f"default planner vs oracle (tail): {name}", min_same=0.97, cap=FUZZ_CAP(want, c), operator_input=bool(c.get('hdr_tonemap')), convert_output=not internal_is_8bit(c), pow_ulps=(POW_ULPS_EOTF_TABLE if 'dovi' in c else None))
            if nb: witnessed.append((nb, nb / (want.shape[0] * want.shape[1] / 1e6), internal_is_8bit(c), i))
    beyond = int((d > lim).sum()); same = float((d == 0).mean())
    worst = max(worst, 1.0 - same)
    if beyond:
        outliers += 1
        print(f"  outlier: {beyond} channel(s) beyond {lim} (max {int(d.max())}) in {name}")
    assert same >= 0.97, name
    # (without a tail: a block-convert texel one code off its plain-kernel value can come out of a Lanczos tap sum 1.2 codes off,
    # i.e. two 10-bit codes after both roundings — seen once per ~1e6 channels; never beyond lim + 1)
    worst_ok = (8 * (4 if c.get("output_format", 0) == 1 else 1)) if has_tail(c) else lim + 1      # ill-conditioned channels: 8 eight-bit codes
    if not has_tail(c) and (c.get("output_format", 0) != 1 or internal_is_8bit(c)):
        assert beyond == 0, name          # (round 5: nothing beyond the bar where no transcendental decides the last code)
    if beyond and has_tail(c):
        # Behind a tail a channel beyond the bar needs its WITNESS, case by case (round 6: until round 5 this was a blanket count — at most 4 per frame —
        # which held while the plain tier shared the fused tiers' v_log_f32 / v_exp_f32; now that the plain tier carries the oracle's bits the
        # comparison is made against the oracle itself: every such channel inside the oracle's own +-4 ulp pow() interval, their number capped)
        from tests.test_parity_gpu import compare_behind_tail
        po = oracle_params(oracle, c)
        fr, pit = case_frame(c)
        want_w = oracle.process(po, fr, pit, dst=np.full((got.shape[0], got.shape[1], 4), BG, dtype=np.uint8))
        _, nb = compare_behind_tail(oracle, po, fr, pit, got, want_w, f"default planner vs oracle (witness for the outliers): {name}", min_same=0.97,
                                    ten_bit=c.get("output_format", 0) == 1, lim=lim, cap=FUZZ_CAP(want_w, c), operator_input=bool(c.get('hdr_tonemap')), convert_output=not internal_is_8bit(c), pow_ulps=(POW_ULPS_EOTF_TABLE if 'dovi' in c else None))
        if nb and i % 5: witnessed.append((nb, nb / (want_w.shape[0] * want_w.shape[1] / 1e6), internal_is_8bit(c), i))
        # how FAR such a channel may lie is bounded where a pow() feeds a smooth tail (a few codes: worst_ok).  Behind an HDR10 tone-mapping
        # operator there is no bound to offer: the operators branch (soak case 3549 of seed 4006, operator 2: the ORACLE's red moves from 511 to
        # 582 under +-4 ulp of pow(), profiles/r06/case3549.txt) and map a code of their input with any slope near black — the witness is the check
        if not c.get("hdr_tonemap"):
            assert d.max() <= worst_ok, name
    else:
        assert beyond == 0 or (beyond <= max(4, 2e-5 * d.size) and d.max() <= worst_ok), name
if witnessed:
    for eight in (False, True):
        w = [t for t in witnessed if t[2] == eight]
        if w: print(f"channels beyond the bar, each inside the oracle's +-4 ulp pow() interval ({'8' if eight else '10 / 16'}-bit internal format): {len(w)} cases, "
                    f"largest count {max(w)[0]} (case {max(w)[3]}), largest rate {max(t[1] for t in w if t[0] > 5) if any(t[0] > 5 for t in w) else 0:.1f} per M pixels among counts > 5")
if plain_stats: print("plain tier vs oracle, max |delta| -> cases:", dict(sorted(plain_stats.items())))
print("cases", n, "of which also as 3-frame batches", batches, "against the CPU oracle", oracle_cases, "refused", refused, "kernels", dict(paths), "largest differing fraction", round(worst, 5), "cases with an ill-conditioned channel", outliers)
