#!/usr/bin/env python3
"""tests/tools/fuzz_errdiff.py [n] [seed] — the error-diffusion pass (EXTENSION, bUseDither = 2) on n random (source format, size, ratio,
scaler, window, video-rect offset / clipping, batch or single frame) combinations: the product twice on the same sample — as the 10-bit
swap chain, and with the pass — and the pass's output against the serial model (oracle.error_diffusion) of the product's own 10-bit
frame, BIT FOR BIT, background untouched.  Shapes the suite does not hold: regions a few columns wide, thousands of rows (bands that
wait for each other at every group), odd first and last columns, batches of frames with their bands interleaved in one launch."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from oracle import oracle
from tests.golden.cases import HDR10, case_frame
from tests.test_parity_gpu import BG, make_vp
from videorenderer_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260926)
paths = collections.Counter()
bad = 0; refused = 0
for i in range(n):
    # sources above 8 bits (the internal format is 10-bit, the pass runs): bi-planar / planar 4:2:0, packed 4:4:4 and 4:2:2, gray, v210 (repacked
    # first), interleaved r210 / RGB48 / BGRA64 (no convert draw: the source texture feeds the resize)
    cf = int(rng.choice([2, 2, 20, 21, 12, 24, 38, 8, 10, 32, 33, 35, 27]))
    shape = rng.random()
    if shape < 0.15: w, h = int(rng.integers(4, 24)) * 2, int(rng.integers(300, 1300)) * 2        # narrow and tall: short bands, long chains
    elif shape < 0.3: w, h = int(rng.integers(300, 900)) * 2, int(rng.integers(4, 40)) * 2         # wide and flat: one or two bands
    else: w, h = int(rng.integers(12, 260)) * 2, int(rng.integers(10, 200)) * 2
    c = dict(cformat=cf, w=w, h=h, kind=str(rng.choice(["noise", "structure", "hdr"])), seed=int(rng.integers(1, 1 << 30)),
             iUpscaling=int(rng.choice([1, 2, 3, 4, 6])), iDownscaling=int(rng.integers(0, 6)), bUseDither=2)
    if cf in (2, 20, 21) and rng.random() < 0.4: c["exfmt"] = HDR10
    if cf == 20 and rng.random() < 0.3:          # Dolby Vision: reshaping in the convert stage, the pass behind it
        c["kind"] = "hdr"; c["dovi"] = dict(kind=str(rng.choice(["poly", "mmr"])))
        c.pop("exfmt", None)
    if rng.random() < 0.15: c["rotation"] = int(rng.choice([90, 180, 270]))
    if rng.random() < 0.15: c["flip"] = 1
    if cf == 10: w = (w + 5) // 6 * 6; c["w"] = w          # v210: six pixels per 16-byte group
    mode = rng.random()
    fx = fy = 1.0 if mode < 0.25 else 2.0 if mode < 0.45 else float(rng.uniform(0.5, 2.2))
    if mode >= 0.45 and rng.random() < 0.6: fy = float(rng.uniform(0.5, 2.2))
    dw, dh = max(4, int(round(w * fx))), max(4, int(round(h * fy)))
    if mode >= 0.45 and dw == w: dw += 1
    c["dst"] = (dw, dh)
    if rng.random() < 0.5:      # a window of its own: offsets of either parity, clipping on any side
        ww, wh = max(8, dw + int(rng.integers(-dw // 3, dw // 3 + 1))), max(8, dh + int(rng.integers(-dh // 3, dh // 3 + 1)))
        c["window"] = (ww, wh); c["offset"] = (int(rng.integers(-dw // 4, ww // 3 + 1)), int(rng.integers(-dh // 4, wh // 3 + 1)))
    (ww, wh) = c.get("window", c["dst"]); ox, oy = c.get("offset", (0, 0))
    x0, y0, x1, y1 = max(ox, 0), max(oy, 0), min(ox + dw, ww), min(oy + dh, wh)
    nb = int(rng.choice([1, 1, 2, 5])) if w * h < 200000 else 1
    try:
        vp10, _ = make_vp(None, dict(c, output_format=1))
        vp8, _ = make_vp(None, c)
    except api.MpcvrError:
        continue
    pitch = case_frame(c)[1]
    frames = [torch.from_numpy(synth.make_frame(cf, w, h, c["kind"], seed=c["seed"] + k)[0]).cuda() for k in range(nb)]
    tens = [torch.full((wh, ww), BG * 0x01010101, dtype=torch.int32, device="cuda") for _ in frames]
    outs = [torch.full((wh, ww, 4), BG, dtype=torch.uint8, device="cuda") for _ in frames]
    try:
        if nb == 1:
            for vp, d in ((vp10, tens[0]), (vp8, outs[0])):
                vp.CopySample(frames[0], pitch); vp.Process(d, ww * 4); vp.Synchronize()
        else:
            vp10.ProcessBatch(frames, tens, ww * 4); vp10.Synchronize()
            vp8.ProcessBatch(frames, outs, ww * 4); vp8.Synchronize()
    except api.MpcvrError as e:         # (a ratio outside the resize kernels' range)
        refused += 1; vp10.close(); vp8.close()
        continue
    info = vp8.GetVPInfo(); paths[info.split(";")[0] + (f" x{nb}" if nb > 1 else "")] += 1
    vp10.close(); vp8.close()
    for k in range(nb):
        got = outs[k].cpu().numpy()
        want = np.full((wh, ww, 4), BG, dtype=np.uint8)
        if x1 > x0 and y1 > y0:
            assert "errdiff" in info, info
            oracle.error_diffusion(tens[k].cpu().numpy().view(np.uint32), (x0, y0, x1, y1), dst=want)
        if not np.array_equal(got, want):
            bad += 1
            print(f"MISMATCH case {i} frame {k}: {c} region {(x0, y0, x1, y1)} [{info}]: {(got != want).any(axis=2).sum()} pixels")
print(f"{n} cases ({refused} refused by the planner), {bad} mismatches; paths:", dict(paths.most_common()))
sys.exit(1 if bad else 0)
