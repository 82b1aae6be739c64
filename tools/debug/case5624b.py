"""Case 5624: where is the convert texel that differs?  The same frame converted at the same size (no draw / a rotated copy draw), default tier against the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from videorenderer_amd import api
from oracle import oracle as O
from tests.golden.cases import case_frame, oracle_params
from tests.test_parity_gpu import run_product, _codes10, BG
np.set_printoptions(linewidth=220)
c0 = {'cformat': 3, 'w': 574, 'h': 264, 'kind': 'noise', 'seed': 931924931, 'exfmt': 2051155200, 'iChromaScaling': 1, 'iUpscaling': 1, 'iDownscaling': 3, 'bInterpolateAt50pct': 0, 'dst': (125, 1378), 'rotation': 270, 'hdr_output': 1, 'output_format': 1, 'hdr_tonemap': 0, 'hdr_display': 400.0, 'hdr_meta': (0.005, 4000.0, 800.0, 0.0)}
fr, pit = case_frame(c0)
print("frame bytes", fr.size, "pitch", pit)
for label, c in (("same size, no rotation", {k: v for k, v in dict(c0, dst=(574, 264)).items() if k != "rotation"}), ("same size, rotated 270", dict(c0, dst=(264, 574))),
                 ("as found but nearest chroma", dict(c0, iChromaScaling=0)), ("as found but Catmull-Rom chroma", dict(c0, iChromaScaling=2)),
                 ("x-only resize, no rotation", {k: v for k, v in dict(c0, dst=(1378, 264)).items() if k != "rotation"}),
                 ("y-only resize, no rotation", {k: v for k, v in dict(c0, dst=(574, 125)).items() if k != "rotation"})):
    p = oracle_params(O, c)
    want = _codes10(O.process(p, fr, pit, dst=np.full((p.window_h, p.window_w, 4), BG, np.uint8)))
    for fl in (0, api.FLAG_NO_FAST_CONVERT):
        got, info = run_product(api, torch, c, extra_flags=fl)
        g = _codes10(got); d = np.abs(g - want)
        print(f"{label:34s} flags {fl} [{info}]: differing {int((d > 0).sum())}, max {int(d.max())}")
        for y, x, ch in np.argwhere(d > 1)[:6]:
            print("      (y, x, ch) =", (int(y), int(x), int(ch)), "got", g[y, x], "oracle", want[y, x])
# the sample around source column 546 / 28, rows 65 / 198 (where the rotated output pixel comes from)
w, h = 574, 264
Y = fr[:w * h * 2].view(np.uint16).reshape(h, w) if pit == w * 2 else None
print("luma plane view:", None if Y is None else Y.shape)
