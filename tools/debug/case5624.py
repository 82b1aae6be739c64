"""Soak4 cases: 5624 (default mode, seed 5001: PQ -> HDR10 passthrough, rotated 270, two channels 7 ten-bit codes off with no transcendental in the
plan) and 1428 (Jinc2m mode, seed 5102: Dolby Vision + level-2 trims + ProcAmp -> Jinc2m 2x, window clipping: got 6, oracle 0, +-4 ulp interval 0..2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from videorenderer_amd import api
from oracle import oracle as O
from tests.golden.cases import case_frame, oracle_params
from tests.test_parity_gpu import run_product, _codes10, BG
np.set_printoptions(linewidth=200)
cases = {
    "5624": {'cformat': 3, 'w': 574, 'h': 264, 'kind': 'noise', 'seed': 931924931, 'exfmt': 2051155200, 'iChromaScaling': 1, 'iUpscaling': 1, 'iDownscaling': 3, 'bInterpolateAt50pct': 0, 'dst': (125, 1378), 'rotation': 270, 'hdr_output': 1, 'output_format': 1, 'hdr_tonemap': 0, 'hdr_display': 400.0, 'hdr_meta': (0.005, 4000.0, 800.0, 0.0)},
    "1428": {'cformat': 2, 'w': 90, 'h': 216, 'kind': 'noise', 'seed': 897013636, 'exfmt': 2051155200, 'iChromaScaling': 0, 'iUpscaling': 5, 'iDownscaling': 3, 'bInterpolateAt50pct': 0, 'dst': (180, 432), 'window': (171, 432), 'offset': (15, 20), 'procamp': (-3.9689549383766405, 1.1823064992043373, -6.779328347755538, 1.053739126351976), 'dovi': {'kind': 'poly', 'l2': (100, 600, 1000)}},
}
for name, c in cases.items():
    fr, pit = case_frame(c)
    p = oracle_params(O, c)
    ten = c.get("output_format", 0) == 1
    codes = _codes10 if ten else (lambda a: a[..., :3].astype(np.int32))
    want = codes(O.process(p, fr, pit, dst=np.full((p.window_h, p.window_w, 4), BG, np.uint8)))
    lim = 2 if ten else 1
    for fl in (0, api.FLAG_NO_FUSED, api.FLAG_NO_FAST_CONVERT, api.FLAG_NO_LUT, api.FLAG_NO_STRIP, api.FLAG_NO_FAST_CONVERT | api.FLAG_NO_STRIP):
        got, info = run_product(api, torch, c, extra_flags=fl)
        g = codes(got); d = np.abs(g - want)
        print(f"case {name} flags {fl:3d} [{info}]: differing {int((d > 0).sum())}, beyond {lim}: {int((d > lim).sum())}, max {int(d.max())}")
        for y, x, ch in np.argwhere(d > lim)[:5]:
            print("      (y, x, ch) =", (int(y), int(x), int(ch)), "got", g[y, x], "oracle", want[y, x], " column neighbours got", g[max(y - 2, 0):y + 3, x, ch].tolist(), "oracle", want[max(y - 2, 0):y + 3, x, ch].tolist())
    for label, cc in (("no rotation", {k: v for k, v in c.items() if k != "rotation"}), ("8-bit target", dict(c, output_format=0)), ("no procamp", {k: v for k, v in c.items() if k != "procamp"}),
                      ("no level-2 trims", dict(c, dovi=dict(c["dovi"], l2=())) if "dovi" in c else None), ("whole window", {k: v for k, v in c.items() if k not in ("window", "offset")})):
        if cc is None or cc == c: continue
        try:
            pp = oracle_params(O, cc); t10 = cc.get("output_format", 0) == 1
            cd = _codes10 if t10 else (lambda a: a[..., :3].astype(np.int32))
            w2 = cd(O.process(pp, fr, pit, dst=np.full((pp.window_h, pp.window_w, 4), BG, np.uint8)))
            got, info = run_product(api, torch, cc)
            d = np.abs(cd(got) - w2)
            print(f"case {name} {label:18s} [{info}]: beyond {2 if t10 else 1}: {int((d > (2 if t10 else 1)).sum())}, max {int(d.max())}")
        except Exception as e:
            print(f"case {name} {label}: {type(e).__name__} {str(e)[:160]}")
