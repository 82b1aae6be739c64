"""Soak case 2367 (MPCVR_FUZZ_JINC=1 MPCVR_FUZZ_FLAGS=8, seed 2102): HLG -> one-draw Jinc2m 2x -> HDR10 tone-mapping operator 6 -> R10G10B10A2;
with MPCVR_FLAG_NO_FAST_CONVERT one channel is 7 ten-bit codes where the oracle (and its +-4 ulp pow runs) says 0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from videorenderer_amd import api
from oracle import oracle as O
from tests.golden.cases import case_frame, oracle_params
from tests.test_parity_gpu import run_product, _codes10, BG
np.set_printoptions(linewidth=220)
c = {'cformat': 13, 'w': 290, 'h': 436, 'kind': 'noise', 'seed': 116969516, 'exfmt': 2185372928, 'iChromaScaling': 0, 'iUpscaling': 5, 'iDownscaling': 2, 'bInterpolateAt50pct': 1, 'dst': (580, 872), 'iTexFormat': 10, 'hdr_output': 1, 'output_format': 1, 'hdr_tonemap': 6, 'hdr_display': 400.0, 'hdr_meta': (0.005, 4000.0, 800.0, 200.0)}
fr, pit = case_frame(c)


def run(label, cc, flags):
    p = oracle_params(O, cc)
    want = _codes10(O.process(p, fr, pit, dst=np.full((p.window_h, p.window_w, 4), BG, np.uint8)))
    got, info = run_product(api, torch, cc, extra_flags=flags)
    g = _codes10(got)
    d = np.abs(g - want)
    print(f"{label:44s} flags {flags:3d} [{info}]: differing {int((d > 0).sum())}, beyond 4: {int((d > 4).sum())}, max {int(d.max())}")
    for y, x, ch in np.argwhere(d > 4)[:6]:
        print("      (y, x, ch) =", (int(y), int(x), int(ch)), "got", g[y, x], "oracle", want[y, x], " neighbours got", g[y, max(x - 2, 0):x + 3, ch].tolist(), "oracle", want[y, max(x - 2, 0):x + 3, ch].tolist())
    return g, want


for fl in (0, api.FLAG_NO_FUSED, api.FLAG_NO_FAST_CONVERT, api.FLAG_NO_LUT, api.FLAG_NO_STRIP):
    run("as found", c, fl)
for label, cc in (("no tone mapping operator", {k: v for k, v in c.items() if k not in ("hdr_tonemap", "hdr_display", "hdr_meta")}),
                  ("operator 5", dict(c, hdr_tonemap=5)), ("operator 1", dict(c, hdr_tonemap=1)),
                  ("Lanczos3 instead of Jinc2m", dict(c, iUpscaling=4)), ("same size", dict(c, dst=(290, 436))),
                  ("16-bit float internal format", dict(c, iTexFormat=16)), ("8-bit target", dict(c, output_format=0))):
    try:
        run(label, cc, api.FLAG_NO_FAST_CONVERT)
    except Exception as e:
        print(label, "->", type(e).__name__, str(e)[:200])
