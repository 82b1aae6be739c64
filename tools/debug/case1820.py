"""Fuzz case 1820 of profiles/r05/fuzz_2500_jinc_flags64.txt (Dolby Vision MMR -> two-draw Jinc2m 2x -> R10G10B10A2): the plain tier's
pixel (231, 107) against the oracle's, and the convert texels under its 4 x 4 taps on both sides (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from videorenderer_amd import api
from oracle import oracle as O
from tests.golden.cases import case_frame, oracle_params
from tests.test_parity_gpu import run_product, _codes10, BG
np.set_printoptions(linewidth=220)
c = {'cformat': 3, 'w': 184, 'h': 166, 'kind': 'noise', 'seed': 612561346, 'exfmt': 2051155200, 'iChromaScaling': 0, 'iUpscaling': 5, 'iDownscaling': 1, 'bInterpolateAt50pct': 0, 'src_rect': (76, 36, 184, 166), 'dst': (216, 260), 'window': (236, 284), 'offset': (5, 19), 'output_format': 1, 'dovi': {'kind': 'mmr', 'l2': ()}}
fr, pit = case_frame(c)
p = oracle_params(O, c)
want = _codes10(O.process(p, fr, pit, dst=np.full((p.window_h, p.window_w, 4), BG, np.uint8)))
for flags in (api.FLAG_NO_FUSED, 0):
    got, info = run_product(api, torch, c, extra_flags=flags)
    g = _codes10(got)
    d = np.abs(g - want)
    print(f"flags {flags} [{info}]: max {d.max()}, beyond 0: {(d > 0).sum()}, beyond 2: {(d > 2).sum()} at", np.argwhere(d > 2)[:6].tolist(), "pixel (231,107): got", g[231, 107], "oracle", want[231, 107])
# the convert texels: the same source rect at 1:1 into a 10-bit target without dither = m_TexConvertOutput's codes
cs = dict(c, dst=(108, 130), bUseDither=0); cs.pop("window"); cs.pop("offset"); cs["iUpscaling"] = 2
cv, fmt = O.convert_only(p, fr, pit)
ocv = np.floor(np.clip(cv[..., :3], 0, 1) * 1023 + 0.5).astype(int)
for flags in (api.FLAG_NO_FUSED, 0):
    got, info = run_product(api, torch, cs, extra_flags=flags)
    g = _codes10(got)
    d = np.abs(g - ocv)
    print(f"same-size flags {flags} [{info}]: convert texels vs oracle: max {d.max()}, differing {(d > 0).sum()} of {d.size}; hist {np.bincount(d.ravel())[:12]}")
    print("  texels rows 104..107 x cols 49..52, ch0: product\n", g[104:108, 49:53, 0], "\n  oracle\n", ocv[104:108, 49:53, 0])
