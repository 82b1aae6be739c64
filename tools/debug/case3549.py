"""Soak case 3549 (default mode, seed 4006): HLG -> HDR10 output with tone-mapping operator 2, rotated 270: one channel 71 ten-bit codes from the
plain tier — inside the oracle's own answers under +-4 ulp of pow() (the operator branches there)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from videorenderer_amd import api
from oracle import oracle as O
from tests.golden.cases import case_frame, oracle_params
from tests.test_parity_gpu import run_product, _codes10, BG
c = {'cformat': 24, 'w': 606, 'h': 258, 'kind': 'noise', 'seed': 476592806, 'exfmt': 2185372928, 'iChromaScaling': 2, 'iUpscaling': 4, 'iDownscaling': 5, 'bInterpolateAt50pct': 0, 'dst': (123, 837), 'rotation': 270, 'hdr_output': 1, 'output_format': 1, 'hdr_tonemap': 2, 'hdr_display': 400.0, 'hdr_meta': (0.005, 1000.0, 0.0, 200.0)}
fr, pit = case_frame(c)
p = oracle_params(O, c)
bg = lambda: np.full((p.window_h, p.window_w, 4), BG, np.uint8)
want = _codes10(O.process(p, fr, pit, dst=bg()))
for fl in (api.FLAG_NO_FUSED, 0):
    got, info = run_product(api, torch, c, extra_flags=fl)
    g = _codes10(got); d = np.abs(g - want)
    print(f"flags {fl} [{info}]: differing {int((d > 0).sum())}, beyond 4: {int((d > 4).sum())}, max {int(d.max())}")
    for y, x, ch in np.argwhere(d > 4)[:4]:
        lo, hi = want[y, x, ch], want[y, x, ch]
        for b, chn, seed in [(b, chn, 0) for b in (-1, 1) for chn in (-1, 0, 1, 2)] + [(1, -1, k) for k in range(1, 9)]:
            v = _codes10(O.process_with_tonemap_input_bias(p, fr, pit, b, channel=chn, seed=seed, dst=bg()))[y, x, ch]
            lo, hi = min(lo, v), max(hi, v)
        plo, phi = want[y, x, ch], want[y, x, ch]
        for bias, seed in [(-4, 0), (4, 0)] + [(4, k) for k in range(1, 9)]:
            v = _codes10(O.process_with_pow_bias(p, fr, pit, bias, dst=bg(), seed=seed))[y, x, ch]
            plo, phi = min(plo, v), max(phi, v)
        print("   (y, x, ch) =", (int(y), int(x), int(ch)), "got", g[y, x], "oracle", want[y, x], f"oracle with the operator's input one code off: {lo} .. {hi}; oracle with every pow() +-4 ulp: {plo} .. {phi}")
