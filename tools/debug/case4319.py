"""Soak5 case 4319 (left-out-scalers mode, seed 6701): Dolby Vision MMR + ProcAmp (contrast 1.17, saturation 1.44), 8-bit internal format, nearest 1.7x, 10-bit target."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from videorenderer_amd import api
from oracle import oracle as O
from tests.golden.cases import case_frame, oracle_params
from tests.test_parity_gpu import run_product, _codes10, BG
c = {'cformat': 3, 'w': 598, 'h': 288, 'kind': 'noise', 'seed': 546315119, 'exfmt': 2051155200, 'iChromaScaling': 0, 'iUpscaling': 0, 'iDownscaling': 2, 'bInterpolateAt50pct': 1, 'dst': (1027, 495), 'output_format': 1, 'iTexFormat': 8, 'procamp': (-3.8196396258474365, 1.1730855112312484, -6.805298891179945, 1.4386824544891605), 'dovi': {'kind': 'mmr', 'l2': ()}}
fr, pit = case_frame(c)
lib = os.environ.get("MPCVR_LIB", "default library")[-22:]
for label, cc in (("as found", c), ("contrast 1", dict(c, procamp=(c["procamp"][0], 1.0) + tuple(c["procamp"][2:]))), ("saturation 1", dict(c, procamp=tuple(c["procamp"][:3]) + (1.0,))),
                  ("10-bit internal format", dict(c, iTexFormat=10)), ("same size", dict(c, dst=(598, 288)))):
    p = oracle_params(O, cc)
    want = _codes10(O.process(p, fr, pit, dst=np.full((p.window_h, p.window_w, 4), BG, np.uint8)))
    for fl in (0, api.FLAG_NO_FUSED):
        got, info = run_product(api, torch, cc, extra_flags=fl)
        d = np.abs(_codes10(got) - want)
        print(f"{lib:22s} {label:24s} flags {fl} [{info}]: differing {int((d > 0).sum())}, beyond 5: {int((d > 5).sum())}, max {int(d.max())}")
