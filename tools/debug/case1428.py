"""Soak case 1428 with whatever library MPCVR_LIB names: the default tier against the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from videorenderer_amd import api
from oracle import oracle as O
from tests.golden.cases import case_frame, oracle_params
from tests.test_parity_gpu import run_product, BG, FUZZ_1428 as c
fr, pit = case_frame(c)
p = oracle_params(O, c)
want = O.process(p, fr, pit, dst=np.full((p.window_h, p.window_w, 4), BG, np.uint8))[..., :3].astype(int)
for label, cc in (("as found", c), ("brightness 0", dict(c, procamp=(0.0,) + tuple(c["procamp"][1:]))), ("contrast 1", dict(c, procamp=(c["procamp"][0], 1.0) + tuple(c["procamp"][2:]))),
                  ("hue 0", dict(c, procamp=tuple(c["procamp"][:2]) + (0.0, c["procamp"][3]))), ("saturation 1", dict(c, procamp=tuple(c["procamp"][:3]) + (1.0,))),
                  ("Lanczos3 instead of Jinc2m", dict(c, iUpscaling=4)), ("same size", {k: v for k, v in dict(c, dst=(90, 216)).items() if k not in ("window", "offset")})):
    pp = oracle_params(O, cc)
    w = O.process(pp, fr, pit, dst=np.full((pp.window_h, pp.window_w, 4), BG, np.uint8))[..., :3].astype(int)
    got, info = run_product(api, torch, cc)
    d = np.abs(got[..., :3].astype(int) - w)
    print(f"{os.environ.get('MPCVR_LIB', 'default library')[-24:]:24s} {label:28s} [{info}]: beyond 1: {int((d > 1).sum())}, max {int(d.max())}")
