"""One case through every tier against the oracle (debug aid: run on the GPU box).  usage: case_tiers.py "<dict literal>" """
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import videorenderer_amd as V
from videorenderer_amd import api
api.load_library()
from oracle import oracle as O
import tests.test_parity_gpu as T
from tests.golden.cases import case_frame, oracle_params
np.set_printoptions(linewidth=250)
c = eval(sys.argv[1])
frame, pitch = case_frame(c)
p = oracle_params(O, c)
want = O.process(p, frame, pitch, dst=np.full((p.window_h, p.window_w, 4), T.BG, dtype=np.uint8))
for name, flags in (("default", 0), ("NO_FAST_CONVERT", api.FLAG_NO_FAST_CONVERT), ("NO_STRIP", api.FLAG_NO_STRIP), ("NO_FUSED", api.FLAG_NO_FUSED)):
    got, info = T.run_product(V, torch, c, extra_flags=flags)
    if c.get("output_format", 0) == 1:      # R10G10B10A2: ten-bit codes
        g, w = got.view(np.uint32)[..., 0], want.view(np.uint32)[..., 0]
        d = np.stack([np.abs(((g >> sh) & 1023).astype(int) - ((w >> sh) & 1023).astype(int)) for sh in (0, 10, 20)], -1).max(axis=2)
    else:
        d = np.abs(got[..., :3].astype(int) - want[..., :3].astype(int)).max(axis=2)
    print("   histogram of |delta|:", np.bincount(d.reshape(-1))[:8])
    bad = d > 1
    print(f"{name:16s} [{info}] max {d.max()} beyond 1: {int(bad.sum())} of {d.size}", "rows", np.nonzero(bad.any(axis=1))[0][:12], "cols", np.nonzero(bad.any(axis=0))[0][:12])
