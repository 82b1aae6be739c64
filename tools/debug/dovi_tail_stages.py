"""The plain tier's Dolby Vision tail against the oracle's, stage by stage, on random PQ-coded triples (GPU box): where do the two first differ?"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from videorenderer_amd import api, synth
from oracle import oracle as O
L = api.load_library()
rng = np.random.default_rng(11)
n = 2_000_000
rgb = np.concatenate([rng.uniform(0, 1, (n, 3)), rng.uniform(-0.05, 1.2, (n // 4, 3)), rng.uniform(0, 0.1, (n // 4, 3))]).astype(np.float32)
fp = C.POINTER(C.c_float)
for kind, l2 in (("mmr", (100, 600, 1000)), ("poly", ()), ("mixed", (100, 600, 1000))):
    pd = api.plan_dovi(synth.dovi_metadata(kind, l2=l2), 1000)
    lms, k = np.ascontiguousarray(pd["lms"], np.float32), np.ascontiguousarray(pd["l2k"], np.float32)
    print(kind, "l2_enabled", pd["l2_enabled"], "k", k.tolist())
    d_in = torch.from_numpy(rgb).cuda()
    for stage in range(6):
        d_out = torch.empty_like(d_in)
        assert L.mpcvr_eval_dovi_tail(stage, C.c_void_p(d_in.data_ptr()), C.c_void_p(d_out.data_ptr()), rgb.shape[0], lms.ctypes.data_as(fp), k.ctypes.data_as(fp), int(pd["l2_enabled"]), 80.0, None) == 0
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        want = np.empty_like(rgb)
        O.lib().orc_eval_dovi_tail(stage, rgb.ctypes.data, want.ctypes.data, rgb.shape[0], lms.ctypes.data_as(fp), k.ctypes.data_as(fp), int(pd["l2_enabled"]), 80.0)
        nan = np.isnan(want)
        bad = ((got.view(np.uint32) != want.view(np.uint32)) & ~(nan & np.isnan(got)))
        print(f"  stage {stage}: {int(bad.sum())} of {bad.size} values differ; NaNs oracle {int(nan.sum())} device {int(np.isnan(got).sum())}")
        for i, j in np.argwhere(bad)[:4]:
            print("     in", rgb[i].tolist(), "ch", int(j), "device", repr(got[i, j]), "oracle", repr(want[i, j]))
