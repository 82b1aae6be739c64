#!/usr/bin/env python3
"""tools/ed_probe.py — how long the error-diffusion pass takes on single frames of growing height (1, 2, 4 ... bands of 21 rows): the time of one
band = steps x step time, the increment per band = the lag a band runs behind the band above.  Prints JSON lines."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from videorenderer_amd import api, synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 3840
for h_src in (10, 20, 42, 84, 168, 336, 672, 2160):
    c = dict(cformat=2, w=W, h=h_src, dst=(2 * W, 2 * h_src))
    out = {}
    for dither in (1, 2):
        vp = api.VideoProcessor(api.default_settings(iUpscaling=6, bUseDither=dither), device=0)
        vp.InitMediaType(2, W, h_src, extfmt=0)
        vp.SetWindowRect((0, 0, 2 * W, 2 * h_src)); vp.SetVideoRect((0, 0, 2 * W, 2 * h_src))
        f, pitch = synth.make_frame(2, W, h_src, "noise", seed=5)
        dev = torch.from_numpy(np.ascontiguousarray(f)).cuda()
        dst = torch.zeros((2 * h_src, 2 * W, 4), dtype=torch.uint8, device="cuda")
        vp.CopySample(dev, pitch)
        for _ in range(3):
            vp.Process(dst, 2 * W * 4)
        vp.Synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            vp.Process(dst, 2 * W * 4); vp.Synchronize()
        out[dither] = (time.perf_counter() - t0) / n * 1e3
        vp.close()
    rows = 2 * h_src
    print(json.dumps({"rows": rows, "bands": (rows + 20) // 21, "ms_ordered": round(out[1], 4), "ms_errdiff": round(out[2], 4), "pass_ms": round(out[2] - out[1], 4)}), flush=True)
