#!/usr/bin/env python3
"""Where do the > 1 LSB channels of the PQ-tail whole-frame cases come from?  (GPU box.)  Runs the Dolby Vision and Catmull-Rom-chroma
whole-frame cases of tests/test_parity_gpu.py on every tier, lists each channel more than one code from the oracle with its position,
both values and the source codes around it, and saves the list (gpurun_out/outliers.json) for an offline look with the oracle."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import oracle as O  # noqa: E402
import videorenderer_amd as V  # noqa: E402
from videorenderer_amd import api  # noqa: E402
from tests.golden.cases import GOLDEN_CASES, case_frame, oracle_params  # noqa: E402
from tests import test_parity_gpu as T  # noqa: E402

BG = 7
CASES = []
base = dict(cformat=2, w=1920, h=1080, kind="hdr", seed=401, dst=(1920, 1080), exfmt=GOLDEN_CASES["dovi_poly_sdr"]["exfmt"])
for label, extra, _ in T.DOVI_FULL:
    if extra.get("output_format", 0) == 1 or "dst" in extra:
        continue
    CASES.append(("dovi_" + label, dict(base, **extra)))
for label, c in T._cr_cases():
    if T.has_tail(dict(c, iChromaScaling=2)) and c["dst"] == (c["w"], c["h"]):
        CASES.append(("cr_" + label, dict(c, iChromaScaling=2)))

out = []
for label, c in CASES:
    frame, pitch = case_frame(c)
    p = oracle_params(O, c)
    want = O.process(p, frame, pitch, dst=np.full((p.window_h, p.window_w, 4), BG, dtype=np.uint8))
    for flags in (api.FLAG_NO_FUSED, 0):
        got, info = T.run_product(V, torch, c, extra_flags=flags)
        d = got[..., :3].astype(np.int16) - want[..., :3].astype(np.int16)
        ys, xs, cs = np.nonzero(np.abs(d) > 1)
        print(f"{label} flags={flags} [{info}]: {len(ys)} channels beyond 1 LSB, max {np.abs(d).max()}")
        y16 = frame.view(np.uint16) if c["cformat"] in (2, 20) else None
        for y, x, ch in list(zip(ys, xs, cs))[:40]:
            rec = dict(case=label, flags=int(flags), x=int(x), y=int(y), ch=int(ch), got=[int(v) for v in got[y, x, :3]], want=[int(v) for v in want[y, x, :3]])
            if y16 is not None and c["cformat"] == 2:
                w, h = c["w"], c["h"]
                Y = y16[: w * h].reshape(h, w); UV = y16[w * h:].reshape(h // 2, w)
                y0, x0 = max(0, (y & ~1) - 2), max(0, (x & ~1) - 2)
                rec["luma"] = (Y[y0:y0 + 6, x0:x0 + 6] >> 6).tolist(); rec["chroma"] = (UV[y0 // 2:y0 // 2 + 3, x0:x0 + 6] >> 6).tolist(); rec["origin"] = [int(x0), int(y0)]
            out.append(rec)
            print("   ", {k: rec[k] for k in ("x", "y", "ch", "got", "want")})
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "outliers.json"), "w"))
