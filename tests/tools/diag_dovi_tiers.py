#!/usr/bin/env python3
"""How many channels of the Dolby Vision whole-frame cases sit more than one code from the oracle, per kernel tier?  (GPU box.)
Prints one JSON line per (case, flags): the count for the library MPCVR_LIB selects (experiment builds of tools/build_variant.sh:
which step of the block convert — the PQ EOTF table, v_rcp_f32 in Hable's quotient — moves the cancelling red channel)."""
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
base = dict(cformat=2, w=1920, h=1080, kind="hdr", seed=401, dst=(1920, 1080), exfmt=GOLDEN_CASES["dovi_poly_sdr"]["exfmt"])
tag = os.path.basename(os.environ.get("MPCVR_LIB", "libmpcvr.so"))
for label, extra, _ in T.DOVI_FULL:
    if extra.get("output_format", 0) == 1 or "dst" in extra:
        continue
    c = dict(base, **extra)
    frame, pitch = case_frame(c)
    p = oracle_params(O, c)
    want = O.process(p, frame, pitch, dst=np.full((p.window_h, p.window_w, 4), BG, dtype=np.uint8))
    for flags in (api.FLAG_NO_FUSED, api.FLAG_NO_FAST_CONVERT, 0):
        got, info = T.run_product(V, torch, c, extra_flags=flags)
        d = np.abs(got[..., :3].astype(np.int16) - want[..., :3].astype(np.int16))
        print(json.dumps({"lib": tag, "case": label, "flags": int(flags), "path": info, "beyond_1lsb": int((d > 1).sum()), "max": int(d.max()),
                          "identical": float((d == 0).mean())}), flush=True)
