"""Fuzz case 6375 of the round-6 final fuzz_8000 run: plain tier != oracle on ONE channel (Dolby Vision MMR + level-2 trims + ProcAmp, same size)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from videorenderer_amd import api
from oracle import oracle as O
from tests.golden.cases import case_frame, oracle_params
from tests.test_parity_gpu import run_product, _codes10, BG
np.set_printoptions(linewidth=220)
c = {'cformat': 2, 'w': 240, 'h': 444, 'kind': 'noise', 'seed': 598584363, 'exfmt': 2051155200, 'iChromaScaling': 1, 'iUpscaling': 1, 'iDownscaling': 3, 'bInterpolateAt50pct': 0, 'src_rect': (92, 78, 240, 444), 'dst': (148, 366), 'procamp': (-5.010828386886953, 1.1960476848026689, 26.2518030673696, 0.8615709989981972), 'dovi': {'kind': 'mmr', 'l2': (100, 600, 1000)}}
fr, pit = case_frame(c)
for variant, cc in (("8-bit target + final pass", c), ("10-bit target, no dither", dict(c, output_format=1, bUseDither=0))):
    p = oracle_params(O, cc)
    want = O.process(p, fr, pit, dst=np.full((p.window_h, p.window_w, 4), BG, np.uint8))
    got, info = run_product(api, torch, cc, extra_flags=api.FLAG_NO_FUSED)
    if cc.get("output_format") == 1:
        g, w = _codes10(got), _codes10(want)
    else:
        g, w = got[..., :3].astype(int), want[..., :3].astype(int)
    d = np.abs(g - w)
    print(variant, f"[{info}]: differing channels {int((d > 0).sum())}, max {int(d.max())}")
    for y, x, ch in np.argwhere(d > 0)[:8]:
        print("   (y, x, ch) =", (int(y), int(x), int(ch)), "got", g[y, x], "oracle", w[y, x])
        sy, sx = y + 78, x + 92
        Y = fr.view(np.uint16)[:240 * 444].reshape(444, 240); UV = fr.view(np.uint16)[240 * 444:].reshape(222, 240)
        print("   luma code", int(Y[sy, sx]) >> 6, "chroma row", sy // 2, "U,V codes around", (UV[max(sy // 2 - 1, 0):sy // 2 + 2, (sx // 2) * 2 - 2:(sx // 2) * 2 + 4] >> 6).tolist())

# which ingredient makes the difference?
from tests.test_parity_gpu import make_vp
base = dict(c, output_format=1, bUseDither=0)
for label, cc in (("as found", base), ("no procamp", {k: v for k, v in base.items() if k != "procamp"}), ("no level-2 trims", dict(base, dovi={'kind': 'mmr', 'l2': ()})),
                  ("nearest chroma", dict(base, iChromaScaling=0)), ("poly curves", dict(base, dovi={'kind': 'poly', 'l2': (100, 600, 1000)})),
                  ("whole frame, no source rect", {k: v for k, v in dict(base, dst=(240, 444)).items() if k != "src_rect"}),
                  ("no Dolby Vision (HDR10)", {k: v for k, v in base.items() if k != "dovi"})):
    p = oracle_params(O, cc)
    want = _codes10(O.process(p, fr, pit, dst=np.full((p.window_h, p.window_w, 4), BG, np.uint8)))
    got, info = run_product(api, torch, cc, extra_flags=api.FLAG_NO_FUSED)
    d = np.abs(_codes10(got) - want)
    vp, _ = make_vp(api, cc, api.FLAG_NO_FUSED)
    cm = np.array(vp.GetColorMatrix(), np.float32); vp.close()
    ocm = np.array(O.color_matrix(p), np.float32).ravel()
    print(f"{label:32s} differing channels {int((d > 0).sum())} max {int(d.max())}; colour matrix product == oracle: {np.array_equal(cm.view(np.uint32), ocm.view(np.uint32))}", "" if np.array_equal(cm, ocm) else (cm - ocm).tolist())
