"""tools/diff_case.py <golden case> [flagsA] [flagsB] — run one golden case through two flag settings on the GPU and show where they differ
(debugging aid: which pixels, which codes)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from videorenderer_amd import api
from tests.golden.cases import GOLDEN_CASES, case_frame
from tests.test_parity_gpu import make_vp, BG
name = sys.argv[1]; fa = int(sys.argv[2]) if len(sys.argv) > 2 else 0; fb = int(sys.argv[3]) if len(sys.argv) > 3 else api.FLAG_NO_FUSED
c = GOLDEN_CASES[name]
outs = []
for fl in (fa, fb):
    vp, (ww, wh) = make_vp(api, c, fl)
    frame, pitch = case_frame(c)
    dst = torch.full((wh, ww, 4), BG, dtype=torch.uint8, device="cuda")
    vp.CopySample(torch.from_numpy(frame).cuda(), pitch)
    vp.Process(dst, ww * 4); vp.Synchronize()
    print(fl, vp.GetVPInfo())
    outs.append(dst.cpu().numpy()); vp.close()
a, b = outs
if c.get("output_format", 0) == 1:
    ua, ub = a.view(np.uint32)[..., 0], b.view(np.uint32)[..., 0]
    ch = lambda u: np.stack([(u >> s) & 1023 for s in (0, 10, 20)], -1).astype(np.int32)
    a, b = ch(ua), ch(ub)
else:
    a, b = a[..., :3].astype(np.int32), b[..., :3].astype(np.int32)
d = a - b
ys, xs, cs = np.nonzero(d)
print("differing channels", len(ys), "of", d.size, "max", np.abs(d).max())
for y, x, k in list(zip(ys, xs, cs))[:40]:
    print(f"  y={y} x={x} ch={k}: A={a[y, x, k]} B={b[y, x, k]}   A px={a[y, x]} B px={b[y, x]}")
print("rows hist", np.bincount(ys, minlength=a.shape[0])[:40])
