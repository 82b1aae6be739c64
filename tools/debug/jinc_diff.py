"""Which texel does an output pixel of the fused Jinc2m kernel read with a one-tap filter?  (debug aid: run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import videorenderer_amd as V
from videorenderer_amd import api
api.load_library()
import tests.test_parity_gpu as T
np.set_printoptions(linewidth=250)
def run(c, frame, flags):
    vp, (ww, wh) = T.make_vp(V, c, flags)
    f0, pitch = T.case_frame(c)
    dst = torch.full((wh, ww, 4), T.BG, dtype=torch.uint8, device="cuda")
    vp.CopySample(torch.from_numpy(frame).cuda(), pitch); vp.Process(dst, ww * 4); vp.Synchronize()
    info = vp.GetVPInfo(); out = dst.cpu().numpy(); vp.close()
    return out, info
c = dict(cformat=1, w=64, h=48, kind="noise", seed=5, dst=(128, 96), iUpscaling=5, exfmt=T._SDR)
f0, pitch = T.case_frame(c)
tap = int(os.environ.get("MPCVR_JINC_DBG", "5")); j, i = tap // 4, tap % 4
for name, ygen in (("vramp", lambda y, x: 16 + 4 * y + 0 * x), ("hramp", lambda y, x: 16 + 3 * x + 0 * y)):
    fr = np.full_like(f0, 128)
    yy, xx = np.mgrid[0:48, 0:64]
    fr[:64 * 48] = ygen(yy, xx).astype(np.uint8).reshape(-1)
    got, info = run(c, fr, 0)
    g = got[..., 1].astype(float); Y = g * 219 / 255 + 16
    coord = (Y - 16) / (4 if name == "vramp" else 3)
    print(name, info, "tap (j, i) =", (j, i))
    if name == "vramp":
        col = np.rint(coord[:24, 8]).astype(int); want = np.array([np.clip(oy // 2 - 2 + (oy & 1) + j, 0, 47) for oy in range(24)])
        print(" source row read by output rows 0..23:", col); print(" expected                            :", want)
    else:
        row = np.rint(coord[8, :24]).astype(int); want = np.array([np.clip(ox // 2 - 2 + (ox & 1) + i, 0, 63) for ox in range(24)])
        print(" source col read by output cols 0..23:", row); print(" expected                            :", want)
