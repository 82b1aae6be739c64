"""Debug helper: the fused 2x kernel's two tap engines side by side on simple synthetic P010 frames (run on the GPU box)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videorenderer_amd import api

def run(frame, w, h, flags, up=4, ex=0):
    vp = api.VideoProcessor(api.default_settings(iUpscaling=up, flags=flags))
    vp.InitMediaType(2, w, h, extfmt=ex)
    vp.SetWindowRect((0, 0, 2 * w, 2 * h)); vp.SetVideoRect((0, 0, 2 * w, 2 * h))
    dst = torch.zeros((2 * h, 2 * w, 4), dtype=torch.uint8, device="cuda")
    vp.CopySample(torch.from_numpy(frame.view(np.uint8)).cuda(), w * 2)
    vp.Process(dst, 2 * w * 4); vp.Synchronize()
    info = vp.GetVPInfo(); vp.close()
    return dst.cpu().numpy(), info

def p010(w, h, y, u, v):
    buf = np.zeros(w * h * 3 // 2, np.uint16)
    buf[:w * h] = (np.broadcast_to(y, (h, w)).astype(np.uint16) << 6).ravel()
    uv = np.zeros((h // 2, w), np.uint16)
    uv[:, 0::2] = np.broadcast_to(u, (h // 2, w // 2)); uv[:, 1::2] = np.broadcast_to(v, (h // 2, w // 2))
    buf[w * h:] = (uv << 6).ravel()
    return buf

w, h = 256, 64
ex = api.make_extfmt(chroma=5, nominal_range=2, matrix=1, primaries=2, transfer=5)
tests = {
    "flat grey": p010(w, h, 500, 512, 512),
    "flat red-ish": p010(w, h, 300, 400, 800),
    "h ramp": p010(w, h, (64 + np.arange(w) * 3)[None, :], 512, 512),
    "v ramp": p010(w, h, (64 + np.arange(h) * 12)[:, None], 512, 512),
}
for up in (4, 2):
    for name, fr in tests.items():
        a, ia = run(fr, w, h, api.FLAG_FUSED_VALU, up, ex)
        b, ib = run(fr, w, h, api.FLAG_FUSED_MFMA, up, ex)
        d = np.abs(a.astype(int) - b.astype(int))
        print(f"up={up} {name:14s} [{ia}|{ib}] max diff {d.max()} differing {100*(d>0).mean():.2f}%")
        if d.max() > 1:
            print("  valu row0 :", a[0, :12, :3].tolist())
            print("  mfma row0 :", b[0, :12, :3].tolist())
            print("  valu col0 :", a[:10, 0, :3].tolist())
            print("  mfma col0 :", b[:10, 0, :3].tolist())
            print("  valu mid  :", a[40, 200:208, :3].tolist())
            print("  mfma mid  :", b[40, 200:208, :3].tolist())
