import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from videorenderer_amd import api
from tests.golden.cases import GOLDEN_CASES, case_frame
from tests.test_parity_gpu import make_vp, BG
import tests.test_parity_gpu as T
name = sys.argv[1]; flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
c = GOLDEN_CASES[name]
mp = api
vp, (ww, wh) = make_vp(mp, c, flags)
frame, pitch = case_frame(c)
dst = torch.full((wh, ww, 4), BG, dtype=torch.uint8, device="cuda")
vp.CopySample(torch.from_numpy(frame).cuda(), pitch)
print("info", vp.GetVPInfo(), flush=True)
vp.Process(dst, ww * 4)
vp.Synchronize()
print("ok", dst.float().mean().item())
