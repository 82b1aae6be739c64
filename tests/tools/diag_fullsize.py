"""GPU diagnostic: locate fused-vs-general mismatches > 1 LSB at full size and compare both to the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from videorenderer_amd import api
from oracle import oracle as O
from tests.golden.cases import GOLDEN_CASES, case_frame
from tests.test_parity_gpu import make_vp

exfmt = GOLDEN_CASES["c3hdr_p010_pq_lanczos3_2x"]["exfmt"]
up = int(sys.argv[1]) if len(sys.argv) > 1 else 4
c = dict(cformat=2, w=3840, h=2160, kind="noise", seed=77, dst=(7680, 4320), exfmt=exfmt, iUpscaling=up)
frame, pitch = case_frame(c)
dev = torch.from_numpy(frame).cuda()
outs = {}
for flags in (0, 2, 4, 8):
    vp, (ww, wh) = make_vp(None, c, flags)
    dst = torch.empty((wh, ww, 4), dtype=torch.uint8, device="cuda")
    vp.CopySample(dev, pitch); vp.Process(dst, ww * 4); vp.Synchronize()
    outs[flags] = dst.cpu().numpy(); print(flags, vp.GetVPInfo()); vp.close()
g = outs[2].astype(np.int16)
for f in (0, 4, 8):
    d = np.abs(outs[f].astype(np.int16) - g)
    print("flags", f, "vs general: max", d.max(), "count>1", int((d > 1).sum()), "same", float((d == 0).mean()))
d = np.abs(outs[0].astype(np.int16) - g)
pos = np.argwhere(d > 1)
print("positions (y,x,ch) first 20:", pos[:20].tolist())
if len(pos):
    print("x%240:", sorted(set((pos[:, 1] % 240).tolist()))[:40])
    print("y%144:", sorted(set((pos[:, 0] % 144).tolist()))[:40], "y%16", sorted(set((pos[:,0]%16).tolist())))
    y, x, ch = pos[0]
    sx0 = max(0, (x // 2 - 64) // 16 * 16); sy0 = max(0, (y // 2 - 32) // 16 * 16)
    cw, chh = 128, 64
    p = O.default_params(cformat=2, width=3840, height=2160, exfmt=exfmt, iUpscaling=up, src_rect=(sx0, sy0, sx0 + cw, sy0 + chh),
                         window_w=2 * cw, window_h=2 * chh, video_rect=(0, 0, 2 * cw, 2 * chh))
    want = O.process(p, frame, pitch)
    ly, lx = y - 2 * sy0, x - 2 * sx0
    print("at", (y, x, ch), "oracle", want[ly, lx], "general", outs[2][y, x], "fused", outs[0][y, x], "nolut", outs[4][y, x], "slowconv", outs[8][y, x])
    print("neigh oracle ", want[ly - 1:ly + 2, lx - 1:lx + 2, ch].tolist())
    print("neigh general", outs[2][y - 1:y + 2, x - 1:x + 2, ch].tolist())
    print("neigh fused  ", outs[0][y - 1:y + 2, x - 1:x + 2, ch].tolist())
    # oracle convert output around source pixel
    conv, fmt = O.convert_only(p, frame, pitch)
    print("oracle conv10 around:", np.round(conv[ly // 2 - 3: ly // 2 + 4, lx // 2 - 3: lx // 2 + 4, ch] * 1023).astype(int).tolist())
