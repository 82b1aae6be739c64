"""Experiment: is it co-residency that two launches in flight recover?  One 32-frame launch per step against two 16-frame launches side by side
(two contexts, a stream each), every step closed by a host synchronize in both modes (no overlap ACROSS steps)."""
import sys, os, time
os.environ["MPCVR_NO_BATCH_LANES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from videorenderer_amd import api

wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3hdr"])
w, h, s = wl["w"], wl["h"], wl["scale"]
dw, dh = wl.get("dst", (w * s, h * s))
extfmt = api.make_extfmt(**wl["ext"])
st = api.default_settings(iUpscaling=wl["iUpscaling"], iDownscaling=wl.get("iDownscaling", 2), output_format=wl.get("output_format", 0), bUseDither=wl.get("bUseDither", 1))
B = 32


def make():
    vp = api.VideoProcessor(st, device=0, use_torch_stream=False)
    vp.InitMediaType(wl["cformat"], w, h, extfmt=extfmt)
    vp.SetWindowRect((0, 0, dw, dh)); vp.SetVideoRect((0, 0, dw, dh))
    return vp


a, b = make(), make()
nbytes, pitch = a.GetFrameBytes()
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
srcs = [bench.noise_frame_gpu(torch, wl, nbytes, pitch, gen) for _ in range(B)]
dsts = [torch.empty((dh, dw, 4), dtype=torch.uint8, device="cuda") for _ in range(B)]
full = a.PrepareBatch(srcs, dsts)
ha, hb = a.PrepareBatch(srcs[:16], dsts[:16]), b.PrepareBatch(srcs[16:], dsts[16:])
qa = [a.PrepareBatch(srcs[8 * k:8 * k + 8], dsts[8 * k:8 * k + 8]) for k in range(4)]


def one():
    a.ProcessBatch(full, None, dw * 4); a.Synchronize()


def two():
    a.ProcessBatch(ha, None, dw * 4); b.ProcessBatch(hb, None, dw * 4); a.Synchronize(); b.Synchronize()


def halves_in_order():
    a.ProcessBatch(ha, None, dw * 4); a.ProcessBatch(a.PrepareBatch(srcs[16:], dsts[16:]) if False else hb2, None, dw * 4); a.Synchronize()


hb2 = a.PrepareBatch(srcs[16:], dsts[16:])
for rep in range(3):
    for name, fn in (("one 32-frame launch", one), ("two 16-frame launches side by side", two), ("two 16-frame launches one after the other", halves_in_order)):
        for _ in range(20): fn()
        t = time.perf_counter()
        for _ in range(150): fn()
        dt = (time.perf_counter() - t) / 150
        print(f"{name:45s} {dt * 1e3:.4f} ms per step  {B / dt:9.1f} frames/s", flush=True)
