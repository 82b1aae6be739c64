"""Experiment: do consecutive 32-frame launches of the exact-2x kernel gain from overlapping (the tail of one with the head of the next)?
One context on one stream (the bench's shape) against two contexts, each on a stream of its own, taking batches alternately."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from videorenderer_amd import api

wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3hdr"])
w, h, s = wl["w"], wl["h"], wl["scale"]
dw, dh = wl.get("dst", (w * s, h * s))
extfmt = api.make_extfmt(**wl["ext"])
st = api.default_settings(iUpscaling=wl["iUpscaling"], iDownscaling=wl.get("iDownscaling", 2), output_format=wl.get("output_format", 0), bUseDither=wl.get("bUseDither", 1))
B, ring = 32, 64


def make(own):
    vp = api.VideoProcessor(st, device=0, use_torch_stream=not own)
    vp.InitMediaType(wl["cformat"], w, h, extfmt=extfmt)
    vp.SetWindowRect((0, 0, dw, dh)); vp.SetVideoRect((0, 0, dw, dh))
    return vp


vps = [make(True), make(True)]
nbytes, pitch = vps[0].GetFrameBytes()
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
srcs = [bench.noise_frame_gpu(torch, wl, nbytes, pitch, gen) for _ in range(ring)]
dsts = [torch.empty((dh, dw, 4), dtype=torch.uint8, device="cuda") for _ in range(ring)]
batches = [[vp.PrepareBatch(srcs[k:k + B], dsts[k:k + B]) for k in (0, 32)] for vp in vps]


def run(n_ctx, steps):
    for i in range(steps):
        vp = vps[i % n_ctx]
        vp.ProcessBatch(batches[i % n_ctx][i & 1 if n_ctx == 1 else (i >> 1) & 1], None, dw * 4)
    for vp in vps:
        vp.Synchronize()


for rep in range(3):
    for n_ctx in (1, 2):
        run(n_ctx, 40)
        t = time.perf_counter(); run(n_ctx, 200); dt = time.perf_counter() - t
        print(f"{n_ctx} stream(s): {200 * B / dt:9.1f} frames/s  ({dt / 200 * 1e3:.4f} ms per batch)", flush=True)
