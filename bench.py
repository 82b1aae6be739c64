#!/usr/bin/env python3
"""bench.py — headline benchmark of the shader-video-processor path on MI355X.

Metric (BASELINE.json): 4K frames/sec/GPU for P010 -> Lanczos3 2x -> PQ->SDR -> dither, and % of HBM roofline.
Workload "c3hdr" (SURVEY.md §8d): 3840x2160 P010 (BT.2020NC, PQ, TV range) -> UPSCALE_Lanczos3 2x (D3D11
quirks as written) -> 7680x4320 B8G8R8A8 with the 32x32 ordered dither; algorithmic bytes/frame =
24,883,200 (in) + 132,710,400 (out) = 157,593,600.

A "step" = one pass of the hot path over one batch of `--batch` frames (one mpcvr_process_batch call; the
fused kernel covers the whole batch in ONE launch).  Inputs are resident in HBM before the timed region; a
ring of distinct noise frames (48 x 24.9 MB = 1.2 GB, far beyond the 256 MiB Infinity Cache) is cycled so no step
re-reads cached input, and every frame is written to its own 132.7 MB output buffer.

    python bench.py                       # N=1, finishes in a few minutes incl. the CPU baseline
    python bench.py --gpus 8 --steps 50 --warmup 5      # starts its own 8 ranks (one per GPU) under torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        bench.py --gpus 8 --steps 50 --warmup 5         # the driver's form: the ranks exist already, nothing is spawned

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes this file under torch.distributed.run with N
ranks (refusing when fewer than N devices are visible); a rank whose WORLD_SIZE differs from --gpus FAILS.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured achievable

WORKLOADS = {
    # name: (cformat, src_w, src_h, scale, extfmt fields, iUpscaling, description)
    "c3hdr": dict(cformat=2, w=3840, h=2160, scale=2, ext=dict(chroma=5, nominal_range=2, matrix=4, primaries=9, transfer=15),
                  iUpscaling=4, desc="4K P010 BT.2020/PQ -> Lanczos3 2x -> PQ->SDR(Hable,125nits) -> ordered dither -> 8K BGRA8"),
    "c3": dict(cformat=2, w=3840, h=2160, scale=2, ext=dict(chroma=5, nominal_range=2, matrix=1, primaries=2, transfer=5),
               iUpscaling=4, desc="4K P010 BT.709 SDR -> Lanczos3 2x -> ordered dither -> 8K BGRA8"),
    "c4": dict(cformat=2, w=3840, h=2160, scale=2, ext=dict(chroma=5, nominal_range=2, matrix=4, primaries=9, transfer=15),
               iUpscaling=1, desc="4K P010 HDR10 -> Mitchell 2x -> PQ->SDR -> ordered dither -> 8K BGRA8"),
    # config 4's optional extension run (the reference has no Spline36: IVideoRenderer.h:54-62; unpinned, reported separately)
    "c4ext": dict(cformat=2, w=3840, h=2160, scale=2, ext=dict(chroma=5, nominal_range=2, matrix=4, primaries=9, transfer=15),
                  iUpscaling=6, desc="4K P010 HDR10 -> Spline36 (extension) 2x -> PQ->SDR -> ordered dither -> 8K BGRA8"),
    # config 4 as BASELINE.json words it: Spline36 (extension) + error-diffusion dither (extension, bUseDither = 2: no reference counterpart, parity unpinned) —
    # the 10-bit plan's fused launch per batch + ONE k_error_diffusion launch (a serial chain of W + 2H steps per frame: VALU-issue and depth bound, not HBM)
    "c4ed": dict(cformat=2, w=3840, h=2160, scale=2, ext=dict(chroma=5, nominal_range=2, matrix=4, primaries=9, transfer=15),
                 iUpscaling=6, bUseDither=2, desc="4K P010 HDR10 -> Spline36 (extension) 2x -> PQ->SDR -> error-diffusion dither (extension) -> 8K BGRA8"),
    "c5": dict(cformat=2, w=3840, h=2160, scale=2, ext=dict(chroma=5, nominal_range=2, matrix=4, primaries=9, transfer=16),
               iUpscaling=4, desc="4K P010 HLG -> Lanczos3 2x -> HLG->SDR -> ordered dither -> 8K BGRA8"),
    "c2": dict(cformat=20, w=1920, h=1080, scale=2, ext=dict(chroma=5, nominal_range=2, matrix=1, primaries=2, transfer=5),
               iUpscaling=2, desc="1080p YUV420P10 BT.709 -> Catmull-Rom 2x -> ordered dither -> 4K BGRA8"),
    # BASELINE.json configs[0] and the native-resolution HDR case: no resize, one block-convert launch per batch
    "c1": dict(cformat=1, w=1920, h=1080, scale=1, ext=dict(chroma=5, nominal_range=2, matrix=1, primaries=2, transfer=5),
               iUpscaling=2, batch=128, desc="1080p NV12 BT.709 -> BGRA8, no resize (BASELINE configs[0])"),
    "hdr4k": dict(cformat=2, w=3840, h=2160, scale=1, ext=dict(chroma=5, nominal_range=2, matrix=4, primaries=9, transfer=15),
                  iUpscaling=4, desc="4K P010 BT.2020/PQ -> PQ->SDR(Hable,125nits) -> ordered dither -> 4K BGRA8, no resize"),
    # everyday non-integer geometries: one fused kernel per batch (k_fused_period at 4:3 / 3:2 / 2:3 / 1:2 / 3:1 down the rows, k_fused_strip otherwise; --flags 128 = MPCVR_FLAG_NO_PERIOD)
    "up1440": dict(cformat=2, w=1920, h=1080, scale=1, dst=(2560, 1440), ext=dict(chroma=5, nominal_range=2, matrix=4, primaries=9, transfer=15),
                   iUpscaling=4, desc="1080p P010 BT.2020/PQ -> Lanczos3 1.33x -> PQ->SDR -> ordered dither -> 1440p BGRA8"),
    "down1440": dict(cformat=2, w=3840, h=2160, scale=1, dst=(2560, 1440), ext=dict(chroma=5, nominal_range=2, matrix=4, primaries=9, transfer=15),
                     iUpscaling=4, iDownscaling=2, desc="4K P010 BT.2020/PQ -> Lanczos3 1.5x down (the interpolation shader: bInterpolateAt50pct keeps ps_convolution for > 2x, DX11VideoProcessor.cpp:3108) -> PQ->SDR -> ordered dither -> 1440p BGRA8"),
    "up1080": dict(cformat=1, w=1280, h=720, scale=1, dst=(1920, 1080), ext=dict(chroma=5, nominal_range=2, matrix=1, primaries=2, transfer=5),
                   iUpscaling=2, desc="720p NV12 BT.709 -> Catmull-Rom 1.5x -> 1080p BGRA8 (8-bit internal format, no dither)"),
    "down1080": dict(cformat=2, w=3840, h=2160, scale=1, dst=(1920, 1080), ext=dict(chroma=5, nominal_range=2, matrix=4, primaries=9, transfer=15),
                     iUpscaling=4, desc="4K P010 BT.2020/PQ -> Lanczos3 at exactly 50 % (interpolation shader) -> PQ->SDR -> ordered dither -> 1080p BGRA8"),
    "up2160": dict(cformat=2, w=1280, h=720, scale=1, dst=(3840, 2160), ext=dict(chroma=5, nominal_range=2, matrix=4, primaries=9, transfer=15),
                   iUpscaling=4, desc="720p P010 BT.2020/PQ -> Lanczos3 3x -> PQ->SDR -> ordered dither -> 2160p BGRA8"),
    # the everyday SDR case (8-bit source, 8-bit internal format, no final pass) and HDR passthrough to a 10-bit swap chain
    "up1440_nv12": dict(cformat=1, w=1920, h=1080, scale=1, dst=(2560, 1440), ext=dict(chroma=5, nominal_range=2, matrix=1, primaries=2, transfer=5),
                        iUpscaling=2, desc="1080p NV12 BT.709 -> Catmull-Rom 1.33x -> 1440p BGRA8 (8-bit internal format, no dither)"),
    "hdrpass_2x": dict(cformat=2, w=3840, h=2160, scale=2, ext=dict(chroma=5, nominal_range=2, matrix=4, primaries=9, transfer=15),
                       iUpscaling=4, hdr_output=1, output_format=1, desc="4K P010 BT.2020/PQ -> Lanczos3 2x -> HDR10 passthrough -> 8K R10G10B10A2"),
    "hdrpass_1440": dict(cformat=2, w=1920, h=1080, scale=1, dst=(2560, 1440), ext=dict(chroma=5, nominal_range=2, matrix=4, primaries=9, transfer=15),
                         iUpscaling=4, hdr_output=1, output_format=1, desc="1080p P010 BT.2020/PQ -> Lanczos3 1.33x -> HDR10 passthrough -> 1440p R10G10B10A2"),
    # the two slowest reference-pinned rows (round-5 counter digests: profiles/r05/jinc1080_*, dovi4k_*)
    "jinc1080": dict(cformat=2, w=1920, h=1080, scale=2, ext=dict(chroma=5, nominal_range=2, matrix=4, primaries=9, transfer=15),
                     iUpscaling=5, desc="1080p P010 BT.2020/PQ -> Jinc2m 2x (ps_resize_onepass_jinc2) -> PQ->SDR -> ordered dither -> 4K BGRA8"),
    "jinc1080_nv12": dict(cformat=1, w=1920, h=1080, scale=2, ext=dict(chroma=5, nominal_range=2, matrix=1, primaries=2, transfer=5),
                          iUpscaling=5, desc="1080p NV12 BT.709 -> Jinc2m 2x -> 4K BGRA8 (8-bit internal format: the exact form of the convert stage, no dither)"),
    "dovi4k": dict(cformat=2, w=3840, h=2160, scale=1, ext=dict(chroma=5, nominal_range=2), iUpscaling=4, dovi=("mmr", (100, 600, 1000)),
                   desc="4K P010 Dolby Vision (MMR chroma curves, level-2 trims) -> SDR -> ordered dither -> 4K BGRA8, no resize"),
    "c3hdr_1080p": dict(cformat=2, w=1920, h=1080, scale=2, ext=dict(chroma=5, nominal_range=2, matrix=4, primaries=9, transfer=15),
                        iUpscaling=4, desc="1080p P010 BT.2020/PQ -> Lanczos3 2x -> PQ->SDR -> dither -> 4K BGRA8 (alternative reading)"),
}


def csrc_digest(sources=None):
    """sha256 over kernel sources: ties an entry of profiles/hbm_traffic.json to the code it was measured on.  `sources` = the
    files (under videorenderer_amd/csrc) the workload's kernel is built from, as recorded with the profile; None = all of them."""
    import hashlib
    d = os.path.join(ROOT, "videorenderer_amd", "csrc")
    h = hashlib.sha256()
    names = sorted(sources) if sources else sorted(f for f in os.listdir(d) if f.endswith((".hip", ".h", ".cpp", ".inc")))
    for f in names:
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def noise_frame_gpu(torch, wl, nbytes, pitch, gen):
    """Legal-range noise directly on the GPU (incompressible; throughput only — parity uses synth.py)."""
    w, h = wl["w"], wl["h"]
    if wl["cformat"] == 2:                                   # P010: 10-bit codes in the MSBs
        y = torch.randint(64, 941, (h, w), device="cuda", generator=gen, dtype=torch.int32) << 6
        c = torch.randint(64, 961, (h // 2, w), device="cuda", generator=gen, dtype=torch.int32) << 6
        buf = torch.cat([y.reshape(-1), c.reshape(-1)]).to(torch.int16).view(torch.uint8)
    elif wl["cformat"] == 20:                                # YUV420P10: raw codes, three planes
        y = torch.randint(64, 941, (h * w,), device="cuda", generator=gen, dtype=torch.int32)
        c = torch.randint(64, 961, (2 * (h // 2) * (w // 2),), device="cuda", generator=gen, dtype=torch.int32)
        buf = torch.cat([y, c]).to(torch.int16).view(torch.uint8)
    elif wl["cformat"] == 1:                                 # NV12: 8-bit codes, luma plane + interleaved chroma
        y = torch.randint(16, 236, (h * w,), device="cuda", generator=gen, dtype=torch.int32)
        c = torch.randint(16, 241, ((h // 2) * w,), device="cuda", generator=gen, dtype=torch.int32)
        buf = torch.cat([y, c]).to(torch.uint8)
    else:
        raise ValueError("workload format")
    assert buf.numel() == nbytes, (buf.numel(), nbytes)
    return buf.contiguous()


def _oracle_cdll(O, native=False):
    """The oracle library; native=True: the same source rebuilt HERE with -march=native (the box that runs the bench decides the ISA)."""
    import ctypes as C
    import subprocess
    import tempfile
    if not native:
        return O.lib()
    out = os.path.join(tempfile.gettempdir(), f"libmpcvr_oracle_native_{os.getpid()}.so")
    subprocess.check_call(["gcc", "-O3", "-march=native", "-std=c11", "-ffp-contract=off", "-fno-math-errno", "-fPIC", "-fopenmp",
                           "-shared", "-o", out, os.path.join(ROOT, "oracle", "mpcvr_oracle.c"), "-lm"])
    L = C.CDLL(out)
    L.orc_process.restype = C.c_int
    L.orc_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_set_num_threads.argtypes = [C.c_int]
    L.orc_num_threads.restype = C.c_int
    return L


def cpu_baseline(wl, extfmt, seconds_budget=10.0):
    """The oracle (our literal C restatement of the reference HLSL; SURVEY.md F1: the reference has NO CPU pixel path) timed on
    this box's host cores on bounded samples of the same workload — SURVEY.md 8d's rows: -O3 -msse2 on all cores (the headline
    row), one thread, -march=native, and the reference's only real per-frame CPU work, the upload repack
    (CopyPlaneAsIs / CopyPlane10to16, Helper.cpp:414-428,789-803).  A failure is printed, never swallowed."""
    import traceback
    rows = {}
    try:
        import ctypes as C
        import numpy as np
        from oracle import oracle as O
        from videorenderer_amd import synth
        w, h, s = wl["w"], wl["h"], wl["scale"]
        dw, dh = wl.get("dst", (w * s, h * s))
        frame, pitch = synth.make_frame(wl["cformat"], w, h, "noise", seed=1)
        frame = np.ascontiguousarray(frame).view(np.uint8).ravel()
        dither = O.dither_table()

        def time_frames(L, p, dst, budget):
            run = lambda: L.orc_process(C.byref(p), frame.ctypes.data, pitch, dither.ctypes.data, dst.ctypes.data, p.window_w * 4)
            t0 = time.perf_counter()
            assert run() == 0
            first = time.perf_counter() - t0
            n = max(1, min(8, int(budget / max(first, 1e-3))))
            t0 = time.perf_counter()
            for _ in range(n):
                run()
            return (time.perf_counter() - t0) / n, n

        def params(rect=None):
            pw, ph = (dw, dh) if rect is None else ((rect[2] - rect[0]) * dw // w, (rect[3] - rect[1]) * dh // h)
            p = O.default_params(cformat=wl["cformat"], width=w, height=h, exfmt=extfmt, iUpscaling=wl["iUpscaling"],
                                 iDownscaling=wl.get("iDownscaling", 2), window_w=pw, window_h=ph, video_rect=(0, 0, pw, ph))
            if rect is not None:
                O.set_params(p, src_rect=rect)
            return p, np.zeros((ph, pw, 4), dtype=np.uint8)

        L = _oracle_cdll(O)
        # threads: what the host lets this process use — the affinity mask and the cgroup's CPU quota (oracle.host_cpus)
        avail = O.host_cpus()
        threads = max(1, min(int(L.orc_num_threads()), avail))
        L.orc_set_num_threads(threads)
        p, dst = params()
        dt, n = time_frames(L, p, dst, seconds_budget)
        rows["sse2_all_threads"] = {"frames_per_s": round(1.0 / dt, 4), "threads": int(threads), "flags": "-O3 -msse2 -fopenmp",
                                    "sample": f"{n} full {w}x{h}->{dw}x{dh} frames, {dt*1e3:.0f} ms/frame"}
        # one thread: a 1/16-height band of the same frame (full width), scaled to whole frames
        band = max(16, (h // 16) & ~1)
        L.orc_set_num_threads(1)
        p1, d1 = params((0, 0, w, band))
        dt1, n1 = time_frames(L, p1, d1, seconds_budget / 2)
        L.orc_set_num_threads(threads)
        rows["sse2_one_thread"] = {"frames_per_s": round(band / h / dt1, 5), "threads": 1, "flags": "-O3 -msse2",
                                   "sample": f"{n1} bands of {w}x{band} source rows ({band}/{h} of a frame), {dt1*1e3:.0f} ms/band"}
        try:
            Ln = _oracle_cdll(O, native=True)
            Ln.orc_set_num_threads(threads)
            dtn, nn = time_frames(Ln, p, dst, seconds_budget / 2)
            rows["native_all_threads"] = {"frames_per_s": round(1.0 / dtn, 4), "threads": int(Ln.orc_num_threads()),
                                          "flags": "-O3 -march=native -fopenmp", "sample": f"{nn} full frames, {dtn*1e3:.0f} ms/frame"}
        except Exception:
            traceback.print_exc(file=sys.stderr)
            rows["native_all_threads"] = {"frames_per_s": None, "sample": "failed, see stderr"}
        # the upload repack of one sample: plane-wise CopyPlaneAsIs (P010 / NV12: the bytes as they are) and CopyPlane10to16
        # (planar 10-bit: << 6) into a "mapped texture" with a 256-byte aligned pitch, one thread as in the reference
        tp = (pitch + 255) & ~255
        lines = frame.size // pitch
        tex = np.zeros(tp * lines, np.uint8)
        L.orc_copy_plane_as_is.argtypes = [C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, C.c_int]
        L.orc_copy_plane_10to16.argtypes = [C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, C.c_int]
        rep = {}
        for nm, fn in (("CopyPlaneAsIs", L.orc_copy_plane_as_is), ("CopyPlane10to16", L.orc_copy_plane_10to16)):
            fn(lines, tex.ctypes.data, tp, frame.ctypes.data, pitch)
            t0 = time.perf_counter()
            for _ in range(20):
                fn(lines, tex.ctypes.data, tp, frame.ctypes.data, pitch)
            d = (time.perf_counter() - t0) / 20
            rep[nm] = {"frames_per_s": round(1.0 / d, 1), "GBps": round(frame.size / d / 1e9, 2)}
        rows["upload_repack_one_thread"] = dict(rep, sample=f"20 x one {frame.size}-byte sample ({w}x{h}, pitch {pitch} -> {tp}), Helper.cpp:414-428 / 789-803 restated in C")
        main = rows["sse2_all_threads"]
        # what the threads could run on: the scheduler's affinity mask, the cgroup's CPU quota (cpu.max: "<quota> <period>" or "max"), the
        # machine's logical CPUs — 128 OpenMP threads on a quota of a few cores is not a 128-core baseline
        host = {"logical_cpus": os.cpu_count()}
        try:
            host["affinity_cpus"] = len(os.sched_getaffinity(0))
        except Exception:
            pass
        for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            try:
                host["cgroup_" + os.path.basename(f)] = open(f).read().strip()
                break
            except OSError:
                pass
        rows["host"] = host
        return {"value": main["frames_per_s"], "unit": "frames/s", "cores": main["threads"], "kind": "port",
                "sample": main["sample"] + ", oracle C (" + main["flags"] + ")", "rows": rows}
    except Exception as e:      # reported side figure: the bench line still goes out, but the failure is loud
        traceback.print_exc(file=sys.stderr)
        return {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": f"FAILED: {type(e).__name__}: {e} (traceback on stderr)", "rows": rows}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, argv):
    """`bench.py --gpus N` from a plain shell: start the N ranks ourselves — one process per GPU under torch.distributed.run on
    127.0.0.1, the same command line the driver uses — and hand back its exit code.  Refuses when fewer than N devices are visible
    (MPCVR_DIST_BACKEND=gloo is the single-GPU rehearsal of the flow: there the ranks share what devices there are, on purpose,
    and the line says so).  --rendezvous-only needs no device at all."""
    import subprocess
    rehearsal = os.environ.get("MPCVR_DIST_BACKEND") == "gloo"
    if "--rendezvous-only" not in argv:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < 1:
            raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
        if have < n and not rehearsal:
            raise SystemExit(f"bench.py: --gpus {n} but only {have} device(s) visible: refusing to put two ranks on one GPU "
                             f"(MPCVR_DIST_BACKEND=gloo rehearses the multi-rank flow on fewer devices)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL between processes needs on this driver
    env["MPCVR_BENCH_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def rendezvous_only(args, vdist):
    """--rendezvous-only: everything of the N-rank flow that is not pixels — rendezvous, the parameter-blob broadcast (a stand-in blob),
    MAX over ranks, every rank's self-description gathered on rank 0 — and a line of the same shape as the real one.  Runs without a
    GPU (gloo), so the launcher and the rank bookkeeping are covered by the CPU suite."""
    import torch
    rank, world, local = vdist.init_from_env()
    if world != max(1, args.gpus):
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}")
    dev = torch.device("cuda", local % torch.cuda.device_count()) if torch.cuda.is_available() else torch.device("cpu")
    blob = vdist.broadcast_blob(bytes(range(256)) * 28 if rank == 0 else None, device=dev if dev.type == "cuda" and vdist.describe_backend().get("backend") == "nccl" else torch.device("cpu"))
    assert blob == bytes(range(256)) * 28, "rank %d holds a different parameter blob" % rank
    frames = vdist.shard_frames(world * args.steps, rank, world)
    elapsed = vdist.max_over_ranks(1.0 + 0.001 * rank, device=torch.device("cpu") if vdist.describe_backend().get("backend") != "nccl" else None)
    ranks = vdist.gather_objects({"rank": rank, "local_rank": local, "device_index": dev.index, "device": str(dev), "pci_bus_id": None,
                                  "visible": None, "frames": len(frames), "pid": os.getpid()})
    if rank == 0:
        print(json.dumps({"metric": "rendezvous only (no pixels)", "value": None, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "scaling": "weak", "max_over_ranks_check": elapsed,
                          "config": {"workload": "none", "spawned_by_bench": bool(os.environ.get("MPCVR_BENCH_SPAWNED")),
                                     "distributed": dict(vdist.describe_backend(), devices=ranks)}}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3hdr", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="frames per step (per GPU) = frames per fused launch; default 32 (c1: 128 — a 1080p NV12 frame is 11 MB, "
                    "32 of them are an 80 us launch whose ramp-up, drain and the gap to the next launch cost a sixth of the step)")
    ap.add_argument("--ring", type=int, default=None, help="distinct input/output frame buffers cycled (>= batch); default batch + 16")
    ap.add_argument("--flags", type=int, default=0, help="mpcvr_settings.flags (2 = pass-per-kernel path)")
    ap.add_argument("--settle", type=float, default=1.0,
                    help="seconds of untimed launches BEFORE the counted warm-up (clock ramp: a cold part reads ~5 %% low over a 40 ms run); stated in config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive (host sample) measurement")
    ap.add_argument("--src", default=None, help="WxH: override the workload's source size (secondary rows, e.g. 1920x1080)")
    ap.add_argument("--scale", type=int, default=None, help="integer upscale factor override (the fused kernel covers 2)")
    ap.add_argument("--rendezvous-only", action="store_true", help="the N-rank flow without pixels (rendezvous, blob broadcast, reductions, the line's "
                    "distributed block): runs without a GPU; the CPU suite's check of `--gpus N` starting its own ranks")
    args = ap.parse_args()

    # `--gpus N` from a plain shell starts its own N ranks (round 6: it used to warn and measure ONE GPU)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    if args.rendezvous_only:
        from videorenderer_amd import dist as vdist
        return rendezvous_only(args, vdist)

    import torch
    from videorenderer_amd import api, dist as vdist

    rank, world, local = vdist.init_from_env()
    if world != max(1, args.gpus):
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: the line would not describe the run that was asked for")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    if world > torch.cuda.device_count() and os.environ.get("MPCVR_DIST_BACKEND") != "gloo":
        raise SystemExit(f"bench.py: {world} ranks but {torch.cuda.device_count()} visible device(s)")
    torch.cuda.set_device(local % torch.cuda.device_count())
    dev = torch.cuda.current_device()

    wl = dict(WORKLOADS[args.workload])
    if args.batch is None:
        args.batch = wl.get("batch", 32)
    if args.ring is None:
        args.ring = args.batch + 16
    if args.src:
        wl["w"], wl["h"] = (int(v) for v in args.src.lower().split("x"))
        wl["desc"] += f" [source overridden to {wl['w']}x{wl['h']}]"
    if args.scale:
        wl["scale"] = args.scale
    w, h, s = wl["w"], wl["h"], wl["scale"]
    dw, dh = wl.get("dst", (w * s, h * s))
    if args.src or args.scale:
        dw, dh = w * s, h * s
    extfmt = api.make_extfmt(**wl["ext"])
    settings = api.default_settings(iUpscaling=wl["iUpscaling"], iDownscaling=wl.get("iDownscaling", 2), flags=args.flags,
                                    output_format=wl.get("output_format", 0), bUseDither=wl.get("bUseDither", 1))
    # One explicit (non-default) HIP stream shared by torch and the context: the kernels are launched on it and the
    # torch.cuda.Event pairs below are recorded on it, so they bracket exactly the launches of a step.
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    vp = api.VideoProcessor(settings, device=dev)          # picks up torch's current stream (mpcvr_set_stream)
    assert stream.cuda_stream != 0
    vp.InitMediaType(wl["cformat"], w, h, extfmt=extfmt)
    if wl.get("hdr_output"):
        vp.SetHdrOutput(True)
    if wl.get("dovi"):
        from videorenderer_amd import synth
        vp.SetDoviMetadata(synth.dovi_metadata(wl["dovi"][0], l2=wl["dovi"][1]))
    vp.SetWindowRect((0, 0, dw, dh))
    vp.SetVideoRect((0, 0, dw, dh))
    vdist.sync_params(vp)                                  # RCCL broadcast of rank 0's parameter blob (few KiB)
    nbytes, pitch = vp.GetFrameBytes()
    out_bytes = dw * dh * 4
    algo_bytes = nbytes + out_bytes

    ring = max(args.ring, args.batch)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0x4D50 + rank)
    srcs = [noise_frame_gpu(torch, wl, nbytes, pitch, gen) for _ in range(ring)]
    dsts = [torch.empty((dh, dw, 4), dtype=torch.uint8, device="cuda") for _ in range(ring)]

    prepared = {}

    def step(i):
        k = (i * args.batch) % ring
        b = prepared.get(k)
        if b is None:                 # the ring revisits a handful of offsets: their pointer arrays are built once
            idx = [(k + j) % ring for j in range(args.batch)]
            b = prepared[k] = vp.PrepareBatch([srcs[j] for j in idx], [dsts[j] for j in idx])
        vp.ProcessBatch(b, None, dw * 4)

    # settle: a fixed DURATION of the same launches, untimed, so a fresh box has ramped its clocks before the counted warm-up
    # (the driver's 20-step run is 30 ms of GPU work; profiles/r02: short runs read a few % under long ones)
    settle_steps = 0
    t_settle = time.perf_counter()
    while args.settle > 0 and time.perf_counter() - t_settle < args.settle:
        for _ in range(4):
            step(settle_steps)
            settle_steps += 1
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
        torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()                                  # torch's current stream IS the context's stream (SetStream)
        step(args.warmup + i)
        ev[i][1].record()
    torch.cuda.synchronize()
    my_elapsed = time.perf_counter() - t0                  # this rank's own K steps (before the closing barrier)
    if world > 1:
        torch.distributed.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = vdist.max_over_ranks(elapsed)
    launch_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps       # per step == per launch on the fused path
    my_launch_ms = launch_ms
    launch_ms = vdist.max_over_ranks(launch_ms)
    path = vp.GetVPInfo()
    # every rank describes itself (device identity, its own rate): rank 0 checks that no two ranks shared a GPU and puts the
    # spread into the line, so an N > 1 result proves its own shape
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "local_rank": local, "device_index": int(dev), "device": props.name,
          "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")),
          "visible": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES"),
          "fps": args.batch * args.steps / my_elapsed, "kernel_ms_per_launch": my_launch_ms, "path": path}
    ranks = vdist.gather_objects(me)

    # empirical HBM ceiling beside the 8 TB/s vendor figure: a large device-to-device copy, read + write bytes
    copy_gbps = None
    if rank == 0:
        n_el = 1 << 30
        a_ = torch.empty(n_el, dtype=torch.uint8, device="cuda")
        b_ = torch.empty(n_el, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            b_.copy_(a_)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b_.copy_(a_)
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = 10 * 2 * n_el / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a_, b_
    # ... and the ceiling for THIS workload's traffic shape (the fused kernels write several times what they read; a copy reads as much as
    # it writes and flatters them): the library's own probe kernel — each 16-byte word of a sample read once, fanned out to `fan` words of
    # a target, no arithmetic — over the same ring of samples and targets as the timed steps
    shape_gbps, shape_fan = None, None
    if rank == 0:
        import ctypes as C
        Lp = api.load_library()
        src16 = nbytes // 16 * 16
        shape_fan = max(1, min(64, int(out_bytes // src16)))
        if shape_fan * src16 > out_bytes:           # a downscale writes less than it reads: the probe's shape is then a plain copy of a target's size
            src16 = out_bytes // 16 * 16
        n_probe = min(ring, 24)
        def probe(i):
            k = i % n_probe
            hr = Lp.mpcvr_bandwidth_probe(C.c_void_p(srcs[k].data_ptr()), C.c_void_p(dsts[k].data_ptr()), C.c_size_t(src16), shape_fan, C.c_void_p(stream.cuda_stream))
            if hr < 0:
                raise SystemExit(f"mpcvr_bandwidth_probe failed: {hr:#x}")
        for i in range(n_probe):
            probe(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 4 * n_probe
        e0.record()
        for i in range(reps):
            probe(i)
        e1.record()
        torch.cuda.synchronize()
        shape_gbps = reps * src16 * (1 + shape_fan) / (e0.elapsed_time(e1) * 1e-3) / 1e9

    # ... and, for the exact-2x workloads on bi-planar 16-bit samples (the headline), the kernel's OWN traffic shape over a whole batch in one
    # launch (round 6: mpcvr_bandwidth_probe_up2x — strips of 120 source columns marching down a segment, two luma rows + a chroma row read and
    # four target rows written per step as 16-byte pieces, no arithmetic), plus the same stores alone and the same loads alone
    up2x_shape = None
    if rank == 0 and wl["cformat"] == 2 and s == 2 and (dw, dh) == (2 * w, 2 * h):
        import ctypes as C
        Lp = api.load_library()
        nb = min(args.batch, 64)
        arr_t = C.c_void_p * nb
        up2x_shape = {}
        for mode, label, cols in ((0, "read_write", 120), (1, "write_only", 120), (2, "read_only", 120), (0, "read_write_1KiB_aligned_strips", 128), (1, "write_only_1KiB_aligned_strips", 128)):
            def launch(i):
                k = (i * nb) % ring
                idx = [(k + j) % ring for j in range(nb)]
                hr = Lp.mpcvr_bandwidth_probe_up2x(mode, nb, arr_t(*[srcs[j].data_ptr() for j in idx]), arr_t(*[dsts[j].data_ptr() for j in idx]), w, h, 90, cols,
                                                   C.c_void_p(stream.cuda_stream))
                if hr < 0:
                    raise SystemExit(f"mpcvr_bandwidth_probe_up2x failed: {hr:#x}")
            for i in range(3):
                launch(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 12
            e0.record()
            for i in range(reps):
                launch(i)
            e1.record()
            torch.cuda.synchronize()
            moved = nb * ((nbytes if mode != 1 else 0) + (out_bytes if mode != 2 else 0))
            up2x_shape[label + "_GBps"] = round(reps * moved / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
            up2x_shape[label + "_ms_per_launch"] = round(e0.elapsed_time(e1) / reps, 4)
        up2x_shape["shape"] = (f"one launch per {nb}-frame batch; a wavefront = 120 source columns x 90 source rows; per step 3 x 240 B read, 4 rows x 960 B written as "
                               "16-byte pieces (csrc/vp_probe.hip: k_probe_up2x); *_1KiB_aligned_strips: the same with 128-column strips (whole 1 KiB-aligned row pieces per wavefront); "
                               "bytes counted = the bytes each mode moves")

    # PCIe-inclusive rate of the reference's own calling pattern (CopySample from host memory, then Process), frame by
    # frame through the 3-slot upload ring: reported beside `value`, never as `value` (inputs-resident is the metric)
    host_path = None
    if rank == 0 and world == 1 and not args.no_host_path:
        nf = 96
        host_frame = srcs[0].cpu()
        pinned = host_frame.clone().pin_memory()
        pageable = host_frame.numpy().copy()
        res_h = {}
        for label, buf, kind in (("pinned", pinned, api.MEM_HOST_PINNED), ("pageable", pageable, api.MEM_HOST)):
            for i in range(6):
                vp.CopySample(buf, pitch, kind)
                vp.Process(dsts[i % ring], dw * 4)
            vp.Synchronize()
            th = time.perf_counter()
            for i in range(nf):
                vp.CopySample(buf, pitch, kind)
                vp.Process(dsts[i % ring], dw * 4)
            vp.Synchronize()
            res_h[label] = nf / (time.perf_counter() - th)
            res_h[label + "_timings"] = {k: round(v, 4) for k, v in vp.GetLastTimings().items()}
        host_path = {"frames_per_s_pinned_host_sample": round(res_h["pinned"], 1),
                     "last_frame_timings_ms_pinned": res_h["pinned_timings"], "last_frame_timings_ms_pageable": res_h["pageable_timings"],
                     "frames_per_s_pageable_host_sample": round(res_h["pageable"], 1),
                     "upload_GBps_pinned": round(res_h["pinned"] * nbytes / 1e9, 2),
                     "note": "mpcvr_copy_sample(host) + mpcvr_process per frame; the output stays in HBM (the reference "
                             "presents it); single-frame launches, 3-slot upload ring on a copy stream"}

    # The reference's own call pattern (Render -> Process per frame, DX11VideoProcessor.cpp:2730): mpcvr_copy_sample (device sample, used in
    # place) + mpcvr_process, one frame per call, on a context that owns its stream like a C / C++ host's — reported beside `value`, which
    # is the mpcvr_process_batch rate.  Two figures: frames dealt to the context's frame lanes (default) and strictly one after the other.
    per_frame = None
    if rank == 0 and world == 1 and not args.no_host_path:
        import ctypes as C
        L = api.load_library()
        sp = [C.c_void_p(t.data_ptr()) for t in srcs]
        dp = [C.c_void_p(t.data_ptr()) for t in dsts]
        res_pf = {}
        for label, extra in (("frame_lanes", 0), ("one_after_the_other", api.FLAG_NO_FRAME_LANES)):
            st2 = api.default_settings(iUpscaling=wl["iUpscaling"], iDownscaling=wl.get("iDownscaling", 2), flags=args.flags | extra,
                                       output_format=wl.get("output_format", 0), bUseDither=wl.get("bUseDither", 1))
            vp2 = api.VideoProcessor(st2, device=dev, use_torch_stream=False)
            vp2.InitMediaType(wl["cformat"], w, h, extfmt=extfmt)
            if wl.get("hdr_output"):
                vp2.SetHdrOutput(True)
            if wl.get("dovi"):
                from videorenderer_amd import synth
                vp2.SetDoviMetadata(synth.dovi_metadata(wl["dovi"][0], l2=wl["dovi"][1]))
            vp2.SetWindowRect((0, 0, dw, dh))
            vp2.SetVideoRect((0, 0, dw, dh))
            ctx = vp2._ctx
            nf = max(256, 8 * ring)

            def frames(n, i0=0):
                for i in range(i0, i0 + n):
                    k = i % ring
                    if L.mpcvr_copy_sample(ctx, sp[k], pitch, api.MEM_DEVICE) < 0 or L.mpcvr_process(ctx, dp[k], dw * 4, None, None, 0) < 0:
                        raise SystemExit("per-frame path: " + L.mpcvr_last_error(ctx).decode())
            frames(2 * ring)
            vp2.Synchronize()
            tp = time.perf_counter()
            frames(nf)
            vp2.Synchronize()
            res_pf[label] = nf / (time.perf_counter() - tp)
            res_pf[label + "_last_process_ms"] = round(vp2.GetLastTimings()["process_ms"], 4)
            # the same frames through ONE call (mpcvr_process_frames: the loop of copy_sample + process inside the library, as a C / C++ render
            # thread runs it): what the path costs without two ctypes calls per frame
            arr = C.c_void_p * nf
            sa = arr(*[sp[i % ring].value for i in range(nf)])
            da = arr(*[dp[i % ring].value for i in range(nf)])
            if L.mpcvr_process_frames(ctx, nf, sa, pitch, api.MEM_DEVICE, da, dw * 4) < 0:
                raise SystemExit("per-frame path (one call): " + L.mpcvr_last_error(ctx).decode())
            vp2.Synchronize()
            tp = time.perf_counter()
            if L.mpcvr_process_frames(ctx, nf, sa, pitch, api.MEM_DEVICE, da, dw * 4) < 0:
                raise SystemExit("per-frame path (one call): " + L.mpcvr_last_error(ctx).decode())
            vp2.Synchronize()
            res_pf[label + "_native_loop"] = nf / (time.perf_counter() - tp)
            vp2.close()
        per_frame = {"frames_per_s": round(res_pf["frame_lanes"], 1), "frames_per_s_one_after_the_other": round(res_pf["one_after_the_other"], 1),
                     "frames_per_s_native_loop": round(res_pf["frame_lanes_native_loop"], 1),
                     "frames_per_s_one_after_the_other_native_loop": round(res_pf["one_after_the_other_native_loop"], 1),
                     "last_process_ms": res_pf["frame_lanes_last_process_ms"], "last_process_ms_one_after_the_other": res_pf["one_after_the_other_last_process_ms"],
                     "hbm_frac": round(res_pf["frame_lanes"] * algo_bytes / 1e9 / HBM_PEAK_GBS, 4),
                     "note": "mpcvr_copy_sample(device) + mpcvr_process per frame, wall clock over frames queued back to back, one mpcvr_synchronize at the end; "
                             "the context owns its stream (no mpcvr_set_stream), so consecutive frames overlap on its frame lanes (MPCVR_FLAG_NO_FRAME_LANES: off); "
                             "*_native_loop: the same frames through mpcvr_process_frames (the per-frame loop inside the library: no ctypes call per frame)"}

    # The same batches on a context that OWNS its stream (a C / C++ host's shape): consecutive mpcvr_process_batch calls take turns on two
    # internal lanes, so two launches are in flight and fill each other's ramp-up and tail (round 6).  Reported beside `value`, which stays
    # the stream-ordered rate the roofline block's per-launch duration belongs to.  Two batches' worth of distinct render targets (consecutive
    # batches that share a target are kept in order, i.e. would not overlap).
    batch_lanes = None
    if rank == 0 and world == 1 and not args.no_host_path:
        st3 = api.default_settings(iUpscaling=wl["iUpscaling"], iDownscaling=wl.get("iDownscaling", 2), flags=args.flags,
                                   output_format=wl.get("output_format", 0), bUseDither=wl.get("bUseDither", 1))
        vp3 = api.VideoProcessor(st3, device=dev, use_torch_stream=False)
        vp3.InitMediaType(wl["cformat"], w, h, extfmt=extfmt)
        if wl.get("hdr_output"):
            vp3.SetHdrOutput(True)
        if wl.get("dovi"):
            from videorenderer_amd import synth
            vp3.SetDoviMetadata(synth.dovi_metadata(wl["dovi"][0], l2=wl["dovi"][1]))
        vp3.SetWindowRect((0, 0, dw, dh))
        vp3.SetVideoRect((0, 0, dw, dh))
        more = [torch.empty((dh, dw, 4), dtype=torch.uint8, device="cuda") for _ in range(max(0, 2 * args.batch - len(dsts)))]
        d2 = dsts + more
        pair = [vp3.PrepareBatch([srcs[j % ring] for j in range(args.batch)], d2[:args.batch]),
                vp3.PrepareBatch([srcs[(j + 16) % ring] for j in range(args.batch)], d2[args.batch:2 * args.batch])]
        for i in range(max(args.warmup, 8)):
            vp3.ProcessBatch(pair[i & 1], None, dw * 4)
        vp3.Synchronize()
        lane_set = set()
        for i in range(4):
            vp3.ProcessBatch(pair[i & 1], None, dw * 4)
            lane_set.add(vp3.GetLastBatchInfo()["lane"])
        vp3.Synchronize()
        # at least a quarter of a second of batches: a wall clock around 30 launches of 0.3 ms measures the closing synchronize as much as them
        steps3 = max(args.steps, int(0.25 / max(elapsed / max(args.steps, 1), 1e-5)))
        tl = time.perf_counter()
        for i in range(steps3):
            vp3.ProcessBatch(pair[i & 1], None, dw * 4)
        vp3.Synchronize()
        dt3 = time.perf_counter() - tl
        fps3 = args.batch * steps3 / dt3
        batch_lanes = {"frames_per_s": round(fps3, 1), "hbm_frac": round(fps3 * algo_bytes / 1e9 / HBM_PEAK_GBS, 4), "ms_per_step": round(dt3 / steps3 * 1e3, 4),
                       "steps": steps3, "lanes": sorted(lane_set),
                       "note": "the timed loop's batches through mpcvr_process_batch on a context that owns its stream: consecutive batches take turns on two "
                               "lanes (lanes [-1] = stream order: the plan is not one launch with nothing shared, or two launches in flight were measured not to pay for its kernel); wall clock over `steps` batches"}
        vp3.close()
        del more, d2

    if rank == 0:
        # one process per GPU means one GPU per process: two ranks on one device would double-count its throughput.
        # (MPCVR_DIST_BACKEND=gloo is the single-GPU rehearsal of the N > 1 flow and shares the device on purpose.)
        ids = [(r["visible"], r["device_index"], r["pci_bus_id"]) for r in ranks]
        if len(set(ids)) != len(ids) and os.environ.get("MPCVR_DIST_BACKEND") != "gloo":
            raise SystemExit(f"bench.py: ranks share a GPU: {ids}")
        frames = world * args.batch * args.steps
        fps = frames / elapsed
        achieved = algo_bytes * args.batch / (launch_ms * 1e-3) / 1e9          # GB/s, algorithmic bytes per launch
        # HBM traffic from the PMC counters is NOT measured by this run: it comes from the committed rocprofv3 --pmc passes
        # (profiles/hbm_traffic.json), and only when the kernels have not changed since (sha256 over videorenderer_amd/csrc)
        traffic, traffic_source, valu_issue = None, None, None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                doc = json.load(open(tf))
                t = doc.get(args.workload)
                if t and t.get("batch") == args.batch:
                    if t.get("csrc_sha256") == csrc_digest(t.get("sources")):
                        traffic = t["bytes_per_launch"]
                        traffic_source = f"profiles/hbm_traffic.json [{t.get('profile', '?')}]: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes"
                        if t.get("valu_issue_frac"):
                            # the roof that binds the fused kernels: the share of the launch during which a SIMD's VALU pipe is issuing
                            # (1.0 = no free issue slot left).  From the same committed PMC passes as `traffic`, tied to the same sources.
                            valu_issue = {"bound": "valu_issue", "frac": t["valu_issue_frac"], "wait_inst_any_share": t.get("wait_inst_any_share"),
                                          # the clock the kernel sustained in that pass (SQ_BUSY_CYCLES / 32 over the kernel's duration): issue slots x clock
                                          "sustained_mhz": t.get("sustained_mhz"),
                                          "source": f"profiles/hbm_traffic.json [{t.get('profile', '?')}]: SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs over SQ_BUSY_CYCLES / 32, rocprofv3 --pmc pass of tools/pmc_traffic.sh"}
                    else:
                        traffic_source = f"omitted: kernels changed since profile {t.get('profile', '?')} (csrc hash differs)"
            except Exception as e:
                traffic_source = f"unreadable profiles/hbm_traffic.json: {e}"
        res = {
            # BASELINE.json's metric names the default workload; other --workload values are side measurements
            "metric": ("4K frames/sec/GPU (P010->Lanczos3 2x->PQ-SDR->dither); % HBM roofline" if args.workload == "c3hdr"
                       else f"frames/sec/GPU ({args.workload}); % HBM roofline"),
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (u16 in, u8 out)", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['desc']}", "frames_per_step_per_gpu": args.batch,
                       "input_ring_frames": ring, "path": path, "sharding": "frames by index, no data-path collective",
                       "fps_per_gpu": round(fps / world, 2), "algorithmic_bytes_per_frame": algo_bytes,
                       "settle_s": args.settle, "settle_steps": settle_steps,
                       "distributed": dict(vdist.describe_backend(), collective="one broadcast of rank 0's parameter blob at set-up",
                                           per_rank_fps_min=round(min(r["fps"] for r in ranks), 2), per_rank_fps_max=round(max(r["fps"] for r in ranks), 2),
                                           devices=[{k: r[k] for k in ("rank", "device_index", "device", "pci_bus_id", "visible")} for r in ranks],
                                           paths=sorted(set(r["path"] for r in ranks)))},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_ms_per_launch": round(launch_ms, 4), "bytes_per_launch": algo_bytes * args.batch,
                         "empirical_copy_peak_GBps": round(copy_gbps, 1) if copy_gbps else None,
                         "frac_of_empirical_copy_peak": round(achieved / copy_gbps, 4) if copy_gbps else None,
                         # the same box's rate for this workload's read : write shape with no arithmetic (mpcvr_bandwidth_probe, csrc/vp_probe.hip)
                         "empirical_shape_peak_GBps": round(shape_gbps, 1) if shape_gbps else None,
                         "empirical_shape": f"1 : {shape_fan} bytes read : written, 16-byte accesses, {min(ring, 24)}-frame ring" if shape_gbps else None,
                         "frac_of_empirical_shape_peak": round(achieved / shape_gbps, 4) if shape_gbps else None,
                         # the exact-2x kernel's own access pattern with no arithmetic, whole batch per launch (read + write / stores alone / loads alone)
                         "empirical_up2x_shape": up2x_shape,
                         "frac_of_empirical_up2x_shape": round(achieved / up2x_shape["read_write_GBps"], 4) if up2x_shape else None},
        }
        # the fused kernels are bound by VALU issue slots, not by HBM (traffic is ~1.04x algorithmic): the measured occupancy of that
        # roof rides beside the HBM fraction — frac_at_full_issue is what the HBM fraction would be with every issue slot used
        if valu_issue:
            valu_issue["hbm_frac_at_full_issue"] = round(achieved / HBM_PEAK_GBS / valu_issue["frac"], 4)
            res["roofline"]["co_limit"] = valu_issue
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(wl, extfmt)
        if host_path:
            res["host_sample_path"] = host_path
        if per_frame:
            res["process_per_frame"] = per_frame
        if batch_lanes:
            res["process_batch_on_lanes"] = batch_lanes
        print(json.dumps(res), flush=True)
    vp.close()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
