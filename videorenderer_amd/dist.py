"""Multi-GPU use of the path: one process per GPU, frames sharded by index, no data-path collective.

Frames (and streams) are independent — no halo, no reduction (the reference itself is
strictly single-GPU, DX11Helper.cpp:81-112).  The only exchange is one broadcast (RCCL over xGMI when
the backend is "nccl") of the parameter blob rank 0 computed — colour matrix, luminance scale, gamut
matrix, resize phase weights, dither table, PQ->SDR LUT (a few KiB) — at context set-up and after a
`Configure`.  Per-frame traffic between GPUs: zero.
"""
import os

import torch
import torch.distributed as dist


def _forced():
    """MPCVR_DIST_FORCE=1: run the collectives even in a world of ONE rank — a single-GPU box then executes the nccl (= RCCL) branch of every
    function below for real (init_process_group("nccl"), broadcast, all_reduce, all_gather_object), which is all the multi-GPU code there is."""
    v = os.environ.get("MPCVR_DIST_FORCE", "")
    return bool(v) and v != "0"


def _single():
    return not dist.is_initialized() or (dist.get_world_size() == 1 and not _forced())


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or _forced()) and not dist.is_initialized():
        if backend is None:      # MPCVR_DIST_BACKEND=gloo lets a single-GPU box exercise the multi-rank flow
            backend = os.environ.get("MPCVR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_frames(n_frames, rank, world):
    """Frame indices owned by `rank`: index mod world (round-robin keeps every rank's load within one frame)."""
    return list(range(rank, n_frames, world))


def broadcast_blob(blob, device=None, src=0):
    """Broadcast rank `src`'s parameter blob (bytes).  Non-source ranks pass None (or anything) and get bytes back.
    `device`: torch device holding the staging tensor — a CUDA device for RCCL, CPU for gloo."""
    if _single():
        return blob
    rank = dist.get_rank()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    size = int(n.item())
    if rank == src:
        buf = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    else:
        buf = torch.empty(size, dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=src)
    return bytes(buf.cpu().numpy().tobytes())


def sync_params(vp, device=None, src=0):
    """Make every rank's processor use rank `src`'s parameter blob (mpcvr_get/set_param_blob)."""
    if _single():
        return
    blob = vp.GetParamBlob() if dist.get_rank() == src else None
    blob = broadcast_blob(blob, device=device, src=src)
    if dist.get_rank() != src:
        vp.SetParamBlob(blob)


def max_over_ranks(value, device=None):
    """MAX-reduce a python float (timing) over ranks."""
    if _single():
        return value
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if _single():
        return value
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_objects(obj):
    """Every rank's `obj` (picklable) as a list on every rank, rank order; [obj] in a single process."""
    if _single():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def describe_backend():
    """Backend name and, for nccl (= RCCL on ROCm), the library version — for the bench line's self-description."""
    if not dist.is_initialized():
        return {"backend": None, "world_size": 1}
    d = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
    if d["backend"] == "nccl":
        try:
            d["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:       # noqa: BLE001 — a description only
            d["rccl_version"] = f"unavailable: {e}"
    return d
