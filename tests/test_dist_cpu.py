"""The N>1 path on CPU: 2 processes, gloo.  Frames shard by index with no data-path collective; the only
exchange is the parameter-blob broadcast from rank 0."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from videorenderer_amd import dist as vdist
    r, w, _ = vdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    blob = bytes(range(256)) * 25 if rank == 0 else None          # 6400 B stand-in for the parameter blob
    got = vdist.broadcast_blob(blob, device=torch.device("cpu"))
    mine = vdist.shard_frames(11, rank, world)
    # each rank "processes" its frames independently; timing = max over ranks, throughput = sum of frames
    t = vdist.max_over_ranks(1.0 + rank, device=torch.device("cpu"))
    n = vdist.sum_over_ranks(float(len(mine)), device=torch.device("cpu"))
    q.put((rank, got == bytes(range(256)) * 25, mine, t, n))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_broadcast_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)                         # every rank holds rank 0's blob
    assert res[0][2] == [0, 2, 4, 6, 8, 10] and res[1][2] == [1, 3, 5, 7, 9]
    assert sorted(res[0][2] + res[1][2]) == list(range(11))   # a partition: nothing dropped, nothing twice
    assert res[0][3] == res[1][3] == 2.0 and res[0][4] == res[1][4] == 11.0


def test_single_process_is_a_noop():
    from videorenderer_amd import dist as vdist
    assert vdist.shard_frames(5, 0, 1) == [0, 1, 2, 3, 4]
    assert vdist.broadcast_blob(b"abc") == b"abc"
    assert vdist.max_over_ranks(3.5) == 3.5
