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


def _bench(args, env_extra=None, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)


def test_bench_gpus_n_spawns_n_ranks_by_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment starts its own two ranks under torch.distributed.run (round 6: it
    used to warn and measure one GPU).  --rendezvous-only is that flow without pixels, so it runs here: rendezvous over gloo, the blob
    broadcast, MAX over ranks, every rank's self-description in the line."""
    import json
    out = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rendezvous-only"])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                     # ONE line, from rank 0
    r = json.loads(lines[0])
    d = r["config"]["distributed"]
    assert r["n_gpus"] == 2 and d["world_size"] == 2 and d["backend"] == "gloo"
    assert [x["rank"] for x in d["devices"]] == [0, 1] and len({x["pid"] for x in d["devices"]}) == 2      # two processes
    assert r["config"]["spawned_by_bench"] is True
    assert r["max_over_ranks_check"] == pytest.approx(1.001)          # rank 1's value reached rank 0
    assert [x["frames"] for x in d["devices"]] == [3, 3]


def test_bench_fails_when_world_size_and_gpus_disagree():
    """A rank whose WORLD_SIZE is not --gpus fails (it used to print a warning and go on): the line must describe the run that was asked for."""
    out = _bench(["--gpus", "4", "--rendezvous-only"], dict(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert out.returncode != 0 and "WORLD_SIZE=1 but --gpus 4" in out.stderr, out.stderr[-500:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_without_a_gpu_refuses_to_spawn():
    """No device visible: the product path has no CPU fallback, and the launcher says so before starting any rank."""
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    out = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert out.returncode != 0 and "needs a GPU" in out.stderr, out.stderr[-500:]
