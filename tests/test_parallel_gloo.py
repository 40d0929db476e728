"""N>1 path on CPU: two processes over gloo (the GPU path is the same code over RCCL).

Covers what bench.py --gpus N does per step besides the (per-rank, unshared) rasterizer call:
frame sharding, the flat mean all-reduce of the parameter-gradient buffer, the per-parameter
all-reduce helper, and the max-over-ranks timing reduction."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ggrt_official_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, l = parallel.init_from_env(world, backend="gloo")
    assert (r, w) == (rank, world) and parallel.world_size() == world
    # 1. frames shard one-per-rank (8 frames / iteration in BASELINE config 5)
    mine = parallel.shard_frames(8, rank, world)
    # 2. flat gradient buffer: mean over ranks
    buf = torch.full((1000,), float(rank + 1))
    parallel.allreduce_mean_(buf)
    # 3. per-parameter helper (encoder / pose-net stand-ins + a [4,4] pose gradient)
    torch.manual_seed(0)
    lin = torch.nn.Linear(7, 3)
    pose = torch.nn.Parameter(torch.zeros(4, 4))
    for p in list(lin.parameters()) + [pose]:
        p.grad = torch.full_like(p, float(10 * (rank + 1)))
    parallel.allreduce_gradients(list(lin.parameters()) + [pose])
    # 3b. a parameter that got no gradient on ONE rank (unused for that rank's frame) must not change the message
    #     layout: it counts as zeros there (ranks would otherwise all-reduce different sizes)
    extra = torch.nn.Parameter(torch.zeros(5))
    if rank == 1:
        extra.grad = torch.full_like(extra, 8.0)
    parallel.allreduce_gradients([extra, pose])
    # 3c. the bucketed exchange of bench.py: K asynchronous chunk collectives, ragged last chunk, waited chunk by chunk
    chunked = torch.arange(1003, dtype=torch.float32) * (rank + 1)
    red = parallel.ChunkedMeanAllReduce(chunks=8)
    red.issue(chunked)
    n_pending = len(red.pending)
    red.wait()
    # 3d. an empty buffer (range() with step 0 raised) still goes through the collective sequence
    empty = torch.zeros(0)
    red.issue(empty)
    red.wait()
    # 4. timing reduction used by bench.py
    t = parallel.max_over_ranks(0.5 + rank, "cpu")
    parallel.barrier()
    out[rank] = dict(frames=mine, buf=float(buf[0]), grad=float(lin.weight.grad[0, 0]), pose=float(pose.grad[3, 3]), t=t,
                     extra=float(extra.grad[2]), chunked_ok=bool(torch.allclose(chunked, torch.arange(1003.0) * 1.5)),
                     n_pending=n_pending)
    parallel.shutdown()


@pytest.mark.timeout(120)
def test_two_rank_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert sorted(out.keys()) == [0, 1]
    assert out[0]["frames"] == [0, 2, 4, 6] and out[1]["frames"] == [1, 3, 5, 7]
    for r in (0, 1):
        assert out[r]["buf"] == pytest.approx(1.5)
        assert out[r]["grad"] == pytest.approx(15.0) and out[r]["pose"] == pytest.approx(15.0)
        assert out[r]["t"] == pytest.approx(1.5)
        assert out[r]["extra"] == pytest.approx(4.0)
        assert out[r]["chunked_ok"] and out[r]["n_pending"] == 8


def _forced_worker(rank, world, port, out):
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    parallel.init_from_env(1, backend="gloo", force=True)
    buf = torch.arange(100.0)
    red = parallel.ChunkedMeanAllReduce(chunks=3)
    red.issue(buf)
    red.wait()
    out["ok"] = bool(dist.is_initialized() and torch.equal(buf, torch.arange(100.0)))
    parallel.shutdown()


@pytest.mark.timeout(120)
def test_forced_one_rank_group():
    """`bench.py --force-dist`: a one-rank process group so that the exchange code path runs on a 1-GPU box."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_forced_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    assert out["ok"]


def test_single_process_is_a_noop():
    buf = torch.arange(4.0)
    assert torch.equal(parallel.allreduce_mean_(buf.clone()), buf)
    assert parallel.shard_frames(3, 0, 1) == [0, 1, 2]
    assert parallel.max_over_ranks(2.0, "cpu") == 2.0
