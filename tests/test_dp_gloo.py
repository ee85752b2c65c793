"""CPU, world_size 2 over gloo: the data-parallel exchange of fateavatar_amd/dp.py (the N > 1 path of
bench.py with the rasterizer replaced by a deterministic per-rank gradient)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fateavatar_amd import dp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # replicated parameters: rank 0's values win
    params = torch.full((1000,), float(rank + 1))
    dp.broadcast_(params, 0)
    # each rank "renders" its own frames and produces a flat gradient
    frames = dp.shard_frames(8, rank, world)
    grad = torch.zeros(1000)
    for f in frames:
        grad += torch.arange(1000, dtype=torch.float32) * (f + 1)
    grad /= len(frames)
    dp.allreduce_mean_(grad)
    stats = torch.tensor([float(len(frames)), float(sum(frames))])
    dp.allreduce_sum_(stats)
    dp.barrier()
    q.put((rank, params[:3].tolist(), grad[:4].tolist(), stats.tolist(), frames))
    dist.destroy_process_group()


def test_two_rank_flat_gradient_allreduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # frames 0..7: weights (f+1) -> rank0 has 1,3,5,7 (mean 4), rank1 has 2,4,6,8 (mean 5); batch mean 4.5
    for rank, params, grad, stats, frames in res:
        assert params == [1.0, 1.0, 1.0]
        np.testing.assert_allclose(grad, [0.0, 4.5, 9.0, 13.5])
        assert stats == [8.0, 28.0]
        assert frames == list(range(rank, 8, 2))


def test_single_process_is_a_no_op():
    g = torch.ones(10)
    assert dp.world_size() == 1
    assert torch.equal(dp.allreduce_mean_(g.clone()), g)
    assert dp.shard_frames(5, 0, 1) == [0, 1, 2, 3, 4]
