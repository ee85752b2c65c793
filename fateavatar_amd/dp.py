"""Data-parallel frame sharding (BASELINE.json config 4).

The reference is single-process; it loops `for bs_ in range(bs): render(...)` over the frames of a
batch with shared Gaussian parameters and averages the loss (model/fateavatar.py:251-276,
train/loss.py:92-105).  Here: one process per GPU, parameters replicated, rank r renders frame r,
and ONE all-reduce(AVG) of the flat gradient buffer over RCCL/xGMI makes every rank's gradient the
batch mean; the per-view densification statistics (‖means2D.grad‖ sums and visibility counts,
model/fateavatar.py:734-737) are summed with a second small all-reduce.  No collective runs
inside the rasterizer: a frame never leaves its GPU.

Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise torch.distributed from torchrun's environment.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # FR_DP_GROUP_OF_ONE=1: a one-rank process group, so that the N > 1 machinery can be run on the REAL backend on one GPU
    global _diagnostic_group_of_one
    if (world > 1 or os.environ.get("FR_DP_GROUP_OF_ONE") == "1") and not dist.is_initialized():
        _diagnostic_group_of_one = world == 1
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # FR_DIST_BACKEND=gloo lets the N > 1 control flow be exercised on a box with fewer GPUs than ranks
            backend = os.environ.get("FR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        elif torch.cuda.is_available():
            local = local % torch.cuda.device_count()
            dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


_diagnostic_group_of_one = False   # set by init_from_env when IT created the one-rank group for FR_DP_GROUP_OF_ONE=1


def group_of_one() -> bool:
    """True for the DIAGNOSTIC one-rank process group only (FR_DP_GROUP_OF_ONE=1, created by init_from_env): the N > 1 code
    paths on one GPU.  A one-rank group somebody else initialised (a single-GPU launch through torchrun, say) is a plain
    single-process run: its steps exchange nothing and keep their captured graph."""
    return _diagnostic_group_of_one and dist.is_initialized() and dist.get_world_size() == 1


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


_avg_supported = None


def _collective_avg_ok(like: torch.Tensor) -> bool:
    """Does this backend average inside the collective?  RCCL does (ncclAvg); gloo does not.  Probed once with a
    one-element all-reduce so that an older RCCL without AVG degrades to SUM + divide instead of failing the step."""
    global _avg_supported
    if _avg_supported is None:
        ok = False
        if dist.get_backend() == "nccl":
            try:
                probe = torch.ones(1, dtype=like.dtype, device=like.device)
                dist.all_reduce(probe, op=dist.ReduceOp.AVG)
                ok = abs(float(probe.item()) - 1.0) < 1e-6
            except Exception:
                ok = False
        _avg_supported = ok
    return _avg_supported


def allreduce_mean_(flat_grad: torch.Tensor) -> torch.Tensor:
    """In-place batch-mean of a flat gradient buffer across ranks (one collective)."""
    w = world_size()
    if w > 1:
        if _collective_avg_ok(flat_grad):
            dist.all_reduce(flat_grad, op=dist.ReduceOp.AVG)  # averaged inside the collective: no extra pass
        else:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
            flat_grad.div_(w)
    return flat_grad


class PendingReduce:
    """Handle of an asynchronous all-reduce(mean).  `wait()` orders the CURRENT stream behind the collective (RCCL: no
    host block; gloo: blocks) and applies the division when the backend cannot average inside the collective."""

    def __init__(self, work, tensor: torch.Tensor, divide_by: int):
        self.work, self.tensor, self.divide_by = work, tensor, divide_by

    def wait(self) -> torch.Tensor:
        if self.work is not None:
            self.work.wait()
            self.work = None
            if self.divide_by:
                self.tensor.div_(self.divide_by)
        return self.tensor


def allreduce_mean_async(flat_grad: torch.Tensor) -> PendingReduce:
    """Start the batch-mean all-reduce of a flat gradient buffer and return at once.  RCCL runs it on its own
    stream, ordered behind the work already enqueued on the current stream, so kernels launched afterwards (the next
    frame) overlap the exchange; nothing may touch `flat_grad` until `wait()`."""
    w = world_size()
    if w == 1:
        return PendingReduce(None, flat_grad, 0)
    if _collective_avg_ok(flat_grad):
        return PendingReduce(dist.all_reduce(flat_grad, op=dist.ReduceOp.AVG, async_op=True), flat_grad, 0)
    return PendingReduce(dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, async_op=True), flat_grad, w)


def allreduce_sum_(stats: torch.Tensor) -> torch.Tensor:
    """In-place sum of a buffer across ranks (flat gradients, densification accumulators).  On the one-rank diagnostic
    group the collective is still issued: that is what the group is for."""
    if world_size() > 1 or group_of_one():
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    return stats


def shard_frames(n_frames: int, rank: int, world: int) -> list[int]:
    """Frames of a batch handled by `rank` (round-robin, like a DistributedSampler without shuffling)."""
    return list(range(rank, n_frames, world))


def broadcast_(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    """Make a tensor identical on every rank (initial parameters, densification draws)."""
    if world_size() > 1:
        dist.broadcast(t, src)
    return t


def barrier():
    if world_size() > 1:
        dist.barrier()
