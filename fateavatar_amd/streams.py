"""Streams that really run side by side.

The HIP runtime multiplexes its streams onto a few hardware queues (four by default, `GPU_MAX_HW_QUEUES`), assigning a
queue to a stream round-robin when the stream is first used.  Two streams that land on the same queue are serialised —
frames that were meant to be in flight together (bench.py --in-flight, AvatarBatchStep) then run one after the other
(measured: three lanes, one of them sharing the caller's queue: 6.8 k frames/s instead of 9.1 k).  Which queue a new
stream gets depends on everything the process created before it, so it is measured: `concurrent_streams` creates
candidates and keeps those that overlap with each other and with the streams the caller names."""
from __future__ import annotations

from typing import List, Sequence

import torch


def _overlap(a: torch.cuda.Stream, b: torch.cuda.Stream, dev, spin_cycles: int) -> bool:
    """Does a short kernel on `b` finish while a long one on `a` is still running?"""
    x = torch.zeros(64, device=dev)
    end_a, end_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(a):
        torch.cuda._sleep(spin_cycles)
        end_a.record(a)
    with torch.cuda.stream(b):
        x.add_(1.0)
        end_b.record(b)
    torch.cuda.synchronize(dev)
    return end_b.elapsed_time(end_a) > 0.0      # b's end lies BEFORE a's end


def concurrent_streams(n: int, device, also_with: Sequence[torch.cuda.Stream] = (), candidates: int = 12) -> List[torch.cuda.Stream]:
    """`n` streams of `device` that overlap pairwise and with every stream in `also_with` (as far as the hardware queues
    allow: with fewer free queues than asked for, the best-effort remainder is plain new streams)."""
    dev = torch.device(device)
    spin = 400_000                                # ~0.2 ms: long against a launch, short against anything that matters
    picked: List[torch.cuda.Stream] = []
    pool = []
    with torch.cuda.device(dev):
        for _ in range(candidates):
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):            # first use fixes the stream's hardware queue
                torch.zeros(8, device=dev).add_(1.0)
            pool.append(s)
        torch.cuda.synchronize(dev)
        for s in pool:
            if len(picked) == n:
                break
            others = list(also_with) + picked
            if all(_overlap(o, s, dev, spin) and _overlap(s, o, dev, spin) for o in others):
                picked.append(s)
        while len(picked) < n:                    # not enough queues: whatever is left
            picked.append(pool[len(picked)] if len(picked) < len(pool) else torch.cuda.Stream(device=dev))
    return picked
