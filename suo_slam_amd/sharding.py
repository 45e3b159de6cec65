"""Multi-GPU layout of the hot path: single-view frames are independent units, so one
process per GPU processes its own shard of the frame stream with NO data-path collective; the only
communication is one all-reduce of the metric accumulators (RCCL over xGMI on GPUs, gloo in the CPU tests)
plus the max-over-ranks wall time that bench.py reports."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_frames(n_frames: int, rank: int, world: int):
    """Frame i -> rank i mod world: returns this rank's frame indices."""
    return list(range(rank, n_frames, world))


def reduce_metrics(elapsed_s: float, sums, device="cpu"):
    """(max over ranks of elapsed, element-wise sum over ranks of `sums`).  No-op without a process group."""
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    s = torch.tensor([float(v) for v in sums], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(v) for v in s.tolist()]
