"""One-process-per-GPU plumbing shared by bench.py and the multi-rank tests.

The hot paths shard by independent pairs: no data-path collective exists.
torch.distributed (backend "nccl" = RCCL on the GPU box, "gloo" in CPU tests) is used only to
line ranks up around the timed region, to take the max of their clocks, and -- when one host
batch is split over ranks -- to bring the per-rank result slices back to rank 0.
"""
from __future__ import annotations

import os

import numpy as np


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_bounds(cost, world: int):
    """Contiguous shards of units (pairs / regions) balanced by cost: the Python mirror of the
    cut rule in agx_sw_score_multi / agx_phmm_forward_multi.  Returns world+1 boundaries."""
    cost = np.asarray(cost, dtype=np.float64) + 1.0
    n = cost.size
    cut = np.full(world + 1, n, dtype=np.int64)
    cut[0] = 0
    total = float(cost.sum())
    acc, d = 0.0, 1
    for p in range(n):
        if d >= world:
            break
        acc += float(cost[p])
        while d < world and acc >= total * d / world:
            cut[d] = p + 1
            d += 1
    return cut


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_slices(local: np.ndarray, cut, device=None):
    """Rank 0 receives the concatenation of every rank's slice (sizes from `cut`); others get None."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [int(cut[r + 1] - cut[r]) for r in range(world)]
    width = max(sizes) if sizes else 0
    buf = torch.zeros(max(width, 1), dtype=torch.float64, device=device)
    buf[: local.size] = torch.from_numpy(local.astype(np.float64)).to(buf.device)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    if rank != 0:
        return None
    return np.concatenate([out[r][: sizes[r]].cpu().numpy() for r in range(world)]) if world else np.zeros(0)
