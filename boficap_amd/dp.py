"""Data-parallel helpers: one process per GPU, images sharded by rank.

The decode path needs no collective (images are independent units; quirk Q1 couples rows inside a
local batch only), so these helpers only split work and combine measurements.  The reference's
single-process nn.DataParallel (tools/train.py:99-101) is replaced by torch.distributed processes.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from torchrun's environment; initialises the process group if world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"))
    return rank, local_rank, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, near-equal shards covering [0, n_items) exactly once."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def reduce_scalar(value: float, op: str = "max", device: str | torch.device = "cpu") -> float:
    """MAX / SUM of a python scalar over all ranks (identity for a single process)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op={"max": dist.ReduceOp.MAX, "sum": dist.ReduceOp.SUM}[op])
    return float(t.item())


def gather_rows(x: torch.Tensor) -> torch.Tensor:
    """Concatenate equally shaped per-rank results along dim 0 on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return x
    out = [torch.empty_like(x) for _ in range(dist.get_world_size())]
    dist.all_gather(out, x.contiguous())
    return torch.cat(out, 0)
