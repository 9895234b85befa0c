"""Multi-GPU plumbing: tile sharding + the one collective of the scan path.

Tiles are independent (count_well_duplicates.py:207-226: one lane_dupl entry per tile, only
summed in output_writer, :63-106), so the (lane, tile) list is block-partitioned over ranks
with no data-path exchange.  The only collective is one in-place int64 SUM all-reduce of the
zero-initialised [all tiles, 1 + 5*levels] counter block in which every rank has filled its
own rows: integer addition is order-independent, so the merged block is bit-identical to a
single-GPU run, and every rank (rank 0 prints) then holds all per-tile rows for the verbose
report and the lane sums.

One process per GPU over torch.distributed: backend "nccl" is RCCL on ROCm (xGMI inside a
node); "gloo" runs the same code on CPU tensors (tests/test_dist.py, world_size 2).
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple



def env_rank() -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment (1 process if unset)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; sizes differ by at most one."""
    return (n_items * rank) // world, (n_items * (rank + 1)) // world


def shard(items: Sequence, rank: int, world: int) -> List:
    lo, hi = shard_bounds(len(items), rank, world)
    return list(items[lo:hi])


def merge_blocks(local_rows, n_items: int, rank: int, world: int, device=None):
    """All ranks' per-tile counter rows -> the full [n_items, ncnt] block on every rank.

    local_rows: this rank's [hi - lo, ncnt] int64 rows (numpy array or torch tensor, on the
    CPU for gloo or on the GPU for nccl).  Returns a torch tensor on the same device.
    """
    import torch
    import torch.distributed as dist

    rows = torch.as_tensor(local_rows) if not isinstance(local_rows, torch.Tensor) else local_rows
    lo, hi = shard_bounds(n_items, rank, world)
    assert rows.shape[0] == hi - lo and rows.dtype == torch.int64
    full = torch.zeros((n_items, rows.shape[1]), dtype=torch.int64, device=device or rows.device)
    full[lo:hi] = rows.to(full.device)
    if world > 1:
        dist.all_reduce(full, op=dist.ReduceOp.SUM)
    return full


def max_over_ranks(value: float, world: int, device=None) -> float:
    """Largest `value` of any rank (wall-clock of the slowest rank)."""
    if world <= 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
