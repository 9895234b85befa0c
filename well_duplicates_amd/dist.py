"""Multi-GPU plumbing: tile sharding + the one collective of the scan path.

Tiles are independent (count_well_duplicates.py:207-226: one lane_dupl entry per tile, only
summed in output_writer, :63-106), so the (lane, tile) list is block-partitioned over ranks
with no data-path exchange.  The only collective is one in-place int64 SUM all-reduce of the
zero-initialised [all tiles, 1 + 5*levels] counter block in which every rank has filled its
own rows: integer addition is order-independent, so the merged block is bit-identical to a
single-GPU run, and every rank (rank 0 prints) then holds all per-tile rows for the verbose
report and the lane sums.

One process per GPU over torch.distributed: backend "nccl" is RCCL on ROCm (xGMI inside a
node); "gloo" runs the same code on CPU tensors (tests/test_dist.py, world_size 2); "wd" sums
the block with libwelldup's own RCCL binding (wd_comm_* / wd_allreduce_counts, include/welldup.h)
and uses the process group only to hand the communicator's unique id around.
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


def _tensor_device(backend: str, device):
    import torch
    return torch.device("cuda", device) if backend == "nccl" and isinstance(device, int) else \
        (device if backend == "nccl" and device is not None else torch.device("cpu"))


def any_rank_failed(failed: bool, world: int, backend: str = "gloo", device=None) -> bool:
    """True on every rank if any rank says so (MAX all-reduce of a flag): a rank that raised must
    not leave the others waiting in the merge."""
    if world <= 1:
        return bool(failed)
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if failed else 0], dtype=torch.int32, device=_tensor_device(backend, device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


def gather_dicts(mine: dict, world: int) -> dict:
    """Union of every rank's dict (keys are disjoint: each (lane, tile) has one owner)."""
    if world <= 1:
        return dict(mine)
    import torch.distributed as dist
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    return {key: val for part in parts for key, val in part.items()}


def wd_comm_bootstrap(scanner, rank: int, world: int):
    """libwelldup's own RCCL communicator (wd_comm_*): rank 0 draws the unique id, the process
    group that torchrun started carries it to the others, every rank joins."""
    import torch.distributed as dist
    box = [scanner.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    scanner.comm_init(rank, world, box[0])


def merge_blocks(local_rows, n_items: int, rank: int, world: int, device=None, backend: str = "torch",
                 scanner=None):
    """All ranks' per-tile counter rows -> the full [n_items, ncnt] block on every rank.

    local_rows: this rank's [hi - lo, ncnt] int64 rows (numpy array or torch tensor).
    backend "torch" / "gloo" / "nccl": one torch.distributed all_reduce (on `device`: CPU
    tensors for gloo, GPU tensors for nccl = RCCL).  backend "wd": the block is staged in the
    scanner's device memory and summed by wd_allreduce_counts - RCCL bound by libwelldup itself,
    for hosts whose only GPU runtime is the library.  Returns a numpy array for numpy input
    (and always for "wd"), else a torch tensor.
    """
    import numpy as np
    lo, hi = shard_bounds(n_items, rank, world)
    if backend == "wd":
        if world > 1 and scanner is None:
            raise ValueError("backend 'wd' sums the block in the scanner's device memory: pass scanner=")
        rows = np.ascontiguousarray(local_rows, dtype=np.int64)
        assert rows.shape[0] == hi - lo
        full = np.zeros((n_items, rows.shape[1]), dtype=np.int64)
        full[lo:hi] = rows
        if world > 1 or scanner is not None:
            wd_comm_bootstrap(scanner, rank, world)
            buf = scanner.malloc(max(8, full.nbytes))
            try:
                scanner.h2d(buf, full)
                scanner.allreduce_counts(buf, full.size)
                scanner.synchronize()
                full = scanner.d2h(buf, full.nbytes, np.int64).reshape(full.shape)
            finally:
                scanner.free(buf)
                scanner.comm_destroy()
        return full
    import torch
    import torch.distributed as dist

    as_numpy = not isinstance(local_rows, torch.Tensor)
    rows = torch.as_tensor(local_rows) if as_numpy else local_rows
    assert rows.shape[0] == hi - lo and rows.dtype == torch.int64
    dev = device if isinstance(device, torch.device) else \
        (_tensor_device(backend, device) if backend in ("nccl", "gloo") else (device or rows.device))
    full = torch.zeros((n_items, rows.shape[1]), dtype=torch.int64, device=dev)
    full[lo:hi] = rows.to(full.device)
    if world > 1:
        dist.all_reduce(full, op=dist.ReduceOp.SUM)
    return full.cpu().numpy() if as_numpy else full


def max_over_ranks(value: float, world: int, device=None) -> float:
    """Largest `value` of any rank (wall-clock of the slowest rank)."""
    if world <= 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
