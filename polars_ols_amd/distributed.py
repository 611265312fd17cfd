"""Group sharding across the GPUs of one node: one process per GPU over ``torch.distributed`` (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" on CPU for the tests).

Groups are independent in the reference (each plugin call sees one group's rows; no cross-group state anywhere in
src/least_squares.rs), so the data path needs NO collective: every rank owns a contiguous range of groups, balanced
by row count, and runs the same kernels on its shard.  The only exchange is re-assembling an output column: the
per-group coefficient table (small: G x k) or, if a caller really wants it in one place, the per-row predictions.
xGMI is point-to-point (7 links per GPU), so a gather-to-root pulls from 7 peers over 7 distinct links at once, while
a ring all-gather is bound by one link; coefficients are tiny so `all_gather` is fine, predictions should stay
sharded next to the rows they belong to (``gather_rows`` exists for completeness and gathers to ONE root).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist


@dataclass
class Shard:
    rank: int
    world: int
    group_lo: int          # first group owned
    group_hi: int          # one past the last group owned
    row_lo: int
    row_hi: int
    offsets: np.ndarray    # local group offsets (start at 0), len = n_local_groups + 1
    group_counts: List[int]  # groups per rank, all ranks
    row_counts: List[int]    # rows per rank, all ranks


def partition_groups(offsets: Sequence[int], world: int) -> List[int]:
    """Boundaries b[0..world] (group indices) of contiguous ranges with near-equal ROW counts.

    Deterministic and identical on every rank (pure function of the offsets), so no communication is needed to
    agree on the partition."""
    offs = np.asarray(offsets, dtype=np.int64)
    G = len(offs) - 1
    total = int(offs[-1])
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        # first group boundary whose cumulative row count reaches the target
        g = int(np.searchsorted(offs, target, side="left"))
        g = min(max(g, bounds[-1]), G)
        bounds.append(g)
    bounds.append(G)
    return bounds


def shard_for_rank(offsets: Sequence[int], world: int, rank: int) -> Shard:
    offs = np.asarray(offsets, dtype=np.int64)
    b = partition_groups(offs, world)
    lo, hi = b[rank], b[rank + 1]
    local = offs[lo:hi + 1] - offs[lo]
    return Shard(rank=rank, world=world, group_lo=lo, group_hi=hi, row_lo=int(offs[lo]), row_hi=int(offs[hi]),
                 offsets=np.ascontiguousarray(local),
                 group_counts=[b[r + 1] - b[r] for r in range(world)],
                 row_counts=[int(offs[b[r + 1]] - offs[b[r]]) for r in range(world)])


def _all_gather_ragged(local: torch.Tensor, counts: Sequence[int], group=None) -> torch.Tensor:
    """all_gather of tensors whose dim-0 lengths differ per rank: pad to the longest, gather, trim, concatenate in
    rank order (= group order, because shards are contiguous ranges)."""
    world = dist.get_world_size(group)
    mx = max(counts) if counts else 0
    pad_shape = (mx,) + tuple(local.shape[1:])
    padded = torch.zeros(pad_shape, dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    if all(c == mx for c in counts):
        out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, padded, group=group)      # one RCCL call, no list bookkeeping
        return out
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    return torch.cat([bufs[r][: counts[r]] for r in range(world)], dim=0)


def gather_coefficients(local_coef: torch.Tensor, shard: Shard, group=None) -> torch.Tensor:
    """Every rank receives the full [G x k] coefficient table, rows in group order."""
    return _all_gather_ragged(local_coef, shard.group_counts, group)


def gather_rows(local_rows: torch.Tensor, shard: Shard, dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """Per-row outputs (predictions / residuals) gathered to ONE root, rows in frame order; None on other ranks."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mx = max(shard.row_counts)
    padded = torch.zeros((mx,) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype, device=local_rows.device)
    padded[: local_rows.shape[0]] = local_rows
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([bufs[r][: shard.row_counts[r]] for r in range(world)], dim=0)


def slice_columns(columns: Sequence, shard: Shard):
    """The rank's rows of each (host or device) column: plain views, nothing is copied."""
    return [c[shard.row_lo:shard.row_hi] for c in columns]
