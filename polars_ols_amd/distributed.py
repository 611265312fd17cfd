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
    agree on the partition.  The C-ABI carries the same function for non-Python hosts (``pols_partition_groups``;
    tests/test_distributed_cpu.py checks the two against each other)."""
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


def create_comm(eng, group=None):
    """The product's own communicator (``pols_comm_*`` over RCCL / xGMI, polars_ols_amd.engine.Comm) for an initialised
    ``torch.distributed`` job: rank 0 makes the unique id, ``torch.distributed`` -- whatever its backend -- only carries those 128
    bytes to the other ranks.  Failing to build it raises: a multi-GPU run never continues without its collective."""
    from .engine import Comm

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    return Comm(eng, world, rank, box[0])


def collective_identity(comm=None, group=None) -> dict:
    """Who took part in the collective, for a measurement line that has to prove it: ``nranks_seen`` and one entry per rank (rank,
    device, PCI bus id) as the communicator's library reports them -- ``Comm.info()`` (RCCL behind the C-ABI) when ``comm`` is given,
    else the ``torch.distributed`` group itself (the CPU twin the gloo tests run).  Collective: every rank of the group calls it."""
    world, rank = (dist.get_world_size(group), dist.get_rank(group)) if dist.is_initialized() else (1, 0)
    if comm is not None:
        mine = comm.info()
        backend = "rccl (pols_comm)"
    else:
        mine = {"nranks_seen": world, "rank_seen": rank, "device": -1, "pci_bus_id": "", "rccl_version": 0}
        backend = dist.get_backend(group) if dist.is_initialized() else "none"
    table = [mine]
    if world > 1:
        table = [None] * world
        dist.all_gather_object(table, mine, group=group)
    seen = sorted({int(e["nranks_seen"]) for e in table})
    return {"library": backend, "nranks_seen": seen[0] if len(seen) == 1 else seen, "rccl_version": int(mine["rccl_version"]),
            "ranks": [{"rank": int(e["rank_seen"]), "device": int(e["device"]), "pci_bus_id": e["pci_bus_id"]} for e in table],
            "distinct_devices": len({(e["device"], e["pci_bus_id"]) for e in table})}


def check_gathered_table(gathered: torch.Tensor, local: torch.Tensor, counts: Sequence[int], group=None) -> dict:
    """The re-assembled table against its parts: every rank's checksum of its own ``local`` rows (sum of the f64 bit patterns, exact) is
    exchanged, and the root compares it with the checksum of that rank's slice of ``gathered`` -- the gathered table IS the concatenation
    of the shards, in rank order.  Collective; returns {"ok": bool, "per_rank": [...]} on the ranks that hold ``gathered`` (None: {"ok": None})."""
    def cks(t):
        b = t.detach().contiguous().to(torch.float64).cpu().numpy().view(np.uint64)
        return int(b.sum(dtype=np.uint64))                    # (wraps modulo 2^64: exact, order-independent)

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    mine = cks(local)
    sums = [mine]
    if world > 1:
        sums = [None] * world
        dist.all_gather_object(sums, mine, group=group)
    if gathered is None:
        return {"ok": None, "per_rank": []}
    lo, per = 0, []
    for r in range(world):
        per.append(cks(gathered[lo: lo + int(counts[r])]) == sums[r])
        lo += int(counts[r])
    return {"ok": bool(all(per)) and lo == gathered.shape[0], "per_rank": per}


class CoefficientRing:
    """Fewer, larger collectives for the per-step coefficient tables (the one exchange step of the path): the tables of ``n_slots``
    consecutive steps are written into one ring buffer and ONE all-gather moves the whole ring -- a ring all-gather over
    point-to-point xGMI pays per-hop latency seven times whatever the payload.  Two rings alternate; a ring is rewritten only
    after the gather that read it has finished.

    ``gather(ring_view, n_used) -> None`` performs the collective on ``ring_view`` (the first ``n_used`` slots of a ring);
    ``produced(r)`` is called right before it (order the collective behind the kernels that filled ring r), ``consumed(r)`` right
    after it, and ``wait_consumed(r)`` before ring r is rewritten.  The stream / event plumbing lives in those callbacks, so the same logic runs under gloo on the CPU
    (tests/test_distributed_cpu.py) and over RCCL in bench.py."""

    def __init__(self, make_ring, n_slots: int, gather, produced=None, wait_consumed=None, consumed=None):
        self.n_slots = int(n_slots)
        self.rings = [make_ring(self.n_slots) for _ in range(2)]
        self._gather, self._produced, self._wait, self._consumed = gather, produced, wait_consumed, consumed
        self.step_no = 0
        self.exchanges = 0

    def begin_step(self):
        """The buffer this step's table goes to."""
        i = self.step_no
        slot, r = i % self.n_slots, (i // self.n_slots) & 1
        if slot == 0 and self._wait is not None:
            self._wait(r)
        return self.rings[r][slot]

    def end_step(self):
        i = self.step_no
        self.step_no += 1
        if i % self.n_slots == self.n_slots - 1:
            self._exchange((i // self.n_slots) & 1, self.n_slots)

    def flush(self):
        """Gather the partly filled ring so that every step's table has been reassembled; the next step starts a fresh ring."""
        i = self.step_no
        if i % self.n_slots:
            self._exchange((i // self.n_slots) & 1, i % self.n_slots)
        self.step_no = ((i + self.n_slots - 1) // self.n_slots) * self.n_slots

    def _exchange(self, r: int, used: int):
        if self._produced is not None:
            self._produced(r)
        self._gather(self.rings[r][:used], used)           # raises on failure: the caller does not continue without the collective
        if self._consumed is not None:
            self._consumed(r)                              # e.g. record an event behind the collective on its stream
        self.exchanges += 1


def slice_columns(columns: Sequence, shard: Shard):
    """The rank's rows of each (host or device) column: plain views, nothing is copied."""
    return [c[shard.row_lo:shard.row_hi] for c in columns]
