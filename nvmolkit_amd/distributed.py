"""Multi-GPU sharding of the similarity path: one process per GPU, ``torch.distributed`` (nccl = RCCL on ROCm).

BASELINE.json configs[4]: the QUERY set is sharded (each rank owns its rows of the result), the REFERENCE set starts
sharded and is assembled with a single all-gather over xGMI — 32 MB per rank / 256 MB total at 1M x 2048 bit, ~2 ms
on a ring, invisible next to >1 s of similarity work per rank.  The reference has no collectives at
all; this is new capability.  Molecule batches (conformers, fingerprints) shard with no collective.
"""

from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous, near-equal split of n rows; the first n % world_size ranks get one extra row."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def padded_shard_rows(n: int, world_size: int) -> int:
    """Rows every rank contributes to the all-gather (shards are padded to the largest one)."""
    return (n + world_size - 1) // world_size


def all_gather_rows(shard: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Assemble the (n_total, W) matrix from per-rank row shards laid out by :func:`shard_bounds`.

    One ``all_gather_into_tensor`` of equal (padded) shards, then the padding rows are dropped."""
    world = dist.get_world_size(group)
    rows = padded_shard_rows(n_total, world)
    buf = torch.zeros((rows, shard.shape[1]), dtype=shard.dtype, device=shard.device)
    buf[: shard.shape[0]] = shard
    gathered = torch.empty((rows * world, shard.shape[1]), dtype=shard.dtype, device=shard.device)
    dist.all_gather_into_tensor(gathered, buf, group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, world, r)
        parts.append(gathered[r * rows: r * rows + (hi - lo)])
    return torch.cat(parts, dim=0)
