"""Multi-GPU: independent trajectory instances are sharded across ranks (one process per GPU).

The hot path has no cross-instance coupling, so there is NO collective inside an iteration; the only
exchange is the all-gather of converged trajectories (and their status) at the end of a solve / MPC
step, over RCCL (backend "nccl" on ROCm) on xGMI -- or gloo on CPU in the tests.  Payloads are small
(N_z doubles per instance), so the cost is collective latency, not link bandwidth.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_range(num_instances: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of instances owned by `rank`: b in [r*B/R, (r+1)*B/R)."""
    # the rule lives on the C-ABI (dto_shard_range) so that every host language shards identically
    import ctypes as C
    from . import capi
    first, count = C.c_int64(), C.c_int64()
    capi.check(capi.lib().dto_shard_range(int(num_instances), int(rank), int(world), C.byref(first), C.byref(count)))
    return first.value, first.value + count.value


def gather_trajectories(z_local, status_local, dist):
    """All-gather [B_local, Nz] trajectories (+ status as an extra column) from every rank.

    Shards may have different sizes (B not divisible by the world size): they are padded to the largest
    shard for the collective and trimmed afterwards.  Returns a [B_total, Nz + 1] tensor on every rank,
    rows in global instance order; the last column is the per-instance solver status.

    Memory: one staging copy of the local shard and ONE receive buffer of world x largest shard (the
    collective writes into it directly); equal shards are returned as a view of that buffer.
    """
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return torch.cat([z_local, status_local.to(z_local.dtype).reshape(-1, 1)], dim=1)
    world = dist.get_world_size()
    n_local = torch.tensor([z_local.shape[0]], device=z_local.device, dtype=torch.int64)
    counts = torch.zeros(world, device=z_local.device, dtype=torch.int64)
    dist.all_gather_into_tensor(counts, n_local)
    counts = [int(c) for c in counts.tolist()]
    nmax, width = max(counts), z_local.shape[1] + 1
    packed = torch.zeros((nmax, width), device=z_local.device, dtype=z_local.dtype)
    packed[: z_local.shape[0], :-1] = z_local
    packed[: z_local.shape[0], -1] = status_local.to(z_local.dtype)
    out = torch.empty((world * nmax, width), device=z_local.device, dtype=z_local.dtype)
    dist.all_gather_into_tensor(out, packed)
    del packed
    if all(c == nmax for c in counts):
        return out
    return torch.cat([out[r * nmax: r * nmax + c] for r, c in enumerate(counts)], dim=0)
