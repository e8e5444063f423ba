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
    """
    import torch
    packed = torch.cat([z_local, status_local.to(z_local.dtype).reshape(-1, 1)], dim=1).contiguous()
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return packed
    world = dist.get_world_size()
    n_local = torch.tensor([packed.shape[0]], device=packed.device, dtype=torch.int64)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    if packed.shape[0] < nmax:
        pad = torch.zeros((nmax - packed.shape[0], packed.shape[1]), device=packed.device, dtype=packed.dtype)
        packed = torch.cat([packed, pad], dim=0)
    out = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(out, packed)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)
