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


def gather_trajectories(z_local, status_local, dist, sink=None, max_bytes=8 << 30, force_collective=False):
    """All-gather [B_local, Nz] trajectories (+ status as an extra column) from every rank.

    Shards may have different sizes (B not divisible by the world size): they are padded to the largest
    shard for the collective and trimmed afterwards.  Without `sink` the result is a [B_total, Nz + 1] tensor on every
    rank, rows in global instance order, the last column the per-instance solver status.

    Memory plan (at the bench's default batch every rank would otherwise hold 8 x 21 GB of receive buffer plus a
    21 GB staging copy).  The exchange runs in row chunks sized so that the receive buffer of one collective stays below
    `max_bytes` (default 8 GiB) -- world x rows x (Nz + 1) x 8 bytes -- plus one staging chunk of rows x (Nz + 1) x 8:
      * `sink(first_global_row, rows)` given: every gathered chunk is handed to it (a [n, Nz + 1] view, valid until the next
        chunk) and nothing else is kept: resident extra memory <= max_bytes * (1 + 1 / world).  Returns the row count.
      * no sink: the chunks are assembled into the full result -- on the device if it fits `max_bytes`, else in host memory.
    """
    import torch
    # (force_collective: run the collectives with a single rank too -- bench.py under DTO_BENCH_FORCE_DIST=1, the N > 1 path on a
    #  one-GPU box)
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collective):
        full = torch.cat([z_local, status_local.to(z_local.dtype).reshape(-1, 1)], dim=1)
        if sink is not None:
            sink(0, full)
            return int(full.shape[0])
        return full
    world = dist.get_world_size()
    n_local = torch.tensor([z_local.shape[0]], device=z_local.device, dtype=torch.int64)
    counts = torch.zeros(world, device=z_local.device, dtype=torch.int64)
    dist.all_gather_into_tensor(counts, n_local)
    counts = [int(c) for c in counts.tolist()]
    first = [sum(counts[:r]) for r in range(world)]
    nmax, width = max(counts), z_local.shape[1] + 1
    total = sum(counts)
    rows_per = max(1, int(max_bytes // (world * width * z_local.element_size())))
    if sink is None and nmax <= rows_per:
        # small payload: one collective, the result stays on the device
        packed = torch.zeros((nmax, width), device=z_local.device, dtype=z_local.dtype)
        packed[: z_local.shape[0], :-1] = z_local
        packed[: z_local.shape[0], -1] = status_local.to(z_local.dtype)
        out = torch.empty((world * nmax, width), device=z_local.device, dtype=z_local.dtype)
        dist.all_gather_into_tensor(out, packed)
        del packed
        if all(c == nmax for c in counts):
            return out
        return torch.cat([out[r * nmax: r * nmax + c] for r, c in enumerate(counts)], dim=0)
    result = None
    if sink is None:
        result = torch.empty((total, width), dtype=z_local.dtype)   # host memory: the full result does not fit the budget
        sink = lambda g0, rows: result[g0:g0 + rows.shape[0]].copy_(rows)
    nloc = z_local.shape[0]
    packed = torch.zeros((min(rows_per, nmax), width), device=z_local.device, dtype=z_local.dtype)
    out = torch.empty((world * packed.shape[0], width), device=z_local.device, dtype=z_local.dtype)
    for r0 in range(0, nmax, rows_per):
        nr = min(rows_per, nmax - r0)
        mine = max(0, min(nloc - r0, nr))
        pk, ob = packed[:nr], out[: world * nr]
        if mine < nr:
            pk[mine:].zero_()
        if mine > 0:
            pk[:mine, :-1] = z_local[r0:r0 + mine]
            pk[:mine, -1] = status_local[r0:r0 + mine].to(z_local.dtype)
        dist.all_gather_into_tensor(ob, pk)
        for r in range(world):
            valid = max(0, min(counts[r] - r0, nr))
            if valid > 0:
                sink(first[r] + r0, ob[r * nr: r * nr + valid])
    return result if result is not None else total
