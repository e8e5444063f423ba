"""The joins of the time-partitioned form inside the launches (csrc/dto_kkt_kernels.hpp: tile_last_arrival -- k_kkt_fwd_sep,
k_kkt_bwd_post, k_part_reduce_conv, k_part_reduce_ls): the chunk wavefront of a tile that finishes LAST runs the tile's joining
step (separator system + inertia verdict, step partials, convergence test, step size) instead of a launch of its own -- 15
dependent launches per iteration instead of 27, which is what a batch of one (the reference's own use,
examples/acrobot/acrobot.jl:126-133) spends its time on.

Same arithmetic in the same order: every state vector must be bit-identical to the separate launches (DTO_FUSE_JOIN=0, read
by the library at every call), over iterations that include failed inertia probes, for one instance, a few, a full tile and
several tiles with a ragged last one, automatic and forced chunk counts."""
import os

import numpy as np
import pytest

from conftest import product_solver
from test_fused_update_gpu import NAMES, _guesses

pytestmark = pytest.mark.gpu


def _run(s, Z, fused, calls, partitions):
    import torch
    old = os.environ.get("DTO_FUSE_JOIN")
    os.environ["DTO_FUSE_JOIN"] = "1" if fused else "0"
    os.environ["DTO_FUSE_JOIN_TILES"] = "64"          # (read at every call, like the switch)
    s.set_partitions(partitions)
    try:
        d = torch.tensor(Z, device="cuda")
        s.begin_batch(d.data_ptr(), Z.shape[0], Z.shape[1])
        assert s.partitions() > 1
        for n in calls:
            s.iterate_batch(n)
        out = {k: s.peek_batch(k) for k in NAMES}
        out["stats"] = s.stats_batch()
        out["nfact"] = s.scalar_batch("nfact").copy()
        out["partitions"] = s.partitions()
        return out
    finally:
        s.set_partitions(0)
        del os.environ["DTO_FUSE_JOIN_TILES"]
        if old is None:
            del os.environ["DTO_FUSE_JOIN"]
        else:
            os.environ["DTO_FUSE_JOIN"] = old


# (the library uses the in-launch joins for batches of at most two tiles -- larger ones were measured 0 - 5 % slower with them:
#  profiles/r05/join_in_launch_sc1_ab.txt -- so DTO_FUSE_JOIN_TILES lifts the limit for the multi-tile cases of this file)
@pytest.mark.parametrize("model,T,B,partitions", [("acrobot", 101, 1, 0), ("acrobot", 1000, 1, 0), ("acrobot", 101, 5, 16),
                                                  ("acrobot", 101, 64, 8), ("acrobot", 101, 130, 5), ("pendulum", 50, 1, 0),
                                                  ("car", 51, 3, 0), ("cartpole", 200, 2, 10), ("acrobot_bounds", 101, 2, 7)])
def test_joins_inside_the_launches_do_not_change_a_bit(model, T, B, partitions):
    s, p = product_solver(model, T, evaluate_hessian=True)
    Z = _guesses(s, p, B, seed=5)
    calls = [6, 1, 9, 4]
    a = _run(s, Z, False, calls, partitions)
    b = _run(s, Z, True, calls, partitions)
    assert a["partitions"] == b["partitions"]
    for k in ("iterations", "status", "objective", "alpha", "delta_w", "mu"):
        assert np.array_equal(a["stats"][k], b["stats"][k]), k
    assert np.array_equal(a["nfact"], b["nfact"])
    if B >= 64:
        assert a["nfact"][:B].max() > sum(calls)       # some instance needed more than one round in some iteration
    for n in NAMES:
        assert np.array_equal(a[n], b[n]), n
