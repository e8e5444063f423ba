"""The fused UPDATE+EVAL pass (csrc/dto_kkt_kernels.hpp: k_update_eval; csrc/dto_solver.cpp: dto_solver_iterate) takes the
step of iteration k while it evaluates iteration k+1 -- one pass over the iterate instead of two.  It must not change a
bit: the same expressions, the values used from registers instead of being re-read.  DTO_FUSE_UPDATE=0 runs k_update and
k_stage_eval one after the other (the library reads the switch at every dto_solver_iterate call)."""
import os

import numpy as np
import pytest

from conftest import product_solver

pytestmark = pytest.mark.gpu

NAMES = ["z", "multipliers", "dz", "dmultipliers", "z_lower", "z_upper", "slack", "slack_multipliers", "dslack"]


def _guesses(s, p, B, seed=0):
    import dto_amd
    rng = np.random.Generator(np.random.PCG64(seed))
    Z = np.zeros((B, s.nlp.num_variables))
    for b in range(B):
        xs, us = p["guess"](rng)
        dto_amd.initialize_states(s, xs)
        dto_amd.initialize_controls(s, us)
        Z[b] = s._z0
    return Z


def _run(s, Z, fused, calls):
    import torch
    old = os.environ.get("DTO_FUSE_UPDATE")
    os.environ["DTO_FUSE_UPDATE"] = "1" if fused else "0"
    try:
        d = torch.tensor(Z, device="cuda")
        s.begin_batch(d.data_ptr(), Z.shape[0], Z.shape[1])
        assert s.fused_update() == fused      # a silent fall-back to the two-kernel sequence would compare it with itself
        for n in calls:
            s.iterate_batch(n)
        out = {k: s.peek_batch(k) for k in NAMES}
        out["stats"] = s.stats_batch()
        return out
    finally:
        if old is None:
            del os.environ["DTO_FUSE_UPDATE"]
        else:
            os.environ["DTO_FUSE_UPDATE"] = old


# barrier models (bounds: cartpole; inequality rows + bounds: car), equality-only (pendulum, acrobot), fixed end points as
# bounds (acrobot_bounds), quasi-Newton records (pendulum without Hessians); several tiles with a ragged last one
@pytest.mark.parametrize("model,T,B,hess", [("pendulum", 50, 70, True), ("acrobot", 101, 130, True), ("car", 51, 66, True),
                                            ("cartpole", 200, 3, True), ("acrobot_bounds", 101, 65, True),
                                            ("pendulum", 50, 64, False)])
def test_fused_update_does_not_change_a_bit(model, T, B, hess):
    if hess:
        s, p = product_solver(model, T, evaluate_hessian=True)
    else:   # per-stage SR1 records; the default of evaluate_hessian=False (limited-memory) does not fuse UPDATE with EVAL at all
        import dto_amd
        from dto_amd import problems as P
        p = getattr(P, f"build_{model}")(T=T, evaluate_hessian=False)
        s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=False,
                           options=dto_amd.Options(hessian_approximation="sr1"), name=model)
    Z = _guesses(s, p, B, seed=3)
    # calls of 1 and 2 iterations run nothing fused; 7 -> 6 fused passes, 4 -> 2: the number per call is even
    calls = [7, 1, 4, 2, 9]
    a = _run(s, Z, False, calls)
    b = _run(s, Z, True, calls)
    for k in ("iterations", "status", "objective", "alpha", "delta_w", "mu"):
        assert np.array_equal(a["stats"][k], b["stats"][k]), k
    assert int(a["stats"]["iterations"].max()) >= 9
    for n in NAMES:
        assert np.array_equal(a[n], b[n]), n


def test_fused_update_over_full_solves_with_repacking():
    """Full solves (instances finish at different iterations and leave the tiles: dto_solver_repack between the calls)."""
    import torch
    s, p = product_solver("acrobot", 101)
    B = 300
    Z = _guesses(s, p, B, seed=9)
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    res = {}
    for fused in (False, True):
        os.environ["DTO_FUSE_UPDATE"] = "1" if fused else "0"
        try:
            d = torch.tensor(Z, device="cuda")
            xo = torch.zeros((B, nz), device="cuda", dtype=torch.float64)
            mo = torch.zeros((B, nc), device="cuda", dtype=torch.float64)
            st, it = s.solve_batch(d.data_ptr(), B, nz, xo.data_ptr(), nz, mo.data_ptr(), nc)
            torch.cuda.synchronize()
            res[fused] = (st.copy(), it.copy(), xo.cpu().numpy(), mo.cpu().numpy())
        finally:
            del os.environ["DTO_FUSE_UPDATE"]
    assert np.all(res[True][0] == 1)
    for x, y in zip(res[False], res[True]):
        assert np.array_equal(x, y)
