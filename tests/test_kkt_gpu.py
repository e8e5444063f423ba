"""KKT assembly + block-tridiagonal LDL^T solve against a dense numpy solve of the same system (oracle).

The system is the one sketched in the reference at examples/pendulum/pendulum.jl:138-198:
    [ H + dw I   J' ; J   -dc I ] [dx; dmu] = -[ grad f + J' mu ; c ]
with H, J, grad f, c taken from the ORACLE evaluator.  Tolerance: 1e-8 relative to the solution norm
(the dense and the block-tridiagonal factorisations round differently; cond(K) ~ 1e3..1e6 here).
"""
import numpy as np
import pytest

from conftest import product_solver

pytestmark = pytest.mark.gpu


def dense_kkt_solve(onlp, z, mu, dw, dc):
    nz, nc = onlp.num_variables, onlp.num_constraint
    H = np.zeros((nz, nz))
    for (r, c), v in zip(onlp.hessian_lagrangian_structure(), onlp.eval_hessian_lagrangian(z, 1.0, mu)):
        H[r - 1, c - 1] = v
    J = np.zeros((nc, nz))
    for (r, c), v in zip(onlp.jacobian_structure(), onlp.eval_constraint_jacobian(z)):
        J[r - 1, c - 1] = v
    g = onlp.eval_objective_gradient(z)
    cv = onlp.eval_constraint(z)
    K = np.block([[H + dw * np.eye(nz), J.T], [J, -dc * np.eye(nc)]])
    rhs = -np.concatenate([g + J.T @ mu, cv])
    sol = np.linalg.solve(K, rhs)
    eig = np.linalg.eigvalsh(K)
    inertia = (int(np.sum(eig > 0)), int(np.sum(eig < 0)))
    return sol[:nz], sol[nz:], inertia, np.linalg.cond(K)


@pytest.mark.parametrize("model,T,B,dw", [("pendulum", 6, 3, 30.0), ("pendulum", 50, 2, 30.0), ("acrobot", 5, 3, 60.0),
                                          ("acrobot", 70, 2, 60.0), ("cartpole", 5, 2, 400.0), ("car", 6, 3, 10.0),
                                          ("acrobot_bounds", 4, 2, 60.0)])
def test_kkt_step_matches_dense_solve(model, T, B, dw):
    import torch
    from oracle import dto_oracle as O, sympy_models as S
    s, _ = product_solver(model, T)
    n = s.nlp
    p = S.build(model, T, evaluate_hessian=True)
    onlp = O.NLPData(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True)
    nz, nc = n.num_variables, n.num_constraint
    rng = np.random.default_rng(42 + T)
    Z, MU = rng.random((B, nz)), rng.random((B, nc))
    dz, dmu = torch.tensor(Z, device="cuda"), torch.tensor(MU, device="cuda")
    dx = torch.full((B, nz), float("nan"), device="cuda", dtype=torch.float64)
    dl = torch.full((B, nc), float("nan"), device="cuda", dtype=torch.float64)
    dc = 1e-5
    ok = s.kkt_step_batch(dz.data_ptr(), B, nz, dmu.data_ptr(), nc, dw, dc, dx.data_ptr(), nz, dl.data_ptr(), nc)
    torch.cuda.synchronize()
    dx, dl = dx.cpu().numpy(), dl.cpu().numpy()
    for b in range(B):
        rx, rl, inertia, cond = dense_kkt_solve(onlp, Z[b], MU[b], dw, dc)
        assert inertia == (nz, nc), "test point must be quasi-definite; raise dw"
        scale = max(np.max(np.abs(rx)), np.max(np.abs(rl)))
        assert np.max(np.abs(dx[b] - rx)) <= 1e-8 * scale, (np.max(np.abs(dx[b] - rx)), scale, cond)
        assert np.max(np.abs(dl[b] - rl)) <= 1e-8 * scale, (np.max(np.abs(dl[b] - rl)), scale, cond)
    assert ok


@pytest.mark.parametrize("model,T,dw", [("pendulum", 50, 30.0), ("acrobot", 70, 60.0), ("car", 51, 10.0), ("cartpole", 40, 400.0)])
@pytest.mark.parametrize("partitions", [1, 2, 3, 7, 16])
def test_time_partitioned_factorisation_matches_dense_solve(model, T, dw, partitions):
    """The parallel-in-time factorisation (chunks + spikes + separator system) solves the same system as the
    sequential sweep and as numpy's dense solve, for any number of chunks (including chunk lengths that do
    not divide T and chunks that start at a constrained stage)."""
    import torch
    from oracle import dto_oracle as O, sympy_models as S
    s, _ = product_solver(model, T)
    n = s.nlp
    p = S.build(model, T, evaluate_hessian=True)
    onlp = O.NLPData(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True)
    nz, nc = n.num_variables, n.num_constraint
    rng = np.random.default_rng(7 * T + partitions)
    B = 2
    Z, MU = rng.random((B, nz)), rng.random((B, nc))
    dz, dmu = torch.tensor(Z, device="cuda"), torch.tensor(MU, device="cuda")
    dx = torch.full((B, nz), float("nan"), device="cuda", dtype=torch.float64)
    dl = torch.full((B, nc), float("nan"), device="cuda", dtype=torch.float64)
    s.set_partitions(partitions)
    try:
        ok = s.kkt_step_batch(dz.data_ptr(), B, nz, dmu.data_ptr(), nc, dw, 1e-5, dx.data_ptr(), nz, dl.data_ptr(), nc)
        assert s.partitions() == partitions
    finally:
        s.set_partitions(0)
    torch.cuda.synchronize()
    dx, dl = dx.cpu().numpy(), dl.cpu().numpy()
    for b in range(B):
        rx, rl, inertia, cond = dense_kkt_solve(onlp, Z[b], MU[b], dw, 1e-5)
        assert inertia == (nz, nc)
        scale = max(np.max(np.abs(rx)), np.max(np.abs(rl)))
        assert np.max(np.abs(dx[b] - rx)) <= 1e-8 * scale, (np.max(np.abs(dx[b] - rx)), scale, cond)
        assert np.max(np.abs(dl[b] - rl)) <= 1e-8 * scale, (np.max(np.abs(dl[b] - rl)), scale, cond)
    assert ok


def test_inertia_flag_matches_dense_inertia():
    """The negative-pivot count of the block-tridiagonal LDL^T (Sylvester) must agree with the dense
    eigenvalue inertia of the same matrix: correct for small multipliers, wrong once the constraint
    curvature dominates and delta_w = 0."""
    import torch
    from oracle import dto_oracle as O, sympy_models as S
    s, _ = product_solver("pendulum", 50)
    n = s.nlp
    p = S.build("pendulum", 50, evaluate_hessian=True)
    onlp = O.NLPData(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True)
    nz, nc = n.num_variables, n.num_constraint
    rng = np.random.default_rng(1)
    z = rng.random(nz)
    base = rng.standard_normal(nc)
    seen = set()
    for scale, dw in [(0.0, 1e-3), (0.01, 1e-3), (1.0, 0.0), (30.0, 0.0), (300.0, 0.0), (300.0, 1e4)]:
        mu = scale * base
        _, _, inertia, _ = dense_kkt_solve(onlp, z, mu, dw, 1e-8)
        dense_ok = inertia == (nz, nc)
        dz, dmu = torch.tensor(z[None, :], device="cuda"), torch.tensor(mu[None, :], device="cuda")
        dx = torch.zeros((1, nz), device="cuda", dtype=torch.float64)
        dl = torch.zeros((1, nc), device="cuda", dtype=torch.float64)
        ok = s.kkt_step_batch(dz.data_ptr(), 1, nz, dmu.data_ptr(), nc, dw, 1e-8, dx.data_ptr(), nz, dl.data_ptr(), nc)
        assert ok == dense_ok, (scale, dw, inertia, ok)
        seen.add(dense_ok)
    assert seen == {True, False}


@pytest.mark.parametrize("seed", [3, 8])
def test_kkt_step_random_heterogeneous_problem(seed):
    """Random stage models with different stage kinds (see test_layout.random_heterogeneous_problem): one regularised KKT
    step against numpy's dense solve of the oracle's system."""
    import torch
    import dto_amd
    from oracle import dto_oracle as O
    from test_layout import random_heterogeneous_problem
    s = dto_amd.Solver(*random_heterogeneous_problem(seed, "product"), evaluate_hessian=True, name=f"random{seed}")
    onlp = O.NLPData(*random_heterogeneous_problem(seed, "oracle"), evaluate_hessian=True)
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    rng = np.random.default_rng(seed)
    B, dw, dc = 3, 40.0, 1e-5
    Z, MU = rng.random((B, nz)), rng.random((B, nc))
    dz, dmu = torch.tensor(Z, device="cuda"), torch.tensor(MU, device="cuda")
    dx = torch.full((B, nz), float("nan"), device="cuda", dtype=torch.float64)
    dl = torch.full((B, nc), float("nan"), device="cuda", dtype=torch.float64)
    ok = s.kkt_step_batch(dz.data_ptr(), B, nz, dmu.data_ptr(), nc, dw, dc, dx.data_ptr(), nz, dl.data_ptr(), nc)
    torch.cuda.synchronize()
    dx, dl = dx.cpu().numpy(), dl.cpu().numpy()
    for b in range(B):
        rx, rl, inertia, cond = dense_kkt_solve(onlp, Z[b], MU[b], dw, dc)
        assert inertia == (nz, nc), "test point must be quasi-definite; raise dw"
        scale = max(np.max(np.abs(rx)), np.max(np.abs(rl)))
        assert np.max(np.abs(dx[b] - rx)) <= 1e-8 * scale and np.max(np.abs(dl[b] - rl)) <= 1e-8 * scale
    assert ok


def primal_dual_system(onlp, n, x, l, zl, zu, sl, zs, mu, dw, gam, dc=1e-8):
    """Dense primal-dual system of one interior-point iteration from the ORACLE's derivatives (the docstring of
    test_interior_point_step_matches_dense_primal_dual_system spells it out); `n`: the product-side NLPData (bounds only)."""
    nz, nc = n.num_variables, n.num_constraint
    lo, hi = n.variable_bounds
    clo, _ = n.constraint_bounds
    ineq = np.where(np.isneginf(clo))[0]
    H = np.zeros((nz, nz))
    lam_h = l if gam != 0.0 else np.zeros(nc)      # Gauss-Newton fallback drops the constraint curvature
    for (r, c), v in zip(onlp.hessian_lagrangian_structure(), onlp.eval_hessian_lagrangian(x, 1.0, lam_h)):
        H[r - 1, c - 1] = v
    J = np.zeros((nc, nz))
    for (r, c), v in zip(onlp.jacobian_structure(), onlp.eval_constraint_jacobian(x)):
        J[r - 1, c - 1] = v
    g, cv = onlp.eval_objective_gradient(x), onlp.eval_constraint(x)
    fixed = lo == hi
    flo, fhi = np.isfinite(lo) & ~fixed, np.isfinite(hi) & ~fixed
    sig = np.zeros(nz); rz = g + J.T @ l
    sig[flo] += zl[flo] / (x[flo] - lo[flo]); rz[flo] -= mu / (x[flo] - lo[flo])
    sig[fhi] += zu[fhi] / (hi[fhi] - x[fhi]); rz[fhi] += mu / (hi[fhi] - x[fhi])
    D = np.full(nc, dc); rc = cv.copy()
    if len(ineq):
        sv, zv, nu = sl, zs, l[ineq]
        D[ineq] += sv / zv
        rc[ineq] = cv[ineq] + sv - (sv / zv) * (nu - mu / sv)
    K = np.block([[H + np.diag(sig) + dw * np.eye(nz), J.T], [J, -np.diag(D)]])
    rhs = -np.concatenate([rz, rc])
    for i in np.where(fixed)[0]:
        K[i, :] = 0.0; K[:, i] = 0.0; K[i, i] = 1.0; rhs[i] = 0.0
    return K, rhs


@pytest.mark.parametrize("model,T", [("cartpole", 6), ("car", 6), ("car", 40)])
def test_interior_point_step_matches_dense_primal_dual_system(model, T):
    """The step of a REAL interior-point iteration (bounds, barrier, slack-eliminated inequality rows, variables fixed by
    equal bounds) re-derived in numpy from first principles with the oracle's derivatives:
        (H + Sigma + dw I) dz + J' dlam = -(grad f + J' lam - mu/(x - lo) + mu/(hi - x)),   Sigma = z_L/(x - lo) + z_U/(hi - x)
        J_r dz - dc dlam_r           = -c_r                                    (equality rows)
        J_r dz - (s/z_s + dc) dnu_r  = -(c_r + s) + (s/z_s)(nu_r - mu/s)       (rows c_r(x) <= 0 with slack s)
        ds = -(s/z_s)(nu + dnu - mu/s),  fixed variables: dz_i = 0.
    State (after two full iterations) and step are read back through dto_solver_peek."""
    import torch
    import dto_amd
    from oracle import dto_oracle as O, sympy_models as S
    s, p = product_solver(model, T)
    n = s.nlp
    op = S.build(model, T, evaluate_hessian=True)
    onlp = O.NLPData(op["dynamics"], op["objective"], op["constraints"], op["bounds"], evaluate_hessian=True)
    nz, nc = n.num_variables, n.num_constraint
    B = 2
    Z = np.zeros((B, nz))
    for b in range(B):
        xs, us = p["guess"](np.random.Generator(np.random.PCG64(20 + b)))
        dto_amd.initialize_states(s, xs)
        dto_amd.initialize_controls(s, us)
        Z[b] = s._z0
    z0 = torch.tensor(Z, device="cuda")
    s.begin_batch(z0.data_ptr(), B, nz)
    s.iterate_batch(2)
    for op_name in ("eval", "conv", "factor_solve"):
        s.launch_op(op_name)
    torch.cuda.synchronize()
    z, lam, dz, dlam = (s.peek_batch(k) for k in ("z", "multipliers", "dz", "dmultipliers"))
    zl, zu, sl, zs, ds = (s.peek_batch(k) for k in ("z_lower", "z_upper", "slack", "slack_multipliers", "dslack"))
    mu, dw, gam = s.scalar_batch("mu"), s.scalar_batch("delta_w"), s.scalar_batch("gamma")
    lo, hi = n.variable_bounds
    clo, _ = n.constraint_bounds
    ineq = np.where(np.isneginf(clo))[0]
    assert len(ineq) == sl.shape[1]
    fixed = lo == hi
    flo, fhi = np.isfinite(lo) & ~fixed, np.isfinite(hi) & ~fixed
    for b in range(B):
        x, l = z[b], lam[b]
        K, rhs = primal_dual_system(onlp, n, x, l, zl[b], zu[b], sl[b], zs[b], mu[b], dw[b], gam[b])
        if len(ineq):
            sv, zv, nu = sl[b], zs[b], l[ineq]
        sol = np.linalg.solve(K, rhs)
        scale = np.max(np.abs(sol))
        # forward error: 1e-8 of the step, relaxed in proportion to the conditioning of this particular system (a
        # 6-knot swing-up has multiplier steps of 1e7 and cond(K) ~ 1e12); backward error: always at rounding level
        tol = max(1e-8, 1e-15 * np.linalg.cond(K)) * scale
        assert np.max(np.abs(dz[b] - sol[:nz])) <= tol, (np.max(np.abs(dz[b] - sol[:nz])), scale)
        assert np.max(np.abs(dlam[b] - sol[nz:])) <= tol, (np.max(np.abs(dlam[b] - sol[nz:])), scale)
        got = np.concatenate([dz[b], dlam[b]])
        assert np.max(np.abs(K @ got - rhs)) <= 1e-10 * (np.max(np.abs(K)) * np.max(np.abs(got)) + np.max(np.abs(rhs)))
        if len(ineq):
            ds_ref = -(sv / zv) * (nu + sol[nz:][ineq] - mu[b] / sv)
            assert np.max(np.abs(ds[b] - ds_ref)) <= 1e-8 * max(scale, np.max(np.abs(ds_ref)))
        assert np.all(x[flo] > lo[flo]) and np.all(x[fhi] < hi[fhi]) and np.all(sl[b] > 0) and np.all(zs[b] > 0)
