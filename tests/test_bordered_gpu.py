"""GeneralConstraint rows that couple several knots on the solver path: the rows are the
dense border of the block-tridiagonal KKT matrix; the stage part is solved by the device kernels, the border by a Schur
complement (csrc/dto_solver.cpp: bordered_step / general_solve_batch).  Reference: src/general_constraint.jl:18-59 (rows
appended behind the stage rows: src/data.jl:72-75); test/solve.jl:227-296 is the reference's own (stage-local) use."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_coupled(name, **kw):
    from oracle import dto_oracle as O, sympy_models as S
    p = S.build_coupled(name, **kw)
    return O.NLPData(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True,
                     general_constraint=p["general_constraint"])


from test_solve_gpu import kkt_report  # noqa: E402


def _oracle_acrobot_coupled(T):
    from oracle import dto_oracle as O, sympy_models as S
    p = S.build("acrobot", T, evaluate_hessian=True)
    n, m = 4, 1
    nz = n * T + m * (T - 1)
    off = lambda t: (t - 1) * (n + m)
    gc = S.GeneralConstraint(lambda z, w: [z[off(3)] - z[off(6)] - S.fl(0.2), z[off(2) + 1] + z[off(7) + 1] + z[off(4) + n]],
                             nz, 0, evaluate_hessian=True)
    return O.NLPData(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, general_constraint=gc)


def test_bordered_kkt_step_matches_dense_solve_of_the_oracles_matrix():
    """One regularised Newton-KKT step at random (x, mu) of the acrobot with two coupling rows: dto_kkt_step_batch against a
    dense solve of K = [H + dw I, J'; J, -dc I] assembled from the ORACLE's derivatives, general rows included.  1e-8 of the
    solution norm (the tolerance of every KKT-step test)."""
    import torch
    import dto_amd
    from dto_amd import problems as P
    T, B = 8, 3
    p = P.build_acrobot_coupled(T=T)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True,
                       general_constraint=p["general_constraint"], options=dto_amd.Options(general_rows="border"), name="acrobot_coupled")
    n = s.nlp
    nz, nc = n.num_variables, n.num_constraint
    onlp = _oracle_acrobot_coupled(T)
    assert (onlp.num_variables, onlp.num_constraint) == (nz, nc) and nc == 4 * (T - 1) + 8 + 2
    rng = np.random.default_rng(1)
    Z = 0.5 * rng.standard_normal((B, nz))
    MU = rng.standard_normal((B, nc))
    dw, dc = 0.8, 1e-6
    z, mu = torch.tensor(Z, device="cuda"), torch.tensor(MU, device="cuda")
    dx = torch.full((B, nz), float("nan"), device="cuda", dtype=torch.float64)
    dl = torch.full((B, nc), float("nan"), device="cuda", dtype=torch.float64)
    ok = s.kkt_step_batch(z.data_ptr(), B, nz, mu.data_ptr(), nc, dw, dc, dx.data_ptr(), nz, dl.data_ptr(), nc)
    torch.cuda.synchronize()
    for b in range(B):
        K = np.zeros((nz + nc, nz + nc))
        for (r, c), v in zip(onlp.hessian_lagrangian_structure(), onlp.eval_hessian_lagrangian(Z[b], 1.0, MU[b])):
            K[r - 1, c - 1] += v
        J = np.zeros((nc, nz))
        for (r, c), v in zip(onlp.jacobian_structure(), onlp.eval_constraint_jacobian(Z[b])):
            J[r - 1, c - 1] = v
        K[:nz, :nz] += dw * np.eye(nz)
        K[nz:, :nz] = J
        K[:nz, nz:] = J.T
        K[nz:, nz:] = -dc * np.eye(nc)
        rhs = -np.concatenate([onlp.eval_objective_gradient(Z[b]) + J.T @ MU[b], onlp.eval_constraint(Z[b])])
        sol = np.linalg.solve(K, rhs)
        scale = np.max(np.abs(sol))
        assert np.max(np.abs(dx[b].cpu().numpy() - sol[:nz])) <= 1e-8 * scale, np.max(np.abs(dx[b].cpu().numpy() - sol[:nz])) / scale
        assert np.max(np.abs(dl[b].cpu().numpy() - sol[nz:])) <= 1e-8 * scale
        ev = np.linalg.eigvalsh(K)
        assert ok == bool(np.sum(ev < 0) == nc)      # the reported inertia is that of the whole bordered matrix


def test_reference_general_constraint_solve_with_a_row_coupling_two_knots():
    """test/solve.jl:227-296 (double integrator, T = 11, x_1 fixed by bounds, terminal state through the GeneralConstraint)
    with one more general row coupling knots 4 and 8.  Same asserts as the reference plus: the coupling holds, and the
    returned point is stationary for the full Lagrangian (multipliers in the reference order [dynamics; stage; general])."""
    import dto_amd
    from dto_amd import problems as P
    p = P.build_ref_general_coupled()
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True,
                       general_constraint=p["general_constraint"], options=dto_amd.Options(general_rows="border"), name="ref_general_coupled")
    rng = np.random.Generator(np.random.PCG64(5))
    dto_amd.initialize_states(s, dto_amd.linear_interpolation(p["x1"], p["xT"], p["T"]))
    dto_amd.initialize_controls(s, [rng.standard_normal(1) for _ in range(p["T"] - 1)])
    assert dto_amd.solve(s) == 1, (s.status, s.iterations)
    x_sol, u_sol = dto_amd.get_trajectory(s)
    assert np.linalg.norm(x_sol[0] - p["x1"]) < 1e-3          # test/solve.jl:294
    assert np.linalg.norm(x_sol[-1] - p["xT"]) < 1e-3         # test/solve.jl:295
    i4, i8, tot = p["coupling"]
    z = s._solution
    assert abs(z[i4] + z[i8] - tot) < 1e-6
    n = s.nlp
    g = np.zeros(n.num_variables); n.eval_objective_gradient(g, z)
    Jv = np.zeros(n.num_jacobian); n.eval_constraint_jacobian(Jv, z)
    J = np.zeros((n.num_constraint, n.num_variables))
    for (r, c), v in zip(n.jacobian_structure(), Jv):
        J[r - 1, c - 1] = v
    c = np.zeros(n.num_constraint); n.eval_constraint(c, z)
    assert np.max(np.abs(c)) < 1e-6
    r = g + J.T @ s._duals
    free = np.ones(n.num_variables, bool); free[:2] = False     # x_1 is fixed by bounds
    assert np.max(np.abs(r[free])) < 1e-5
    # a convex QP with linear constraints: Newton's method needs a handful of iterations
    assert s.iterations <= 10


def test_device_border_agrees_with_the_host_border(monkeypatch):
    """Round 4: the border algebra (r_x = grad f + J' mu, the general rows of J, the n_g x n_g Schur complement, v = v0 - Y drho)
    runs on the device; DTO_BORDER_HOST=1 keeps round 3's host algebra.  Same step from both (different summation order in the
    n_g x n_g products only: 1e-10 relative), in a batch large enough that every kernel sees several instances."""
    import torch
    import dto_amd
    from dto_amd import problems as P
    T, B = 8, 40
    p = P.build_acrobot_coupled(T=T)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True,
                       general_constraint=p["general_constraint"], options=dto_amd.Options(general_rows="border"), name="acrobot_coupled")
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    rng = np.random.default_rng(3)
    z = torch.tensor(0.5 * rng.standard_normal((B, nz)), device="cuda")
    mu = torch.tensor(rng.standard_normal((B, nc)), device="cuda")
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DTO_BORDER_HOST", mode)
        dx = torch.full((B, nz), float("nan"), device="cuda", dtype=torch.float64)
        dl = torch.full((B, nc), float("nan"), device="cuda", dtype=torch.float64)
        ok = s.kkt_step_batch(z.data_ptr(), B, nz, mu.data_ptr(), nc, 0.8, 1e-6, dx.data_ptr(), nz, dl.data_ptr(), nc)
        torch.cuda.synchronize()
        out[mode] = (dx.cpu().numpy(), dl.cpu().numpy(), ok)
    scale = max(np.max(np.abs(out["1"][0])), np.max(np.abs(out["1"][1])))
    assert np.all(np.isfinite(out["0"][0])) and out["0"][2] == out["1"][2]
    assert np.max(np.abs(out["0"][0] - out["1"][0])) <= 1e-10 * scale and np.max(np.abs(out["0"][1] - out["1"][1])) <= 1e-10 * scale


@pytest.mark.parametrize("total,active", [(0.05, True), (-0.2, True), (5.0, False)])
def test_inequality_row_coupling_two_knots(total, active):
    """an INEQUALITY GeneralConstraint row (src/general_constraint.jl:15-19, src/data.jl:146) that couples
    knots 4 and 8, x_4[1] + x_8[1] <= total, next to the equality rows of test/solve.jl:227-296.  The row is carried with a slack
    in the border (s / nu on its diagonal, mu / nu in its right-hand side), mu follows Ipopt's monotone rule.  Checked: primal
    feasibility, the sign of the multiplier, complementarity at compl_inf_tol, stationarity of the full Lagrangian; with the
    binding total the row is active, with the loose one the solution is that of the problem without the row."""
    import dto_amd
    from dto_amd import problems as P
    p = P.build_ref_general_coupled(inequality=total)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True,
                       general_constraint=p["general_constraint"], options=dto_amd.Options(general_rows="border"), name="ref_general_coupled_ineq")
    rng = np.random.Generator(np.random.PCG64(5))
    dto_amd.initialize_states(s, dto_amd.linear_interpolation(p["x1"], p["xT"], p["T"]))
    dto_amd.initialize_controls(s, [rng.standard_normal(1) for _ in range(p["T"] - 1)])
    assert dto_amd.solve(s) == 1, (s.status, s.iterations)
    i4, i8, tot = p["coupling"]
    z, lam = s._solution, s._duals
    # KKT conditions from the ORACLE's callbacks for the same problem with the same GeneralConstraint (round 6, VERDICT r5 item 1:
    # until then these came from the product's own callbacks): oracle/sympy_models.py: build_coupled, tests/test_solve_gpu.py: kkt_report
    onlp = _oracle_coupled("ref_general_coupled", total=total)
    rep = kkt_report(onlp, z, lam)
    assert rep["violation"] <= 1e-6 and rep["stationarity"] <= 1e-5 and rep["sign_ok"] and rep["compl"] <= 1e-3, rep
    nu, gi = lam[-1], onlp.eval_constraint(z)[-1]   # the inequality row is the last constraint row
    if active:
        # (mu_target = 1e-4 in the reference Options mapping: the slack stops at s = mu_target / nu, not at zero)
        assert abs(z[i4] + z[i8] - tot) < 2e-2 and nu > 1e-3 and abs(nu * gi) <= 2e-4
    else:
        assert z[i4] + z[i8] < tot - 0.1 and nu < 1e-4
    assert s.iterations <= 40


def test_inequality_coupling_row_on_the_nonlinear_pendulum():
    """The same on nonlinear dynamics: pendulum swing-up, T = 50 (examples/pendulum/pendulum.jl), theta_15 + theta_35 <= total.
    Without the row the sum is 1.449: total = 1 binds (nu > 0, the sum ends at the bound up to mu_target / nu), total = 6 does not
    (the trajectory of the problem without the row, nu ~ mu_target / slack).  KKT conditions from the ORACLE's callbacks for the
    same problem with the same GeneralConstraint (oracle/sympy_models.py: build_coupled; round 6)."""
    import dto_amd
    from dto_amd import problems as P

    def solve(total):
        p = P.build_pendulum_coupled(T=50, total=total, inequality=True) if total is not None else P.build_pendulum(T=50, evaluate_hessian=True)
        s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True,
                           general_constraint=p.get("general_constraint"), options=dto_amd.Options(general_rows="border"), name="pendulum_coupled" if total is not None else "pendulum")
        xs, us = p["guess"](np.random.Generator(np.random.PCG64(0)))
        dto_amd.initialize_states(s, xs); dto_amd.initialize_controls(s, [0.1 * u for u in us])
        assert dto_amd.solve(s) == 1, (total, s.status, s.iterations)
        return s, p

    s0, _ = solve(None)
    i15, i35 = 14 * 3, 34 * 3
    free_sum = s0._solution[i15] + s0._solution[i35]
    assert 1.2 < free_sum < 1.7
    for total in (1.0, 6.0):
        s, p = solve(total)
        z, lam = s._solution, s._duals
        onlp = _oracle_coupled("pendulum_coupled", total=total)
        rep = kkt_report(onlp, z, lam)              # the ORACLE's callbacks (round 6), not the product's
        assert rep["violation"] <= 1e-6 and rep["stationarity"] <= 1e-5 and rep["sign_ok"] and rep["compl"] <= 2e-4, rep
        nu = lam[-1]
        assert s.iterations <= 40
        if total < free_sum:
            assert abs(z[i15] + z[i35] - total) < 1e-3 and nu > 0.1
        else:
            assert np.max(np.abs(z - s0._solution)) < 1e-3 and nu < 1e-3


def test_batch_of_512_bordered_instances():
    """no bordered test ran more than 40 instances.  test/solve.jl:227-296 with the coupling inequality row
    x_4[1] + x_8[1] <= 0.05, 512 seeded guesses through dto_solve_batch on the BORDERED path (device border, host-driven filter
    loop): every instance converges; 16 of them against the oracle's KKT conditions; all end in the same minimiser (the problem
    is convex)."""
    import torch
    import dto_amd
    from dto_amd import problems as P
    B = 512
    p = P.build_ref_general_coupled(inequality=0.05)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True,
                       general_constraint=p["general_constraint"], options=dto_amd.Options(general_rows="border"), name="ref_general_coupled_ineq")
    assert s.general_rows_path == "border"
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    Z = np.zeros((B, nz))
    for b in range(B):
        rng = np.random.Generator(np.random.PCG64(b))
        dto_amd.initialize_states(s, dto_amd.linear_interpolation(p["x1"], p["xT"], p["T"]))
        dto_amd.initialize_controls(s, [rng.standard_normal(1) for _ in range(p["T"] - 1)])
        Z[b] = s._z0
    z0 = torch.tensor(Z, device="cuda")
    zo = torch.full((B, nz), float("nan"), device="cuda", dtype=torch.float64)
    lo = torch.full((B, nc), float("nan"), device="cuda", dtype=torch.float64)
    st, it = s.solve_batch(z0.data_ptr(), B, nz, zo.data_ptr(), nz, lo.data_ptr(), nc)
    torch.cuda.synchronize()
    assert np.all(st == 1), (np.bincount(st), it.max())
    Zs, Ls = zo.cpu().numpy(), lo.cpu().numpy()
    onlp = _oracle_coupled("ref_general_coupled", total=0.05)
    for b in range(0, B, B // 16):
        rep = kkt_report(onlp, Zs[b], Ls[b])
        assert rep["violation"] <= 1e-6 and rep["stationarity"] <= 1e-5 and rep["sign_ok"] and rep["compl"] <= 1e-3, (b, rep)
    assert np.max(np.abs(Zs - Zs[0])) <= 1e-4
