"""The C-ABI entry points added in round 2 (include/dto.h), through ctypes:

* dto_kkt_assemble / dto_kkt_factor / dto_kkt_solve -- the block-tridiagonal LDL^T as a linear solver for a caller that
  keeps its own outer iteration (Ipopt's augmented system with Sigma_x / Sigma_c / delta_w / delta_c on the diagonals),
  against numpy dense solves of the ORACLE's matrices, several right-hand sides per factorisation, inertia against the
  dense eigenvalue count;
* dto_solver_begin_warm / dto_solver_run -- receding-horizon re-solves that keep the interior-point state on the device.
"""
import numpy as np
import pytest

from conftest import product_solver

pytestmark = pytest.mark.gpu


def dense_blocks(onlp, z, mu):
    nz, nc = onlp.num_variables, onlp.num_constraint
    H = np.zeros((nz, nz))
    for (r, c), v in zip(onlp.hessian_lagrangian_structure(), onlp.eval_hessian_lagrangian(z, 1.0, mu)):
        H[r - 1, c - 1] = v
    J = np.zeros((nc, nz))
    for (r, c), v in zip(onlp.jacobian_structure(), onlp.eval_constraint_jacobian(z)):
        J[r - 1, c - 1] = v
    return H, J


@pytest.mark.parametrize("model,T,dw,partitions", [("pendulum", 6, 30.0, 0), ("acrobot", 70, 60.0, 0), ("acrobot", 70, 60.0, 7),
                                                   ("car", 6, 10.0, 0), ("cartpole", 40, 400.0, 3)])
def test_linear_solver_entry_points_match_dense_solves(model, T, dw, partitions):
    import torch
    from oracle import dto_oracle as O, sympy_models as S
    s, _ = product_solver(model, T)
    p = S.build(model, T, evaluate_hessian=True)
    onlp = O.NLPData(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True)
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    rng = np.random.default_rng(17 * T + partitions)
    B, dc = 3, 1e-6
    Z, MU = rng.random((B, nz)), rng.random((B, nc))
    SX, SC = rng.random((B, nz)) * 3.0, rng.random((B, nc)) * 0.5
    SX[:, ::3] = 0.0                                           # some variables without a barrier term
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
    dZ, dMU, dSX, dSC = dev(Z), dev(MU), dev(SX), dev(SC)
    s.set_partitions(partitions)
    try:
        s.kkt_assemble(dZ.data_ptr(), B, nz, dMU.data_ptr(), nc, dw, dc, dSX.data_ptr(), nz, dSC.data_ptr(), nc)
        ok, neg = s.kkt_factor()
        Ks = []
        for b in range(B):
            H, J = dense_blocks(onlp, Z[b], MU[b])
            K = np.block([[H + np.diag(SX[b]) + dw * np.eye(nz), J.T], [J, -np.diag(SC[b]) - dc * np.eye(nc)]])
            Ks.append(K)
            eig = np.linalg.eigvalsh(K)
            assert int(np.sum(eig < 0)) == neg[b], (int(np.sum(eig < 0)), neg[b])      # Sylvester: pivots count the inertia
            assert bool(ok[b]) == (int(np.sum(eig < 0)) == nc)
        for _ in range(3):                                     # several right-hand sides on one assembled system
            RX, RC = rng.standard_normal((B, nz)), rng.standard_normal((B, nc))
            dRX, dRC = dev(RX), dev(RC)
            oX = torch.full((B, nz), float("nan"), device="cuda", dtype=torch.float64)
            oC = torch.full((B, nc), float("nan"), device="cuda", dtype=torch.float64)
            s.kkt_solve(dRX.data_ptr(), nz, dRC.data_ptr(), nc, oX.data_ptr(), nz, oC.data_ptr(), nc)
            torch.cuda.synchronize()
            oX, oC = oX.cpu().numpy(), oC.cpu().numpy()
            for b in range(B):
                sol = np.linalg.solve(Ks[b], np.concatenate([RX[b], RC[b]]))
                scale = np.max(np.abs(sol))
                assert np.max(np.abs(oX[b] - sol[:nz])) <= 1e-8 * scale and np.max(np.abs(oC[b] - sol[nz:])) <= 1e-8 * scale
    finally:
        s.set_partitions(0)
    # an indefinite case: no regularisation, large multipliers -> the factorisation reports the wrong inertia
    # (also with the plain sequential sweep, whose lanes stop storing carries once an attempt is lost -- the
    # single attempt of these entry points must be swept through: the pivot count is the matrix's inertia whatever the
    # neighbouring lanes do, and the solve is K^-1 rhs although the inertia is "wrong")
    MU2 = 300.0 * rng.standard_normal((B, nc))
    dMU2 = dev(MU2)
    for part in (0, 1):
        s.set_partitions(part)
        try:
            s.kkt_assemble(dZ.data_ptr(), B, nz, dMU2.data_ptr(), nc, 0.0, dc)
            ok2, neg2 = s.kkt_factor()
            RX, RC = rng.standard_normal((B, nz)), rng.standard_normal((B, nc))
            dRX, dRC = dev(RX), dev(RC)
            oX = torch.full((B, nz), float("nan"), device="cuda", dtype=torch.float64)
            oC = torch.full((B, nc), float("nan"), device="cuda", dtype=torch.float64)
            s.kkt_solve(dRX.data_ptr(), nz, dRC.data_ptr(), nc, oX.data_ptr(), nz, oC.data_ptr(), nc)
            torch.cuda.synchronize()
            oX, oC = oX.cpu().numpy(), oC.cpu().numpy()
            for b in range(B):
                H, J = dense_blocks(onlp, Z[b], MU2[b])
                K = np.block([[H, J.T], [J, -dc * np.eye(nc)]])
                eig = np.linalg.eigvalsh(K)
                if np.min(np.abs(eig)) > 1e-6:                     # away from singular matrices the counts agree exactly
                    assert int(np.sum(eig < 0)) == neg2[b], (part, b)
                    assert bool(ok2[b]) == (int(np.sum(eig < 0)) == nc)
                    sol = np.linalg.solve(K, np.concatenate([RX[b], RC[b]]))
                    scale = np.max(np.abs(sol))
                    # an indefinite K without pivoting: 1e-6 of the solution (condition numbers up to 1e10 here)
                    assert np.max(np.abs(oX[b] - sol[:nz])) <= 1e-6 * scale and np.max(np.abs(oC[b] - sol[nz:])) <= 1e-6 * scale, (part, b)
        finally:
            s.set_partitions(0)


def test_linear_solver_entry_points_reject_misuse():
    import torch
    import dto_amd
    from dto_amd import capi
    s, _ = product_solver("pendulum", 6)
    s2 = dto_amd.Solver(*[product_solver("pendulum", 6)[1][k] for k in ("dynamics", "objective", "constraints", "bounds")],
                        evaluate_hessian=True, name="pendulum")
    with pytest.raises(capi.DtoError, match="dto_kkt_assemble has not been called"):
        s2._B = 1
        s2.kkt_factor()


def test_warm_started_mpc_resolve_keeps_the_interior_point_state():
    """Receding horizon: solve a batch of pendulum MPC problems, move every measured initial state a little, and re-solve
    (a) cold from the previous solution as the guess and (b) warm with dto_solver_begin_warm.  Both reach the same solutions;
    the warm start, which keeps multipliers, slack/bound multipliers and the barrier parameter, needs clearly fewer iterations."""
    import torch
    import dto_amd
    from dto_amd import problems as P
    T, B = 30, 6
    rng = np.random.default_rng(5)
    p = P.build_mpc_pendulum(T=T)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True,
                       parameters=p["parameters"], name="mpc_pendulum")
    nz, nc, nw = s.nlp.num_variables, s.nlp.num_constraint, s.nlp.num_parameters
    x1s = 0.3 * rng.standard_normal((B, 2))
    goals = np.pi * (0.5 + 0.5 * rng.random(B))
    W = np.stack([np.tile([x1s[b, 0], x1s[b, 1], goals[b]], T) for b in range(B)])
    Z = np.zeros((B, nz))
    for b in range(B):
        dto_amd.initialize_states(s, dto_amd.linear_interpolation(x1s[b], np.array([goals[b], 0.0]), T))
        dto_amd.initialize_controls(s, [0.1 * rng.standard_normal(1) for _ in range(T - 1)])
        Z[b] = s._z0
    z0, w = torch.tensor(Z, device="cuda"), torch.tensor(W, device="cuda")
    z1 = torch.empty_like(z0)
    st, it0 = s.solve_batch(z0.data_ptr(), B, nz, z1.data_ptr(), nz, params_ptr=w.data_ptr(), ldp=nw)
    assert np.all(st == 1)
    mu_end = s.scalar_batch("mu").copy()
    # the plant moved: new measured states
    W2 = W.copy()
    for b in range(B):
        W2[b].reshape(T, 3)[:, :2] += 0.02 * rng.standard_normal(2)
    w2 = torch.tensor(W2, device="cuda")
    # (b) warm: device state kept, only the parameters change
    zw = torch.empty_like(z0)
    s.begin_warm_batch(B, params_ptr=w2.data_ptr(), ldp=nw)
    assert np.allclose(s.scalar_batch("mu"), mu_end)           # the barrier parameter was kept
    st_w, it_w = s.run_batch(zw.data_ptr(), nz)
    # (a) cold: same guess (the previous solution), fresh interior-point state
    zc = torch.empty_like(z0)
    st_c, it_c = s.solve_batch(z1.data_ptr(), B, nz, zc.data_ptr(), nz, params_ptr=w2.data_ptr(), ldp=nw)
    torch.cuda.synchronize()
    assert np.all(st_w == 1) and np.all(st_c == 1), (st_w, st_c)
    assert np.max(np.abs(zw.cpu().numpy() - zc.cpu().numpy())) < 1e-5
    assert it_w.sum() < it_c.sum(), (it_w, it_c)
    # ... and the warm-started solutions are KKT points by the ORACLE's derivatives (this test used to compare
    # HIP solves with each other only).  The MPC family is the oracle's pendulum with the pins x_1 = w[0:2], x_T = (w[2], 0):
    # same Jacobian and gradient, constraint values shifted by the pin data.
    from test_solve_gpu import oracle_for
    lw = torch.empty((B, nc), device="cuda", dtype=torch.float64)
    s.end_batch(zw.data_ptr(), nz, lw.data_ptr(), nc)
    torch.cuda.synchronize()
    onlp = oracle_for("pendulum", T)
    nd = 2 * (T - 1)
    for b in range(B):
        z, lam = zw[b].cpu().numpy(), lw[b].cpu().numpy()
        J = np.zeros((nc, nz))
        for (r, c_), v in zip(onlp.jacobian_structure(), onlp.eval_constraint_jacobian(z)):
            J[r - 1, c_ - 1] = v
        c = onlp.eval_constraint(z).copy()
        x1b, goal = W2[b][:2], W2[b][2]
        c[nd:nd + 2] -= x1b                       # oracle pins x_1 at the origin
        c[nd + 2] += np.pi - goal                 # ... and x_T at (pi, 0)
        assert np.max(np.abs(c)) <= 1e-6, (b, np.max(np.abs(c)))
        assert np.max(np.abs(onlp.eval_objective_gradient(z) + J.T @ lam)) <= 1e-5, b
    # a batch of another size has no state to continue from
    from dto_amd import capi
    with pytest.raises(capi.DtoError, match="same batch size"):
        s.begin_warm_batch(B + 1, params_ptr=w2.data_ptr(), ldp=nw)


def test_repack_of_running_instances_changes_nothing_but_the_cost():
    """dto_solver_repack moves the still-running instances to the leading tiles.  Per-instance arithmetic does not depend on
    the slot an instance sits in, so a solve with repacking (dto_solve_batch does it by itself) returns bit for bit what a
    plain iterate loop without it returns, in the caller's instance order -- including statuses and iteration counts."""
    import torch
    from bench import make_guesses
    s, p = product_solver("acrobot", 101)
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    B = 200                                                    # 4 tiles, the last one partially filled
    Z = make_guesses(s, p, B, seed=77)
    z0 = torch.tensor(Z, device="cuda")
    # (a) no repacking: drive the iteration by hand
    s.begin_batch(z0.data_ptr(), B, nz)
    for _ in range(100):
        s.iterate_batch(10)
        if not np.any(s.scalar_batch("status") == 0):
            break
    za = torch.empty_like(z0); la = torch.empty((B, nc), device="cuda", dtype=torch.float64)
    s.end_batch(za.data_ptr(), nz, la.data_ptr(), nc)
    torch.cuda.synchronize()
    st_a, it_a = s.scalar_batch("status").copy(), s.scalar_batch("iter").copy()
    # (b) explicit repacking between slices
    s.begin_batch(z0.data_ptr(), B, nz)
    counts = []
    for _ in range(100):
        s.iterate_batch(10)
        counts.append(s.repack_batch())
        if counts[-1] == 0:
            break
    zb = torch.empty_like(z0); lb = torch.empty((B, nc), device="cuda", dtype=torch.float64)
    s.end_batch(zb.data_ptr(), nz, lb.data_ptr(), nc)
    torch.cuda.synchronize()
    st_b, it_b = s.scalar_batch("status").copy(), s.scalar_batch("iter").copy()
    assert counts[0] > counts[-1] and sorted(counts, reverse=True) == counts      # instances only ever leave
    # (nearly every instance converges; the few that a run leaves at the iteration limit must be the same ones both ways)
    assert np.array_equal(st_a, st_b) and np.array_equal(it_a, it_b) and np.mean(st_a == 1) >= 0.97, np.mean(st_a == 1)
    assert torch.equal(za, zb) and torch.equal(la, lb)
    # the repacked solves are KKT points by the oracle (not only equal to the unrepacked HIP solves)
    from test_solve_gpu import kkt_report, oracle_for
    onlp = oracle_for("acrobot", 101)
    for b in [b_ for b_ in range(0, B, 25) if st_b[b_] == 1]:
        rep = kkt_report(onlp, zb[b].cpu().numpy(), lb[b].cpu().numpy())
        assert rep["violation"] <= 1e-6 and rep["stationarity"] <= 1e-5, (b, rep)
    # (c) the one-call solve (repacks by itself)
    zc = torch.empty_like(z0)
    st_c, it_c = s.solve_batch(z0.data_ptr(), B, nz, zc.data_ptr(), nz)
    torch.cuda.synchronize()
    assert np.array_equal(st_c, st_a.astype(np.int32)) and np.array_equal(it_c, it_a.astype(np.int32)) and torch.equal(zc, za)
    assert len(set(it_a.tolist())) > 10                         # the instances really finished at different times



def test_repack_on_a_tiny_model_with_more_than_one_tile():
    """with a short horizon the widest vector that dto_solver_repack moves is the filter (48 rows per instance),
    not the iterate: pendulum T = 8 has 23 variables.  130 instances (three tiles) that finish at different iterations,
    solved through dto_solve_batch (which repacks by itself); every solution is checked with the oracle."""
    import torch
    from test_solve_gpu import kkt_report, oracle_for
    s, p = product_solver("pendulum", 8)
    B = 130
    rng = np.random.Generator(np.random.PCG64(21))
    import dto_amd
    Z = np.zeros((B, s.nlp.num_variables))
    for b in range(B):
        xs, us = p["guess"](rng)
        # spread the difficulty: instances start at different distances from the solution
        us = [u * (0.1 + 3.0 * (b % 7)) for u in us]
        xs = [x + 1.5 * (b % 5) * rng.standard_normal(len(x)) for x in xs]
        dto_amd.initialize_states(s, xs)
        dto_amd.initialize_controls(s, us)
        Z[b] = s._z0
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    assert nz < 48
    d = torch.tensor(Z, device="cuda")
    xo = torch.zeros((B, nz), device="cuda", dtype=torch.float64)
    mo = torch.zeros((B, nc), device="cuda", dtype=torch.float64)
    # check_every = 1: a repack opportunity after every iteration
    st, it = s.solve_batch(d.data_ptr(), B, nz, xo.data_ptr(), nz, mo.data_ptr(), nc, check_every=1)
    torch.cuda.synchronize()
    assert np.all(st == 1), np.bincount(st)
    assert it.max() > it.min()            # a staggered finish
    X, MU = xo.cpu().numpy(), mo.cpu().numpy()
    onlp = oracle_for("pendulum", 8)
    for b in range(0, B, 9):
        rep = kkt_report(onlp, X[b], MU[b])
        assert rep["violation"] <= 1e-6 and rep["stationarity"] <= 1e-5 and rep["sign_ok"], (b, rep)

def test_varying_state_dimensions_callbacks_kkt_step_and_solve():
    """Per-stage state / action dimensions on the whole path (dimensions(), src/dynamics.jl:206-211; round-1 verdict: the
    solver refused them): the five callbacks and one regularised KKT step against the oracle, then a converged solve whose
    KKT conditions are evaluated with the oracle.  Such problems run the sequential sweep (one chunk)."""
    import torch
    import dto_amd
    from dto_amd import capi
    from oracle import dto_oracle as O
    from test_layout import varying_dimension_problem
    from test_solve_gpu import kkt_report
    s = dto_amd.Solver(*varying_dimension_problem("product"), evaluate_hessian=True, name="varydims")
    onlp = O.NLPData(*varying_dimension_problem("oracle"), evaluate_hessian=True)
    n = s.nlp
    nz, nc = n.num_variables, n.num_constraint
    rng = np.random.default_rng(12)
    z, mu = rng.random(nz), rng.random(nc)
    rel = lambda got, ref: np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3 * max(1e-300, np.max(np.abs(ref)))))
    g = np.zeros(nz); n.eval_objective_gradient(g, z)
    c = np.zeros(nc); n.eval_constraint(c, z)
    J = np.zeros(n.num_jacobian); n.eval_constraint_jacobian(J, z)
    H = np.zeros(int(n.sizes.nnz_hess_key)); n.eval_hessian_lagrangian(H, z, 0.7, mu)
    assert abs(n.eval_objective(z) - onlp.eval_objective(z)) <= 1e-8 * abs(onlp.eval_objective(z))
    assert rel(g, onlp.eval_objective_gradient(z)) < 1e-8 and rel(c, onlp.eval_constraint(z)) < 1e-8
    assert rel(J, onlp.eval_constraint_jacobian(z)) < 1e-8 and rel(H, onlp.eval_hessian_lagrangian(z, 0.7, mu)) < 1e-8
    # KKT step
    B, dw, dc = 2, 20.0, 1e-6
    Z, MU = rng.random((B, nz)), rng.random((B, nc))
    dZ, dMU = torch.tensor(Z, device="cuda"), torch.tensor(MU, device="cuda")
    dx = torch.full((B, nz), float("nan"), device="cuda", dtype=torch.float64)
    dl = torch.full((B, nc), float("nan"), device="cuda", dtype=torch.float64)
    ok = s.kkt_step_batch(dZ.data_ptr(), B, nz, dMU.data_ptr(), nc, dw, dc, dx.data_ptr(), nz, dl.data_ptr(), nc)
    torch.cuda.synchronize()
    assert s.partitions() == 1
    for b in range(B):
        Hd, Jd = dense_blocks(onlp, Z[b], MU[b])
        K = np.block([[Hd + dw * np.eye(nz), Jd.T], [Jd, -dc * np.eye(nc)]])
        rhs = -np.concatenate([onlp.eval_objective_gradient(Z[b]) + Jd.T @ MU[b], onlp.eval_constraint(Z[b])])
        sol = np.linalg.solve(K, rhs)
        assert np.max(np.abs(np.concatenate([dx[b].cpu().numpy(), dl[b].cpu().numpy()]) - sol)) <= 1e-8 * np.max(np.abs(sol))
    assert ok
    with pytest.raises(capi.DtoError, match="uniform state dimension"):
        s.set_partitions(3)
        try:
            s.kkt_step_batch(dZ.data_ptr(), B, nz, dMU.data_ptr(), nc, dw, dc, dx.data_ptr(), nz, dl.data_ptr(), nc)
        finally:
            s.set_partitions(0)
    # solve
    s._z0[:] = 0.1 * rng.standard_normal(nz)
    assert dto_amd.solve(s) == 1, (s.status, s.iterations)
    rep = kkt_report(onlp, s._solution, s._duals)
    assert rep["violation"] <= 1e-6 and rep["stationarity"] <= 1e-5 and rep["compl"] <= 1e-3 and rep["sign_ok"], rep


def test_receding_horizon_stream_shift_and_warm_resolve():
    """An MPC loop on the device-resident state ("no shift/stream loop"): solve, apply the first
    action to the plant, dto_solver_shift by one knot, re-solve warm with the measured state as the new parameter -- ten times,
    a batch of rollouts at once.  Every re-solve is checked against the oracle's KKT conditions (pendulum with the pins moved),
    the plant follows the plan (the plant is the model plus a small disturbance), and the streamed re-solves need far fewer
    iterations than the first, cold one."""
    import torch
    import dto_amd
    from dto_amd import problems as P
    from test_solve_gpu import oracle_for
    T, B, STEPS, h = 30, 5, 10, 0.05
    rng = np.random.default_rng(11)
    p = P.build_mpc_pendulum(T=T)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True,
                       parameters=p["parameters"], name="mpc_pendulum")
    nz, nc, nw = s.nlp.num_variables, s.nlp.num_constraint, s.nlp.num_parameters
    onlp = oracle_for("pendulum", T)
    nd = 2 * (T - 1)

    def plant(x, u):   # the model's implicit midpoint step, solved by fixed-point iteration (the "real" system of this test)
        y = x.copy()
        for _ in range(50):
            m = 0.5 * (x + y)
            f = np.array([m[1], u / 0.25 - 9.81 * np.sin(m[0]) / 0.5 - 0.1 * m[1] / 0.25])
            y = x + h * f
        return y

    def kkt_ok(z, lam, x1b, goal):
        J = np.zeros((nc, nz))
        for (r, c_), v in zip(onlp.jacobian_structure(), onlp.eval_constraint_jacobian(z)):
            J[r - 1, c_ - 1] = v
        c = onlp.eval_constraint(z).copy()
        c[nd:nd + 2] -= x1b
        c[nd + 2] += np.pi - goal
        return np.max(np.abs(c)) <= 1e-6 and np.max(np.abs(onlp.eval_objective_gradient(z) + J.T @ lam)) <= 1e-5

    x = 0.2 * rng.standard_normal((B, 2))
    goals = np.pi * (0.6 + 0.4 * rng.random(B))
    W = np.stack([np.tile([x[b, 0], x[b, 1], goals[b]], T) for b in range(B)])
    Z = np.zeros((B, nz))
    for b in range(B):
        dto_amd.initialize_states(s, dto_amd.linear_interpolation(x[b], np.array([goals[b], 0.0]), T))
        dto_amd.initialize_controls(s, [0.1 * rng.standard_normal(1) for _ in range(T - 1)])
        Z[b] = s._z0
    z0, w = torch.tensor(Z, device="cuda"), torch.tensor(W, device="cuda")
    zo = torch.empty_like(z0)
    lo = torch.empty((B, nc), device="cuda", dtype=torch.float64)
    st, it_first = s.solve_batch(z0.data_ptr(), B, nz, zo.data_ptr(), nz, lo.data_ptr(), nc, params_ptr=w.data_ptr(), ldp=nw)
    assert np.all(st == 1)
    warm_iters = []
    for k in range(STEPS):
        Zk, Lk = zo.cpu().numpy(), lo.cpu().numpy()
        for b in range(B):
            assert kkt_ok(Zk[b], Lk[b], W[b][:2], goals[b]), (k, b)
        # plant step with the first action (+ a disturbance), horizon shifted by one knot, measured state as the new pin
        for b in range(B):
            x[b] = plant(x[b], Zk[b][2]) + 0.002 * rng.standard_normal(2)
            assert np.linalg.norm(x[b] - Zk[b][3:5]) < 0.02, (k, b)      # the plan's x_2 is where the plant went
            W[b].reshape(T, 3)[:, :2] = x[b]
        w = torch.tensor(W, device="cuda")
        s.shift_batch(1)
        zs = s.peek_batch("z")
        assert np.allclose(zs[:, :nz - 3 - 2], Zk[:, 3:nz - 2])          # knots moved forward by one (x, u stride 3)
        s.begin_warm_batch(B, params_ptr=w.data_ptr(), ldp=nw)
        st, itw = s.run_batch(zo.data_ptr(), nz, lo.data_ptr(), nc)
        assert np.all(st == 1), (k, st)
        warm_iters.append(itw.sum())
    # streaming pays: a warm re-solve costs a fraction of the first (cold) solve
    assert np.mean(warm_iters) < 0.6 * it_first.sum(), (warm_iters, it_first.sum())
