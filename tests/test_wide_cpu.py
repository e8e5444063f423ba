"""CPU-side checks of the wide-stage (configs[4], n = 64) model: symbolic model vs the oracle restatement, the
constant-table / variable-entry / nonlinear-remainder split the plugin generator relies on, layout totals."""
import numpy as np
import pytest

from _dag_eval import evaluate

import dto_amd
from dto_amd import problems as P
from dto_amd.plugin import Structure, generate_source
from dto_amd.symbolic import expr as E


@pytest.fixture(scope="module")
def padded():
    return P.build_acrobot_padded(T=3)


def _env(rng):
    x, u, y, lam = rng.random(64), rng.random(1), rng.random(64), rng.random(64)
    env = {}
    for nm, v in (("x", x), ("u", u), ("y", y), ("lam", lam)):
        for i, val in enumerate(v):
            env[(nm, i)] = float(val)
    return env, x, u, y, lam


def test_symbolic_model_matches_oracle_restatement(padded):
    from oracle.padded_model import PaddedAcrobot
    d, c = padded["dynamics"][0], padded["objective"][0]
    om = PaddedAcrobot(64)
    env, x, u, y, lam = _env(np.random.default_rng(0))
    assert (d.num_jacobian, d.num_hessian) == (64 * 129, 52)
    assert np.max(np.abs(np.array(evaluate(d.evaluate_expr, env)) - om.residual(x, u, y))) < 1e-13
    J = np.zeros((64, 129))
    J[np.array(d.jacobian_sparsity[0]) - 1, np.array(d.jacobian_sparsity[1]) - 1] = evaluate(d.jacobian_expr, env)
    assert np.max(np.abs(J - om.jacobian(x, u, y))) < 1e-13
    H = np.zeros((129, 129))
    H[np.array(d.hessian_sparsity[0]) - 1, np.array(d.hessian_sparsity[1]) - 1] = evaluate(d.hessian_expr, env)
    assert np.max(np.abs(H - om.hessian(x, u, y, lam))) < 1e-13
    g, _ = om.cost_grad_hess(x, u)
    assert np.max(np.abs(np.array(evaluate(c.gradient_expr, env)) - g)) < 1e-15


def test_constant_part_plus_remainder_reproduces_residual(padded):
    """What csrc/dto_wide_kernels.hpp computes: d = FE_const [x;u;y] + remainder, remainder by symbolic substitution."""
    d = padded["dynamics"][0]
    x, u, y = E.variables("x", 64), E.variables("u", 1), E.variables("y", 64)
    wrt = list(x) + list(u) + list(y)
    fe = np.zeros((64, 129))
    const_cols = {r: [] for r in range(64)}
    nvar = 0
    for r1, c1, e in zip(d.jacobian_sparsity[0], d.jacobian_sparsity[1], d.jacobian_expr):
        if e.is_const:
            fe[r1 - 1, c1 - 1] = float(e.value)
            const_cols[r1 - 1].append(c1 - 1)
        else:
            nvar += 1
    assert nvar == 18
    rem = [E.substitute([d.evaluate_expr[r]], {wrt[c]: E.const(0.0) for c in const_cols[r]})[0] for r in range(64)]
    assert [r for r in range(64) if not rem[r].is_zero()] == [2, 3]  # the two velocity rows of the acrobot are linear
    env, xv, uv, yv, _ = _env(np.random.default_rng(3))
    want = np.array(evaluate(d.evaluate_expr, env))
    got = fe @ np.concatenate([xv, uv, yv]) + np.array(evaluate(rem, env))
    assert np.max(np.abs(got - want)) < 1e-13


def test_wide_plugin_source_and_structure_rules(padded):
    st = Structure(padded["dynamics"], padded["objective"], padded["constraints"], None, True)
    assert st.wide and len(st.kinds) == 3
    src = generate_source(st, "acrobot_padded")
    assert '#include "dto_wide_kernels.hpp"' in src and "dto_kkt_kernels.hpp" not in src
    assert "NJV = 18" in src and "NNL = 2" in src and "WIDE_N = 64" in src
    # 17 .. 63 states (round 4): a callbacks-only plugin of the problem's own size (the KKT kernels are built for 64 states: the
    # solver embeds such a problem, solver.py: pad_to_wide) -- no k_wide_step instantiation in it
    d20 = dto_amd.Dynamics(lambda y, x, u, w: y - x, 20, 20, 1, evaluate_hessian=True)
    cost = dto_amd.Cost(lambda x, u, w: dto_amd.dot(x, x), 20, 1, evaluate_hessian=True)
    costT = dto_amd.Cost(lambda x, u, w: dto_amd.dot(x, x), 20, 0, evaluate_hessian=True)
    st20 = Structure([d20], [cost, costT], [dto_amd.Constraint(), dto_amd.Constraint()], None, True)
    assert st20.wide and st20.wide_n == 20 and not st20.wide_solver
    src20 = generate_source(st20, "m20")
    assert "WIDE_N = 20" in src20 and "launch_wide<Model>" not in src20 and "launch_wide_eval<Model>" in src20
    # up to four actions per knot (round 4: the action block is factored in place, csrc/dto_wide_kernels.hpp phase 5) ...
    d2 = dto_amd.Dynamics(lambda y, x, u, w: y - x, 20, 20, 2, evaluate_hessian=True)
    cost2 = dto_amd.Cost(lambda x, u, w: dto_amd.dot(x, x), 20, 2, evaluate_hessian=True)
    st2 = Structure([d2], [cost2, costT], [dto_amd.Constraint(), dto_amd.Constraint()], None, True)
    assert st2.wide_nu == 2 and "WIDE_NU = 2" in generate_source(st2, "m20u2")
    # ... more than that is still refused
    bad = dto_amd.Dynamics(lambda y, x, u, w: y - x, 20, 20, 5, evaluate_hessian=True)
    cost5 = dto_amd.Cost(lambda x, u, w: dto_amd.dot(x, x), 20, 5, evaluate_hessian=True)
    with pytest.raises(ValueError):
        Structure([bad], [cost5, costT], [dto_amd.Constraint(), dto_amd.Constraint()], None, True)
    # stage constraints in the wide range (round 6): the plugin gets `Con` classes for the evaluator callbacks; the tile KKT kernels
    # have no stage rows, so such a structure is never a solver plugin itself (solver.py: pad_to_wide carries the rows as
    # auxiliary states of the embedding)
    con = dto_amd.Constraint(lambda x, u, w: [x[0] - 1.0, x[1] * u[0]], 20, 1, indices_inequality=[2], evaluate_hessian=True)
    stc = Structure([d20], [cost, costT], [con, dto_amd.Constraint()], None, True)
    srcc = generate_source(stc, "m20c")
    assert stc.wide and not stc.wide_solver and "struct Model::Con<0>" in srcc and "NC = 2, NJ = 3, NH = 2" in srcc and "CON = 0" in srcc
    d64 = dto_amd.Dynamics(lambda y, x, u, w: y - x, 64, 64, 1, evaluate_hessian=True)
    c64 = dto_amd.Cost(lambda x, u, w: dto_amd.dot(x, x), 64, 1, evaluate_hessian=True)
    c64T = dto_amd.Cost(lambda x, u, w: dto_amd.dot(x, x), 64, 0, evaluate_hessian=True)
    con64 = dto_amd.Constraint(lambda x, u, w: [x[0] - 1.0], 64, 1, evaluate_hessian=True)
    assert not Structure([d64], [c64, c64T], [con64, dto_amd.Constraint()], None, True).wide_solver


def test_full_horizon_layout_totals():
    """configs[4]: T = 2000 -> N_z, N_c, nnz_J as derived in SURVEY.md Appendix C terms (n=64, m=1, dense stage Jacobian)."""
    p = P.build_acrobot_padded(T=2000)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot_padded")
    T, n, m = 2000, 64, 1
    assert s.nlp.num_variables == (T - 1) * (n + m) + n
    assert s.nlp.num_constraint == (T - 1) * n
    assert s.nlp.num_jacobian == (T - 1) * n * (2 * n + m)
    lo, hi = s.nlp.variable_bounds
    assert np.all(lo[:n] == hi[:n]) and np.all(lo[-n:] == hi[-n:]) and hi[-n] == pytest.approx(np.pi)


def test_embedding_of_a_24_state_problem_in_the_64_state_kernels():
    """solver.py: pad_to_wide -- the padded dynamics reproduce the original residuals in their first 24 rows and y_k - x_k in
    the padding rows, the cost is unchanged, the padding states are fixed at zero, and zmap / mumap pick the original
    variables / rows out of the padded layout in the original order."""
    from dto_amd.solver import pad_to_wide
    n, T = 24, 5
    p = P.build_acrobot_padded(T=T, n=n, terminal="physical")
    out = pad_to_wide(p["dynamics"], p["objective"], p["constraints"], p["bounds"], True)
    assert out is not None
    dyn, obj, cons, bnds, zmap, mumap, musign = out
    assert np.all(musign == 1.0)
    assert dyn[0] is dyn[1] and obj[0] is obj[1] and obj[-1] is not obj[0]           # object sharing preserved (stage classes)
    assert (dyn[0].num_state, dyn[0].num_next_state, dyn[0].num_action) == (64, 64, 1) and obj[-1].num_state == 64
    rng = np.random.default_rng(2)
    x, u, y = rng.random(64), rng.random(1), rng.random(64)
    env = {("x", i): float(v) for i, v in enumerate(x)}
    env.update({("y", i): float(v) for i, v in enumerate(y)}); env[("u", 0)] = float(u[0])
    r_pad = np.array(evaluate(dyn[0].evaluate_expr, env))
    r_org = np.array(evaluate(p["dynamics"][0].evaluate_expr, env))
    assert np.max(np.abs(r_pad[:n] - r_org)) == 0.0 and np.max(np.abs(r_pad[n:] - (y[n:] - x[n:]))) == 0.0
    assert evaluate(obj[0].evaluate_expr, env) == evaluate(p["objective"][0].evaluate_expr, env)
    for b in bnds:
        assert np.all(b.state_lower[n:] == 0.0) and np.all(b.state_upper[n:] == 0.0)
    nz_pad = (T - 1) * 65 + 64
    z_pad = np.arange(nz_pad, dtype=float)
    z = z_pad[zmap]
    assert len(zmap) == (T - 1) * (n + 1) + n and len(mumap) == (T - 1) * n
    assert np.array_equal(z[:n], np.arange(n)) and z[n] == 64 and np.array_equal(z[n + 1:2 * n + 1], 65 + np.arange(n))
    assert np.array_equal(mumap[:n], np.arange(n)) and mumap[n] == 64
    # two actions keep their places behind the 64 padded states
    d2 = dto_amd.Dynamics(lambda y, x, u, w: y - x - 0.1 * u[0] * x - 0.2 * u[1], 20, 20, 2, evaluate_hessian=True)
    c2 = dto_amd.Cost(lambda x, u, w: dto_amd.dot(x, x) + dto_amd.dot(u, u), 20, 2, evaluate_hessian=True)
    cT = dto_amd.Cost(lambda x, u, w: dto_amd.dot(x, x), 20, 0, evaluate_hessian=True)
    out2 = pad_to_wide([d2, d2], [c2, c2, cT], [dto_amd.Constraint() for _ in range(3)],
                       [dto_amd.Bound(20, 2), dto_amd.Bound(20, 2), dto_amd.Bound(20, 0)], True)
    assert out2 is not None and out2[0][0].num_action == 2
    assert np.array_equal(out2[4][20:22], [64, 65]) and out2[4][22] == 66 and len(out2[4]) == 2 * 22 + 20
    # not eligible: more than four actions, or already 64 states
    assert pad_to_wide(P.build_acrobot_padded(T=3)["dynamics"], *[P.build_acrobot_padded(T=3)[k] for k in ("objective", "constraints", "bounds")], True) is None


def test_three_action_model_matches_oracle_restatement():
    """The several-action variant of the padded model (problems.py: padded_torque / padded_action_cost; a test model of the action
    block, not a BASELINE.json configuration) against the oracle's own restatement of it (oracle/padded_model.py: torque)."""
    from oracle.padded_model import PaddedAcrobot
    m = 3
    p = P.build_acrobot_padded(T=3, m=m)
    d, c = p["dynamics"][0], p["objective"][0]
    om = PaddedAcrobot(64, m)
    rng = np.random.default_rng(0)
    x, u, y, lam = rng.random(64), rng.random(m), rng.random(64), rng.random(64)
    env = {}
    for nm, v in (("x", x), ("u", u), ("y", y), ("lam", lam)):
        for i, val in enumerate(v):
            env[(nm, i)] = float(val)
    assert np.max(np.abs(np.array(evaluate(d.evaluate_expr, env)) - om.residual(x, u, y))) < 1e-13
    J = np.zeros((64, 128 + m))
    J[np.array(d.jacobian_sparsity[0]) - 1, np.array(d.jacobian_sparsity[1]) - 1] = evaluate(d.jacobian_expr, env)
    assert np.max(np.abs(J - om.jacobian(x, u, y))) < 1e-13
    H = np.zeros((128 + m, 128 + m))
    H[np.array(d.hessian_sparsity[0]) - 1, np.array(d.hessian_sparsity[1]) - 1] = evaluate(d.hessian_expr, env)
    assert np.max(np.abs(H - om.hessian(x, u, y, lam))) < 1e-13
    assert np.count_nonzero(H[64:64 + m, 64:64 + m]) >= 2          # the action block of the dynamics Hessian is not empty
    g, W = om.cost_grad_hess(x, u)
    assert np.max(np.abs(np.array(evaluate(c.gradient_expr, env)) - g)) < 1e-15
    Hc = np.zeros((64 + m, 64 + m))
    Hc[np.array(c.solver_sparsity[0]) - 1, np.array(c.solver_sparsity[1]) - 1] = evaluate(c.solver_hessian_expr, env)
    assert np.max(np.abs(Hc - W)) < 1e-15


def test_blockwise_kkt_residual_matches_dense_assembly():
    """oracle/padded_model.py: kkt_residual_blockwise (the full-horizon check of tests/test_wide_gpu.py) against the dense
    assembly of the same oracle on a small case."""
    from oracle.padded_model import PaddedAcrobot, dense_derivatives, kkt_residual_blockwise
    om, T = PaddedAcrobot(64, 2), 4
    rng = np.random.default_rng(0)
    z, lam = rng.random((T - 1) * 66 + 64), rng.random((T - 1) * 64)
    _, g, c, J, _ = dense_derivatives(om, T, z, lam, 1.0)
    c2, r2 = kkt_residual_blockwise(om, T, z, lam)
    assert np.max(np.abs(c - c2)) == 0.0 and np.max(np.abs(g + J.T @ lam - r2)) < 1e-14


def test_parametric_model_matches_oracle_restatement():
    """Stage parameters w_t = [torque gain, state-cost weight] of the 64-state test model (problems.py: build_acrobot_padded(parameters=...))
    against the oracle's restatement with the same numbers (oracle/padded_model.py: PaddedAcrobot(parameters=...))."""
    from oracle.padded_model import PaddedAcrobot
    par = (1.3, 0.7)
    p = P.build_acrobot_padded(T=3, parameters=par)
    d, c = p["dynamics"][0], p["objective"][0]
    assert d.num_parameter == 2 and c.num_parameter == 2 and len(p["parameters"]) == 3
    om = PaddedAcrobot(64, 1, par)
    rng = np.random.default_rng(0)
    x, u, y, lam = rng.random(64), rng.random(1), rng.random(64), rng.random(64)
    env = {}
    for nm, v in (("x", x), ("u", u), ("y", y), ("lam", lam), ("w", np.array(par))):
        for i, val in enumerate(v):
            env[(nm, i)] = float(val)
    assert np.max(np.abs(np.array(evaluate(d.evaluate_expr, env)) - om.residual(x, u, y))) < 1e-13
    J = np.zeros((64, 129))
    J[np.array(d.jacobian_sparsity[0]) - 1, np.array(d.jacobian_sparsity[1]) - 1] = evaluate(d.jacobian_expr, env)
    assert np.max(np.abs(J - om.jacobian(x, u, y))) < 1e-13
    g, _ = om.cost_grad_hess(x, u)
    assert np.max(np.abs(np.array(evaluate(c.gradient_expr, env)) - g)) < 1e-15
    # the generator accepts parameters on the tile path (the rule that refused them is gone) and passes them to the model code
    st = Structure(p["dynamics"], p["objective"], p["constraints"], None, True)
    src = generate_source(st, "acrobot_padded_par")
    assert st.wide and "NW = 2" in src


def test_stage_constraints_ride_the_embedding_as_auxiliary_states():
    """solver.py: pad_to_wide with `Constraint`s (round 6): row j of c_t becomes the dynamics row y_{n+j} - c_j(x, u) of stage t, the
    rows of the last knot ride on the last stage as functions of its next state; the auxiliary states are fixed at 0 (equality) or
    bounded above by 0 (inequality); mumap / musign pick the rows out in the reference order [dynamics; stage rows] with the sign
    flipped for the stage rows.  Values against the ORACLE's closed forms (oracle/padded_model.py: PaddedStageRows)."""
    from dto_amd.solver import pad_to_wide
    from oracle.padded_model import PaddedAcrobot, PaddedStageRows
    n, T, disc = 24, 5, (0.4, -2.56, 0.1)
    p = P.build_acrobot_padded(T=T, n=n, target=0.4, terminal="physical", stage_constraints=disc)
    out = pad_to_wide(p["dynamics"], p["objective"], p["constraints"], p["bounds"], True)
    assert out is not None
    dyn, obj, cons, bnds, zmap, mumap, musign = out
    assert all(c.num_constraint == 0 for c in cons) and len(dyn) == T - 1
    # three dynamics classes: the first stage (endpoint rows + obstacle row of knot 1), the interior ones, the last (its own
    # obstacle row + the five rows of knot T)
    assert dyn[1] is dyn[2] and dyn[0] is not dyn[1] and dyn[3] is not dyn[1]
    rows = PaddedStageRows(n, 1, T, p["x1"], p["xT"], *disc)
    om = PaddedAcrobot(n)
    rng = np.random.default_rng(11)
    X = rng.random((T, 64)); U = rng.random((T - 1, 1))
    z = np.concatenate([np.concatenate([X[t, :n], U[t]]) if t < T - 1 else X[t, :n] for t in range(T)])
    cs = rows.values(z)
    QS = n + 1
    for t in range(T - 1):
        env = {("x", i): float(v) for i, v in enumerate(X[t])}
        env.update({("y", i): float(v) for i, v in enumerate(X[t + 1])}); env[("u", 0)] = float(U[t, 0])
        r = np.array(evaluate(dyn[t].evaluate_expr, env))
        assert np.max(np.abs(r[:n] - om.residual(X[t, :n], U[t], X[t + 1, :n]))) < 1e-13
        q = rows.rows_of[t]
        assert np.max(np.abs(r[n:n + q] - (X[t + 1, n:n + q] - cs[rows.off[t]:rows.off[t + 1]]))) < 1e-14     # y_aux - c_t(x, u)
        k = n + q
        if t == T - 2:   # the rows of the LAST knot, evaluated at the next state
            assert np.max(np.abs(r[k:k + 5] - (X[t + 1, k:k + 5] - cs[rows.off[T - 1]:]))) < 1e-14
            k += 5
        assert k == n + QS or np.max(np.abs(r[k:n + QS] - X[t + 1, k:n + QS])) == 0.0                        # unused slots: a = 0
        assert np.max(np.abs(r[n + QS:] - (X[t + 1, n + QS:] - X[t, n + QS:]))) == 0.0                       # padding
    # bounds of the auxiliary states: knot 1 none feeds (fixed 0); knot 2: n fixed + one <= 0; interior: one <= 0; last: 1 + 4 fixed + 1
    assert np.all(bnds[0].state_lower[n:] == 0.0) and np.all(bnds[0].state_upper[n:] == 0.0)
    assert np.all(bnds[1].state_lower[n:2 * n] == 0.0) and np.isneginf(bnds[1].state_lower[2 * n]) and np.all(bnds[1].state_upper[n:] == 0.0)
    assert np.isneginf(bnds[2].state_lower[n]) and np.all(bnds[2].state_lower[n + 1:] == 0.0)
    lo_T = bnds[T - 1].state_lower[n:]
    assert np.isneginf(lo_T[0]) and np.all(lo_T[1:5] == 0.0) and np.isneginf(lo_T[5]) and np.all(lo_T[6:] == 0.0)
    # multiplier map: dynamics rows first (+), then the stage rows knot by knot (-)
    nd = (T - 1) * n
    assert len(mumap) == nd + rows.num and np.all(musign[:nd] == 1.0) and np.all(musign[nd:] == -1.0)
    assert np.array_equal(mumap[:n], np.arange(n)) and mumap[n] == 64
    assert np.array_equal(mumap[nd:nd + n + 1], n + np.arange(n + 1))                       # knot 1: aux rows of stage 1
    assert mumap[nd + n + 1] == 64 + n                                                       # knot 2: first aux row of stage 2
    assert np.array_equal(mumap[-5:], (T - 2) * 64 + n + 1 + np.arange(5))                  # knot T: behind stage T-1's own row
    # the problem's own plugin (callbacks) carries the Constraint classes; the embedding's is a plain tile-solver model
    st = Structure(p["dynamics"], p["objective"], p["constraints"], None, True)
    src = generate_source(st, "acrobot24c")
    assert st.wide and not st.wide_solver and "struct Model::Con<2>" in src and "N_CON = 3" in src
    st2 = Structure(dyn, obj, cons, None, True)
    assert st2.wide_solver and len(st2.dyn) == 3
    # too many rows for the padding states: not eligible (a 60-state model with its n endpoint rows)
    p60 = P.build_acrobot_padded(T=4, n=60, terminal="physical", stage_constraints=disc)
    assert pad_to_wide(p60["dynamics"], p60["objective"], p60["constraints"], p60["bounds"], True) is None


def test_pins_to_bounds_restates_single_variable_rows():
    """solver.py: pins_to_bounds -- rows affine in one variable become bounds (equalities on states only, inequalities where that
    side is free), everything else stays a row; positions and coefficients of the restated rows are reported for the multipliers."""
    import numpy as np
    from dto_amd import Bound, Constraint
    from dto_amd.solver import pins_to_bounds
    n, m = 3, 1
    c1 = Constraint(lambda x, u, w: np.array([x[0] - 0.5, 2.0 * x[1] + 1.0, x[0] * x[2] - 1.0, u[0] - 0.3, -x[2] + 0.25, x[0] + 1.0], dtype=object), n, m,
                    indices_inequality=[2, 4, 5], evaluate_hessian=True)
    c2 = Constraint()
    c3 = Constraint(lambda x, u, w: np.array([x[1] - 2.0, x[1] + x[2]], dtype=object), n, 0, evaluate_hessian=True)
    b = [Bound(n, m), Bound(n, m, state_lower=np.array([-1.0, -1.0, -1.0])), Bound(n, 0)]
    out = pins_to_bounds([c1, c2, c3], b, True)
    assert out is not None
    con, bnd, pins, keep = out
    # knot 1: row 0 (x0 = 0.5) pinned; row 1 (2 x1 + 1 <= 0 -> x1 <= -0.5); row 2 nonlinear: stays; row 3 (u0 - 0.3 <= 0 -> u0 <= 0.3);
    # row 4 (-x2 + 0.25 <= 0 -> x2 >= 0.25); row 5 is a second row in x0: stays
    assert [(p[0], p[1], p[2]) for p in pins] == [(0, 0, 0), (1, 0, 1), (3, 0, 3), (4, 0, 2), (6, 2, 1)]
    assert [p[3] for p in pins] == [1.0, 2.0, 1.0, -1.0, 1.0]
    assert keep == [2, 5, 7]
    assert con[0].num_constraint == 2 and con[0].indices_inequality == [] and con[1].num_constraint == 0 and con[2].num_constraint == 1
    assert np.array_equal(bnd[0].state_lower, [0.5, -np.inf, 0.25]) and np.array_equal(bnd[0].state_upper, [0.5, -0.5, np.inf])
    assert np.array_equal(bnd[0].action_upper, [0.3]) and np.array_equal(bnd[0].action_lower, [-np.inf])
    assert np.array_equal(bnd[2].state_lower, [-np.inf, 2.0, -np.inf]) and np.array_equal(bnd[2].state_upper, [np.inf, 2.0, np.inf])
    assert bnd[1] is b[1]
    # nothing to restate: None; an equality on an action stays a row (a fixed action has no interior for the barrier)
    c4 = Constraint(lambda x, u, w: np.array([u[0] - 0.3, x[0] * x[1]], dtype=object), n, m, evaluate_hessian=True)
    assert pins_to_bounds([c4, c2, c2], b, True) is None
