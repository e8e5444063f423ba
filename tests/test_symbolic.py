"""Symbolic front end (product) against the oracle's independently derived patterns and values."""
import numpy as np
import pytest

from conftest import load_golden
from _dag_eval import evaluate

import dto_amd
from dto_amd import problems as P


@pytest.mark.parametrize("fixture,builder", [
    ("pendulum_T6.json", P.build_pendulum), ("cartpole_T5.json", P.build_cartpole),
    ("acrobot_T5.json", P.build_acrobot), ("car_T6.json", P.build_car)])
def test_local_patterns_match_oracle(fixture, builder):
    g = load_golden(fixture)
    d = builder(T=3, evaluate_hessian=True)["dynamics"][0]
    # bit-exact, 1-based, CSC order (src/dynamics.jl:29,35)
    assert d.jacobian_sparsity == g["dynamics_jacobian_sparsity"]
    assert d.hessian_sparsity == g["dynamics_hessian_sparsity"]


def test_appendix_c_counts():
    # nnz of local Jacobian / Hessian
    exp = {"pendulum": (9, 4), "cartpole": (19, 9), "acrobot": (26, 52), "car": (13, 8)}
    for name, (nj, nh) in exp.items():
        d = getattr(P, f"build_{name}")(T=3, evaluate_hessian=True)["dynamics"][0]
        assert (d.num_jacobian, d.num_hessian) == (nj, nh)


def test_pendulum_euler_known_answer():
    """test/dynamics.jl:37-46 at x = u = y = ones: residual and Jacobian entries in closed form."""
    d = dto_amd.Dynamics(P.euler_implicit_test, 2, 2, 1)
    env = {("x", 0): 1.0, ("x", 1): 1.0, ("u", 0): 1.0, ("y", 0): 1.0, ("y", 1): 1.0}
    val = evaluate(d.evaluate_expr, env)
    assert np.allclose(val, [-0.1, 0.7354830360965464], rtol=0, atol=1e-12)
    assert list(zip(*d.jacobian_sparsity)) == [(1, 1), (2, 2), (2, 3), (1, 4), (2, 4), (1, 5), (2, 5)]
    jv = evaluate(d.jacobian_expr, env)
    assert np.allclose(jv, [-1, -1, -0.1, 1, 0.5300365620566452, -0.1, 1.01], rtol=0, atol=1e-12)


def test_derivatives_match_finite_differences():
    rng = np.random.default_rng(0)
    d = P.build_acrobot(T=3, evaluate_hessian=True)["dynamics"][0]
    pt = rng.random(13)
    names = [("x", i) for i in range(4)] + [("u", 0)] + [("y", i) for i in range(4)] + [("lam", i) for i in range(4)]
    env = dict(zip(names, pt))
    J = np.zeros((4, 9))
    for (r, c), v in zip(zip(*d.jacobian_sparsity), evaluate(d.jacobian_expr, env)):
        J[r - 1, c - 1] = v
    h = 1e-6
    for j in range(9):
        ep, em = dict(env), dict(env)
        ep[names[j]] += h
        em[names[j]] -= h
        fd = (np.array(evaluate(d.evaluate_expr, ep)) - np.array(evaluate(d.evaluate_expr, em))) / (2 * h)
        assert np.allclose(J[:, j], fd, atol=1e-7)
    # Hessian symmetric and consistent with the Jacobian's directional derivative
    H = np.zeros((9, 9))
    for (r, c), v in zip(zip(*d.hessian_sparsity), evaluate(d.hessian_expr, env)):
        H[r - 1, c - 1] = v
    assert np.allclose(H, H.T, atol=1e-12)
    lam = pt[9:]
    for j in range(9):
        ep, em = dict(env), dict(env)
        ep[names[j]] += h
        em[names[j]] -= h
        def jt_lam(e):
            Jm = np.zeros((4, 9))
            for (r, c), v in zip(zip(*d.jacobian_sparsity), evaluate(d.jacobian_expr, e)):
                Jm[r - 1, c - 1] = v
            return Jm.T @ lam
        fd = (jt_lam(ep) - jt_lam(em)) / (2 * h)
        assert np.allclose(H[:, j], fd, atol=1e-6)


def test_zero_folding_controls_sparsity():
    # `0.0 * dot(x - xT, x - xT) + dot(u, u)` (examples/car/car.jl:37): the x terms fold away
    c = P.build_car(T=3, evaluate_hessian=True)["objective"][0]
    assert list(zip(*c.sparsity)) == [(4, 4), (5, 5)]


def test_user_jacobian_ctor_dense_column_major():
    """src/dynamics.jl:59-101 / test/solve.jl:182: dense 2x5 pattern, column-major."""
    d = dto_amd.Dynamics(P.double_integrator, P.double_integrator_grad, 2, 2, 1)
    assert d.num_jacobian == 10 and d.num_hessian == 0
    assert list(zip(*d.jacobian_sparsity)) == [(i, j) for j in range(1, 6) for i in range(1, 3)]
    vals = evaluate(d.jacobian_expr, {})
    assert vals == [-1.0, 0.0, -1.0, -1.0, 0.0, -1.0, 1.0, 0.0, 0.0, 1.0]


def test_general_constraint_folding_for_the_solver():
    """solver.py:fold_general_constraint -- stage-local general rows become stage rows (test/solve.jl:273's use);
    multipliers map back to the reference order [dynamics; stage; general]; coupling rows are refused."""
    from dto_amd.solver import fold_general_constraint
    p = P.build_ref_general(user_jacobian=False)
    new_cons, mu_map = fold_general_constraint(p["dynamics"], p["objective"], p["constraints"], p["general_constraint"], True)
    T, n = p["T"], p["n"]
    assert [c.num_constraint for c in new_cons] == [0] * (T - 1) + [2]
    n_dyn = (T - 1) * n
    assert list(mu_map) == list(range(n_dyn)) + [n_dyn, n_dyn + 1]
    env = {("x", 0): 0.3, ("x", 1): -0.2}
    assert np.allclose(evaluate(new_cons[-1].evaluate_expr, env), [0.3 - 1.0, -0.2 - 0.0])
    nz = n * T + (T - 1)
    coupling = dto_amd.GeneralConstraint(lambda z, w: z[0:1] + z[nz - 1:nz], nz, 0, evaluate_hessian=True)
    assert fold_general_constraint(p["dynamics"], p["objective"], p["constraints"], coupling, True) is None
    # mixed with an existing stage constraint and an inequality row
    con = dto_amd.Constraint(lambda x, u, w: x[0:1] - 2.0, n, 0, evaluate_hessian=True)
    gen = dto_amd.GeneralConstraint(lambda z, w: np.array([z[nz - 1] * z[nz - 2], z[3] - 1.0], dtype=object), nz, 0,
                                    indices_inequality=[1], evaluate_hessian=True)
    cons = [dto_amd.Constraint() for _ in range(T - 1)] + [con]
    new_cons, mu_map = fold_general_constraint(p["dynamics"], p["objective"], cons, gen, True)
    assert new_cons[1].num_constraint == 1 and new_cons[-1].num_constraint == 2 and new_cons[-1].indices_inequality == [2]
    # internal rows: dyn..., stage 2: general row 2, stage T: own row, general row 1
    assert list(mu_map[n_dyn:]) == [n_dyn + 1 + 1, n_dyn + 0, n_dyn + 1 + 0]
