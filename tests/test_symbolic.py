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
