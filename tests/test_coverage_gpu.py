"""§8 rows not covered elsewhere, through the C-ABI on the GPU: GeneralConstraint rows (a9), the user-Jacobian
dynamics constructor (a2), per-stage parameters, against the oracle built from the same formulas in sympy."""
import numpy as np
import pytest

from conftest import product_solver
from test_eval_gpu import close

pytestmark = pytest.mark.gpu


def _oracle_general(user_jacobian):
    from oracle import dto_oracle as O, sympy_models as S
    T, n, m = 11, 2, 1
    xT = [1.0, 0.0]
    dt = S.Dynamics(S.double_integrator, n, n, m, evaluate_hessian=not user_jacobian)
    ct = S.Cost(lambda x, u, w: S.fl(0.1) * S.dot(x, x) + S.fl(0.1) * S.dot(u, u), n, m, evaluate_hessian=not user_jacobian)
    cT = S.Cost(lambda x, u, w: S.fl(0.1) * S.dot(x, x), n, 0, evaluate_hessian=not user_jacobian)
    nz = n * T + m * (T - 1)
    gc = S.GeneralConstraint(lambda z, w: [z[nz - 2] - S.fl(xT[0]), z[nz - 1] - S.fl(xT[1])], nz, 0,
                             evaluate_hessian=not user_jacobian)
    bounds = [S.Bound(n, m, state_lower=[0, 0], state_upper=[0, 0])] + [S.Bound(n, m)] * (T - 2) + [S.Bound(n, 0)]
    return O.NLPData([dt] * (T - 1), [ct] * (T - 1) + [cT], [S.Constraint() for _ in range(T)], bounds,
                     evaluate_hessian=not user_jacobian, general_constraint=gc)


def test_general_constraint_rows_and_exact_hessian():
    """test/solve.jl:227-296 problem: values/Jacobian rows of the GeneralConstraint are appended last
    (src/data.jl:72-75); its Hessian is empty (linear), so the exact-Hessian callback works."""
    s, _ = product_solver("ref_general", 11, evaluate_hessian=True)
    n = s.nlp
    onlp = _oracle_general(False)
    assert n.jacobian_structure() == onlp.jacobian_structure()
    assert n.hessian_lagrangian_structure() == onlp.hessian_lagrangian_structure()
    assert n.num_constraint == onlp.num_constraint == 22 and int(n.sizes.num_constraint_general) == 2
    rng = np.random.default_rng(0)
    z, mu = rng.random(n.num_variables), rng.random(n.num_constraint)
    c = np.zeros(n.num_constraint); n.eval_constraint(c, z); close(c, onlp.eval_constraint(z))
    J = np.zeros(n.num_jacobian); n.eval_constraint_jacobian(J, z); close(J, onlp.eval_constraint_jacobian(z))
    H = np.zeros(int(n.sizes.nnz_hess_key)); n.eval_hessian_lagrangian(H, z, 0.7, mu)
    close(H, onlp.eval_hessian_lagrangian(z, 0.7, mu))
    lo, hi = n.constraint_bounds
    assert np.all(lo == 0) and np.all(hi == 0)


def test_user_jacobian_dynamics_dense_pattern():
    """test/solve.jl:140-225: Dynamics(constraint, constraint_jacobian, ...) -> dense column-major 2x5 blocks."""
    s, _ = product_solver("ref_general", 11, evaluate_hessian=False)
    n = s.nlp
    onlp = _oracle_general(True)
    assert n.features_available() == ["Grad", "Jac"]
    assert n.num_jacobian == 10 * 10 + 2          # dense 2x5 per stage + the general rows
    rng = np.random.default_rng(1)
    z = rng.random(n.num_variables)
    J = np.zeros(n.num_jacobian); n.eval_constraint_jacobian(J, z)
    dense = np.zeros((n.num_constraint, n.num_variables))
    for (r, c), v in zip(n.jacobian_structure(), J):
        dense[r - 1, c - 1] = v
    dense_o = np.zeros_like(dense)
    for (r, c), v in zip(onlp.jacobian_structure(), onlp.eval_constraint_jacobian(z)):
        dense_o[r - 1, c - 1] = v
    assert np.max(np.abs(dense - dense_o)) < 1e-12    # same matrix; the user-Jacobian pattern also stores the zeros
    c = np.zeros(n.num_constraint); n.eval_constraint(c, z); close(c, onlp.eval_constraint(z))
    with pytest.raises(Exception):
        n.eval_hessian_lagrangian(np.zeros(1), z, 1.0, np.zeros(n.num_constraint))


def test_per_stage_parameters():
    """num_parameter > 0 in every object, parameters passed like Solver(...; parameters=...) (src/solver.jl:10)."""
    import sympy as sp
    from oracle import dto_oracle as O, sympy_models as S
    T, n, m, nw = 8, 2, 1, 2
    s, p = product_solver("param_pendulum", T)
    nlp = s.nlp
    params = p["parameters"]

    def pend(x, u, w):
        return [x[1], u[0] / (w[0] * S.fl(0.5) * S.fl(0.5)) - S.fl(9.81) * sp.sin(x[0]) / S.fl(0.5)
                - S.fl(0.1) * x[1] / (w[0] * S.fl(0.5) * S.fl(0.5))]

    def dyn(y, x, u, w):
        xm = [S.fl(0.5) * (a + b) for a, b in zip(x, y)]
        f = pend(xm, u, w)
        return [yi - (xi + S.fl(0.05) * fi) for yi, xi, fi in zip(y, x, f)]

    dt = S.Dynamics(dyn, n, n, m, num_parameter=nw, evaluate_hessian=True)
    ct = S.Cost(lambda x, u, w: S.fl(0.1) * (x[0] - w[1]) ** 2 + S.fl(0.1) * x[1] ** 2 + S.fl(0.1) * u[0] ** 2, n, m,
                num_parameter=nw, evaluate_hessian=True)
    cT = S.Cost(lambda x, u, w: S.fl(10.0) * (x[0] - w[1]) ** 2, n, 0, num_parameter=nw, evaluate_hessian=True)
    con = S.Constraint(lambda x, u, w: [x[0] ** 2 + u[0] - w[0]], n, m, num_parameter=nw, indices_inequality=[1], evaluate_hessian=True)
    conT = S.Constraint(lambda x, u, w: [x[0] - w[1]], n, 0, num_parameter=nw, evaluate_hessian=True)
    onlp = O.NLPData([dt] * (T - 1), [ct] * (T - 1) + [cT], [con] * (T - 1) + [conT], [S.Bound(n, m)] * (T - 1) + [S.Bound(n, 0)],
                     evaluate_hessian=True, parameters=params)
    assert nlp.num_parameters == nw * T
    assert nlp.jacobian_structure() == onlp.jacobian_structure()
    assert nlp.hessian_lagrangian_structure() == onlp.hessian_lagrangian_structure()
    rng = np.random.default_rng(2)
    z, mu = rng.random(nlp.num_variables), rng.random(nlp.num_constraint)
    assert abs(nlp.eval_objective(z) - onlp.eval_objective(z)) <= 1e-8 * max(1.0, abs(onlp.eval_objective(z)))
    g = np.zeros(nlp.num_variables); nlp.eval_objective_gradient(g, z); close(g, onlp.eval_objective_gradient(z))
    c = np.zeros(nlp.num_constraint); nlp.eval_constraint(c, z); close(c, onlp.eval_constraint(z))
    J = np.zeros(nlp.num_jacobian); nlp.eval_constraint_jacobian(J, z); close(J, onlp.eval_constraint_jacobian(z))
    H = np.zeros(int(nlp.sizes.nnz_hess_key)); nlp.eval_hessian_lagrangian(H, z, 1.3, mu)
    close(H, onlp.eval_hessian_lagrangian(z, 1.3, mu))
    # and the solver consumes the same parameters (one inequality row per stage, barrier path)
    import torch
    z0 = torch.tensor(0.1 * np.ones((1, nlp.num_variables)), device="cuda")
    zo = torch.zeros_like(z0)
    lam = torch.zeros((1, nlp.num_constraint), device="cuda", dtype=torch.float64)
    status, iters = s.solve_batch(z0.data_ptr(), 1, nlp.num_variables, zo.data_ptr(), nlp.num_variables, lam.data_ptr(), nlp.num_constraint)
    assert status[0] == 1, (status, iters)
    zs = zo.cpu().numpy()[0]
    cs = onlp.eval_constraint(zs)
    lo, _ = onlp.constraint_bounds
    viol = np.where(np.isneginf(lo), np.maximum(cs, 0), np.abs(cs))
    assert np.max(viol) <= 1e-6
