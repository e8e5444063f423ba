"""An independent check of the SOLVER half (missing 7): Ipopt is not in this image, so the minimisers the GPU
iteration reports are compared with what a third-party NLP solver -- scipy.optimize.minimize(method="trust-constr"), a
trust-region interior-point / SQP method that shares no code and no algorithm with this repository -- finds on the ORACLE's
callbacks (objective, gradient, constraints, Jacobian, Hessian of the Lagrangian: oracle/dto_oracle.py, the restatement of
src/moi.jl:1-120) from the same initial guess, as the reference's own solve tests do with Ipopt (test/solve.jl:128-137).

  * pendulum T = 11 (the reference example's horizon) and T = 50 (BASELINE configs[0]): the same minimiser to 1e-6;
  * acrobot T = 25: the problem has many local minimisers (a swing-up in 1.25 s) and the two methods end in different
    ones from the same guess -- documented, not hidden, not ranked; trust-constr STARTED AT the GPU's minimiser must stay
    there (it confirms a local minimiser of the oracle's NLP).
  (BASELINE configs[2], acrobot T = 1000, from the bench's guesses: tools/third_party_cfg3.py, profiles/r05/.)
"""
import numpy as np
import pytest

from conftest import product_solver

pytestmark = pytest.mark.gpu


def _oracle_problem(model, T):
    import scipy.sparse as sp
    from scipy.optimize import NonlinearConstraint
    from test_solve_gpu import oracle_for
    onlp = oracle_for(model, T)
    nz, nc = onlp.num_variables, onlp.num_constraint
    js = np.array(onlp.jacobian_structure()) - 1
    hs = np.array(onlp.hessian_lagrangian_structure()) - 1
    mat = lambda v, idx, shape: sp.coo_matrix((v, (idx[:, 0], idx[:, 1])), shape=shape).tocsr()
    con = NonlinearConstraint(onlp.eval_constraint, 0.0, 0.0,
                              jac=lambda z: mat(onlp.eval_constraint_jacobian(z), js, (nc, nz)),
                              hess=lambda z, v: mat(onlp.eval_hessian_lagrangian(z, 0.0, v), hs, (nz, nz)))
    kw = dict(jac=onlp.eval_objective_gradient, hess=lambda z: mat(onlp.eval_hessian_lagrangian(z, 1.0, np.zeros(nc)), hs, (nz, nz)),
              method="trust-constr", constraints=[con], options=dict(gtol=1e-9, xtol=1e-12, maxiter=3000))
    return onlp, kw


def _gpu_solve(model, T, z0):
    import torch
    s, p = product_solver(model, T)
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    d = torch.tensor(z0[None, :], device="cuda")
    xo = torch.zeros((1, nz), device="cuda", dtype=torch.float64)
    mo = torch.zeros((1, nc), device="cuda", dtype=torch.float64)
    st, it = s.solve_batch(d.data_ptr(), 1, nz, xo.data_ptr(), nz, mo.data_ptr(), nc)
    torch.cuda.synchronize()
    assert st[0] == 1, (st, it)
    return xo.cpu().numpy()[0], int(it[0])


@pytest.mark.parametrize("T", [11, 50])
def test_pendulum_minimiser_agrees_with_trust_constr(T):
    from scipy.optimize import minimize
    from oracle.cpu_port import guesses
    onlp, kw = _oracle_problem("pendulum", T)
    z0 = guesses("pendulum", T, 1, 1000)[0][0]            # linear interpolation + u ~ N(0, 1), seeded (examples/pendulum/pendulum.jl:85-86)
    res = minimize(onlp.eval_objective, z0, **kw)
    assert res.status in (1, 2) and res.constr_violation < 1e-10
    z, it = _gpu_solve("pendulum", T, z0)
    assert np.max(np.abs(z - res.x)) <= 1e-6 * max(1.0, np.max(np.abs(res.x))), np.max(np.abs(z - res.x))
    assert abs(onlp.eval_objective(z) - res.fun) <= 1e-8 * abs(res.fun)


def test_acrobot_T25_minimiser_is_confirmed_by_trust_constr():
    from scipy.optimize import minimize
    from oracle.cpu_port import guesses
    T = 25
    onlp, kw = _oracle_problem("acrobot", T)
    z0 = guesses("acrobot", T, 1, 1000)[0][0]
    z, it = _gpu_solve("acrobot", T, z0)
    f_gpu = onlp.eval_objective(z)
    assert np.max(np.abs(onlp.eval_constraint(z))) <= 1e-6
    # (1) trust-constr started at the GPU's point does not leave it and cannot lower the objective
    pol = minimize(onlp.eval_objective, z, **kw)
    assert np.max(np.abs(pol.x - z)) <= 1e-3 * np.max(np.abs(z)), np.max(np.abs(pol.x - z))
    assert pol.fun >= f_gpu - 1e-6 * abs(f_gpu), (pol.fun, f_gpu)
    # (2) from the common guess the two methods end in DIFFERENT local minimisers of this problem (a swing-up in 1.25 s has many):
    #     recorded, not ranked -- which basin a method falls into from a far-away guess is not a property of either (round 4's filter
    #     line search: GPU 4375 / trust-constr 6839; round 5's penalty phase: GPU 8768 / trust-constr 6839); what IS checked is (1):
    #     the GPU's point is a local minimiser of the oracle's NLP that an independent solver confirms.
    res = minimize(onlp.eval_objective, z0, **kw)
    same = np.max(np.abs(res.x - z)) <= 1e-5 * np.max(np.abs(z))
    assert res.constr_violation < 1e-8 and np.isfinite(f_gpu) and np.isfinite(res.fun)
    print(f"[third party] acrobot T=25: GPU f = {f_gpu:.4f} in {it} iterations; trust-constr from the same guess f = {res.fun:.4f} "
          f"({'same point' if same else 'different local minimiser'}); polish moved {np.max(np.abs(pol.x - z)):.1e}")
