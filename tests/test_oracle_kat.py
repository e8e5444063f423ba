"""Pin the ORACLE against the known answers the reference's own tests hold."""
import numpy as np

from oracle import dto_oracle as O
from oracle import sympy_models as S


def test_objective_kat():
    """test/objective.jl:1-38: T=3, n=2, m=1, ot = x'x + 0.1 u'u, oT = 10 x'x at ones."""
    T, n, m = 3, 2, 1
    ct = S.Cost(lambda x, u, w: S.dot(x, x) + S.fl(0.1) * S.dot(u, u), n, m)
    cT = S.Cost(lambda x, u, w: S.fl(10.0) * S.dot(x, x), n, 0)
    assert abs(ct.evaluate([1, 1], [1], [])[0] - 2.1) < 1e-12
    assert np.allclose(ct.gradient([1, 1], [1], []), [2, 2, 0.2], atol=1e-8)
    assert np.allclose(cT.gradient([1, 1], [], []), [20, 20], atol=1e-8)
    dyn = S.Dynamics(S.double_integrator, n, n, m)
    nlp = O.NLPData([dyn] * (T - 1), [ct] * (T - 1) + [cT], [S.Constraint() for _ in range(T)],
                    [S.Bound(n, m)] * (T - 1) + [S.Bound(n, 0)])
    z = np.ones(nlp.num_variables)
    assert abs(nlp.eval_objective(z) - 24.2) < 1e-12
    assert np.allclose(nlp.eval_objective_gradient(z), [2, 2, 0.2, 2, 2, 0.2, 20, 20], atol=1e-8)
    assert nlp.idx_state_action == [[1, 2, 3], [4, 5, 6], [7, 8]]          # idx_xu, test/objective.jl:14


def test_dynamics_kat():
    """test/dynamics.jl:1-83: pendulum implicit Euler at ones; block placement of the global Jacobian."""
    T, n, m = 3, 2, 1
    dt = S.Dynamics(S.euler_implicit_test, n, n, m)
    assert np.allclose(dt.evaluate([1, 1], [1, 1], [1], []), [-0.1, 0.7354830360965464], atol=1e-12)
    assert list(zip(*dt.jacobian_sparsity)) == [(1, 1), (2, 2), (2, 3), (1, 4), (2, 4), (1, 5), (2, 5)]
    jl = dt.jacobian([1, 1], [1, 1], [1], [])
    assert np.allclose(jl, [-1, -1, -0.1, 1, 0.5300365620566452, -0.1, 1.01], atol=1e-12)
    ct = S.Cost(lambda x, u, w: S.dot(x, x), n, m)
    cT = S.Cost(lambda x, u, w: S.dot(x, x), n, 0)
    nlp = O.NLPData([dt] * (T - 1), [ct] * (T - 1) + [cT], [S.Constraint() for _ in range(T)],
                    [S.Bound(n, m)] * (T - 1) + [S.Bound(n, 0)])
    z = np.ones(nlp.num_variables)
    J = np.zeros((nlp.num_constraint, nlp.num_variables))
    for (r, c), v in zip(nlp.jacobian_structure(), nlp.eval_constraint_jacobian(z)):
        J[r - 1, c - 1] = v
    blk = np.zeros((2, 5))
    for (r, c), v in zip(zip(*dt.jacobian_sparsity), jl):
        blk[r - 1, c - 1] = v
    exp = np.zeros((4, 8))
    exp[0:2, 0:5] = blk
    exp[2:4, 3:8] = blk                      # test/dynamics.jl:52-59
    assert np.allclose(J, exp, atol=1e-12)
    xs, us = nlp.trajectory(np.arange(1.0, 9.0))
    assert [list(x) for x in xs] == [[1, 2], [4, 5], [7, 8]] and [list(u) for u in us[:2]] == [[3], [6]]


def test_constraints_kat():
    """test/constraints.jl:1-45: ct = [-1 - x; x - 1] (all rows inequality), cT = x; Jacobian = blockdiag."""
    T, n, m = 5, 2, 1
    cont = S.Constraint(lambda x, u, w: [-S.fl(1.0) - x[0], -S.fl(1.0) - x[1], x[0] - S.fl(1.0), x[1] - S.fl(1.0)],
                        n, m, indices_inequality=[1, 2, 3, 4])
    conT = S.Constraint(lambda x, u, w: [x[0], x[1]], n, 0)
    dyn = S.Dynamics(S.double_integrator, n, n, m)
    ct = S.Cost(lambda x, u, w: S.dot(x, x), n, m)
    cT = S.Cost(lambda x, u, w: S.dot(x, x), n, 0)
    nlp = O.NLPData([dyn] * (T - 1), [ct] * (T - 1) + [cT], [cont] * (T - 1) + [conT],
                    [S.Bound(n, m)] * (T - 1) + [S.Bound(n, 0)])
    rng = np.random.default_rng(0)
    z = rng.random(nlp.num_variables)
    c = nlp.eval_constraint(z)[nlp.num_dynamics:]
    xs, _ = nlp.trajectory(z)
    exp = np.concatenate([np.concatenate([-1 - xs[t], xs[t] - 1]) for t in range(T - 1)] + [xs[T - 1]])
    assert np.allclose(c, exp, atol=1e-12)
    J = np.zeros((nlp.num_constraint, nlp.num_variables))
    for (r, cc), v in zip(nlp.jacobian_structure(), nlp.eval_constraint_jacobian(z)):
        J[r - 1, cc - 1] = v
    Js = J[nlp.num_dynamics:]
    dct = np.vstack([np.hstack([-np.eye(2), np.zeros((2, 1))]), np.hstack([np.eye(2), np.zeros((2, 1))])])
    exp = np.zeros((4 * (T - 1) + 2, 3 * (T - 1) + 2))
    for t in range(T - 1):
        exp[4 * t:4 * t + 4, 3 * t:3 * t + 3] = dct
    exp[-2:, -2:] = np.eye(2)
    assert np.allclose(Js, exp, atol=1e-12)
    lo, _ = nlp.constraint_bounds
    assert np.all(np.isneginf(lo[nlp.num_dynamics:nlp.num_dynamics + 16])) and np.all(lo[-2:] == 0)


def test_hessian_lagrangian_kat():
    """test/hessian_lagrangian.jl:97-205: acrobot midpoint, T=3, nonlinear stage constraints; the evaluator's
    output scattered by the key equals the dense Hessian of the hand-written Lagrangian (both triangles),
    and its ordering is the row-major sorted key."""
    import sympy as sp
    T, n, m = 3, 4, 1
    dt = S.Dynamics(S.acrobot_midpoint, n, n, m, evaluate_hessian=True)
    ot = lambda x, u, w: S.fl(0.1) * S.dot(x[2:4], x[2:4]) + S.fl(0.1) * S.dot(u, u)
    oT = lambda x, u, w: S.fl(0.1) * S.dot(x[2:4], x[2:4])
    ctf = lambda x, u, w: ([-S.fl(5.0) - sp.cos(u[0]) * sum(xi ** 2 for xi in x)]
                           + [sp.cos(xi) * sp.tan(u[0]) - S.fl(5.0) for xi in x])
    cTf = lambda x, u, w: [sp.sin(xi ** 3) for xi in x]
    objt, objT = S.Cost(ot, n, m, evaluate_hessian=True), S.Cost(oT, n, 0, evaluate_hessian=True)
    cont = S.Constraint(ctf, n, m, indices_inequality=list(range(1, m + n + 1)), evaluate_hessian=True)
    conT = S.Constraint(cTf, n, 0, evaluate_hessian=True)
    nlp = O.NLPData([dt] * 2, [objt, objt, objT], [cont, cont, conT], [S.Bound(n, m)] * 2 + [S.Bound(n, 0)],
                    evaluate_hessian=True)
    np_, nd = 14, 4 + 4 + 5 + 5 + 4
    assert nlp.num_variables == np_ and nlp.num_constraint == nd
    zs = S.syms("z", np_ + nd)
    x1, u1, x2, u2, x3 = zs[0:4], zs[4:5], zs[5:9], zs[9:10], zs[10:14]
    l1, l2 = zs[14:18], zs[18:22]
    s1, s2, s3 = zs[22:27], zs[27:32], zs[32:36]
    L = (ot(x1, u1, []) + ot(x2, u2, []) + oT(x3, [], [])
         + S.dot(l1, S.acrobot_midpoint(x2, x1, u1, [])) + S.dot(l2, S.acrobot_midpoint(x3, x2, u2, []))
         + S.dot(s1, ctf(x1, u1, [])) + S.dot(s2, ctf(x2, u2, [])) + S.dot(s3, cTf(x3, [], [])))
    rng = np.random.default_rng(7)
    z0 = rng.random(np_ + nd)
    key = nlp.hessian_lagrangian_structure()
    assert key == sorted(key)
    h0 = nlp.eval_hessian_lagrangian(z0[:np_], 1.0, z0[np_:])
    full = np.zeros((np_, np_))
    for (r, c), v in zip(key, h0):
        full[r - 1, c - 1] = v
    grad = [sp.diff(L, v) for v in zs[:np_]]
    subs = dict(zip(zs, z0))
    dense = np.zeros((np_, np_))
    for i in range(np_):
        row = sp.lambdify(zs, [sp.diff(grad[i], v) for v in zs[i:np_]], modules="math")(*z0)
        dense[i, i:] = row
        dense[i:, i] = row
    assert np.linalg.norm(full - dense) < 1e-8                   # test/hessian_lagrangian.jl:200
    # every structurally present slot of the key is where the dense Hessian is nonzero or a structural zero
    assert np.count_nonzero(dense) <= len(key)
