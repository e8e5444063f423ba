"""Parity of the HIP evaluator path (through the C-ABI) with the oracle.  Needs a real MI355X.

Tolerance (BASELINE.json north_star: values within 1e-8 relative): every entry must satisfy
|got - ref| <= 1e-8 * max(|ref|, 1e-3 * max|ref_vector|); the floor only shields entries that are
themselves the result of cancellation to (near) zero.  Structure/index parity is bit-exact and is
tested on CPU in test_layout.py.
"""
import numpy as np
import pytest

from conftest import load_golden, product_solver

pytestmark = pytest.mark.gpu

RTOL = 1e-8


def close(got, ref, rtol=RTOL):
    got, ref = np.asarray(got, float), np.asarray(ref, float)
    assert got.shape == ref.shape
    if ref.size == 0:
        return True
    floor = 1e-3 * np.max(np.abs(ref)) if np.max(np.abs(ref)) > 0 else 1.0
    err = np.abs(got - ref) / np.maximum(np.abs(ref), floor)
    assert np.max(err) <= rtol, f"max rel err {np.max(err):.3e} at {int(np.argmax(err))}"
    return True


CASES = ["pendulum_T6.json", "cartpole_T5.json", "acrobot_T5.json", "car_T6.json", "acrobot_bounds_T4.json",
         "acrobot_T70.json"]


@pytest.mark.parametrize("fixture", CASES)
def test_five_callbacks_against_golden(fixture):
    g = load_golden(fixture)
    s, _ = product_solver(g["model"], g["T"])
    n = s.nlp
    z, mu = np.array(g["z"]), np.array(g["mu"])
    assert abs(n.eval_objective(z) - g["objective"]) <= RTOL * max(1.0, abs(g["objective"]))
    grad = np.full(n.num_variables, np.nan)
    n.eval_objective_gradient(grad, z)
    close(grad, g["gradient"])
    c = np.full(n.num_constraint, np.nan)
    n.eval_constraint(c, z)
    close(c, g["constraint"])
    J = np.full(n.num_jacobian, np.nan)
    n.eval_constraint_jacobian(J, z)
    close(J, g["jacobian"])
    H = np.full(int(n.sizes.nnz_hess_key), np.nan)
    n.eval_hessian_lagrangian(H, z, g["sigma"], mu)
    close(H, g["hessian_sigma"])
    n.eval_hessian_lagrangian(H, z, 1.0, mu)
    close(H, g["hessian_one"])


def test_reference_objective_test():
    """test/objective.jl:1-38 through the product path."""
    s, _ = product_solver("ref_objective", 3, evaluate_hessian=False)
    z = np.ones(s.nlp.num_variables)
    assert abs(s.nlp.eval_objective(z) - 24.2) < 1e-8
    g = np.zeros(s.nlp.num_variables)
    s.nlp.eval_objective_gradient(g, z)
    assert np.linalg.norm(g - np.array([2, 2, 0.2, 2, 2, 0.2, 20, 20])) < 1e-8


def test_reference_dynamics_test():
    """test/dynamics.jl:37-59 through the product path (pendulum implicit Euler at ones)."""
    s, _ = product_solver("ref_dynamics", 3, evaluate_hessian=False)
    z = np.ones(s.nlp.num_variables)
    c = np.zeros(s.nlp.num_constraint)
    s.nlp.eval_constraint(c, z)
    assert np.linalg.norm(c - np.tile([-0.1, 0.7354830360965464], 2)) < 1e-8
    J = np.zeros(s.nlp.num_jacobian)
    s.nlp.eval_constraint_jacobian(J, z)
    dense = np.zeros((4, 8))
    for (r, cc), v in zip(s.nlp.jacobian_structure(), J):
        dense[r - 1, cc - 1] = v
    blk = np.array([[-1, 0, 0, 1, -0.1], [0, -1, -0.1, 0.5300365620566452, 1.01]])
    exp = np.zeros((4, 8))
    exp[0:2, 0:5] = blk
    exp[2:4, 3:8] = blk
    assert np.linalg.norm(dense - exp) < 1e-8


def test_reference_constraints_test():
    """test/constraints.jl:1-45 through the product path."""
    T = 5
    s, _ = product_solver("ref_constraints", 5, evaluate_hessian=False)
    rng = np.random.default_rng(3)
    z = rng.random(s.nlp.num_variables)
    c = np.zeros(s.nlp.num_constraint)
    s.nlp.eval_constraint(c, z)
    nd = int(s.nlp.sizes.num_constraint_dynamics)
    xs = [z[np.array(i) - 1] for i in s.nlp.indices.states]
    exp = np.concatenate([np.concatenate([-1 - xs[t], xs[t] - 1]) for t in range(T - 1)] + [xs[T - 1]])
    assert np.linalg.norm(c[nd:] - exp) < 1e-8
    J = np.zeros(s.nlp.num_jacobian)
    s.nlp.eval_constraint_jacobian(J, z)
    dense = np.zeros((s.nlp.num_constraint, s.nlp.num_variables))
    for (r, cc), v in zip(s.nlp.jacobian_structure(), J):
        dense[r - 1, cc - 1] = v
    dct = np.vstack([np.hstack([-np.eye(2), np.zeros((2, 1))]), np.hstack([np.eye(2), np.zeros((2, 1))])])
    expJ = np.zeros((4 * (T - 1) + 2, 3 * (T - 1) + 2))
    for t in range(T - 1):
        expJ[4 * t:4 * t + 4, 3 * t:3 * t + 3] = dct
    expJ[-2:, -2:] = np.eye(2)
    assert np.linalg.norm(dense[nd:] - expJ) < 1e-8


def test_reference_hessian_lagrangian_test():
    """test/hessian_lagrangian.jl:97-205 through the product path: acrobot midpoint, T = 3, nonlinear stage
    constraints with tan/cos/pow; compared with the oracle built from the same formulas in sympy."""
    import sympy as sp
    from oracle import dto_oracle as O, sympy_models as S
    T, n, m = 3, 4, 1
    s, _ = product_solver("ref_hesslag", 3, evaluate_hessian=True)
    odt = S.Dynamics(S.acrobot_midpoint, n, n, m, evaluate_hessian=True)
    oot = S.Cost(lambda x, u, w: S.fl(0.1) * S.dot(x[2:4], x[2:4]) + S.fl(0.1) * S.dot(u, u), n, m, evaluate_hessian=True)
    ooT = S.Cost(lambda x, u, w: S.fl(0.1) * S.dot(x[2:4], x[2:4]), n, 0, evaluate_hessian=True)
    octf = lambda x, u, w: ([-S.fl(5.0) - sp.cos(u[0]) * sum(xi ** 2 for xi in x)] + [sp.cos(xi) * sp.tan(u[0]) - S.fl(5.0) for xi in x])
    ocont = S.Constraint(octf, n, m, indices_inequality=list(range(1, m + n + 1)), evaluate_hessian=True)
    oconT = S.Constraint(lambda x, u, w: [sp.sin(xi ** 3) for xi in x], n, 0, evaluate_hessian=True)
    onlp = O.NLPData([odt] * 2, [oot, oot, ooT], [ocont, ocont, oconT], [S.Bound(n, m)] * 2 + [S.Bound(n, 0)], evaluate_hessian=True)
    assert s.nlp.hessian_lagrangian_structure() == onlp.hessian_lagrangian_structure()
    assert s.nlp.jacobian_structure() == onlp.jacobian_structure()
    rng = np.random.default_rng(11)
    z, mu = rng.random(s.nlp.num_variables), rng.random(s.nlp.num_constraint)
    H = np.zeros(len(onlp.hessian_lagrangian_structure()))
    s.nlp.eval_hessian_lagrangian(H, z, 1.0, mu)
    close(H, onlp.eval_hessian_lagrangian(z, 1.0, mu, hp=True))
    J = np.zeros(s.nlp.num_jacobian)
    s.nlp.eval_constraint_jacobian(J, z)
    close(J, onlp.eval_constraint_jacobian(z, hp=True))
    c = np.zeros(s.nlp.num_constraint)
    s.nlp.eval_constraint(c, z)
    close(c, onlp.eval_constraint(z, hp=True))


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("model,T,B", [("acrobot", 70, 5), ("cartpole", 200, 3), ("acrobot", 1000, 2), ("car", 500, 3),
                                       ("pendulum", 50, 4)])
def test_batched_device_path_against_oracle(model, T, B):
    """B instances resident in HBM (instance-major, padded leading dimensions) vs the float64 oracle."""
    torch = _torch()
    from oracle import dto_oracle as O, sympy_models as S
    s, _ = product_solver(model, T)
    n = s.nlp
    p = S.build(model, T, evaluate_hessian=True)
    onlp = O.NLPData(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True)
    rng = np.random.default_rng(100 + T)
    nz, nc, nj, nh = n.num_variables, n.num_constraint, n.num_jacobian, int(n.sizes.nnz_hess_key)
    pad = 3
    Z = rng.random((B, nz + pad))
    MU = rng.random((B, nc + pad))
    dz = torch.tensor(Z, device="cuda")
    dmu = torch.tensor(MU, device="cuda")
    f = torch.full((B,), float("nan"), device="cuda", dtype=torch.float64)
    g = torch.full((B, nz + pad), float("nan"), device="cuda", dtype=torch.float64)
    c = torch.full((B, nc + pad), float("nan"), device="cuda", dtype=torch.float64)
    J = torch.full((B, nj + pad), float("nan"), device="cuda", dtype=torch.float64)
    H = torch.full((B, nh + pad), float("nan"), device="cuda", dtype=torch.float64)
    st = torch.cuda.current_stream().cuda_stream
    n.eval_objective_batch(dz.data_ptr(), B, nz + pad, f.data_ptr(), st)
    n.eval_objective_gradient_batch(dz.data_ptr(), B, nz + pad, g.data_ptr(), nz + pad, st)
    n.eval_constraint_batch(dz.data_ptr(), B, nz + pad, c.data_ptr(), nc + pad, st)
    n.eval_constraint_jacobian_batch(dz.data_ptr(), B, nz + pad, J.data_ptr(), nj + pad, st)
    n.eval_hessian_lagrangian_batch(dz.data_ptr(), B, nz + pad, 0.8, dmu.data_ptr(), nc + pad, H.data_ptr(), nh + pad, st)
    torch.cuda.synchronize()
    f, g, c, J, H = (t.cpu().numpy() for t in (f, g, c, J, H))
    # padding is never written
    for arr, k in ((g, nz), (c, nc), (J, nj), (H, nh)):
        assert np.all(np.isnan(arr[:, k:]))
    for b in range(B):
        z, mu = Z[b, :nz], MU[b, :nc]
        assert abs(f[b] - onlp.eval_objective(z)) <= RTOL * max(1.0, abs(f[b]))
        close(g[b, :nz], onlp.eval_objective_gradient(z))
        close(c[b, :nc], onlp.eval_constraint(z))
        close(J[b, :nj], onlp.eval_constraint_jacobian(z))
        close(H[b, :nh], onlp.eval_hessian_lagrangian(z, 0.8, mu))
    # the single-instance host-pointer callback gives bit-identical values to the batched path
    J1 = np.zeros(nj)
    n.eval_constraint_jacobian(J1, Z[1, :nz])
    assert np.array_equal(J1, J[1, :nj])
    H1 = np.zeros(nh)
    n.eval_hessian_lagrangian(H1, Z[1, :nz], 0.8, MU[1, :nc])
    assert np.array_equal(H1, H[1, :nh])


@pytest.mark.parametrize("model,T", [("acrobot", 1000), ("acrobot", 2000), ("car", 500), ("cartpole", 200)])
def test_full_size_properties(model, T):
    """Size-independent properties at the BASELINE sizes: derivative consistency (directional finite
    differences), Hessian symmetry, linearity of H in (sigma, mu)."""
    s, _ = product_solver(model, T)
    n = s.nlp
    rng = np.random.default_rng(5)
    nz, nc, nj, nh = n.num_variables, n.num_constraint, n.num_jacobian, int(n.sizes.nnz_hess_key)
    z = rng.random(nz)
    d = rng.standard_normal(nz)
    d /= np.linalg.norm(d)
    h = 1e-6
    g = np.zeros(nz)
    n.eval_objective_gradient(g, z)
    fd = (n.eval_objective(z + h * d) - n.eval_objective(z - h * d)) / (2 * h)
    assert abs(fd - g @ d) <= 1e-6 * max(1.0, abs(fd))
    J = np.zeros(nj)
    n.eval_constraint_jacobian(J, z)
    rows, cols = (np.array(v) - 1 for v in zip(*n.jacobian_structure()))
    Jd = np.zeros(nc)
    np.add.at(Jd, rows, J * d[cols])
    cp, cm = np.zeros(nc), np.zeros(nc)
    n.eval_constraint(cp, z + h * d)
    n.eval_constraint(cm, z - h * d)
    assert np.max(np.abs((cp - cm) / (2 * h) - Jd)) <= 1e-6
    mu = rng.random(nc)
    H = np.zeros(nh)
    n.eval_hessian_lagrangian(H, z, 1.3, mu)
    hr, hc = (np.array(v) - 1 for v in zip(*n.hessian_lagrangian_structure()))
    # symmetry: the key holds both triangles (src/data.jl:184)
    lut = {(r, c): v for r, c, v in zip(hr, hc, H)}
    assert all(abs(lut[(c, r)] - v) <= 1e-12 * max(1.0, abs(v)) for (r, c), v in lut.items())
    # H d == d/dz (grad f * sigma + J' mu) . d
    Hd = np.zeros(nz)
    np.add.at(Hd, hr, H * d[hc])
    def lag_grad(zz):
        gg, JJ = np.zeros(nz), np.zeros(nj)
        n.eval_objective_gradient(gg, zz)
        n.eval_constraint_jacobian(JJ, zz)
        out = 1.3 * gg
        np.add.at(out, cols, JJ * mu[rows])
        return out
    fdH = (lag_grad(z + h * d) - lag_grad(z - h * d)) / (2 * h)
    assert np.max(np.abs(fdH - Hd)) <= 1e-5 * max(1.0, np.max(np.abs(Hd)))
    # linearity: H(sigma, mu) = sigma H(1, 0) + H(0, mu)
    H10, H0m = np.zeros(nh), np.zeros(nh)
    n.eval_hessian_lagrangian(H10, z, 1.0, np.zeros(nc))
    n.eval_hessian_lagrangian(H0m, z, 0.0, mu)
    assert np.max(np.abs(H - (1.3 * H10 + H0m))) <= 1e-10 * max(1.0, np.max(np.abs(H)))


@pytest.mark.parametrize("model,T", [("acrobot", 1000), ("cartpole", 200), ("car", 500)])
def test_full_size_sampled_stages_against_oracle(model, T):
    """BASELINE sizes: the dynamics rows and Jacobian slots of sampled stages (first, last, and the stages around the 64-knot
    wavefront tiles and the 63-stage Hessian tiles) equal the oracle's per-stage functions evaluated at that stage's
    (x_t, u_t, x_{t+1}) -- ties the full-size outputs to the oracle without evaluating the whole oracle problem."""
    from oracle import sympy_models as S
    s, _ = product_solver(model, T)
    n = s.nlp
    op = S.build(model, 3, evaluate_hessian=True)
    od = op["dynamics"][0]
    rng = np.random.default_rng(17)
    z, mu = rng.random(n.num_variables), rng.random(n.num_constraint)
    c = np.zeros(n.num_constraint); n.eval_constraint(c, z)
    J = np.zeros(n.num_jacobian); n.eval_constraint_jacobian(J, z)
    idx = n.indices
    stages = sorted({0, 1, 62, 63, 64, 65, 125, 126, 127, 128, T // 2, T - 3, T - 2} & set(range(T - 1)))
    for t in stages:
        x = z[np.array(idx.states[t]) - 1]
        u = z[np.array(idx.actions[t]) - 1]
        y = z[np.array(idx.states[t + 1]) - 1]
        close(c[np.array(idx.dynamics_constraints[t]) - 1], od.evaluate(list(y), list(x), list(u), []))
        close(J[np.array(idx.dynamics_jacobians[t]) - 1], od.jacobian(list(y), list(x), list(u), []))


@pytest.mark.parametrize("seed", [3, 8])
def test_random_heterogeneous_problem_callbacks_match_oracle(seed):
    """Random stage models, stages of different kinds (constraints on some knots, an inequality row, different terminal
    objects): the five callbacks through the HIP kernels equal the oracle at a random point."""
    import dto_amd
    from oracle import dto_oracle as O
    from test_layout import random_heterogeneous_problem
    dyn, obj, cons, bnds = random_heterogeneous_problem(seed, "product")
    s = dto_amd.Solver(dyn, obj, cons, bnds, evaluate_hessian=True, name=f"random{seed}")
    onlp = O.NLPData(*random_heterogeneous_problem(seed, "oracle"), evaluate_hessian=True)
    n = s.nlp
    rng = np.random.default_rng(1000 + seed)
    z, mu = rng.random(n.num_variables), rng.random(n.num_constraint)
    assert abs(n.eval_objective(z) - onlp.eval_objective(z)) <= RTOL * max(1.0, abs(onlp.eval_objective(z)))
    g = np.full(n.num_variables, np.nan); n.eval_objective_gradient(g, z); close(g, onlp.eval_objective_gradient(z))
    c = np.full(n.num_constraint, np.nan); n.eval_constraint(c, z); close(c, onlp.eval_constraint(z))
    J = np.full(n.num_jacobian, np.nan); n.eval_constraint_jacobian(J, z); close(J, onlp.eval_constraint_jacobian(z))
    H = np.full(int(n.sizes.nnz_hess_key), np.nan); n.eval_hessian_lagrangian(H, z, 0.7, mu)
    close(H, onlp.eval_hessian_lagrangian(z, 0.7, mu))


def test_ragged_and_tiny_horizons():
    """T not a multiple of the wave width, T smaller than a wave, T = 2; plus the 63-stage Hessian tiling edge."""
    from oracle import dto_oracle as O, sympy_models as S
    for T in (2, 3, 63, 64, 65, 127, 129):
        s, _ = product_solver("pendulum", T)
        n = s.nlp
        p = S.build("pendulum", T, evaluate_hessian=True)
        onlp = O.NLPData(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True)
        rng = np.random.default_rng(T)
        z, mu = rng.random(n.num_variables), rng.random(n.num_constraint)
        J = np.zeros(n.num_jacobian)
        n.eval_constraint_jacobian(J, z)
        close(J, onlp.eval_constraint_jacobian(z))
        H = np.zeros(int(n.sizes.nnz_hess_key))
        n.eval_hessian_lagrangian(H, z, 0.5, mu)
        close(H, onlp.eval_hessian_lagrangian(z, 0.5, mu))
        c = np.zeros(n.num_constraint)
        n.eval_constraint(c, z)
        close(c, onlp.eval_constraint(z))
        g = np.zeros(n.num_variables)
        n.eval_objective_gradient(g, z)
        close(g, onlp.eval_objective_gradient(z))
        assert abs(n.eval_objective(z) - onlp.eval_objective(z)) < 1e-9
