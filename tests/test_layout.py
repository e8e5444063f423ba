"""Layout contract through the C-ABI, bit-exact against the oracle fixtures."""
import hashlib
import json

import numpy as np
import pytest

from conftest import load_golden, product_solver


def _model_T(fixture):
    g = load_golden(fixture)
    return g, g["model"], g["T"]


CASES = ["pendulum_T6.json", "cartpole_T5.json", "acrobot_T5.json", "car_T6.json", "acrobot_bounds_T4.json",
         "acrobot_T70.json"]


@pytest.mark.parametrize("fixture", CASES)
def test_sizes_and_structures_bit_exact(fixture):
    g, model, T = _model_T(fixture)
    s, _ = product_solver(model, T)
    n = s.nlp
    assert n.num_variables == g["num_variables"]
    assert n.num_constraint == g["num_constraint"]
    assert n.num_jacobian == g["num_jacobian"]
    assert n.num_hessian_lagrangian == g["num_hessian_lagrangian_raw"]      # duplicate-counting (src/data.jl:187)
    assert int(n.sizes.nnz_hess_key) == g["num_hessian_key"]
    assert [list(rc) for rc in n.jacobian_structure()] == g["jacobian_structure"]
    assert [list(rc) for rc in n.hessian_lagrangian_structure()] == g["hessian_structure"]


@pytest.mark.parametrize("fixture", CASES)
def test_index_vectors(fixture):
    g, model, T = _model_T(fixture)
    s, _ = product_solver(model, T)
    idx = s.nlp.indices
    assert idx.states == g["idx_states"]
    assert idx.actions == g["idx_actions"]
    assert idx.dynamics_hessians == g["idx_dynamics_hessians"]
    assert idx.objective_hessians == g["idx_objective_hessians"]
    assert idx.stage_hessians == g["idx_stage_hessians"]
    # test/hessian_lagrangian.jl:191-193: key[idx] == per-component sparsity
    key = s.nlp.hessian_lagrangian_structure()
    d = s.nlp.structure.dyn[0]
    nxu = g["idx_states"][1][0] - 1
    for t, ix in enumerate(idx.dynamics_hessians):
        got = [key[i - 1] for i in ix]
        exp = [(r + t * nxu, c + t * nxu) for r, c in zip(*d.hessian_sparsity)]
        assert got == exp


@pytest.mark.parametrize("fixture", CASES)
def test_bounds(fixture):
    g, model, T = _model_T(fixture)
    s, _ = product_solver(model, T)
    lo, hi = s.nlp.variable_bounds
    exp_lo = np.array([-np.inf if v is None else v for v in g["variable_lower"]], dtype=float)
    exp_hi = np.array([np.inf if v == "inf" else v for v in g["variable_upper"]], dtype=float)
    assert np.array_equal(lo, exp_lo) and np.array_equal(hi, exp_hi)
    clo, chi = s.nlp.constraint_bounds
    assert [bool(np.isneginf(v)) for v in clo] == g["constraint_lower_is_minus_inf"]
    assert np.all(chi == 0.0) and np.all((clo == 0.0) | np.isneginf(clo))


def test_full_size_structures_by_digest():
    """BASELINE sizes: totals of SURVEY.md Appendix C and a SHA-256 of the full (row, col) lists."""
    def digest(pairs):
        return hashlib.sha256(np.asarray(pairs, dtype=np.int64).tobytes()).hexdigest()
    for g in load_golden("full_size_structure.json"):
        s, _ = product_solver(g["model"], g["T"])
        n = s.nlp
        assert (n.num_variables, n.num_constraint, n.num_jacobian) == (g["num_variables"], g["num_constraint"], g["num_jacobian"])
        assert (n.num_hessian_lagrangian, int(n.sizes.nnz_hess_key)) == (g["num_hessian_lagrangian_raw"], g["num_hessian_key"])
        assert digest(n.jacobian_structure()) == g["jacobian_structure_sha256"]
        assert digest(n.hessian_lagrangian_structure()) == g["hessian_structure_sha256"]


def test_appendix_c_headline_numbers():
    s, _ = product_solver("acrobot", 1000)
    n = s.nlp
    assert (n.num_variables, n.num_constraint, n.num_jacobian) == (4999, 4004, 25982)
    assert (n.num_hessian_lagrangian, int(n.sizes.nnz_hess_key)) == (54947, 40971)


def test_features_and_trajectory_roundtrip():
    """src/moi.jl:122 and test/dynamics.jl:62-81 (state/action indices tile z exactly once)."""
    s, _ = product_solver("pendulum", 6)
    assert s.nlp.features_available() == ["Grad", "Jac", "Hess"]
    s2, _ = product_solver("car", 6, evaluate_hessian=False)
    assert s2.nlp.features_available() == ["Grad", "Jac"]
    idx = s.nlp.indices
    seen = sorted(i for v in idx.states + idx.actions for i in v)
    assert seen == list(range(1, s.nlp.num_variables + 1))
    for t, xu in enumerate(idx.state_action):
        assert xu == idx.states[t] + (idx.actions[t] if t < len(idx.actions) else [])


def test_invalid_specs_are_rejected():
    import dto_amd
    from dto_amd import problems as P
    p = P.build_pendulum(T=4)
    with pytest.raises(ValueError):
        dto_amd.Solver(p["dynamics"], p["objective"][:-1], p["constraints"], p["bounds"])
    # mixed evaluate_hessian flags are refused up front instead of failing at call time
    q = P.build_pendulum(T=4, evaluate_hessian=False)
    with pytest.raises(ValueError):
        dto_amd.Solver(q["dynamics"], q["objective"], q["constraints"], q["bounds"], evaluate_hessian=True)


def random_heterogeneous_problem(seed, lib):
    """(dynamics, objective, constraints, bounds) of a random 5-knot problem for lib in {"product", "oracle"}."""
    import dto_amd
    from oracle import sympy_models as S
    from test_symbolic import _random_model
    (n, m), mk_dyn, mk_cost, mk_con = _random_model(seed)
    T = 5
    mod = dto_amd if lib == "product" else S
    d = mod.Dynamics(mk_dyn(lib), n, n, m, evaluate_hessian=True)
    c = mod.Cost(mk_cost(lib), n, m, evaluate_hessian=True)
    if lib == "product":
        cT = mod.Cost(lambda x, u, w: dto_amd.dot(x, x), n, 0, evaluate_hessian=True)
        kT = mod.Constraint(lambda x, u, w: x[0:1] * x[1:2], n, 0, evaluate_hessian=True)
    else:
        cT = mod.Cost(lambda x, u, w: S.dot(x, x), n, 0, evaluate_hessian=True)
        kT = mod.Constraint(lambda x, u, w: [x[0] * x[1]], n, 0, evaluate_hessian=True)
    k_eq = mod.Constraint(mk_con(lib), n, m, evaluate_hessian=True)
    k_in = mod.Constraint(mk_con(lib), n, m, indices_inequality=[2], evaluate_hessian=True)
    cons = [k_eq, mod.Constraint(), k_in, mod.Constraint(), kT]
    bnds = [mod.Bound(n, m, action_lower=[-1.0] * m, action_upper=[1.0] * m)] * (T - 1) + [mod.Bound(n, 0)]
    return [d] * (T - 1), [c] * (T - 1) + [cT], cons, bnds


@pytest.mark.parametrize("seed", [3, 8])
def test_random_heterogeneous_problem_layout_matches_oracle(seed):
    """A random problem whose stages differ (stage constraints on some knots only, one with an inequality row, different
    terminal objects): global Jacobian / Hessian structures, totals and index vectors of the product (C++ layout through
    the C-ABI) equal the oracle's restatement of src/data.jl:61-220 bit for bit."""
    import dto_amd
    from oracle import dto_oracle as O
    build = lambda lib: random_heterogeneous_problem(seed, lib)
    dyn, obj, cons, bnds = build("product")
    s = dto_amd.Solver(dyn, obj, cons, bnds, evaluate_hessian=True, name=f"random{seed}")
    odyn, oobj, ocons, obnds = build("oracle")
    onlp = O.NLPData(odyn, oobj, ocons, obnds, evaluate_hessian=True)
    nl = s.nlp
    assert (nl.num_variables, nl.num_constraint, nl.num_jacobian) == (onlp.num_variables, onlp.num_constraint, onlp.num_jacobian)
    assert nl.num_hessian_lagrangian == onlp.num_hessian_lagrangian
    assert nl.jacobian_structure() == onlp.jacobian_structure()
    assert nl.hessian_lagrangian_structure() == onlp.hessian_lagrangian_structure()
    clo, chi = nl.constraint_bounds
    oclo, ochi = onlp.constraint_bounds
    assert np.array_equal(np.isneginf(clo), np.isneginf(np.asarray(oclo, float))) and np.all(chi == 0.0)


def varying_dimension_problem(lib):
    """(dynamics, objective, constraints, bounds) of a 6-knot problem whose state and action dimensions change along the
    horizon (dimensions(), src/dynamics.jl:206-211): n = [2, 2, 3, 3, 2, 2], m = [1, 1, 2, 1, 1], for lib in {"product", "oracle"}."""
    import dto_amd
    from oracle import sympy_models as S
    mod = dto_amd if lib == "product" else S
    sin = dto_amd.sin if lib == "product" else __import__("sympy").sin
    vec = (lambda v: np.array(v, dtype=object)) if lib == "product" else (lambda v: v)
    nx, nu = [2, 2, 3, 3, 2, 2], [1, 1, 2, 1, 1]
    T = len(nx)

    def make_dyn(ny, n, m):
        def f(y, x, u, w):
            out = []
            for i in range(ny):
                e = y[i] - 0.9 * x[i % n] - 0.1 * sin(x[(i + 1) % n]) * u[i % m] - 0.05 * y[i] * x[0]
                if i == ny - 1 and m > 1:
                    e = e - 0.2 * u[1]
                out.append(e)
            return vec(out)
        return f

    def make_cost(n, m):
        def f(x, u, w):
            e = 0.0
            for i in range(n):
                e = e + 0.5 * x[i] * x[i] + 0.1 * sin(x[i]) * x[(i + 1) % n]
            for j in range(m):
                e = e + 0.05 * u[j] * u[j]
            return e
        return f

    dyn = [mod.Dynamics(make_dyn(nx[t + 1], nx[t], nu[t]), nx[t + 1], nx[t], nu[t], evaluate_hessian=True) for t in range(T - 1)]
    obj = [mod.Cost(make_cost(nx[t], nu[t]), nx[t], nu[t], evaluate_hessian=True) for t in range(T - 1)]
    obj.append(mod.Cost(make_cost(nx[-1], 1), nx[-1], 0, evaluate_hessian=True) if False else
               mod.Cost((lambda x, u, w: 2.0 * x[0] * x[0] + 2.0 * x[1] * x[1]), nx[-1], 0, evaluate_hessian=True))
    first = mod.Constraint((lambda x, u, w: vec([x[0] - 0.3, x[1] + 0.2])), nx[0], nu[0], evaluate_hessian=True)
    mid = mod.Constraint((lambda x, u, w: vec([x[0] * x[2] - 0.1])), nx[2], nu[2], indices_inequality=[1], evaluate_hessian=True)
    cons = [first, mod.Constraint(), mid, mod.Constraint(), mod.Constraint(), mod.Constraint()]
    bnds = [mod.Bound(nx[t], nu[t], action_lower=[-2.0] * nu[t], action_upper=[2.0] * nu[t]) for t in range(T - 1)] + [mod.Bound(nx[-1], 0)]
    return dyn, obj, cons, bnds


def test_varying_dimensions_layout_matches_oracle():
    """Per-stage state / action dimensions: structures, totals and index vectors equal the oracle's restatement of
    src/data.jl:61-220 and src/dynamics.jl:188-211 bit for bit."""
    import dto_amd
    from oracle import dto_oracle as O
    dyn, obj, cons, bnds = varying_dimension_problem("product")
    s = dto_amd.Solver(dyn, obj, cons, bnds, evaluate_hessian=True, name="varydims")
    onlp = O.NLPData(*varying_dimension_problem("oracle"), evaluate_hessian=True)
    nl = s.nlp
    assert (nl.num_variables, nl.num_constraint, nl.num_jacobian) == (onlp.num_variables, onlp.num_constraint, onlp.num_jacobian) == (20, 15, nl.num_jacobian)
    assert nl.jacobian_structure() == onlp.jacobian_structure()
    assert nl.hessian_lagrangian_structure() == onlp.hessian_lagrangian_structure()
    assert nl.state_dimensions == [2, 2, 3, 3, 2, 2] and nl.action_dimensions == [1, 1, 2, 1, 1, 0]
    assert [len(i) for i in nl.indices.states] == [2, 2, 3, 3, 2, 2]
