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


def _random_model(seed):
    """A random smooth stage model as a recipe that both front ends can trace: returns (dims, make_dyn, make_cost, make_con)
    where make_*(lib) builds the closure for `lib` in {"product", "oracle"}."""
    rng = np.random.default_rng(seed)
    n, m = int(rng.integers(2, 4)), int(rng.integers(1, 3))
    nv = 2 * n + m

    def pick(k):
        return [int(i) for i in rng.choice(nv, size=k, replace=False)]

    rows = []
    for i in range(n):
        terms = []
        for _ in range(int(rng.integers(1, 4))):
            kind = int(rng.integers(0, 4))
            coef = float(np.round(rng.uniform(-2, 2), 3))
            vs = pick(2)
            terms.append((kind, coef, vs))
        rows.append(terms)
    cost_terms = [(int(rng.integers(0, 3)), float(np.round(rng.uniform(0.1, 2), 3)), [int(rng.integers(0, n + m)), int(rng.integers(0, n + m))])
                  for _ in range(3)]
    con_terms = [(int(rng.integers(0, 3)), float(np.round(rng.uniform(-1, 1), 3)), [int(rng.integers(0, n + m)), int(rng.integers(0, n + m))])
                 for _ in range(2)]

    def funcs(lib):
        if lib == "product":
            return dto_amd.sin, dto_amd.cos
        import sympy as sp
        return sp.sin, sp.cos

    def term(kind, coef, a, b, sin, cos):
        if kind == 0:
            return coef * a * b
        if kind == 1:
            return coef * sin(a) * b
        if kind == 2:
            return coef * cos(a + b)
        return coef * a - coef * a + coef * b      # exact cancellation: must not leave a structural entry for `a`

    def make_dyn(lib):
        sin, cos = funcs(lib)

        def f(y, x, u, w):
            v = list(x) + list(u) + list(y)
            out = []
            for i, terms in enumerate(rows):
                e = y[i] - x[i]
                for kind, coef, (ia, ib) in terms:
                    e = e - 0.05 * term(kind, coef, v[ia], v[ib], sin, cos)
                out.append(e)
            return np.array(out, dtype=object) if lib == "product" else out
        return f

    def make_cost(lib):
        sin, cos = funcs(lib)

        def f(x, u, w):
            v = list(x) + list(u)
            e = 0.0
            for kind, coef, (ia, ib) in cost_terms:
                e = e + term(kind, coef, v[ia], v[ib], sin, cos) + coef * v[ia] * v[ia]
            return e
        return f

    def make_con(lib):
        sin, cos = funcs(lib)

        def f(x, u, w):
            v = list(x) + list(u)
            out = [term(kind, coef, v[ia], v[ib], sin, cos) + v[0] for kind, coef, (ia, ib) in con_terms]
            return np.array(out, dtype=object) if lib == "product" else out
        return f

    return (n, m), make_dyn, make_cost, make_con


@pytest.mark.parametrize("seed", range(12))
def test_random_models_patterns_and_values_match_oracle(seed):
    """Random smooth stage models (products, sin, cos of sums, exact cancellations): local Jacobian / Hessian patterns of
    the product's front end equal the oracle's (sympy) bit for bit, and the nonzero values agree at a random point."""
    from oracle import sympy_models as S
    (n, m), mk_dyn, mk_cost, mk_con = _random_model(seed)
    pd = dto_amd.Dynamics(mk_dyn("product"), n, n, m, evaluate_hessian=True)
    od = S.Dynamics(mk_dyn("oracle"), n, n, m, evaluate_hessian=True)
    pc = dto_amd.Cost(mk_cost("product"), n, m, evaluate_hessian=True)
    oc = S.Cost(mk_cost("oracle"), n, m, evaluate_hessian=True)
    pk = dto_amd.Constraint(mk_con("product"), n, m, evaluate_hessian=True)
    ok = S.Constraint(mk_con("oracle"), n, m, evaluate_hessian=True)
    assert pd.jacobian_sparsity == od.jacobian_sparsity and pd.hessian_sparsity == od.hessian_sparsity
    assert pc.sparsity == oc.sparsity
    assert pk.jacobian_sparsity == ok.jacobian_sparsity and pk.hessian_sparsity == ok.hessian_sparsity
    rng = np.random.default_rng(100 + seed)
    x, u, y, lam = rng.random(n), rng.random(m), rng.random(n), rng.random(n)
    env = {}
    for nm, vec in (("x", x), ("u", u), ("y", y), ("lam", lam)):
        for i, val in enumerate(vec):
            env[(nm, i)] = float(val)
    assert np.allclose(evaluate(pd.jacobian_expr, env), od.jacobian(list(y), list(x), list(u), []), rtol=1e-12, atol=1e-14)
    assert np.allclose(evaluate(pd.hessian_expr, env), od.hessian(list(y), list(x), list(u), [], list(lam)), rtol=1e-12, atol=1e-14)
    assert np.allclose(evaluate(pc.hessian_expr, env), oc.hessian(list(x), list(u), []), rtol=1e-12, atol=1e-14)
    lamc = rng.random(2)
    for i, val in enumerate(lamc):
        env[("lam", i)] = float(val)
    assert np.allclose(evaluate(pk.hessian_expr, env), ok.hessian(list(x), list(u), [], list(lamc)), rtol=1e-12, atol=1e-14)


def test_plugin_cache_key_does_not_depend_on_tracing_history():
    """The generated text names temporaries after DAG node ids, which depend on what the process traced before; the
    cache key must not, or prebuilt plugins would miss in every process with another history (plugin.py:build_plugin).
    Two fresh processes: one traces the acrobot first, the other a cartpole and a car before it."""
    import subprocess, sys, json, os
    prog = r"""
import sys, json, hashlib
sys.path.insert(0, %r)
from dto_amd import problems as P
from dto_amd.plugin import Structure, build_plugin, generate_source
for b in sys.argv[1:]:
    q = getattr(P, "build_" + b)(T=4, evaluate_hessian=True)
    generate_source(Structure(q["dynamics"], q["objective"], q["constraints"], None, True), b)
p = P.build_acrobot(T=5, evaluate_hessian=True)
st = Structure(p["dynamics"], p["objective"], p["constraints"], None, True)
print(json.dumps([build_plugin(st, "acrobot"), hashlib.sha256(generate_source(st, "acrobot").encode()).hexdigest()]))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = [json.loads(subprocess.run([sys.executable, "-c", prog] + a, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
           for a in ([], ["cartpole", "car"])]
    assert res[0][0] == res[1][0]                  # same plugin
    assert res[0][1] != res[1][1]                  # although the text differs -- which is what the key must ignore


def test_ifelse_min_max_abs_trace_differentiate_and_emit():
    """The reference imports IfElse (src/DirectTrajectoryOptimization.jl:5): models may branch on symbolic values with
    ifelse / min / max / abs.  Values and first derivatives against a plain numpy restatement (central differences),
    patterns by the occurrence / linearity rules, and the emitted code is a select (no branch)."""
    import dto_amd
    from dto_amd.symbolic import codegen as CG, diff as D, expr as E
    from _dag_eval import evaluate
    x = E.variables("x", 3)
    f = (dto_amd.ifelse(x[0] > 0.5, x[0] ** 2 * x[1], np.sin(x[2])) + dto_amd.maximum(x[1], 0.0) ** 2 + abs(x[2])
         + dto_amd.minimum(x[0], x[1] * x[2]))

    def num(p):
        x0, x1, x2 = p
        return (x0 ** 2 * x1 if x0 > 0.5 else np.sin(x2)) + max(x1, 0.0) ** 2 + abs(x2) + min(x0, x1 * x2)

    for pt in ([0.7, 0.3, -0.2], [0.1, -0.4, 0.9], [0.9, 2.0, 1.5]):
        env = {("x", i): pt[i] for i in range(3)}
        assert abs(evaluate([f], env)[0] - num(pt)) < 1e-14
        g = evaluate(D.gradient(f, list(x)), env)
        for i in range(3):
            e = np.zeros(3); e[i] = 1e-6
            fd = (num(np.array(pt) + e) - num(np.array(pt) - e)) / 2e-6
            assert abs(g[i] - fd) < 1e-8, (pt, i, g[i], fd)
    assert D.jacobian_sparsity([f], list(x)) == [(0, 0), (0, 1), (0, 2)]
    # x0^2 x1 | sin x2 | max(x1,0)^2 | |x2| | x1 x2: no (0,2) pair, the conditions contribute nothing
    assert D.hessian_sparsity(f, list(x)) == [(0, 0), (1, 0), (0, 1), (1, 1), (2, 1), (1, 2), (2, 2)]
    body = CG.emit_body(D.gradient(f, list(x)), "g", {"x": "x"})
    assert " ? " in body and "if" not in body
    # constant conditions fold at construction, identical branches collapse
    assert dto_amd.ifelse(E.const(1.0) < E.const(2.0), x[0], x[1]) is x[0]
    assert dto_amd.ifelse(x[0] < x[1], x[2], x[2]) is x[2]
    with pytest.raises(TypeError):
        bool(x[0] < x[1])
    # a traced model closure using them builds (patterns by occurrence)
    d = dto_amd.Dynamics(lambda y, xx, u, w: y - xx - 0.1 * dto_amd.maximum(u, -1.0) * dto_amd.ifelse(xx[0] < 0.0, 1.0, 0.5),
                         1, 1, 1)
    assert d.num_jacobian == 3


def test_parameters_and_bounds_are_validated():
    """a short / long / missing stage parameter vector used to shift every later stage's w_t (or read past the
    end of the buffer) without an error; now the constructor checks it (src/solver.jl:10, src/data.jl:218)."""
    import dto_amd
    from dto_amd import problems as P
    p = P.build_param_pendulum(8)
    pars = [np.asarray(w, dtype=float) for w in p["parameters"]]
    with pytest.raises(ValueError, match="one vector per stage"):
        dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True,
                       parameters=pars[:5], name="param_pendulum")
    short = list(pars); short[3] = short[3][:-1]
    with pytest.raises(ValueError, match=r"parameters\[3\]"):
        dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True,
                       parameters=short, name="param_pendulum")
    with pytest.raises(ValueError, match="one Bound per stage"):
        dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"][:-1], evaluate_hessian=True,
                       parameters=pars, name="param_pendulum")
    # extra trailing entries of a stage vector are never read by the closures: dropped, not shifted into the next stage
    longer = [np.concatenate([w, [123.0]]) for w in pars]
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True,
                       parameters=longer, name="param_pendulum")
    assert s.nlp.num_parameters == sum(len(w) for w in pars)
