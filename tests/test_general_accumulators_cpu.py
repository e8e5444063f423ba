"""CPU checks of solver.py: accumulate_general_constraint (round 6): coupling GeneralConstraint rows as accumulator states.

With the accumulators propagated along the horizon (s_1 = 0, s_{t+1} from the transformed dynamics rows) the last knot's extra
rows must reproduce the original general rows g(z) for any z -- constants, signs and the split into one-knot terms included --,
rows of one knot join that knot's stage constraint, rows that are not sums of one-knot terms are refused, and the maps pick the
original variables / rows out of the transformed layout in the reference order."""
import numpy as np
import pytest

from _dag_eval import evaluate

import dto_amd
from dto_amd import problems as P
from dto_amd.solver import accumulate_general_constraint


def _env(prefix_vals):
    env = {}
    for nm, v in prefix_vals.items():
        for i, val in enumerate(v):
            env[(nm, i)] = float(val)
    return env


@pytest.mark.parametrize("name", ["acrobot_coupled", "ref_general_coupled", "pendulum_coupled"])
def test_accumulators_reproduce_the_general_rows(name):
    p = {"acrobot_coupled": lambda: P.build_acrobot_coupled(T=8), "ref_general_coupled": lambda: P.build_ref_general_coupled(inequality=0.05),
         "pendulum_coupled": lambda: P.build_pendulum_coupled(T=40, total=1.0, u_max=3.0)}[name]()
    gc = p["general_constraint"]
    out = accumulate_general_constraint(p["dynamics"], p["objective"], p["constraints"], p["bounds"], gc, True)
    assert out is not None
    dyn, obj, con, bnd, zmap, mumap, musign = out
    T, n, m = p["T"], p["n"], p["m"]
    N = dyn[0].num_state
    na = N - n
    assert na >= 1 and all(d.num_state == N and d.num_next_state == N for d in dyn) and np.all(musign == 1.0)
    rng = np.random.default_rng(4)
    nz = n * T + m * (T - 1)
    z = rng.standard_normal(nz)
    g_ref = np.array(evaluate(gc.evaluate_expr, _env({"z": z})))
    # propagate the accumulators: the accumulator rows of stage t read  y_acc - x_acc - e_t(x, u) = 0
    xs = [z[t * (n + m):t * (n + m) + n] for t in range(T)]
    us = [z[t * (n + m) + n:t * (n + m) + n + m] for t in range(T - 1)]
    acc = np.zeros(na)
    zt = []
    for t in range(T - 1):
        x_full = np.concatenate([xs[t], acc])
        env = _env({"x": x_full, "u": us[t], "y": np.concatenate([xs[t + 1], np.zeros(na)])})
        r = np.array(evaluate(dyn[t].evaluate_expr, env))
        # original dynamics rows untouched
        r0 = np.array(evaluate(p["dynamics"][t].evaluate_expr, _env({"x": xs[t], "u": us[t], "y": xs[t + 1]})))
        assert np.max(np.abs(r[:n] - r0)) == 0.0
        zt += list(x_full) + list(us[t])
        acc = -r[n:]                      # with y_acc = 0 the row is -(x_acc + e_t): the next accumulator value
    x_full = np.concatenate([xs[T - 1], acc])
    zt += list(x_full)
    # the transformed problem's last-knot rows: original stage rows first, then the extra rows
    cT = np.array(evaluate(con[T - 1].evaluate_expr, _env({"x": x_full})))
    qT = p["constraints"][T - 1].num_constraint
    nd2 = (T - 1) * N
    # every general row sits where mumap says, with the original value
    vals = {}
    off = nd2
    for t in range(T):
        q = con[t].num_constraint
        if q:
            xt = np.array(zt[t * (N + m):t * (N + m) + N]) if t < T - 1 else x_full
            ut = us[t] if t < T - 1 else np.zeros(0)
            v = np.array(evaluate(con[t].evaluate_expr, _env({"x": xt, "u": ut})))
            for j in range(q):
                vals[off + j] = v[j]
        off += q
    n_dyn, n_stage = (T - 1) * n, sum(c.num_constraint for c in p["constraints"])
    got = np.array([vals[int(k)] for k in mumap[n_dyn + n_stage:]])
    assert np.max(np.abs(got - g_ref)) <= 1e-12 * max(1.0, np.max(np.abs(g_ref))), (got, g_ref)
    # variables: the transformed vector restricted by zmap is the original one
    assert np.array_equal(np.array(zt)[zmap], z)
    # inequality flags travel with the rows
    ineq_T = set(con[T - 1].indices_inequality)
    for r1 in gc.indices_inequality:
        k = int(mumap[n_dyn + n_stage + r1 - 1])
        t_of = T - 1 if k >= nd2 + sum(c.num_constraint for c in con[:T - 1]) else None
        if t_of == T - 1:
            assert (k - (nd2 + sum(c.num_constraint for c in con[:T - 1])) + 1) in ineq_T
    # accumulators: fixed at zero at the first knot, free afterwards
    assert np.all(bnd[0].state_lower[n:] == 0.0) and np.all(bnd[0].state_upper[n:] == 0.0)
    assert np.all(np.isneginf(bnd[1].state_lower[n:])) and np.all(np.isposinf(bnd[1].state_upper[n:]))


def test_rows_that_are_not_sums_of_one_knot_terms_are_refused():
    from dto_amd.model import GeneralConstraint
    p = P.build_pendulum(T=6, evaluate_hessian=True)
    n, m, T = 2, 1, 6
    nz = n * T + m * (T - 1)
    i2, i4 = 1 * (n + m), 3 * (n + m)
    prod = GeneralConstraint(lambda z, w: np.array([z[i2] * z[i4] - 0.1], dtype=object), nz, 0, evaluate_hessian=True)
    assert accumulate_general_constraint(p["dynamics"], p["objective"], p["constraints"], p["bounds"], prod, True) is None
    # a nonlinear but separable row is fine: sin(theta_2) + theta_4^2
    sep = GeneralConstraint(lambda z, w: np.array([np.sin(z[i2]) + z[i4] ** 2.0 - 0.3], dtype=object), nz, 0, evaluate_hessian=True)
    out = accumulate_general_constraint(p["dynamics"], p["objective"], p["constraints"], p["bounds"], sep, True)
    assert out is not None and out[0][0].num_state == 3
    # too many coupling rows for the lane-per-instance kernels' 16 states
    many = GeneralConstraint(lambda z, w: np.array([z[i2] + (k + 1) * z[i4] for k in range(15)], dtype=object), nz, 0, evaluate_hessian=True)
    assert accumulate_general_constraint(p["dynamics"], p["objective"], p["constraints"], p["bounds"], many, True) is None
    # rows of one knot only: nothing to accumulate (fold_general_constraint's case)
    one = GeneralConstraint(lambda z, w: np.array([z[i2] - 0.1], dtype=object), nz, 0, evaluate_hessian=True)
    assert accumulate_general_constraint(p["dynamics"], p["objective"], p["constraints"], p["bounds"], one, True) is None
    # Options.general_rows = "border" keeps a separable row on the bordered path (linear rows: the bordered path has no
    # second derivatives of general rows, the reference's own call being broken -- src/general_constraint.jl:87)
    lin = GeneralConstraint(lambda z, w: np.array([z[i2] + z[i4] - 0.3], dtype=object), nz, 0, evaluate_hessian=True)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, general_constraint=lin,
                       options=dto_amd.Options(general_rows="border"), name="pendulum_sep")
    assert s.general_rows_path == "border" and s._pad is None
    with pytest.raises(ValueError):
        dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, general_constraint=lin,
                       options=dto_amd.Options(general_rows="accumulate"), name="pendulum_sep")
