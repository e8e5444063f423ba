"""Iterative refinement of the KKT step (dto_options.kkt_refinement, ABI 4; csrc/dto_kkt_kernels.hpp: k_kkt_refine) and the
extended-precision reference that settles what the T = 1000 step bars should be.

The reference ("truth"): the oracle's K and right-hand side (the system of examples/pendulum/pendulum.jl:138-198) solved by
sparse LU + refinement with residuals in np.longdouble until the correction is below 1e-13 of the step
(tests/extended_precision.py).  Measured on the acrobot T = 1000 bench state (tools/step_truth.py, profiles/r06/step_truth_*):
  * plain float64 sparse LU is within 1e-13 of truth and half an ulp of noise in the data moves truth by 1e-15: the systems are
    well conditioned, 1e-8 is a fair bar for anything that solves them;
  * the sequential sweeps (the bench's path) are within 7.3e-10;
  * the time-partitioned sweeps are not: 1e-8 .. 2.5e-8 in the Gauss-Newton phase, up to 5.2e-6 at delta_w = 0 -- the
    partition's cancellation (csrc/dto_kkt_kernels.hpp, "iterative refinement"); ONE refinement pass takes them below 1e-9.
Same-state comparisons go through dto_solver_launch_op (EVAL, CONV, FACTOR_SOLVE -> step; REFINE -> one pass on that step).
"""
import numpy as np
import pytest

from conftest import product_solver

pytestmark = pytest.mark.gpu


def _truth(K, rhs):
    import scipy.sparse as sp
    from extended_precision import solve_extended
    x, info = solve_extended(sp.csc_matrix(K), rhs)
    assert info["converged"], info
    return np.asarray(x, dtype=np.float64), float(np.max(np.abs(x)))


def _steps_before_and_after(s, names=("dz", "dmultipliers")):
    import torch
    for op_name in ("eval", "conv", "factor_solve"):
        s.launch_op(op_name)
    torch.cuda.synchronize()
    before = [s.peek_batch(k) for k in names]
    state = dict(dw=s.scalar_batch("delta_w"), gam=s.scalar_batch("gamma"), status=s.scalar_batch("status"), mu=s.scalar_batch("mu"),
                 nfact=s.scalar_batch("nfact").copy())
    s.launch_op("kkt_refine")          # one whole pass: residual -> records, factor + solve, step := step + correction
    torch.cuda.synchronize()
    after = [s.peek_batch(k) for k in names]
    # the refined solve is ONE more factorisation of every running lane at the (delta_w, gamma) the iteration had accepted
    assert np.array_equal(s.scalar_batch("delta_w"), state["dw"]) and np.array_equal(s.scalar_batch("gamma"), state["gam"])
    run = state["status"] == 0
    assert np.all(s.scalar_batch("nfact")[run] == state["nfact"][run] + 1)
    return before, after, state


@pytest.mark.parametrize("B,P,it", [(3, 0, 5), (8, 8, 14), (8, 16, 14), (3, 32, 14), (2, 1, 14)])
def test_acrobot_T1000_step_against_extended_precision_truth_before_and_after_one_pass(B, P, it):
    """BASELINE configs[2] sizes (N_z 4 999, N_c 4 004).  P = 0: the library's own chunk count for the batch (64); P = 8 / 16 at
    iteration 14 are the worst cases of the study (delta_w = 0: 5e-6 unrefined); P = 1: the sequential sweeps need no pass."""
    import torch
    from bench import make_guesses
    from test_baseline_sizes_gpu import oracle_for, sparse_kkt
    s, p = product_solver("acrobot", 1000)
    onlp = oracle_for("acrobot", 1000)
    nz = s.nlp.num_variables
    z0 = torch.tensor(make_guesses(s, p, B, seed=1000), device="cuda")
    s.set_partitions(P)
    try:
        s.begin_batch(z0.data_ptr(), B, nz)
        s.iterate_batch(it)
        (dz0, dl0), (dz1, dl1), st = _steps_before_and_after(s)
        z, lam = s.peek_batch("z"), s.peek_batch("multipliers")
        parts = s.partitions()
        assert parts == (P if P else 64)
    finally:
        s.set_partitions(0)
        s.release_state()
    worst0 = worst1 = 0.0
    for b in np.flatnonzero(st["status"] == 0):
        K, rhs, _ = sparse_kkt(onlp, z[b], lam[b], st["dw"][b], 1e-8, gam=st["gam"][b])
        x, scale = _truth(K, rhs)
        e0 = np.max(np.abs(np.concatenate([dz0[b], dl0[b]]) - x)) / scale
        e1 = np.max(np.abs(np.concatenate([dz1[b], dl1[b]]) - x)) / scale
        worst0, worst1 = max(worst0, e0), max(worst1, e1)
        # north_star: primal / dual iterates within 1e-8 relative -- after one pass on every path; the sequential sweeps without
        assert e1 <= 1e-8, (b, parts, e0, e1, st["dw"][b], st["gam"][b])
        if parts == 1:
            assert e0 <= 1e-8, (b, e0)
        else:
            # unrefined time-partitioned sweeps: the measured bar (5.2e-6 worst in profiles/r06/step_truth_chunked_*), and the pass
            # must not make a step worse than rounding
            assert e0 <= 2e-5, (b, parts, e0)
            assert e1 <= max(e0, 1e-10), (b, e0, e1)
    print(f"[refinement] B={B} P={parts} iteration {it}: forward error vs extended-precision truth {worst0:.2e} -> {worst1:.2e}")


@pytest.mark.parametrize("model,T,P", [("car", 40, 0), ("car", 40, 4), ("cartpole", 60, 0), ("cartpole", 60, 1)])
def test_refined_step_of_a_barrier_iteration_against_the_dense_primal_dual_system(model, T, P):
    """Bounds, slack-eliminated inequality rows, variables fixed by equal bounds (car: obstacle rows at every knot, bounded
    actions, fixed endpoints; cartpole: bounded action): the residual the refinement pass forms must be that of the system the
    sweeps solve -- barrier terms on the right-hand side, Sigma on the diagonal, identity rows -- or the refined step drifts
    AWAY from the truth of that system.  Dense system from the oracle's derivatives (tests/test_kkt_gpu.py)."""
    import torch
    import dto_amd
    from oracle import dto_oracle as O, sympy_models as S
    from test_kkt_gpu import primal_dual_system
    s, p = product_solver(model, T)
    n = s.nlp
    op = S.build(model, T, evaluate_hessian=True)
    onlp = O.NLPData(op["dynamics"], op["objective"], op["constraints"], op["bounds"], evaluate_hessian=True)
    nz = n.num_variables
    B = 3
    Z = np.zeros((B, nz))
    for b in range(B):
        xs, us = p["guess"](np.random.Generator(np.random.PCG64(40 + b)))
        dto_amd.initialize_states(s, xs)
        dto_amd.initialize_controls(s, us)
        Z[b] = s._z0
    z0 = torch.tensor(Z, device="cuda")
    s.set_partitions(P)
    try:
        s.begin_batch(z0.data_ptr(), B, nz)
        s.iterate_batch(3)
        (dz0, dl0, ds0), (dz1, dl1, ds1), st = _steps_before_and_after(s, ("dz", "dmultipliers", "dslack"))
        z, lam = s.peek_batch("z"), s.peek_batch("multipliers")
        zl, zu, sl, zs = (s.peek_batch(k) for k in ("z_lower", "z_upper", "slack", "slack_multipliers"))
    finally:
        s.set_partitions(0)
        s.release_state()
    clo, _ = n.constraint_bounds
    ineq = np.where(np.isneginf(clo))[0]
    for b in range(B):
        K, rhs = primal_dual_system(onlp, n, z[b], lam[b], zl[b], zu[b], sl[b], zs[b], st["mu"][b], st["dw"][b], st["gam"][b])
        x, scale = _truth(K, rhs)
        e0 = np.max(np.abs(np.concatenate([dz0[b], dl0[b]]) - x)) / scale
        e1 = np.max(np.abs(np.concatenate([dz1[b], dl1[b]]) - x)) / scale
        assert e1 <= 1e-8 and e1 <= max(2.0 * e0, 1e-10), (model, b, e0, e1)
        if len(ineq):
            ds_ref = -(sl[b] / zs[b]) * (lam[b][ineq] + x[nz:][ineq] - st["mu"][b] / sl[b])
            assert np.max(np.abs(ds1[b] - ds_ref)) <= 1e-8 * max(scale, np.max(np.abs(ds_ref)))


def test_solves_with_refinement_converge_to_kkt_points():
    """Options(kkt_refinement = 1 / 2) through the solver's own loop (dto_solver_iterate keeps a copy of the stage records and
    puts it back): 64 acrobot T = 101 seeds converge as without the pass, sampled solutions are KKT points of the oracle; and
    the step of an iteration of that loop equals, bit for bit, the manual op sequence from the same state."""
    import torch
    import dto_amd
    from dto_amd import problems as P
    from bench import make_guesses
    from test_solve_gpu import kkt_report, oracle_for
    T, B = 101, 64
    onlp = oracle_for("acrobot", T)
    res = {}
    for passes in (0, 1, 2):
        pr = P.build_acrobot(T=T, evaluate_hessian=True)
        s = dto_amd.Solver(pr["dynamics"], pr["objective"], pr["constraints"], pr["bounds"], evaluate_hessian=True, name="acrobot",
                           options=dto_amd.Options(kkt_refinement=passes))
        nz, nc = s.nlp.num_variables, s.nlp.num_constraint
        z0 = torch.tensor(make_guesses(s, pr, B, seed=7), device="cuda")
        zo = torch.zeros((B, nz), device="cuda", dtype=torch.float64)
        lo = torch.zeros((B, nc), device="cuda", dtype=torch.float64)
        st, it = s.solve_batch(z0.data_ptr(), B, nz, zo.data_ptr(), nz, lo.data_ptr(), nc)
        torch.cuda.synchronize()
        res[passes] = (st.copy(), it.copy(), zo.cpu().numpy(), lo.cpu().numpy())
        assert np.all(st == 1), (passes, np.bincount(st))
        for b in range(0, B, 16):
            rep = kkt_report(onlp, res[passes][2][b], res[passes][3][b])
            assert rep["violation"] <= 1e-6 and rep["stationarity"] <= 1e-5, (passes, rep)
        if passes == 1:
            # the loop's pass is the manual sequence EVAL, CONV, FACTOR_SOLVE, REFINE from the same state, bit for bit
            s.begin_batch(z0.data_ptr(), B, nz)
            s.iterate_batch(3)
            z_a = s.peek_batch("z")
            s.iterate_batch(1)
            dz_loop = s.peek_batch("dz")
            s.begin_batch(z0.data_ptr(), B, nz)
            s.iterate_batch(3)
            assert np.array_equal(z_a, s.peek_batch("z"))
            for op_name in ("eval", "conv", "factor_solve", "kkt_refine"):
                s.launch_op(op_name)
            torch.cuda.synchronize()
            assert np.array_equal(dz_loop, s.peek_batch("dz"))
        s.close()
    # the pass changes steps in their ninth digit: iteration counts stay in the same range (individual seeds may part ways)
    m0 = np.median(res[0][1])
    for passes in (1, 2):
        assert abs(np.median(res[passes][1]) - m0) <= 0.25 * max(8.0, m0), (m0, np.median(res[passes][1]))
