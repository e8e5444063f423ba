"""The instance-major engine (csrc/dto_im_kernels.hpp: per-instance stage records, work lists, one factorisation attempt per
instance and pass) against the SoA-tile engine, the oracle and the independent C port.

Both engines call the same block algebra, convergence test, inertia ladder and filter line search; they differ in data
layout, in the schedule (an instance of the instance-major engine retries a rejected factorisation in the next pass instead
of inside one launch) and in the order in which the per-stage partial sums of a residual norm are added (butterfly over the
64 knots of a wavefront instead of a sequential walk).  Expected agreement:
  * one iteration from the same point: step, multipliers step, slack step within 1e-9 of the SoA engine's (the sweeps run
    identical arithmetic on identical inputs);
  * iterate histories: identical decisions while rounding of the summed norms has not tipped a borderline filter decision
    (the first iterations), and KKT points at the end -- checked with the oracle's kkt_report like every other solve test.
Tolerances are written at the asserts.
"""
import os

import numpy as np
import pytest

from conftest import product_solver

pytestmark = pytest.mark.gpu

IM_MODELS = [("pendulum", 50), ("acrobot", 101), ("cartpole", 200), ("car", 51)]
_IM_CACHE = {}


def im_solver(model, T):
    """A Solver whose plugin carries the instance-major kernels: they are compiled in only under DTO_PLUGIN_IM=1
    (plugin.py:with_im_engine; __graft_entry__.build() prebuilds these four variants)."""
    import os
    import dto_amd
    from dto_amd import problems as P
    if (model, T) not in _IM_CACHE:
        old = os.environ.get("DTO_PLUGIN_IM")
        os.environ["DTO_PLUGIN_IM"] = "1"
        try:
            p = getattr(P, f"build_{model}")(T=T, evaluate_hessian=True)
            s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name=model + "_im")
        finally:
            if old is None:
                del os.environ["DTO_PLUGIN_IM"]
            else:
                os.environ["DTO_PLUGIN_IM"] = old
        _IM_CACHE[(model, T)] = (s, p)
    return _IM_CACHE[(model, T)]


def test_default_plugins_carry_no_instance_major_engine():
    import dto_amd
    s, p = product_solver("pendulum", 50)
    with pytest.raises(Exception):
        s.set_engine("im")
    s.set_engine("auto")


def _guesses(s, p, B, seed=0):
    import dto_amd
    rng = np.random.Generator(np.random.PCG64(seed))
    Z = np.zeros((B, s.nlp.num_variables))
    for b in range(B):
        xs, us = p["guess"](rng)
        dto_amd.initialize_states(s, xs)
        dto_amd.initialize_controls(s, us)
        Z[b] = s._z0
    return Z


def _run_iterations(s, Z, engine, n_iter, names):
    import torch
    s.set_engine(engine)
    try:
        d = torch.tensor(Z, device="cuda")
        s.begin_batch(d.data_ptr(), Z.shape[0], Z.shape[1])
        assert s.engine() == engine
        s.iterate_batch(n_iter)
        out = {k: s.peek_batch(k) for k in names}
        out["stats"] = s.stats_batch()
        return out
    finally:
        s.set_engine("auto")


# The engine is frozen (round 6: opt-in, 1.7x slower than the SoA tiles, no further development): the default suite keeps the
# pendulum cases as smoke tests; DTO_IM_ALL=1 (at build time too: __graft_entry__.build) runs the four models of rounds 3 - 5.
_ALL = os.environ.get("DTO_IM_ALL") == "1"


@pytest.mark.parametrize("model,T,B", [("pendulum", 50, 70), ("acrobot", 101, 130), ("cartpole", 200, 3), ("car", 51, 66)] if _ALL else [("pendulum", 50, 70)])
def test_first_iterations_match_the_soa_engine(model, T, B):
    """Same guesses, k iterations on each engine: iterates, multipliers, bound multipliers, slacks and the last step."""
    s, p = im_solver(model, T)
    Z = _guesses(s, p, B, seed=11)
    names = ["z", "multipliers", "dz", "dmultipliers", "z_lower", "z_upper", "slack", "slack_multipliers", "dslack"]
    for k in (1, 3):
        a = _run_iterations(s, Z, "soa", k, names)
        b = _run_iterations(s, Z, "im", k, names)
        assert np.array_equal(a["stats"]["iterations"], b["stats"]["iterations"])
        assert np.array_equal(a["stats"]["status"], b["stats"]["status"])
        # decisions of the iteration: step size and regularisation identical
        # (the fraction-to-the-boundary step length is computed from the step: last bits may differ)
        assert np.allclose(a["stats"]["alpha"], b["stats"]["alpha"], rtol=1e-6, atol=0.0), k
        assert np.allclose(a["stats"]["delta_w"], b["stats"]["delta_w"], rtol=1e-12, atol=0.0), k
        for n in names:
            if a[n].size == 0:
                continue
            scale = max(1.0, np.max(np.abs(a[n])))
            # One iteration from the same point: both engines solve the same regularised KKT system (condition number ~1e8 at a
            # random guess: delta_c = 1e-8 on the dual block) with differently rounded factorisations (the two kernels contract
            # multiply-adds differently): 2e-8 of the vector's scale for the primal quantities -- the bar the KKT-step tests
            # apply against a dense solve is 1e-8 of the solution norm --, 1e-7 for the multipliers.  After three iterations of
            # a nonconvex solve those differences have been fed back through the iterates (multipliers of 3e5 at a random guess,
            # three factorisations of systems conditioned like 1e8 in a row): 1e-5; measured 1e-6 .. 3e-7 depending on how the
            # compiler contracts the multiply-adds of the SoA sweeps.
            # (round 5: cartpole's first steps are Gauss-Newton steps of the penalty phase, delta_w = 1e-4 -- a less well
            #  conditioned system than the ladder's: 5e-8; with the SoA engine's small-batch form -- 50 chunks of four stages,
            #  separator system by cyclic reduction -- 5.4e-8 was observed: 1e-7)
            tol = 1e-5 if k > 1 else 1e-7
            assert np.max(np.abs(a[n] - b[n])) <= tol * scale, (k, n, np.max(np.abs(a[n] - b[n])), scale)


@pytest.mark.parametrize("model,T,B", [("pendulum", 50, 64), ("acrobot", 101, 200), ("car", 51, 100), ("cartpole", 200, 2)] if _ALL else [("pendulum", 50, 64)])
def test_full_solves_are_kkt_points(model, T, B):
    """Solve to the reference tolerances on the instance-major engine; every converged instance is checked against the
    oracle's KKT conditions (tests/test_solve_gpu.py's checker); convergence and iteration counts are compared with the SoA
    engine on the same guesses."""
    import torch
    from test_solve_gpu import kkt_report, oracle_for
    s, p = im_solver(model, T)
    Z = _guesses(s, p, B, seed=5)
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    res = {}
    for eng in ("soa", "im"):
        s.set_engine(eng)
        try:
            d = torch.tensor(Z, device="cuda")
            xo = torch.zeros((B, nz), device="cuda", dtype=torch.float64)
            mo = torch.zeros((B, max(1, nc)), device="cuda", dtype=torch.float64)
            st, it = s.solve_batch(d.data_ptr(), B, nz, xo.data_ptr(), nz, mo.data_ptr(), max(1, nc))
            assert s.engine() == eng
            torch.cuda.synchronize()
            res[eng] = (st.copy(), it.copy(), xo.cpu().numpy(), mo.cpu().numpy()[:, :nc])
        finally:
            s.set_engine("auto")
    st, it, X, MU = res["im"]
    assert np.all(st == 1), (np.bincount(st), it.max())
    assert np.all(res["soa"][0] == 1)
    # same algorithm: the iteration counts have the same distribution (individual instances may part ways once rounding tips a
    # filter decision): medians within 15 %
    # (cartpole: ONE deterministic guess, 236 or 351 iterations depending on the last bits -- see test_sequential_sweeps_gpu.py)
    if B >= 16:
        assert abs(np.median(it) - np.median(res["soa"][1])) <= 0.15 * max(4.0, np.median(res["soa"][1])), (np.median(it), np.median(res["soa"][1]))
    onlp = oracle_for(model, T)
    for b in range(0, B, max(1, B // 8)):
        rep = kkt_report(onlp, X[b], MU[b])
        # the bars of tests/test_solve_gpu.py: equality-constrained models 1e-6 / 1e-5; barrier models (bounds, inequality
        # rows) to the barrier accuracy compl_inf_tol = 1e-3 of the reference Options
        barrier = model in ("car", "cartpole")
        assert rep["violation"] <= (1e-5 if barrier else 1e-6) and rep["bound_viol"] <= 1e-12 and rep["sign_ok"], rep
        assert rep["stationarity"] <= (1e-3 if barrier else 1e-5) and rep["compl"] <= 1e-3, rep
