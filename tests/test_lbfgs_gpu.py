"""The reference's default solve mode on the GPU (Missing 2): a problem built with evaluate_hessian=false --
src/solver.jl:7, and what the reference's own acrobot and car examples run (examples/acrobot/acrobot.jl:122-123,
examples/car/car.jl:63) -- leaves Ipopt on hessian_approximation = limited-memory.  Here: compact L-BFGS (history 6) whose
low-rank part is a dense border of the block-tridiagonal system (csrc/dto_kkt_kernels.hpp, "limited-memory BFGS";
dto_options.hessian_approximation = DTO_HESSIAN_LBFGS), mirrored in the C port (oracle/cpu_port/solver_port.c: qn_*).

  * the four reference configs from 64 seeded guesses each: >= 62 of 64 converge (cartpole: its one deterministic guess), every
    converged point is a KKT point of the ORACLE's problem and passes the reference's own endpoint asserts (test/solve.jl:136-137);
  * no second derivative is used: the step of an iteration equals the port's step with the same history to 1e-6 (first
    iterations of a pendulum solve), and the port's Hessian model is sigma I + low rank by construction;
  * the mode is reachable through dto_options alone (Options(hessian_approximation="lbfgs") on a problem WITH Hessians).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solver(model, T, evaluate_hessian=False, **opts):
    import dto_amd
    from dto_amd import problems as P
    p = getattr(P, f"build_{model}")(T=T, evaluate_hessian=evaluate_hessian)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=evaluate_hessian,
                       options=dto_amd.Options(**opts), name=model)
    return s, p


def _guesses(s, p, B):
    import dto_amd
    nz = s._solve_nlp.num_variables
    Z = np.zeros((B, nz))
    for b in range(B):
        xs, us = p["guess"](np.random.Generator(np.random.PCG64(b)))
        dto_amd.initialize_states(s, xs)
        dto_amd.initialize_controls(s, us)
        Z[b] = s._z0
    return Z


@pytest.mark.parametrize("model,T,B,need", [("pendulum", 50, 64, 64), ("car", 51, 64, 62), ("cartpole", 200, 1, 1), ("acrobot", 101, 64, 62)])
def test_default_mode_converges_on_the_reference_configs(model, T, B, need):
    import torch
    from test_solve_gpu import kkt_report, oracle_for
    s, p = _solver(model, T)
    assert s.hessian_mode == "lbfgs"
    nz, nc = s._solve_nlp.num_variables, s._solve_nlp.num_constraint
    Z = _guesses(s, p, B)
    z0 = torch.tensor(Z, device="cuda")
    zo = torch.full((B, nz), float("nan"), device="cuda", dtype=torch.float64)
    lo = torch.full((B, nc), float("nan"), device="cuda", dtype=torch.float64)
    st, it = s.solve_batch(z0.data_ptr(), B, nz, zo.data_ptr(), nz, lo.data_ptr(), nc)
    torch.cuda.synchronize()
    assert int(np.sum(st == 1)) >= need, (model, np.bincount(st), np.median(it))
    onlp = oracle_for(model, T)
    zo, lo = zo.cpu().numpy(), lo.cpu().numpy()
    idx = s.nlp.indices
    for b in [b_ for b_ in range(0, B, max(1, B // 8)) if st[b_] == 1]:
        rep = kkt_report(onlp, zo[b], lo[b])
        assert rep["violation"] <= 1e-6 and rep["stationarity"] <= 1e-5 and rep["compl"] <= 1e-3, (model, b, rep)
        assert np.linalg.norm(zo[b][np.array(idx.states[0]) - 1] - p["x1"]) < 1e-3      # test/solve.jl:136
        assert np.linalg.norm(zo[b][np.array(idx.states[-1]) - 1] - p["xT"]) < 1e-3     # test/solve.jl:137
    print(f"[lbfgs] {model} T={T}: {int(np.sum(st == 1))}/{B} converged, median {np.median(it):.0f} iterations (max {it.max()})")


def test_steps_match_the_port_in_limited_memory_mode():
    """First iterations of a pendulum T = 50 solve, iterate by iterate against oracle/cpu_port in its lbfgs mode (same secant
    pairs, same sigma, same 12 x 12 border system -- a different route to grad_x L(x_k, lam_{k+1}): J'dlam there, the first block
    row of the solved system here)."""
    import torch
    from oracle.cpu_port import PortSolver
    s, p = _solver("pendulum", 50)
    nz = s._solve_nlp.num_variables
    Z = _guesses(s, p, 1)
    ps = PortSolver("pendulum", 50, max_iter=1000, lbfgs=6)
    ps.begin(Z[0])
    z0 = torch.tensor(Z, device="cuda")
    s.begin_batch(z0.data_ptr(), 1, nz)
    for k in range(12):
        s.iterate_batch(1)
        ps.iterate()
        zg = s.peek_batch("z")[0]
        zp = ps.z
        assert np.max(np.abs(zg - zp)) <= 1e-6 * max(1.0, np.max(np.abs(zp))), (k, np.max(np.abs(zg - zp)))
        assert abs(float(s.scalar_batch("qn_sigma")[0]) - ps.qn_sigma) <= 1e-6 * max(1.0, ps.qn_sigma), k
    s.release_state()


def test_mode_is_an_option_of_the_c_abi():
    """Options(hessian_approximation="lbfgs") on a problem built WITH Hessians: same plugin, the approximation replaces them."""
    import dto_amd
    s, p = _solver("pendulum", 50, evaluate_hessian=True, hessian_approximation="lbfgs")
    e, _ = _solver("pendulum", 50, evaluate_hessian=True)
    assert s.hessian_mode == "lbfgs" and e.hessian_mode == "exact"
    for sol in (s, e):
        xs, us = p["guess"](np.random.Generator(np.random.PCG64(0)))
        dto_amd.initialize_states(sol, xs); dto_amd.initialize_controls(sol, us)
        assert dto_amd.solve(sol) == 1
    assert s.iterations > e.iterations                     # a quasi-Newton iteration count, not the Newton one
    assert np.max(np.abs(s._solution - e._solution)) <= 1e-4 * np.max(np.abs(e._solution))


@pytest.mark.parametrize("model,T,B", [("pendulum", 50, 1), ("acrobot", 101, 3), ("car", 51, 70), ("cartpole", 101, 2)])
def test_columns_as_instances_match_the_columns_one_after_the_other(model, T, B):
    """Small batches solve the 12 systems K0 z_c = u_c of an iteration side by side, as the instances of a second solver state
    (csrc/dto_solver.cpp: k_qn_cols_copy / k_qn_cols_gather, csrc/dto_kkt_kernels.hpp: k_qn_cols_rhs) -- one factor + solve
    instead of twelve.  DTO_QN_COLS=0 (read at every begin) runs them one after the other on the batch's own state: the same
    systems (the column state picks its own chunk count: rounding apart), so the first iterates agree closely and both solves
    converge."""
    import os
    import torch
    s, p = _solver(model, T)
    nz = s._solve_nlp.num_variables
    Z = _guesses(s, p, B)
    z0 = torch.tensor(Z, device="cuda")
    got = {}
    for cols in (False, True):
        os.environ["DTO_QN_COLS"] = "1" if cols else "0"
        try:
            s.begin_batch(z0.data_ptr(), B, nz)
            s.iterate_batch(6)
            zk = s.peek_batch("z")[:B].copy()
            zo = torch.full((B, nz), float("nan"), device="cuda", dtype=torch.float64)
            st, it = s.solve_batch(z0.data_ptr(), B, nz, zo.data_ptr(), nz)
            torch.cuda.synchronize()
            got[cols] = (zk, st.copy(), it.copy())
        finally:
            del os.environ["DTO_QN_COLS"]
    scale = max(1.0, np.max(np.abs(got[False][0])))
    assert np.max(np.abs(got[True][0] - got[False][0])) <= 1e-7 * scale, np.max(np.abs(got[True][0] - got[False][0]))
    assert np.all(got[False][1] == 1) and np.all(got[True][1] == 1), (got[False][1], got[True][1])


def test_history_moves_with_its_instance_when_a_batch_is_repacked():
    """Round 5 did not repack batches in this mode (the limited-memory history lives in per-slot rows that dto_solver_repack did not
    move: a repack of a three-tile batch handed running instances the history of the slots they moved into), so a batch paid for every
    tile until its last lane ended.  Round 6: the history block moves with its instance.  200 acrobot T=101 instances three ways --
    an iterate loop without repacking, the same with dto_solver_repack after every slice, dto_solve_batch (repacks by itself):
    statuses, iteration counts, solutions and multipliers bit for bit the same, in the caller's instance order; converged points
    of every tile are KKT points of the oracle's problem."""
    import torch
    from test_solve_gpu import kkt_report, oracle_for
    s, p = _solver("acrobot", 101)
    nz, nc = s._solve_nlp.num_variables, s._solve_nlp.num_constraint
    B = 200
    Z = _guesses(s, p, B)
    z0 = torch.tensor(Z, device="cuda")
    s.options.max_iter = 600
    # (a) no repacking
    s.begin_batch(z0.data_ptr(), B, nz)
    for _ in range(60):
        s.iterate_batch(10)
        if not np.any(s.scalar_batch("status") == 0):
            break
    za = torch.empty_like(z0); la = torch.empty((B, nc), device="cuda", dtype=torch.float64)
    s.end_batch(za.data_ptr(), nz, la.data_ptr(), nc)
    torch.cuda.synchronize()
    st_a, it_a = s.scalar_batch("status").copy(), s.scalar_batch("iter").copy()
    # (b) repack after every slice
    s.begin_batch(z0.data_ptr(), B, nz)
    counts = []
    for _ in range(60):
        s.iterate_batch(10)
        counts.append(s.repack_batch())
        if counts[-1] == 0:
            break
    zb = torch.empty_like(z0); lb = torch.empty((B, nc), device="cuda", dtype=torch.float64)
    s.end_batch(zb.data_ptr(), nz, lb.data_ptr(), nc)
    torch.cuda.synchronize()
    st_b, it_b = s.scalar_batch("status").copy(), s.scalar_batch("iter").copy()
    assert counts[0] >= 0 and counts[0] > counts[-1] and sorted(counts, reverse=True) == counts
    assert np.array_equal(st_a, st_b) and np.array_equal(it_a, it_b), (np.flatnonzero(st_a != st_b), np.flatnonzero(it_a != it_b))
    assert torch.equal(za, zb) and torch.equal(la, lb)
    assert np.mean(st_a == 1) >= 0.95 and len(set(it_a.tolist())) > 10, (np.bincount(st_a.astype(int)), len(set(it_a.tolist())))
    onlp = oracle_for("acrobot", 101)
    zo, lo = zb.cpu().numpy(), lb.cpu().numpy()
    for b in (0, 63, 64, 100, 128, 199):
        if st_b[b] == 1:
            rep = kkt_report(onlp, zo[b], lo[b])
            assert rep["violation"] <= 1e-6 and rep["stationarity"] <= 1e-5, (b, rep)
    # (c) the one-call solve
    zc = torch.empty_like(z0)
    st_c, it_c = s.solve_batch(z0.data_ptr(), B, nz, zc.data_ptr(), nz)
    torch.cuda.synchronize()
    # (the hand-driven loops stop after 600 iterations without the classifying evaluation dto_solver_run adds: an instance still
    #  running there is at the iteration limit here)
    done = st_a == 1
    assert np.array_equal(st_c[done], st_a[done].astype(np.int32)) and np.all(st_c[~done] == 2), (st_c[~done], st_a[~done])
    assert np.array_equal(it_c[done], it_a[done].astype(np.int32)), np.flatnonzero(done & (it_c != it_a))
    sel = torch.tensor(np.flatnonzero(done), device="cuda")
    assert torch.equal(zc[sel], za[sel])


def test_no_instance_repeats_a_null_step_for_ever():
    """Round 6: 24 of these 4 096 acrobot T = 101 instances (guesses with the actions scaled by 0.01) used to end at max_iter with
    alpha = 0 in every one of their last iterations -- after a null step the regularisation was meant to grow tenfold, but the
    ladder's delta_last is never written in this mode and the rule returned delta_w_init again: same point, same direction.  The
    escalation now starts from the delta_w of the rejected direction (k_conv): all but a handful converge, and none is stuck."""
    import torch
    import dto_amd
    s, p = _solver("acrobot", 101)
    nz = s._solve_nlp.num_variables
    B = 4096
    Z = np.zeros((B, nz))
    for b in range(B):
        xs, us = p["guess"](np.random.Generator(np.random.PCG64(b)))
        dto_amd.initialize_states(s, xs)
        dto_amd.initialize_controls(s, [0.01 * u for u in us])
        Z[b] = s._z0
    z0 = torch.tensor(Z, device="cuda")
    zo = torch.empty_like(z0)
    st, it = s.solve_batch(z0.data_ptr(), B, nz, zo.data_ptr(), nz)
    torch.cuda.synchronize()
    assert int(np.sum(st == 1)) >= 4090, (np.bincount(st), np.median(it))
    alpha = s.stats_batch()["alpha"]
    assert not np.any((st != 1) & (alpha == 0.0)), np.flatnonzero((st != 1) & (alpha == 0.0))
    print(f"[lbfgs] 4 096 x acrobot T=101: {int(np.sum(st == 1))} converged, median {np.median(it):.0f} iterations")
