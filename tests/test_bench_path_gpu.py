"""The oracle on the EXACT kernel path bench.py times.

bench.py's headline runs acrobot T = 1000 in the plain sequential form with more tiles than wavefront slots:
`k_kkt_fwd_seq` (every inertia-correction round of a tile inside one launch) -> `k_kkt_bwd_early` on the second stream
beside it + `k_kkt_bwd_rest` after it -> `k_linesearch` -> `k_update_eval` (UPDATE fused with the next EVAL).  Until now that
combination met the oracle only as one `newton_only` step (partitions = 1, two instances) -- the full-solve test at
T = 1000 runs 64 instances, i.e. the time-partitioned form.

Here: the bench's own seeded guesses, 1 100 tiles (> 1 024 wavefront slots), set_partitions(1).  The step (dz, dlambda) of
the 5th and of the 25th iteration -- computed INSIDE dto_solver_iterate, on that path -- is compared with the solution of
the ORACLE's K (examples/pendulum/pendulum.jl:138-198) at the regularisation (delta_w, Gauss-Newton flag) the device chose,
computed in extended precision (round 6: tests/extended_precision.py; bar 1e-8 of the step); then the whole batch is run to
termination and 64 of its instances are checked against the oracle's KKT conditions.
"""
import os

import numpy as np
import pytest

from conftest import product_solver

pytestmark = pytest.mark.gpu


def _step_check(s, onlp, z, lam, dz, dlam, dw, gam, picks, tag):
    """Round 6: the reference is the solution of the oracle's K in EXTENDED precision (sparse LU + refinement
    with np.longdouble residuals, tests/extended_precision.py), not a float64 LU solve whose own error nobody knew.  Measured on
    24 steps of this path (tools/step_truth.py, profiles/r06/step_truth_acrobot_T1000.json): the sequential sweeps are within
    7.3e-10 of it, plain float64 LU within 7e-13, half an ulp of data noise moves it by 3e-14 -- so the bar is north_star's 1e-8,
    without the 1e-6 / 1e-7 allowances of rounds 4 - 5 (those came from the time-partitioned sweeps of the OTHER T = 1000 test)."""
    from extended_precision import residual_extended, solve_extended
    from test_baseline_sizes_gpu import sparse_kkt
    worst = 0.0
    for b in picks:
        K, rhs, _ = sparse_kkt(onlp, z[b], lam[b], dw[b], 1e-8, gam=gam[b])
        x, info = solve_extended(K, rhs)
        assert info["converged"], (tag, b, info)
        scale = float(np.max(np.abs(x)))
        got = np.concatenate([dz[b], dlam[b]])
        err = float(np.max(np.abs(got - np.asarray(x, dtype=np.float64)))) / scale
        worst = max(worst, err)
        assert err <= 1e-8, (tag, b, err, scale, dw[b], gam[b])
        # backward error against |K||x| + |b| (products accumulated in extended precision): observed <= 2.4e-10
        assert residual_extended(K, got, rhs) <= 2e-9 * (abs(K).max() * np.max(np.abs(got)) + np.max(np.abs(rhs))), (tag, b)
    print(f"[bench path] {tag}: forward error vs extended-precision truth {worst:.2e} (bar 1e-8)")


def test_cfg3_T1000_overlapped_multi_round_sequential_sweeps_step_and_solve_vs_oracle():
    import scipy.sparse as sp
    import torch
    from bench import make_guesses
    from test_baseline_sizes_gpu import oracle_for
    assert os.environ.get("DTO_OVERLAP_SWEEPS", "1") != "0" and os.environ.get("DTO_FUSE_UPDATE", "1") != "0"
    T, tiles = 1000, 1100                       # > 1 024 tiles: dto_solver_iterate takes the two-stream path (dto_solver.cpp: overlap_sweeps)
    B = tiles * 64
    s, p = product_solver("acrobot", T)
    onlp = oracle_for("acrobot", T)
    s.options.max_iter = 1000
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    z0 = torch.empty((B, nz), device="cuda", dtype=torch.float64)
    for b0 in range(0, B, 8192):                # the bench's rank-0 stream of guesses, chunk by chunk
        nb = min(8192, B - b0)
        if b0 == 0:
            rng = np.random.Generator(np.random.PCG64(1000))
        z0[b0:b0 + nb] = torch.from_numpy(make_guesses(s, p, nb, 1000, rng=rng)).cuda()
    s.set_partitions(1)
    try:
        s.begin_batch(z0.data_ptr(), B, nz)
        assert s.partitions() == 1 and s.engine() == "soa"
        done = 0
        for upto in (4, 24):
            s.iterate_batch(upto - done)        # fused UPDATE+EVAL passes inside (an even number per call), ends on UPDATE
            done = upto
            z, lam = s.peek_batch("z"), s.peek_batch("multipliers")
            nf0 = s.scalar_batch("nfact").copy()
            s.iterate_batch(1)                  # EVAL, CONV, k_kkt_fwd_seq || k_kkt_bwd_early, k_kkt_bwd_rest, line search, UPDATE
            done += 1
            nf = s.scalar_batch("nfact") - nf0
            dw, gam = s.scalar_batch("delta_w"), s.scalar_batch("gamma")
            # iteration 5: penalty phase, Gauss-Newton model, one factorisation per lane; iteration 25: the lanes still running are
            # on the filter's exact-Hessian ladder -- multi-round: some lane needed >= 3 factorisations in this launch
            assert nf.max() >= (1 if upto == 4 else 3) and np.median(nf[nf > 0]) >= 1, (upto, nf.max(), np.median(nf))
            # instances: the first of the batch, one of the last (ragged end of the launch), the ones with the most attempts
            # (an instance that has terminated takes no step: its dz is whatever its last iteration left -- by iteration 25 the
            #  fastest ones are done since round 5's penalty phase; nf > 0 = factorised in this iteration = still running)
            live = np.flatnonzero(nf > 0)
            picks = sorted({int(live[0]), int(live[-1]), int(np.argmax(nf)), int(np.argsort(nf)[-2])})
            sub = lambda a: {b: a[b].copy() for b in picks}
            zs, ls = sub(z), sub(lam)
            del z, lam
            dz, dl = sub(s.peek_batch("dz")), sub(s.peek_batch("dmultipliers"))
            _step_check(s, onlp, zs, ls, dz, dl, dw, gam, picks, f"iteration {done}")
        # ---- the rest of the solve on the same path; 64 instances against the oracle's KKT conditions
        zo = torch.full((B, nz), float("nan"), device="cuda", dtype=torch.float64)
        lo = torch.full((B, nc), float("nan"), device="cuda", dtype=torch.float64)
        status, iters = s.run_batch(zo.data_ptr(), nz, lo.data_ptr(), nc)
        torch.cuda.synchronize()
    finally:
        s.set_partitions(0)
    assert np.all((status == 1) | (status == 2)), np.bincount(status)
    assert np.mean(status == 1) >= 0.99, np.bincount(status)          # 70 359 of 70 400 (round 5: 91 %)
    sel = np.arange(0, B, B // 64)[:64]
    Zs, Ls = zo[torch.tensor(sel, device="cuda")].cpu().numpy(), lo[torch.tensor(sel, device="cuda")].cpu().numpy()
    s.release_state()
    rows, cols = np.array(onlp.jacobian_structure()).T - 1
    idx = s.nlp.indices
    n_conv = 0
    for k, b in enumerate(sel):
        if status[b] != 1:
            continue
        n_conv += 1
        z, lam = Zs[k], Ls[k]
        J = sp.coo_matrix((onlp.eval_constraint_jacobian(z), (rows, cols)), shape=(nc, nz)).tocsr()
        stat = np.max(np.abs(onlp.eval_objective_gradient(z) + J.T @ lam))
        viol = np.max(np.abs(onlp.eval_constraint(z)))
        assert viol <= 1e-6 and stat <= 1e-5, (b, stat, viol, iters[b])
        assert np.linalg.norm(z[np.array(idx.states[0]) - 1] - p["x1"]) < 1e-3      # test/solve.jl:136
        assert np.linalg.norm(z[np.array(idx.states[-1]) - 1] - p["xT"]) < 1e-3     # test/solve.jl:137
    assert n_conv >= 60, n_conv
    print(f"[bench path] {int(np.sum(status == 1))}/{B} converged, median iterations {np.median(iters):.0f}; {n_conv}/64 sampled instances KKT-checked")
