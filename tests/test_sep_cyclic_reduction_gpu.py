"""The separator system of the time-partitioned factorisation by block cyclic reduction, lanes = separators
(csrc/dto_kkt_kernels.hpp: kkt_sep_cr) -- the form a batch of at most DTO_SEP_CR_MAX_INST = 4 instances takes: a batch of one, the reference's own use (examples/acrobot/acrobot.jl:126-133), where the
lane-per-instance elimination walks the separators one after the other with one active lane.

It is the same block LDL' under another symmetric permutation: the step must agree with the sequential elimination
(DTO_SEP_CR=0, read by the library at every call) to rounding, with the dense solve of the ORACLE's K to the bar of
tests/test_kkt_gpu.py, and the inertia verdict (Sylvester) must be the same -- including on a matrix with the wrong inertia.
tests/test_kkt_gpu.py::test_time_partitioned_factorisation_matches_dense_solve (2 instances, 2..16 chunks) and
tests/test_baseline_sizes_gpu.py (T = 1000) run through it as well, by default."""
import os

import numpy as np
import pytest

from conftest import product_solver

pytestmark = pytest.mark.gpu


def _step(s, Z, MU, dw, dc, partitions, cr):
    import torch
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    B = Z.shape[0]
    dz, dmu = torch.tensor(Z, device="cuda"), torch.tensor(MU, device="cuda")
    dx = torch.full((B, nz), float("nan"), device="cuda", dtype=torch.float64)
    dl = torch.full((B, nc), float("nan"), device="cuda", dtype=torch.float64)
    old = os.environ.get("DTO_SEP_CR")
    os.environ["DTO_SEP_CR"] = "1" if cr else "0"
    s.set_partitions(partitions)
    try:
        ok = s.kkt_step_batch(dz.data_ptr(), B, nz, dmu.data_ptr(), nc, dw, dc, dx.data_ptr(), nz, dl.data_ptr(), nc)
        assert s.partitions() == partitions
    finally:
        s.set_partitions(0)
        if old is None:
            del os.environ["DTO_SEP_CR"]
        else:
            os.environ["DTO_SEP_CR"] = old
    torch.cuda.synchronize()
    return ok, dx.cpu().numpy(), dl.cpu().numpy()


# separators: 1, 2, 5 (not a power of two minus one), 31, 63 (six full levels); bounds (cartpole), inequality rows (car)
@pytest.mark.parametrize("model,T,dw,partitions,B", [("acrobot", 101, 60.0, 2, 1), ("acrobot", 101, 60.0, 3, 2), ("car", 51, 10.0, 6, 1),
                                                     ("cartpole", 200, 400.0, 25, 3), ("acrobot", 1000, 60.0, 32, 1),
                                                     ("acrobot", 1000, 60.0, 64, 1), ("acrobot", 1000, 60.0, 64, 4),
                                                     ("pendulum", 50, 30.0, 6, 4)])
def test_cyclic_reduction_matches_the_sequential_elimination_and_the_dense_solve(model, T, dw, partitions, B):
    from test_kkt_gpu import dense_kkt_solve
    from test_baseline_sizes_gpu import oracle_for
    s, _ = product_solver(model, T)
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    rng = np.random.default_rng(11 * T + partitions + B)
    Z, MU = rng.random((B, nz)), rng.random((B, nc))
    ok_s, dx_s, dl_s = _step(s, Z, MU, dw, 1e-5, partitions, cr=False)
    ok_c, dx_c, dl_c = _step(s, Z, MU, dw, 1e-5, partitions, cr=True)
    assert ok_s and ok_c
    scale = max(np.max(np.abs(dx_s)), np.max(np.abs(dl_s)))
    # two elimination orders of one positive definite separator matrix: rounding apart, not bit-identical
    # (one or two separators are eliminated in the same order both ways: bit-identical there)
    assert np.max(np.abs(dx_c - dx_s)) <= 1e-9 * scale and np.max(np.abs(dl_c - dl_s)) <= 1e-9 * scale
    assert partitions <= 3 or np.max(np.abs(dx_c - dx_s)) > 0, "the switch did nothing: both runs took the same elimination"
    if T <= 200:
        onlp = oracle_for(model, T)
        for b in range(B):
            rx, rl, inertia, cond = dense_kkt_solve(onlp, Z[b], MU[b], dw, 1e-5)
            assert inertia == (nz, nc)
            sc = max(np.max(np.abs(rx)), np.max(np.abs(rl)))
            assert np.max(np.abs(dx_c[b] - rx)) <= 1e-8 * sc and np.max(np.abs(dl_c[b] - rl)) <= 1e-8 * sc


@pytest.mark.parametrize("model,T,dw,partitions,B", [("acrobot", 101, 60.0, 16, 70), ("acrobot", 1000, 60.0, 32, 5),
                                                     ("acrobot", 1000, 60.0, 64, 130), ("cartpole", 200, 400.0, 25, 64)])
def test_one_wavefront_per_instance_for_larger_batches_with_many_chunks(model, T, dw, partitions, B):
    """More than four instances and at least 16 chunks: the same cyclic reduction on one wavefront per (tile, lane)
    (k_kkt_sep_cr) instead of the lane-per-instance elimination -- same bars as above."""
    from test_kkt_gpu import dense_kkt_solve
    from test_baseline_sizes_gpu import oracle_for
    s, _ = product_solver(model, T)
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    rng = np.random.default_rng(5 * T + partitions + B)
    Z, MU = rng.random((B, nz)), rng.random((B, nc))
    ok_s, dx_s, dl_s = _step(s, Z, MU, dw, 1e-5, partitions, cr=False)
    ok_c, dx_c, dl_c = _step(s, Z, MU, dw, 1e-5, partitions, cr=True)
    assert ok_s and ok_c
    scale = max(np.max(np.abs(dx_s)), np.max(np.abs(dl_s)))
    assert 0 < np.max(np.abs(dx_c - dx_s)) <= 1e-9 * scale and np.max(np.abs(dl_c - dl_s)) <= 1e-9 * scale
    if T <= 200:
        onlp = oracle_for(model, T)
        for b in (0, B // 2, B - 1):
            rx, rl, inertia, cond = dense_kkt_solve(onlp, Z[b], MU[b], dw, 1e-5)
            assert inertia == (nz, nc)
            sc = max(np.max(np.abs(rx)), np.max(np.abs(rl)))
            assert np.max(np.abs(dx_c[b] - rx)) <= 1e-8 * sc and np.max(np.abs(dl_c[b] - rl)) <= 1e-8 * sc


def test_more_running_instances_than_the_threshold_take_the_sequential_elimination():
    """Five instances in a tile and fewer than 16 chunks: the lane-per-instance form, whatever the switch says -- bit-identical
    results."""
    s, _ = product_solver("acrobot", 101)
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    rng = np.random.default_rng(3)
    Z, MU = rng.random((5, nz)), rng.random((5, nc))
    ok_s, dx_s, dl_s = _step(s, Z, MU, 60.0, 1e-5, 12, cr=False)
    ok_c, dx_c, dl_c = _step(s, Z, MU, 60.0, 1e-5, 12, cr=True)
    assert ok_s and ok_c and np.array_equal(dx_s, dx_c) and np.array_equal(dl_s, dl_c)


def test_inertia_verdict_agrees_with_the_dense_inertia():
    """The cases of tests/test_kkt_gpu.py::test_inertia_flag_matches_dense_inertia (constraint curvature against delta_w) with the
    chunk count forced: both eliminations must give the verdict of the dense eigenvalue count."""
    from test_kkt_gpu import dense_kkt_solve
    from test_baseline_sizes_gpu import oracle_for
    s, _ = product_solver("pendulum", 50)
    onlp = oracle_for("pendulum", 50)
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    rng = np.random.default_rng(1)
    z = rng.random(nz)
    base = rng.standard_normal(nc)
    seen = set()
    for scale, dw in [(0.0, 1e-3), (0.01, 1e-3), (1.0, 0.0), (30.0, 0.0), (300.0, 0.0), (300.0, 1e4)]:
        mu = scale * base
        _, _, inertia, _ = dense_kkt_solve(onlp, z, mu, dw, 1e-8)
        want = inertia == (nz, nc)
        for parts in (3, 6):
            ok_s, _, _ = _step(s, z[None, :], mu[None, :], dw, 1e-8, parts, cr=False)
            ok_c, _, _ = _step(s, z[None, :], mu[None, :], dw, 1e-8, parts, cr=True)
            assert bool(ok_s) == bool(ok_c) == want, (scale, dw, parts, ok_s, ok_c, inertia)
        seen.add(want)
    assert seen == {True, False}


@pytest.mark.parametrize("model,T", [("acrobot", 101), ("cartpole", 101), ("car", 51)])
def test_full_solve_of_one_instance(model, T):
    """One solve of the reference example as written, with and without the cyclic reduction: converged, the same minimiser,
    iteration counts within two of each other (the two steps differ by rounding)."""
    import dto_amd
    s, p = product_solver(model, T, evaluate_hessian=True)
    res = {}
    for cr in (False, True):
        os.environ["DTO_SEP_CR"] = "1" if cr else "0"
        try:
            xs, us = p["guess"](np.random.Generator(np.random.PCG64(0)))
            dto_amd.initialize_states(s, xs)
            dto_amd.initialize_controls(s, us)
            st = dto_amd.solve(s)
            x_sol, u_sol = dto_amd.get_trajectory(s)
            res[cr] = (st, s.iterations, np.array(x_sol))
        finally:
            del os.environ["DTO_SEP_CR"]
    assert res[False][0] == 1 and res[True][0] == 1
    assert abs(res[False][1] - res[True][1]) <= 2, (res[False][1], res[True][1])
    assert np.max(np.abs(res[False][2] - res[True][2])) <= 1e-5 * max(1.0, np.max(np.abs(res[False][2])))
