"""GPU solver vs the independent serial C port of the same algorithm (oracle/cpu_port): iterate-level agreement.

Both implement the iteration documented in DESIGN.md (same constants).  They are written independently
(HIP templates over generated device code vs plain C over sympy-generated code), so agreement of the
iteration history pins the GPU path far more tightly than the end-point checks of test_solve_gpu.py.
Tolerance: objective / violation histories agree to 1e-6 relative while both are in the same branch of the
line search; the final points agree to 1e-7 (Newton's quadratic tail erases rounding differences).
"""
import numpy as np
import pytest

from conftest import product_solver

pytestmark = pytest.mark.gpu


def gpu_history(model, T, z0, iters):
    import torch
    s, p = product_solver(model, T)
    nz = s.nlp.num_variables
    d = torch.tensor(z0[None, :], device="cuda")
    s.begin_batch(d.data_ptr(), 1, nz)
    hist = []
    for _ in range(iters):
        s.iterate_batch(1)
        st = s.stats_batch()
        hist.append((int(st["iterations"][0]), st["objective"][0], st["constr_viol"][0], st["alpha"][0], st["delta_w"][0],
                     int(st["status"][0])))
        if st["status"][0] != 0:
            break
    out = torch.zeros((1, nz), device="cuda", dtype=torch.float64)
    lam = torch.zeros((1, s.nlp.num_constraint), device="cuda", dtype=torch.float64)
    s.end_batch(out.data_ptr(), nz, lam.data_ptr(), s.nlp.num_constraint)
    torch.cuda.synchronize()
    return hist, out.cpu().numpy()[0], lam.cpu().numpy()[0]


def port_history(model, T, x1, xT, z0, iters):
    from oracle.cpu_port import PortSolver
    s = PortSolver(model, T, x1, xT)
    s.begin(z0)
    hist = []
    for _ in range(iters):
        ran = s.iterate()
        st = s.stats()
        hist.append((s.iterations, st["objective"], st["constr_viol"], st["alpha"], st["delta_w"], s.status))
        if not ran:
            break
    return hist, s.z, s.lam


@pytest.mark.parametrize("model,T,seed", [("pendulum", 50, 0), ("pendulum", 50, 3), ("acrobot", 101, 7), ("acrobot", 101, 1)])
def test_iteration_history_matches_cpu_port(model, T, seed):
    import dto_amd
    s, p = product_solver(model, T)
    rng = np.random.Generator(np.random.PCG64(seed))
    xs, us = p["guess"](rng)
    dto_amd.initialize_states(s, xs)
    dto_amd.initialize_controls(s, us)
    z0 = s._z0.copy()
    gh, gz, gl = gpu_history(model, T, z0, 400)
    ph, pz, pl = port_history(model, T, p["x1"], p["xT"], z0, 400)
    assert gh[-1][5] == 1 and ph[-1][5] == 1
    # identical decisions (step size, regularisation) and objective history while rounding has not yet
    # tipped a borderline inertia/filter decision: the whole run for the mildly nonconvex pendulum, at
    # least the first 8 iterations of the long nonconvex acrobot path (the watchdog now fires after two shortened steps, so
    # the first borderline filter decision comes early)
    need = len(ph) if model == "pendulum" else 8
    agree = 0
    for a, b in zip(gh, ph):
        same = (a[0] == b[0] and abs(a[1] - b[1]) <= 1e-6 * max(1.0, abs(b[1])) and a[3] == b[3]
                and abs(a[4] - b[4]) <= 1e-12 * max(1.0, b[4]))
        if not same:
            break
        agree += 1
    assert agree >= need, (agree, gh[:agree + 1][-1], ph[:agree + 1][-1])
    if model == "pendulum":
        assert gh[-1][0] == ph[-1][0]
    # same local solution and multipliers (GPU multipliers are in the reference order: dynamics rows, then
    # stage rows = first pin, last pin -- the port uses the same order).  The acrobot swing-up has several
    # local minima (f* = 309.8, 350.1, 423.6, 567.2, ...); once rounding has tipped a decision the two
    # implementations may legitimately end in different ones, so the end points are compared only when the
    # objectives coincide (both are KKT points either way: test_solve_gpu.py / test_cpu_port.py).
    if model == "pendulum" or abs(gh[-1][1] - ph[-1][1]) <= 1e-8 * abs(ph[-1][1]):
        assert np.max(np.abs(gz - pz)) <= 1e-6 * max(1.0, np.max(np.abs(pz)))
        assert np.max(np.abs(gl - pl)) <= 1e-5 * max(1.0, np.max(np.abs(pl)))
