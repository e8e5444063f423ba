"""Cost of one bordered KKT step (multi-knot GeneralConstraint rows), device border against the host border of round 3:
python tools/border_bench.py [T] [B]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dto_amd
from dto_amd import problems as P
T = int(sys.argv[1]) if len(sys.argv) > 1 else 101
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
p = P.build_acrobot_coupled(T=T)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True,
                   general_constraint=p["general_constraint"], options=dto_amd.Options(general_rows="border"), name="acrobot_coupled")
nz, nc = s.nlp.num_variables, s.nlp.num_constraint
rng = np.random.default_rng(0)
z = torch.tensor(0.5 * rng.standard_normal((B, nz)), device="cuda"); mu = torch.tensor(rng.standard_normal((B, nc)), device="cuda")
dx = torch.empty((B, nz), device="cuda", dtype=torch.float64); dl = torch.empty((B, nc), device="cuda", dtype=torch.float64)
res = {}
for mode in ("1", "0"):
    os.environ["DTO_BORDER_HOST"] = mode
    step = lambda: s.kkt_step_batch(z.data_ptr(), B, nz, mu.data_ptr(), nc, 0.8, 1e-6, dx.data_ptr(), nz, dl.data_ptr(), nc)
    step(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    res["host_border_ms" if mode == "1" else "device_border_ms"] = round(min(ts) * 1e3, 2)
print(json.dumps(dict(T=T, B=B, general_rows=2, **res)), flush=True)

# full solves: the bordered problem (host-driven loop around the bordered step) against the same acrobot without the coupling rows
def solve_rate(solver, tag):
    nzs = solver._solve_nlp.num_variables       # the solver's own layout (accumulator states included: pad_batch)
    Z = np.zeros((B, nzs))
    pa = P.build_acrobot(T=T, evaluate_hessian=True)
    for b in range(B):
        xs, us = pa["guess"](np.random.Generator(np.random.PCG64(b)))
        dto_amd.initialize_states(solver, xs); dto_amd.initialize_controls(solver, [0.01 * u for u in us])
        Z[b] = solver.pad_batch(solver._z0)
    z0 = torch.tensor(Z, device="cuda"); zo = torch.empty_like(z0)
    out = {}
    for mode in (("1", "0") if tag == "bordered" else ("0",)):
        os.environ["DTO_BORDER_HOST"] = mode
        t0 = time.perf_counter()
        st, it = solver.solve_batch(z0.data_ptr(), B, nzs, zo.data_ptr(), nzs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[("host_border" if mode == "1" else "device_border") if tag == "bordered" else tag] = dict(
            seconds=round(dt, 3), iterations=int(np.sum(it)), converged=int(np.sum(st == 1)), iterations_per_sec=round(float(np.sum(it)) / dt, 1))
    return out
s.options.max_iter = 60
r = solve_rate(s, "bordered")
pa = P.build_acrobot(T=T, evaluate_hessian=True)
sp = dto_amd.Solver(pa["dynamics"], pa["objective"], pa["constraints"], pa["bounds"], evaluate_hessian=True, name="acrobot",
                    options=dto_amd.Options(max_iter=60))
r.update(solve_rate(sp, "plain"))
# round 6: the same coupling rows carried by accumulator states through the ordinary device loop (solver.py: accumulate_general_constraint)
sa = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True,
                    general_constraint=p["general_constraint"], options=dto_amd.Options(max_iter=60), name="acrobot_coupled")
assert sa.general_rows_path == "accumulators"
r.update(solve_rate(sa, "accumulators"))
print(json.dumps(dict(T=T, B=B, max_iter=60, **r)))
