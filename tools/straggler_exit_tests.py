"""Why Ipopt's acceptable-level exit (src/options.jl:15-20) does not catch the acrobot T = 1000 stragglers.

python tools/straggler_exit_tests.py [instances]   -- GPU.  The bench's seeded guesses, reference Options.  Three solves of the same
batch: max_iter = 990 and 1000 (objective change per iteration at the end), and acceptable_tol = 1e-4 instead of the reference's
1e-6 (= tol: with the reference defaults the acceptable test IS the convergence test).  For every instance that ends at the iteration
limit: the five acceptable-level tests of Ipopt (scaled NLP error <= acceptable_tol; dual infeasibility <= 1e10; constraint violation
<= 1e-2; complementarity <= 1e-2; relative objective change <= 1e-5) evaluated at its last iterate, from dto_solver_stats.
Writes one JSON object (profiles/r06/straggler_exit_tests_T1000.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import dto_amd
from dto_amd import problems as P
from bench import make_guesses_device


def run(s, z0, nz, nc, **opt):
    for k, v in opt.items():
        setattr(s.options, k, v)
    B = z0.shape[0]
    zo = torch.empty_like(z0)
    mo = torch.empty((B, nc), device="cuda", dtype=torch.float64)
    s.begin_batch(z0.data_ptr(), B, nz)
    status, iters = s.run_batch(zo.data_ptr(), nz, mo.data_ptr(), nc)
    torch.cuda.synchronize()
    st = s.stats_batch()
    lam1 = mo.abs().sum(dim=1).cpu().numpy()
    s.release_state()
    return status.copy(), iters.copy(), {k: np.array(v).copy() for k, v in st.items()}, lam1


def q(a):
    a = np.asarray(a, dtype=float)
    return {k: float(np.quantile(a, p)) for k, p in (("min", 0.0), ("p10", 0.1), ("median", 0.5), ("p90", 0.9), ("max", 1.0))} if len(a) else None


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    T = 1000
    p = P.build_acrobot(T=T, evaluate_hessian=True)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot")
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    z0 = make_guesses_device(s, p, B, 1000, "cuda")
    sa, ia, A, _ = run(s, z0, nz, nc, max_iter=990)
    sb, ib, Bst, lam1 = run(s, z0, nz, nc, max_iter=1000)
    strag = np.flatnonzero(sb == 2)
    s_d = np.maximum(100.0, lam1 / nc) / 100.0               # Ipopt's s_d with s_max = 100 (no bound multipliers in this problem)
    E = np.maximum(Bst["dual_inf"] / s_d, Bst["constr_viol"])  # scaled NLP error (no inequality rows / bounds: no complementarity term)
    dfrel = np.abs(Bst["objective"] - A["objective"]) / 10.0 / np.maximum(1.0, np.abs(Bst["objective"]))
    out = dict(workload=f"acrobot T={T}, {B} instances, bench seeds, reference Options", instances=B,
               converged=int(np.sum(sb == 1)), at_iteration_limit=int(len(strag)),
               stragglers=dict(objective=q(Bst["objective"][strag]), constr_viol=q(Bst["constr_viol"][strag]), dual_inf=q(Bst["dual_inf"][strag]),
                               s_d=q(s_d[strag]), scaled_error=q(E[strag]), mu=q(Bst["mu"][strag]),
                               objective_change_per_iteration_relative=q(dfrel[strag]), delta_w=q(Bst["delta_w"][strag]), alpha=q(Bst["alpha"][strag])),
               acceptable_tests_failed=dict(
                   scaled_error_le_acceptable_tol_1e_6=int(np.sum(E[strag] > 1e-6)),
                   dual_inf_le_1e10=int(np.sum(Bst["dual_inf"][strag] > 1e10)),
                   constr_viol_le_1e_2=int(np.sum(Bst["constr_viol"][strag] > 1e-2)),
                   compl_le_1e_2=0,
                   objective_change_le_1e_5=int(np.sum(dfrel[strag] > 1e-5))),
               would_pass_first_test_at=dict((f"acceptable_tol_{t:g}", int(np.sum(E[strag] <= t))) for t in (1e-5, 1e-4, 1e-3)))
    sc, ic, Cst, _ = run(s, z0, nz, nc, max_iter=1000, acceptable_tol=1e-4)
    out["with_acceptable_tol_1e_4"] = dict(converged=int(np.sum(sc == 1)), acceptable_exit=int(np.sum(sc == 4)), at_iteration_limit=int(np.sum(sc == 2)),
                                           other=int(np.sum((sc != 1) & (sc != 2) & (sc != 4))),
                                           iterations_of_acceptable_exits=q(ic[sc == 4]),
                                           objective_of_acceptable_exits=q(Cst["objective"][sc == 4]),
                                           objective_of_converged=q(Cst["objective"][sc == 1]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
