"""The last 48 iterations of the acrobot T = 1000 instances that end at the iteration limit: dual infeasibility, constraint
violation, objective, step length and delta_w per iteration (dto_solver_stats after every dto_solver_iterate(1)).
python tools/straggler_tail_trace.py [instances] -- GPU; companion of tools/straggler_exit_tests.py."""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T, TAIL = 1000, 48
p = P.build_acrobot(T=T, evaluate_hessian=True)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot")
nz, nc = s.nlp.num_variables, s.nlp.num_constraint
z0 = make_guesses_device(s, p, B, 1000, "cuda")
s.options.max_iter = 1000
s.begin_batch(z0.data_ptr(), B, nz)
s.iterate_batch(1000 - TAIL)
rows = []
for k in range(TAIL):
    s.iterate_batch(1)
    st = s.stats_batch()
    rows.append({k2: np.array(v).copy() for k2, v in st.items()})
run = np.flatnonzero(rows[-1]["status"] == 0)
E = np.stack([np.maximum(r["dual_inf"][run], r["constr_viol"][run]) for r in rows])      # [TAIL, n]  (s_d = 1 on these: tools/straggler_exit_tests.py)
F = np.stack([r["objective"][run] for r in rows])
A = np.stack([r["alpha"][run] for r in rows])
DW = np.stack([r["delta_w"][run] for r in rows])
Emin, Emax = E.min(axis=0), E.max(axis=0)
out = dict(instances=B, still_running_at_1000=int(len(run)), tail_iterations=TAIL,
           min_error_over_tail=dict(le_1e_6=int(np.sum(Emin <= 1e-6)), le_2e_6=int(np.sum(Emin <= 2e-6)), le_1e_5=int(np.sum(Emin <= 1e-5)), le_1e_4=int(np.sum(Emin <= 1e-4))),
           max_error_over_tail=dict(le_1e_5=int(np.sum(Emax <= 1e-5)), le_1e_4=int(np.sum(Emax <= 1e-4)), le_1e_3=int(np.sum(Emax <= 1e-3))),
           smallest_step_fraction=float(np.mean(A <= 0.0078125)), null_step_fraction=float(np.mean(A == 0.0)),
           no_step_in_the_whole_tail=int(np.sum(np.all(A == 0.0, axis=0))), full_step_fraction=float(np.mean(A >= 1.0)),
           objective_drop_over_tail=dict(median=float(np.median(F[0] - F[-1])), p10=float(np.quantile(F[0] - F[-1], 0.1)), p90=float(np.quantile(F[0] - F[-1], 0.9))),
           examples=[])
order = np.argsort(Emin)
for j in list(order[:4]) + list(order[len(order) // 2: len(order) // 2 + 3]) + list(order[-2:]):
    out["examples"].append(dict(instance=int(run[j]), error=[float(f"{v:.3e}") for v in E[:, j]], alpha=[float(v) for v in A[:, j]],
                                objective_first_last=[float(F[0, j]), float(F[-1, j])], delta_w=[float(v) for v in DW[-8:, j]]))
print(json.dumps(out))
