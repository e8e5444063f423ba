"""Compare the GPU pivot-sign test with the dense KKT inertia (oracle) along the solver's iterates."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dto_amd
from dto_amd import problems as P
from oracle import dto_oracle as O, sympy_models as S

model, T = sys.argv[1], int(sys.argv[2])
p = getattr(P, f"build_{model}")(T=T, evaluate_hessian=True)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name=model)
n = s.nlp
op = S.build(model, T, evaluate_hessian=True)
onlp = O.NLPData(op["dynamics"], op["objective"], op["constraints"], op["bounds"], evaluate_hessian=True)
nz, nc = n.num_variables, n.num_constraint
rng = np.random.Generator(np.random.PCG64(0))
xs, us = p["guess"](rng)
dto_amd.initialize_states(s, xs); dto_amd.initialize_controls(s, us)
z0 = torch.tensor(s._z0[None, :], device="cuda")
s.begin_batch(z0.data_ptr(), 1, nz)
zo = torch.zeros((1, nz), device="cuda", dtype=torch.float64); lo = torch.zeros((1, nc), device="cuda", dtype=torch.float64)
dx = torch.zeros((1, nz), device="cuda", dtype=torch.float64); dl = torch.zeros((1, nc), device="cuda", dtype=torch.float64)
for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 12):
    s.iterate_batch(1)
    st = s.stats_batch()
    s.end_batch(zo.data_ptr(), nz, lo.data_ptr(), nc); torch.cuda.synchronize()
    z, lam = zo.cpu().numpy()[0], lo.cpu().numpy()[0]
    H = np.zeros((nz, nz))
    for (r, c), v in zip(onlp.hessian_lagrangian_structure(), onlp.eval_hessian_lagrangian(z, 1.0, lam)): H[r-1, c-1] = v
    J = np.zeros((nc, nz))
    for (r, c), v in zip(onlp.jacobian_structure(), onlp.eval_constraint_jacobian(z)): J[r-1, c-1] = v
    row = []
    for dw in (0.0, 1e-3, 1e-2, 1e-1, 1.0):
        K = np.block([[H + dw*np.eye(nz), J.T], [J, -1e-8*np.eye(nc)]])
        e = np.linalg.eigvalsh(K)
        dense_ok = (int(np.sum(e > 0)), int(np.sum(e < 0))) == (nz, nc)
        row.append((dw, dense_ok))
    # min eigenvalue of the reduced Hessian Z' H Z
    u_, sv, vt = np.linalg.svd(J); Zn = vt[nc:].T
    red = np.linalg.eigvalsh(Zn.T @ H @ Zn)
    print(it, "solver dw=%.2e" % st["delta_w"][0], "viol=%.2e dinf=%.2e" % (st["constr_viol"][0], st["dual_inf"][0]),
          "dense inertia ok:", row, "min eig reduced H = %.3e" % red[0], "min eig full H = %.3e" % np.linalg.eigvalsh(H)[0], "|lam|max=%.2f" % np.max(np.abs(lam)))
    # the same struct continues (begin state is preserved by end_batch)
