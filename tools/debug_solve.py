"""Iteration log of the GPU solver for one config (debug aid): python tools/debug_solve.py pendulum 50 [B] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dto_amd
from dto_amd import problems as P

model = sys.argv[1] if len(sys.argv) > 1 else "pendulum"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 60
kw = dict(endpoint="bounds") if model == "acrobot_bounds" else {}
EH = os.environ.get("DBG_EH", "1") == "1"
p = getattr(P, "build_acrobot" if model.startswith("acrobot") else f"build_{model}")(T=T, evaluate_hessian=EH, **kw)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=EH, name=model)
n = s.nlp
nz = n.num_variables
Z = np.zeros((B, nz))
for b in range(B):
    rng = np.random.Generator(np.random.PCG64(b))
    xs, us = p["guess"](rng)
    dto_amd.initialize_states(s, xs)
    dto_amd.initialize_controls(s, us)
    Z[b] = s._z0
dz = torch.tensor(Z, device="cuda")
s.begin_batch(dz.data_ptr(), B, nz)
print("it  status  f            viol        dinf        mu        dw         alpha   | filter_n theta1 dphi apmax lsfail nfact kind gamma")
for it in range(iters):
    s.iterate_batch(1)
    st = s.stats_batch()
    b = int(os.environ.get('DBG_INST', '0'))
    print(f"{it:3d} {st['status'][b]:3d} {st['iterations'][b]:4d} {st['objective'][b]:12.5e} {st['constr_viol'][b]:10.3e} "
          f"{st['dual_inf'][b]:10.3e} {st['mu'][b]:9.2e} {st['delta_w'][b]:9.2e} {st['alpha'][b]:9.3e}   | "
          f"{s.scalar_batch('filter_n')[b]:4.0f} {s.scalar_batch('theta1')[b]:9.2e} {s.scalar_batch('dmerit')[b]:10.2e} {s.scalar_batch('alpha_pmax')[b]:8.2e} "
          f"{s.scalar_batch('ls_fail')[b]:.0f} {s.scalar_batch('nfact')[b]:.0f} {s.scalar_batch('ls_kind')[b]:.0f} {s.scalar_batch('gamma')[b]:.0f} run {int(np.sum(st['status'] == 0))}")
    if np.all(st["status"] != 0):
        break
out = torch.zeros((B, nz), device="cuda", dtype=torch.float64)
s.end_batch(out.data_ptr(), nz)
torch.cuda.synchronize()
z = out.cpu().numpy()
idx = n.indices
print("x1 =", z[0][np.array(idx.states[0]) - 1], " xT =", z[0][np.array(idx.states[-1]) - 1])
print("status", st["status"][:16], "iters", st["iterations"][:16])
for k in ("objective", "constr_viol", "dual_inf", "delta_w"):
    print(k, np.array2string(st[k][:16], precision=3))
print("gamma", s.scalar_batch("gamma")[:16], "nfact", s.scalar_batch("nfact")[:16])
