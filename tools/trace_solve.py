"""Iteration trace of one instance (debug aid): python tools/trace_solve.py cartpole 200 [iters] [every]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dto_amd
from dto_amd import problems as P

model, T = sys.argv[1], int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 300
every = int(sys.argv[4]) if len(sys.argv) > 4 else 10
p = getattr(P, f"build_{model}")(T=T, evaluate_hessian=True)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name=model)
s.options.max_iter = 100000
xs, us = p["guess"](np.random.Generator(np.random.PCG64(0)))
dto_amd.initialize_states(s, xs); dto_amd.initialize_controls(s, us)
nz = s.nlp.num_variables
z0 = torch.tensor(np.asarray(s._z0)[None, :].copy(), device="cuda")
s.begin_batch(z0.data_ptr(), 1, nz)
names = ["iter", "f", "theta_inf", "dinf", "compl", "mu", "delta_w", "gamma", "alpha", "alpha_pmax", "ls_fail", "ls_kind", "nfact", "filter_n", "status"]
print(" ".join(f"{n:>10s}" for n in names))
done = 0
while done < iters:
    s.iterate_batch(every)
    done += every
    v = [float(s.scalar_batch(n)[0]) for n in names]
    print(" ".join(f"{x:10.3e}" if abs(x) > 1e4 or (x != 0 and abs(x) < 1e-2) else f"{x:10.4f}" for x in v))
    if v[-1] != 0:
        break
