"""Iteration trace of the instances of the T=101 side measurement that run into the iteration limit (bench.py:
full_solve_measurement): python tools/trace_stragglers.py [every]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dto_amd
from dto_amd import problems as P
from bench import make_guesses
T, B = 101, 1024
every = int(sys.argv[1]) if len(sys.argv) > 1 else 50
p = P.build_acrobot(T=T, evaluate_hessian=True)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot")
nz = s.nlp.num_variables
z0 = torch.tensor(make_guesses(s, p, B, seed=1000), device="cuda")
zo = torch.empty_like(z0)
status, iters = s.solve_batch(z0.data_ptr(), B, nz, zo.data_ptr(), nz)
bad = [int(i) for i in np.nonzero(status != 1)[0]]
print("not converged:", bad, [int(status[i]) for i in bad])
names = ["iter", "f", "theta_inf", "dinf", "compl", "mu", "delta_w", "gamma", "alpha", "alpha_pmax", "ls_fail", "ls_kind", "nfact", "filter_n", "status"]
s.begin_batch(z0.data_ptr(), B, nz)
done = 0
print(" ".join(f"{n:>10s}" for n in ["inst"] + names))
while done < 1000:
    s.iterate_batch(every)
    done += every
    for i in bad[:2]:
        v = [float(s.scalar_batch(n)[i]) for n in names]
        print(f"{i:10d} " + " ".join(f"{x:10.3e}" if abs(x) > 1e4 or (x != 0 and abs(x) < 1e-2) else f"{x:10.4f}" for x in v))
