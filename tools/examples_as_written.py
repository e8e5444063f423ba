"""The reference's four examples as written (default mode where the example uses it): iterations to convergence."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dto_amd
from dto_amd import problems as P
for model, T, eh in (("pendulum", 11, True), ("cartpole", 101, False), ("acrobot", 101, False), ("car", 51, False)):
    p = getattr(P, f"build_{model}")(T=T, evaluate_hessian=eh)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=eh, name=model)
    its = []
    for seed in range(4):
        xs, us = p["guess"](np.random.Generator(np.random.PCG64(seed)))
        dto_amd.initialize_states(s, xs); dto_amd.initialize_controls(s, us)
        st = dto_amd.solve(s)
        its.append((st, s.iterations))
    print(model, T, "evaluate_hessian", eh, its)
