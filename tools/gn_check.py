"""Iterations of the quasi-Newton (per-stage SR1) mode on the four models: python tools/gn_check.py"""
import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import dto_amd
from dto_amd import problems as P
for model, T in (("pendulum", 50), ("car", 51), ("cartpole", 51), ("acrobot", 101)):
    p = getattr(P, f"build_{model}")(T=T, evaluate_hessian=False)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=False, name=model,
                       options=dto_amd.Options(hessian_approximation="sr1"))   # the quasi-Newton mode, not the default
    s.options.max_iter = 3000
    xs, us = p["guess"](np.random.Generator(np.random.PCG64(0)))
    dto_amd.initialize_states(s, xs); dto_amd.initialize_controls(s, us)
    st = dto_amd.solve(s)
    print(model, T, "status", st, "iters", s.iterations)
