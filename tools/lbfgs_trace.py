"""Where an iteration of the limited-memory mode goes, launch by launch (dto_solver_trace): acrobot T = 101, a batch of one /
64 / 1024, iterations 20 - 39.   python tools/lbfgs_trace.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dto_amd
from dto_amd import problems as P
from bench import make_guesses

OPN = {19: "qn_begin", 20: "qn_rhs", 21: "qn_col", 22: "qn_small", 23: "qn_save", 24: "qn_cols_rhs"}
for T, B in ((101, 1), (101, 64), (101, 1024), (101, 4096)):
    for mode in ("lbfgs", "exact"):
        p = P.build_acrobot(T=T, evaluate_hessian=True)
        s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot",
                           options=dto_amd.Options(hessian_approximation=mode, tol=1e-30, max_iter=100000))
        nz = s.nlp.num_variables
        z0 = torch.tensor(make_guesses(s, p, B, seed=1000), device="cuda")
        s.begin_batch(z0.data_ptr(), B, nz)
        s.iterate_batch(20)
        torch.cuda.synchronize()
        s.trace(True)
        t0 = time.perf_counter()
        s.iterate_batch(20)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 20 * 1e3
        s.trace(False)
        tr = s.read_trace()
        names = [OPN.get(int(o), n) for o, n in zip(tr["op"], tr["name"])]
        tot = {}
        cnt = {}
        for n, d in zip(names, tr["duration_ms"]):
            tot[n] = tot.get(n, 0.0) + d; cnt[n] = cnt.get(n, 0) + 1
        span = (tr["start_ms"] + tr["duration_ms"]).max() / 20
        print(f"T={T} B={B} {mode}: wall {wall:.3f} ms/iteration, traced span {span:.3f}, launches/iteration {len(names) / 20:.1f}")
        for n in sorted(tot, key=tot.get, reverse=True):
            print(f"    {n:14s} {tot[n] / 20:8.4f} ms/iteration in {cnt[n] / 20:5.1f} launches")
        s.close()
