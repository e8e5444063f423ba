"""Wall time of ONE solve (batch of 1) of the reference's examples -- the latency a drop-in user sees: python tools/single_solve_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dto_amd
from dto_amd import problems as P
for model, T, eh in (("pendulum", 11, True), ("pendulum", 50, True), ("cartpole", 101, False), ("acrobot", 101, False), ("car", 51, False),
                     ("acrobot", 101, True), ("acrobot", 1000, True)):   # the last one: ONE instance of the bench workload
    p = getattr(P, f"build_{model}")(T=T, evaluate_hessian=eh)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=eh, name=model)
    s.options.print_level = 0
    ts = []
    for rep in range(4):
        xs, us = p["guess"](np.random.Generator(np.random.PCG64(0)))
        dto_amd.initialize_states(s, xs); dto_amd.initialize_controls(s, us)
        t0 = time.perf_counter()
        st = dto_amd.solve(s)
        ts.append(time.perf_counter() - t0)
    print(f"{model} T={T} eh={eh}: status {st}, {s.iterations} iterations, {1e3*min(ts[1:]):.2f} ms per solve "
          f"({1e3*min(ts[1:])/max(s.iterations,1):.3f} ms per iteration)", flush=True)
