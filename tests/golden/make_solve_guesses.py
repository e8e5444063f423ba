"""Export the initial guesses of the solve comparisons (seeded, the reference uses unseeded randn) so that the reference's
own solve!(solver) can be run from EXACTLY the same starting points (tools/julia_parity_check.jl --solve), together with
what the GPU solver returned from them.  Needs a GPU for the `ours` part:  python tests/golden/make_solve_guesses.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dto_amd  # noqa: E402
from dto_amd import problems as P  # noqa: E402

out = []
for model, T, eh in (("pendulum", 50, True), ("cartpole", 101, False), ("acrobot", 101, False), ("car", 51, False)):
    p = getattr(P, f"build_{model}")(T=T, evaluate_hessian=eh)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=eh, name=model)
    for seed in (0, 1):
        xs, us = p["guess"](np.random.Generator(np.random.PCG64(seed)))
        dto_amd.initialize_states(s, xs)
        dto_amd.initialize_controls(s, us)
        rec = dict(model=model, T=T, evaluate_hessian=eh, seed=seed, states=[list(map(float, x)) for x in xs],
                   actions=[list(map(float, u)) for u in us])
        try:
            st = dto_amd.solve(s)
            rec["ours"] = dict(status=int(st), iterations=int(s.iterations), objective=float(s.nlp.eval_objective(s._solution)))
        except Exception as e:  # no GPU: guesses only
            rec["ours"] = dict(error=str(e)[:80])
        out.append(rec)
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "solve_guesses.json"), "w") as f:
    json.dump(out, f)
print("wrote", len(out), "cases")
