"""Convergence of the quasi-Newton modes against exact Hessians on the reference configs (problems built with
evaluate_hessian=false, the reference default):  python tools/qn_stats.py [n_seeds]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dto_amd
from dto_amd import problems as P

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for model, T in (("pendulum", 50), ("acrobot", 101), ("cartpole", 200), ("car", 51), ("acrobot", 301)):
    # "lbfgs" = the default of evaluate_hessian=false since round 5 (compact L-BFGS, dto_options.hessian_approximation); "exact" =
    # second derivatives of the traced expressions; "sr1" = per-stage SR1 blocks (a partitioned damped BFGS was tried in round 4
    # and never converged: tools/experiments/)
    import time
    for mode in ("lbfgs", "exact", "sr1"):
        p = getattr(P, f"build_{model}")(T=T, evaluate_hessian=False)
        try:
            s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=False,
                               options=dto_amd.Options(hessian_approximation=mode), name=model)
        except Exception as e:
            print(json.dumps(dict(model=model, T=T, mode=mode, error=str(e)[:100])))
            continue
        nz = s._solve_nlp.num_variables
        Z = np.zeros((B, nz))
        for b in range(B):
            xs, us = p["guess"](np.random.Generator(np.random.PCG64(b)))
            dto_amd.initialize_states(s, xs); dto_amd.initialize_controls(s, us)
            Z[b] = s._z0
        z0 = torch.tensor(Z, device="cuda"); zo = torch.empty_like(z0)
        t0 = time.perf_counter()
        st, it = s.solve_batch(z0.data_ptr(), B, nz, zo.data_ptr(), nz)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        f = []
        for b in range(min(B, 8)):
            f.append(round(float(s.nlp.eval_objective(zo[b].cpu().numpy())), 3))
        print(json.dumps(dict(model=model, T=T, mode=mode, converged=int(np.sum(st == 1)), n=B, it_median=float(np.median(it)),
                              it_max=int(it.max()), seconds=round(dt, 3), hessian_mode=s.hessian_mode, status=np.bincount(st, minlength=6).tolist(), f=f)), flush=True)
