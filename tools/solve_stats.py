"""Time-to-solution statistics of a batched solve: python tools/solve_stats.py acrobot 1000 256 [max_iter]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dto_amd
from dto_amd import problems as P
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_guesses

model, T, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
max_iter = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
p = getattr(P, f"build_{model}")(T=T, evaluate_hessian=True)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name=model)
s.options.max_iter = max_iter
if os.environ.get("DTO_PART"):
    s.set_partitions(int(os.environ["DTO_PART"]))
nz, nc = s.nlp.num_variables, s.nlp.num_constraint
if model == "acrobot":
    Z = make_guesses(s, p, B, seed=1000)
else:
    Z = np.zeros((B, nz))
    for b in range(B):
        xs, us = p["guess"](np.random.Generator(np.random.PCG64(b)))
        dto_amd.initialize_states(s, xs); dto_amd.initialize_controls(s, us)
        Z[b] = s._z0
z0 = torch.tensor(Z, device="cuda")
zo = torch.empty_like(z0)
torch.cuda.synchronize()
t0 = time.perf_counter()
status, iters = s.solve_batch(z0.data_ptr(), B, nz, zo.data_ptr(), nz, check_every=20)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
nf = s.scalar_batch("nfact")
f = s.scalar_batch("f")
print(json.dumps(dict(model=model, T=T, B=B, seconds=round(dt, 3), converged=int(np.sum(status == 1)), max_iter_hit=int(np.sum(status == 2)),
                      failed=int(np.sum(status == 3)), iters_median=float(np.median(iters)), iters_mean=float(np.mean(iters)),
                      iters_max=int(np.max(iters)), total_iters=int(np.sum(iters)), factorizations_per_iter=float(np.sum(nf) / max(1, np.sum(iters))),
                      solves_per_sec=round(float(np.sum(status == 1)) / dt, 2), partitions=s.partitions(),
                      objective_quartiles=[float(q) for q in np.percentile(f[status == 1], [0, 25, 50, 75, 100])] if np.any(status == 1) else None)))
