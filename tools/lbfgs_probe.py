"""One limited-memory solve of a reference config from its example guess: python tools/lbfgs_probe.py cartpole 200 [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dto_amd
from dto_amd import problems as P
model, T = sys.argv[1], int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
p = getattr(P, f"build_{model}")(T=T, evaluate_hessian=False)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=False, name=model)
nz = s._solve_nlp.num_variables
Z = np.zeros((B, nz))
for b in range(B):
    xs, us = p["guess"](np.random.Generator(np.random.PCG64(b)))
    dto_amd.initialize_states(s, xs); dto_amd.initialize_controls(s, us)
    Z[b] = s._z0
z0 = torch.tensor(Z, device="cuda"); zo = torch.empty_like(z0)
st, it = s.solve_batch(z0.data_ptr(), B, nz, zo.data_ptr(), nz)
torch.cuda.synchronize()
f = [round(float(s.nlp.eval_objective(zo[b].cpu().numpy())), 4) for b in range(min(B, 4))]
print(f"{model} T={T} B={B} mode={s.hessian_mode} env={ {k: v for k, v in os.environ.items() if k.startswith('DTO_')} }: "
      f"status {np.bincount(st, minlength=7).tolist()} iterations median {np.median(it):.0f} max {it.max()} f {f}", flush=True)
