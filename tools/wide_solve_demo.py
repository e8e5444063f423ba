"""Full solve of the 64-state configuration: python tools/wide_solve_demo.py [T] [B] [terminal] [max_iter]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dto_amd
from dto_amd import problems as P
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
terminal = sys.argv[3] if len(sys.argv) > 3 else "full"
max_iter = int(sys.argv[4]) if len(sys.argv) > 4 else 150
p = P.build_acrobot_padded(T=T, terminal=terminal)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot_padded",
                   options=dto_amd.Options(max_iter=max_iter))
nz = s.nlp.num_variables
Z = np.zeros((B, nz))
for b in range(B):
    xs, us = p["guess"](np.random.Generator(np.random.PCG64(b)))
    dto_amd.initialize_states(s, xs); dto_amd.initialize_controls(s, us)
    Z[b] = s._z0
z0 = torch.tensor(Z, device="cuda"); zo = torch.empty_like(z0)
t0 = time.perf_counter()
status, iters = s.solve_batch(z0.data_ptr(), B, nz, zo.data_ptr(), nz)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps(dict(T=T, B=B, terminal=terminal, status=status.tolist(), iterations=iters.tolist(), seconds=round(dt, 2),
                      iterations_per_sec=round(float(np.sum(iters)) / dt, 2))))
