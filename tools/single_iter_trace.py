"""A batch of ONE under the kernel tracer: which kernels one iteration of a single instance spends its time in.

    rocprofv3 --kernel-trace --stats -d gpurun_out/lat -- python3 tools/single_iter_trace.py acrobot 1000 [iterations] [partitions]

Prints wall time per iteration; the per-kernel averages come from the tracer's stats file."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dto_amd
from dto_amd import problems as P

model, T = sys.argv[1], int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 40
parts = int(sys.argv[4]) if len(sys.argv) > 4 else 0
p = getattr(P, f"build_{model}")(T=T, evaluate_hessian=True)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name=model)
xs, us = p["guess"](np.random.Generator(np.random.PCG64(0)))
dto_amd.initialize_states(s, xs); dto_amd.initialize_controls(s, us)
nz = s._solve_nlp.num_variables
z0 = torch.tensor(s._z0[None, :], device="cuda")
if parts:
    s.set_partitions(parts)
s.begin_batch(z0.data_ptr(), 1, nz)
s.iterate_batch(5)
torch.cuda.synchronize()
nf0 = float(s.scalar_batch("nfact")[0])
t0 = time.perf_counter()
s.iterate_batch(iters)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
nf = float(s.scalar_batch("nfact")[0]) - nf0
print(f"{model} T={T} B=1 partitions={s.partitions()}: {1e3 * dt / iters:.3f} ms per iteration over {iters} iterations, "
      f"{nf / iters:.2f} factorisations per iteration", flush=True)
