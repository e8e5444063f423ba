"""Timing of the wide-stage (dense block, f64 MFMA) KKT step: python tools/wide_bench.py [T] [B] [reps]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dto_amd
from dto_amd import problems as P

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
p = P.build_acrobot_padded(T=T)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot_padded")
nz, nc = s.nlp.num_variables, s.nlp.num_constraint
g = torch.Generator(device="cuda"); g.manual_seed(0)
Z = torch.rand((B, nz), device="cuda", dtype=torch.float64, generator=g)
MU = torch.rand((B, nc), device="cuda", dtype=torch.float64, generator=g)
dx = torch.empty_like(Z); dl = torch.empty_like(MU)
def step():
    return s.kkt_step_batch(Z.data_ptr(), B, nz, MU.data_ptr(), nc, 2.0, 1e-5, dx.data_ptr(), nz, dl.data_ptr(), nc)
ok = step(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
dt = min(ts)
n, m = 64, 1
# algorithmic flops per stage: two 64^3/3 LDL^T, three triangular solves with 64 right-hand sides (64^3 each),
# four 64x64x64 products (2*64^3 each; the symmetric ones counted in full as computed)
flop_stage = 2 * 64**3 / 3 + 3 * 64**3 + 4 * 2 * 64**3
print(json.dumps(dict(T=T, B=B, ok=bool(ok), seconds=round(dt, 4), stages_per_s=round(B * (T - 1) / dt, 1),
                      us_per_stage_per_wg=round(dt / (T - 1) * 1e6, 2), gflops=round(B * (T - 1) * flop_stage / dt / 1e9, 1),
                      factor_bytes_GB=round(B * T * s.footprint_wide() * 8 / 1e9, 2) if hasattr(s, "footprint_wide") else round(B * T * 18128 * 8 / 1e9, 2))))

# ---- evaluator callbacks on the same model: Jacobian nnz/s (HBM-write bound: the stage Jacobian is dense)
Bj = min(B, 16)
nj = s.nlp.num_jacobian
J = torch.empty((Bj, nj), device="cuda", dtype=torch.float64)
s.nlp.eval_constraint_jacobian_batch(Z.data_ptr(), Bj, nz, J.data_ptr(), nj); torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0.record(); s.nlp.eval_constraint_jacobian_batch(Z.data_ptr(), Bj, nz, J.data_ptr(), nj); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e-3)
dtj = min(ts)
print(json.dumps(dict(op="jacobian", B=Bj, nnz=nj, seconds=round(dtj, 6), nnz_per_s=round(Bj * nj / dtj, 1),
                      GBps=round(Bj * (nj + nz) * 8 / dtj / 1e9, 1))))
