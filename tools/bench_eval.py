"""Kernel-level timing of the batched evaluator callbacks (HIP events on the launch stream).

python tools/bench_eval.py [--model acrobot --T 1000 --B 4096 --iters 20]
Prints one JSON line per callback with algorithmic bytes and achieved GB/s.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dto_amd
from dto_amd import problems as P


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="acrobot")
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--B", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--sigma", type=float, default=1.0)
    a = ap.parse_args()
    p = getattr(P, f"build_{a.model}")(T=a.T, evaluate_hessian=True)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name=a.model)
    n = s.nlp
    B = a.B
    nz, nc, nj, nh = n.num_variables, n.num_constraint, n.num_jacobian, int(n.sizes.nnz_hess_key)
    g = torch.Generator(device="cuda").manual_seed(0)
    z = torch.rand((B, nz), device="cuda", dtype=torch.float64, generator=g)
    mu = torch.rand((B, nc), device="cuda", dtype=torch.float64, generator=g)
    out = torch.empty((B, max(nz, nc, nj, nh)), device="cuda", dtype=torch.float64)
    f = torch.empty((B,), device="cuda", dtype=torch.float64)
    st = torch.cuda.current_stream().cuda_stream
    calls = {
        "eval_f": (lambda: n.eval_objective_batch(z.data_ptr(), B, nz, f.data_ptr(), st), 8 * nz + 8),
        "eval_grad_f": (lambda: n.eval_objective_gradient_batch(z.data_ptr(), B, nz, out.data_ptr(), nz, st), 8 * nz + 8 * nz),
        "eval_g": (lambda: n.eval_constraint_batch(z.data_ptr(), B, nz, out.data_ptr(), nc, st), 8 * nz + 8 * nc),
        "eval_jac_g": (lambda: n.eval_constraint_jacobian_batch(z.data_ptr(), B, nz, out.data_ptr(), nj, st), 8 * nz + 8 * nj),
        "eval_h": (lambda: n.eval_hessian_lagrangian_batch(z.data_ptr(), B, nz, a.sigma, mu.data_ptr(), nc, out.data_ptr(), nh, st),
                   8 * (nz + nc + 1) + 8 * nh),
    }
    for name, (fn, bytes_per_inst) in calls.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        gbs = B * bytes_per_inst / (ms * 1e-3) / 1e9
        rec = dict(callback=name, model=a.model, T=a.T, B=B, ms=round(ms, 4), algorithmic_bytes=B * bytes_per_inst,
                   GBps=round(gbs, 1), frac_of_8TBps=round(gbs / 8000, 4))
        if name == "eval_jac_g":
            rec["jacobian_nnz_per_s"] = B * nj / (ms * 1e-3)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
