"""CPU side of the extended-precision step study: who is right at T = 1000, the GPU's block-tridiagonal
LDL^T step or the float64 sparse-LU step the tests compared it with?

Input: gpurun_out/step_dump_acrobot_T1000.npz (tools/dump_bench_step.py, run on the GPU).  For every dumped instance: K and the
right-hand side from the ORACLE's derivatives at the dumped (z, lambda, delta_w, gamma); the solution of that float64 system in
extended precision (tests/extended_precision.py: LU-preconditioned refinement, residual in np.longdouble); forward errors of the
GPU step and of the plain sparse-LU step against it; the sensitivity of the solution to half-ulp noise in the entries of K and
the right-hand side; backward errors.

    python tools/step_truth.py [dump.npz] > profiles/r06/step_truth_acrobot_T1000.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from scipy.sparse.linalg import splu
    from extended_precision import data_sensitivity, residual_extended, solve_extended
    from test_baseline_sizes_gpu import oracle_for, sparse_kkt
    fn = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "step_dump_acrobot_T1000.npz")
    d = np.load(fn)
    nz, nc = int(d["nz"]), int(d["nc"])
    onlp = oracle_for("acrobot", 1000)
    rows = []
    for key in sorted(k for k in d.files if k not in ("nz", "nc")):
        v = d[key]
        dw, gam, nf = v[0], v[1], v[2]
        z, lam, dz, dl = np.split(v[3:], np.cumsum([nz, nc, nz]))
        K, rhs, _ = sparse_kkt(onlp, z, lam, dw, 1e-8, gam=gam)
        x, info = solve_extended(K, rhs)
        scale = float(np.max(np.abs(x)))
        got = np.concatenate([dz, dl])
        lu = splu(K).solve(rhs)
        sens = data_sensitivity(K, rhs, x, trials=2)
        den = float(abs(K).max() * np.max(np.abs(got)) + np.max(np.abs(rhs)))
        rows.append(dict(instance=key, delta_w=float(dw), gauss_newton=bool(gam == 0.0), factorisations_in_launch=int(nf),
                         step_norm=scale, rhs_norm=float(np.max(np.abs(rhs))), refinement_iterations=info["iterations"],
                         truth_residual=info["residual"],
                         gpu_forward_error=float(np.max(np.abs(got - x))) / scale,
                         gpu_forward_error_dz=float(np.max(np.abs(dz - x[:nz]))) / scale,
                         lu_forward_error=float(np.max(np.abs(lu - x))) / scale,
                         half_ulp_data_sensitivity=sens / scale,
                         gpu_backward_error=residual_extended(K, got, rhs) / den,
                         lu_backward_error=residual_extended(K, lu, rhs) / den))
        print(json.dumps(rows[-1]), file=sys.stderr)
    out = dict(what="acrobot T = 1000, steps of the bench state: forward errors against the extended-precision solution of the oracle's K "
                    "(relative to max|step|), tools/step_truth.py", instances=rows,
               worst=dict(gpu_forward_error=max(r["gpu_forward_error"] for r in rows), lu_forward_error=max(r["lu_forward_error"] for r in rows),
                          half_ulp_data_sensitivity=max(r["half_ulp_data_sensitivity"] for r in rows),
                          gpu_backward_error=max(r["gpu_backward_error"] for r in rows)))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
