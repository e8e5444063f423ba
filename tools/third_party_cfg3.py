"""Independent evidence for the end-game of BASELINE configs[2] (acrobot T = 1000): scipy.optimize.minimize(method=
"trust-constr") -- a Byrd-Omojokun trust-region SQP that shares nothing with this repository -- on the ORACLE's callbacks
(oracle/dto_oracle.py, the restatement of src/moi.jl:1-120) from the bench's own seeded guesses, beside the C port of this
repository's iteration (oracle/cpu_port) on the same guesses.  Reports iterations, final objective, violation and which
minimiser each ends in.  CPU only.

    python tools/third_party_cfg3.py --T 1000 --seeds 8 --workers 4 --out profiles/r05/third_party_cfg3_T1000.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run_one(args):
    T, k, maxiter, gtol = args
    import scipy.sparse as sp
    from scipy.optimize import NonlinearConstraint, minimize
    from oracle import dto_oracle as O, sympy_models as S
    from oracle.cpu_port import guesses
    p = S.build("acrobot", T, evaluate_hessian=True)
    onlp = O.NLPData(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True)
    nz, nc = onlp.num_variables, onlp.num_constraint
    js = np.array(onlp.jacobian_structure()) - 1
    hs = np.array(onlp.hessian_lagrangian_structure()) - 1
    mat = lambda v, idx, shape: sp.coo_matrix((v, (idx[:, 0], idx[:, 1])), shape=shape).tocsr()
    con = NonlinearConstraint(onlp.eval_constraint, 0.0, 0.0,
                              jac=lambda z: mat(onlp.eval_constraint_jacobian(z), js, (nc, nz)),
                              hess=lambda z, v: mat(onlp.eval_hessian_lagrangian(z, 0.0, v), hs, (nz, nz)))
    z0 = guesses("acrobot", T, k + 1, 1000)[0][k]            # instance k of the bench's stream (bench.py: seed 1000 + rank)
    hist = []

    def cb(xk, st):
        hist.append((int(st.nit), float(st.fun), float(st.constr_violation), float(st.optimality), float(st.tr_radius)))
        return False
    t0 = time.time()
    res = minimize(onlp.eval_objective, z0, jac=onlp.eval_objective_gradient,
                   hess=lambda z: mat(onlp.eval_hessian_lagrangian(z, 1.0, np.zeros(nc)), hs, (nz, nz)),
                   method="trust-constr", constraints=[con], callback=cb,
                   options=dict(gtol=gtol, xtol=1e-10, maxiter=maxiter))
    dt = time.time() - t0
    # first iteration at which the reference Options' termination levels hold (tol 1e-6 on the scaled optimality, constr_viol_tol
    # 1e-3: src/options.jl:7,13) -- trust-constr's own gtol is tighter
    first = next((h[0] for h in hist if h[3] <= 1e-6 and h[2] <= 1e-3), None)
    xs = res.x.reshape(-1)[: (T - 1) * 5 + 4]
    q1 = np.array([res.x[t * 5] for t in range(T)])
    # time of the swing-up: first knot from which the first link stays within 0.1 rad of pi
    up = np.abs(q1 - np.pi) < 0.1
    t_up = int(T - np.argmin(up[::-1])) if not up.all() else 0
    return dict(instance=k, status=int(res.status), message=str(res.message), nit=int(res.nit), nfev=int(res.nfev),
                cg_niter=int(res.cg_niter), f=float(res.fun), constr_violation=float(res.constr_violation),
                optimality=float(res.optimality), first_iter_at_reference_tolerances=first, seconds=dt,
                swing_up_complete_at_knot=t_up, history_every_50=hist[::50])


def port_runs(T, n):
    """The C port of this repository's iteration on the same guesses (iterations, final objective, status)."""
    from oracle import cpu_port as CP
    Z0, x1, xT = CP.guesses("acrobot", T, n, 1000)
    out = []
    for k in range(n):
        ps = CP.PortSolver("acrobot", T, max_iter=1000)
        t0 = time.time()
        st = ps.solve(Z0[k])
        out.append(dict(instance=k, status=int(st), iterations=int(ps.iterations), factorizations=int(ps.nfact),
                        f=float(ps.stats()["objective"]), constr_viol=float(ps.stats()["constr_viol"]), seconds=time.time() - t0))
        ps.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--maxiter", type=int, default=3000)
    ap.add_argument("--gtol", type=float, default=1e-8)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from multiprocessing import Pool
    with Pool(a.workers) as pool:
        rows = pool.map(run_one, [(a.T, k, a.maxiter, a.gtol) for k in range(a.seeds)], chunksize=1)
    doc = dict(what="scipy trust-constr on the oracle's callbacks, acrobot T=%d, bench guesses (PCG64 seed 1000), instance k" % a.T,
               options=dict(gtol=a.gtol, xtol=1e-10, maxiter=a.maxiter), trust_constr=rows, port=port_runs(a.T, a.seeds))
    s = json.dumps(doc, indent=1)
    if a.out:
        with open(a.out, "w") as f:
            f.write(s + "\n")
    for r in rows:
        print({k: v for k, v in r.items() if k != "history_every_50"})
    print(doc["port"])


if __name__ == "__main__":
    main()
