"""Convergence statistics of the oracle's serial C port over seeded acrobot guesses, one process per core
(experimentation aid for the globalisation; CPU only):  python tools/port_stats.py T n_seeds [max_iter]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multiprocessing import Pool


def run(args):
    T, b, max_iter, model = args
    from oracle.cpu_port import PortSolver, acrobot_guesses, guesses
    Z, x1, xT = guesses(model, T, b + 1, 1000)
    s = PortSolver(model, T, x1, xT, max_iter=max_iter)
    s.begin(Z[b])
    while s.iterate():
        pass
    st = s.stats()
    return dict(seed=b, status=s.status, it=s.iterations, nfact=s.nfact, f=st["objective"], cv=st["constr_viol"], di=st["dual_inf"])


if __name__ == "__main__":
    T, n = int(sys.argv[1]), int(sys.argv[2])
    max_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    model = sys.argv[4] if len(sys.argv) > 4 else "acrobot"
    t0 = time.perf_counter()
    with Pool(min(8, n)) as pool:
        res = pool.map(run, [(T, b, max_iter, model) for b in range(n)])
    dt = time.perf_counter() - t0
    it = np.array([r["it"] for r in res]); st = np.array([r["status"] for r in res])
    print(json.dumps(dict(model=model, T=T, n=n, max_iter=max_iter, converged=int(np.sum(st == 1)), it_median=float(np.median(it)), it_mean=float(np.mean(it)),
                          it_max=int(it.max()), it_conv_median=float(np.median(it[st == 1])) if np.any(st == 1) else None,
                          fact_per_it=float(sum(r["nfact"] for r in res) / max(1, it.sum())), seconds=round(dt, 1),
                          f=[round(r["f"], 3) for r in res][:16])))
    if os.environ.get("VERBOSE"):
        for r in res:
            print(r)
