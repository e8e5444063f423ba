"""Globalisation / inertia-ladder settings scanned on the oracle's C port (CPU; experimentation aid):
    python tools/port_sweep.py NAME='{"option": value, ...}' ...      # one line per variant: acrobot T = 1000 / 301 / 101
Options: everything oracle/cpu_port/solver_port.c: port_set_int / port_set_double take (delta_w_exact_cap, kappa_w_minus,
kappa_w_plus, delta_w_init, delta_w_min, watchdog_trigger, watchdog_trials, max_soc, ls_penalty, pen_gn, ls_switch, lbfgs).
Per horizon: converged / seeds, iterations (mean, median, p90), factorisations in total, and a proxy of what a 64-lane tile pays
in iterations 5 - 24 and 25 - 59: the maximum number of factorisation attempts over groups of 64 seeds."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multiprocessing import Pool


def run(args):
    T, b, opts = args
    from oracle.cpu_port import PortSolver, guesses
    Z, x1, xT = guesses("acrobot", T, b + 1, 1000)
    s = PortSolver("acrobot", T, x1, xT, max_iter=1000)
    for k, v in opts.items():
        s.set(k, v)
    s.begin(Z[b])
    att, nf = [], 0
    while s.iterate():
        att.append(s.nfact - nf); nf = s.nfact
    return (s.status, s.iterations, s.nfact, att[:60])


if __name__ == "__main__":
    variants = [("default", {})]
    for a in sys.argv[1:]:
        name, js = a.split("=", 1)
        variants.append((name, json.loads(js)))
    sizes = ((1000, 128), (301, 64), (101, 64))
    with Pool(8) as pool:
        for name, o in variants:
            row = [f"{name} {json.dumps(o)}"]
            for T, n in sizes:
                res = pool.map(run, [(T, b, o) for b in range(n)])
                st = np.array([r[0] for r in res]); it = np.array([r[1] for r in res]); nf = np.array([r[2] for r in res])
                A = np.zeros((n, 60), dtype=int)
                for i, r in enumerate(res):
                    A[i, :len(r[3])] = r[3]
                g = A[: (n // 64) * 64].reshape(-1, 64, 60).max(axis=1)
                row.append(f"T={T}: {int((st == 1).sum())}/{n} converged, iterations mean {it.mean():.1f} median {np.median(it):.0f} p90 {np.quantile(it, 0.9):.0f}, "
                           f"factorisations {nf.sum()}, tile rounds it 5-24 {g[:, 5:25].mean():.2f} / 25-59 {g[:, 25:60].mean():.2f}")
            print(" | ".join(row), flush=True)
