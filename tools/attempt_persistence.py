"""How predictable is the number of factorisation attempts of an instance-iteration?  ("the discarded half
of the lanes' work" -- a tile of 64 lanes runs as many inertia-correction rounds as its slowest lane.)

Runs N acrobot T=1000 instances of the bench's seeded guesses through oracle/cpu_port (the same iteration as the device),
records attempts[i, k], delta_w[i, k], theta[i, k] and prices tile compositions:

    random      tiles of 64 in guess order (today)
    prev        instances sorted by the attempts of the PREVIOUS iteration before every iteration (upper bound of what a
                per-iteration regrouping on that key could give)
    prev_dw     ... by (attempts, delta_w) of the previous iteration
    every R     regrouped on the key every R iterations only (what a periodic repack would give)
    oracle      sorted by the attempts of THIS iteration (lower bound: mean tile maximum with perfect knowledge)

    python tools/attempt_persistence.py [N] [iterations] > profiles/r05/attempt_persistence_T1000.json
"""
import json
import os
import sys
from multiprocessing import Pool

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

T = 1000


def _run(args):
    lo, hi, K, seed = args
    from oracle.cpu_port import PortSolver, guesses
    Z = guesses("acrobot", T, hi, seed)[0][lo:hi]
    att = np.zeros((hi - lo, K), dtype=np.int32)
    dw = np.zeros((hi - lo, K))
    th = np.zeros((hi - lo, K))
    for i in range(hi - lo):
        ps = PortSolver("acrobot", T, max_iter=1000)
        ps.begin(Z[i])
        prev = 0
        for k in range(K):
            more = ps.iterate()
            nf = ps.nfact
            att[i, k] = nf - prev
            prev = nf
            st = ps.stats()
            dw[i, k], th[i, k] = st["delta_w"], st["constr_viol"]
            if not more:
                break
        ps.close()
    return att, dw, th


def tile_rounds(att_k, order):
    a = att_k[order]
    n = len(a) // 64 * 64
    return float(np.mean(np.max(a[:n].reshape(-1, 64), axis=1)))


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 45
    P = 8
    chunks = [(i * N // P, (i + 1) * N // P, K, 1000) for i in range(P)]
    with Pool(P) as pool:
        res = pool.map(_run, chunks)
    att = np.concatenate([r[0] for r in res])
    dw = np.concatenate([r[1] for r in res])
    th = np.concatenate([r[2] for r in res])
    out = dict(instances=N, iterations=K, horizon=T, windows={})
    for (k0, k1) in ((5, 25), (25, 45)):
        if k1 > K:
            continue
        live = lambda k: att[:, k] > 0
        rows = {}
        rounds = {name: [] for name in ("random", "prev", "prev_dw", "every4", "every8", "oracle")}
        lane_mean = []
        key_at = {}
        for k in range(k0, k1):
            a = att[:, k].copy()
            ident = np.arange(N)
            keyp = att[:, k - 1] * 1.0
            keyd = att[:, k - 1] * 1e6 + np.log10(np.maximum(dw[:, k - 1], 1e-30))
            rounds["random"].append(tile_rounds(a, ident))
            rounds["prev"].append(tile_rounds(a, np.argsort(keyp, kind="stable")))
            rounds["prev_dw"].append(tile_rounds(a, np.argsort(keyd, kind="stable")))
            for R in (4, 8):
                kk = k0 + (k - k0) // R * R      # last regrouping
                kd = att[:, kk - 1] * 1e6 + np.log10(np.maximum(dw[:, kk - 1], 1e-30))
                rounds[f"every{R}"].append(tile_rounds(a, np.argsort(kd, kind="stable")))
            rounds["oracle"].append(tile_rounds(a, np.argsort(a, kind="stable")))
            lane_mean.append(float(np.mean(a[a > 0])) if np.any(a > 0) else 0.0)
        hist = np.bincount(att[:, k0:k1].ravel(), minlength=8)[:8]
        # persistence: P(attempts_k >= 3 | attempts_{k-1} >= 3) against the base rate
        a0, a1 = att[:, k0 - 1:k1 - 1], att[:, k0:k1]
        base = float(np.mean(a1 >= 3))
        cond = float(np.mean(a1[a0 >= 3] >= 3)) if np.any(a0 >= 3) else None
        out["windows"][f"{k0}-{k1}"] = dict(tile_rounds={n: round(float(np.mean(v)), 3) for n, v in rounds.items()},
                                             lane_mean_attempts=round(float(np.mean(lane_mean)), 3),
                                             attempts_histogram=hist.tolist(),
                                             p_ge3=round(base, 4), p_ge3_given_prev_ge3=None if cond is None else round(cond, 4),
                                             corr_prev=round(float(np.corrcoef(a0.ravel(), a1.ravel())[0, 1]), 3))
    print(json.dumps(out, indent=1))
