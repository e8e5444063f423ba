"""Price a change of the inertia-correction ladder on the C port before it goes into the kernels: acrobot T=1000, the bench's
guesses, full solves.  Environment knobs of oracle/cpu_port/solver_port.c (DTO_PIV_JUMP, DTO_KW_PLUS, ...) are passed through.

    DTO_PIV_JUMP=2 python tools/ladder_experiment.py [N]

prints: converged / N, iteration median / p90, attempts per instance-iteration, rounds of a 64-lane tile (mean of the tile
maximum over the iterations in which the tile has a running lane), and the total tile rounds until 85 % of a tile's lanes are done.
"""
import json
import os
import sys
from multiprocessing import Pool

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
T = int(os.environ.get("LADDER_T", "1000"))
KREC = 400


def _run(args):
    lo, hi, seed = args
    from oracle.cpu_port import PortSolver, guesses
    Z = guesses("acrobot", T, hi, seed)[0][lo:hi]
    att = np.zeros((hi - lo, KREC), dtype=np.int16)
    its = np.zeros(hi - lo, dtype=np.int32)
    st = np.zeros(hi - lo, dtype=np.int32)
    f = np.zeros(hi - lo)
    for i in range(hi - lo):
        ps = PortSolver("acrobot", T, max_iter=1000)
        ps.begin(Z[i])
        prev, k = 0, 0
        while True:
            more = ps.iterate()
            nf = ps.nfact
            if k < KREC:
                att[i, k] = nf - prev
            prev = nf
            k += 1
            if not more:
                break
        its[i], st[i], f[i] = ps.iterations, ps.status, ps.stats()["objective"]
        ps.close()
    return att, its, st, f


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    P = 8
    with Pool(P) as pool:
        res = pool.map(_run, [(i * N // P, (i + 1) * N // P, 1000) for i in range(P)])
    att = np.concatenate([r[0] for r in res]).astype(np.int64)
    its = np.concatenate([r[1] for r in res])
    st = np.concatenate([r[2] for r in res])
    f = np.concatenate([r[3] for r in res])
    tiles = att[: N // 64 * 64].reshape(-1, 64, KREC)
    tmax = tiles.max(axis=1)                        # (tiles, K)
    live = tmax > 0
    win = lambda a, k0, k1: float(a[:, k0:k1][live[:, k0:k1]].mean())
    lanes = att[att > 0]
    # iterations until 85 % of a tile's lanes are done, and the tile rounds spent until then
    done_at = np.sort(its[: N // 64 * 64].reshape(-1, 64), axis=1)[:, int(0.85 * 64) - 1]
    cost85 = [int(tmax[i, : min(int(done_at[i]), KREC)].sum()) for i in range(len(done_at))]
    print(json.dumps(dict(knobs={k: v for k, v in os.environ.items() if k.startswith("DTO_")}, N=N, T=T,
                          converged=int(np.sum(st == 1)), iterations_median=float(np.median(its)),
                          iterations_p90=float(np.percentile(its, 90)), f_median=round(float(np.median(f[st == 1])), 2),
                          attempts_per_lane=round(float(lanes.mean()), 3),
                          tile_rounds_5_25=round(win(tmax, 5, 25), 3), tile_rounds_25_60=round(win(tmax, 25, 60), 3),
                          tile_rounds_all=round(float(tmax[live].mean()), 3),
                          iterations_to_85pct_mean=float(done_at.mean()), tile_rounds_to_85pct_mean=float(np.mean(cost85)),
                          attempts_histogram=np.bincount(lanes, minlength=8)[:8].tolist())))
