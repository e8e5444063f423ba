"""GPU side of the extended-precision step study (analysis: tools/step_truth.py, on the CPU).

Runs the acrobot T = 1000 bench state on the kernel path bench.py times (sequential sweeps, more tiles than wavefront slots:
tests/test_bench_path_gpu.py) and, separately, the time-partitioned path of three instances
(tests/test_baseline_sizes_gpu.py::test_cfg3_acrobot_T1000_step_of_the_bench_state), and writes the state and the step of a few
instances -- z, lambda, dz, dlambda, delta_w, gamma -- to gpurun_out/step_dump_*.npz.  Also: which of the eight line-search
trials every lane accepted, per tile (the lazy line-search question of VERDICT r5 item 3).

    python tools/dump_bench_step.py [tiles]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from bench import make_guesses
    from conftest import product_solver
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 1100
    T = 1000
    s, p = product_solver("acrobot", T)
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    s.options.max_iter = 1000
    # ---- (a) the bench's kernel path
    B = tiles * 64
    z0 = torch.empty((B, nz), device="cuda", dtype=torch.float64)
    rng = np.random.Generator(np.random.PCG64(1000))
    for b0 in range(0, B, 8192):
        nb = min(8192, B - b0)
        z0[b0:b0 + nb] = torch.from_numpy(make_guesses(s, p, nb, 1000, rng=rng)).cuda()
    s.set_partitions(1)
    dump = {}
    ls_hist = []
    try:
        s.begin_batch(z0.data_ptr(), B, nz)
        done = 0
        for upto in (4, 24, 60):
            while done < upto:
                # line-search statistics of every iteration on the way: trial index k = log2(alpha_max / alpha) per lane
                s.iterate_batch(1)
                done += 1
                a, am, st, mode = (s.scalar_batch(k) for k in ("alpha", "alpha_pmax", "status", "ls_mode"))
                run = st == 0
                with np.errstate(divide="ignore", invalid="ignore"):
                    k = np.where(a > 0, np.round(np.log2(np.maximum(am, 1e-300) / np.maximum(a, 1e-300))), 99.0)
                kt = np.where(run, k, -1).reshape(tiles, 64).max(axis=1)           # the deepest trial a tile needed
                ls_hist.append(dict(iteration=done, running=int(run.sum()),
                                    lane_hist=np.bincount(np.clip(k[run], 0, 9).astype(int), minlength=10).tolist(),
                                    tile_hist=np.bincount(np.clip(kt[kt >= 0], 0, 9).astype(int), minlength=10).tolist(),
                                    filter_phase=float(np.mean(mode[run] == 2)) if run.any() else 0.0))
            z, lam = s.peek_batch("z"), s.peek_batch("multipliers")
            nf0 = s.scalar_batch("nfact").copy()
            s.iterate_batch(1)
            done += 1
            nf = s.scalar_batch("nfact") - nf0
            dw, gam = s.scalar_batch("delta_w"), s.scalar_batch("gamma")
            live = np.flatnonzero(nf > 0)
            order = np.argsort(nf)
            picks = sorted({int(live[0]), int(live[-1]), int(order[-1]), int(order[-2]), int(live[len(live) // 2]), int(live[len(live) // 3])})
            dz, dl = s.peek_batch("dz"), s.peek_batch("dmultipliers")
            for b in picks:
                dump[f"seq_it{done}_b{b}"] = np.concatenate([[dw[b], gam[b], nf[b]], z[b], lam[b], dz[b], dl[b]])
            del z, lam, dz, dl
    finally:
        s.set_partitions(0)
    s.release_state()
    # ---- (b) the time-partitioned path, three instances, 5 iterations in
    B = 3
    Z = make_guesses(s, p, B, seed=1000)
    z0 = torch.tensor(Z, device="cuda")
    for it in (5, 30):
        s.begin_batch(z0.data_ptr(), B, nz)
        s.iterate_batch(it)
        for op_name in ("eval", "conv", "factor_solve"):
            s.launch_op(op_name)
        torch.cuda.synchronize()
        z, lam, dz, dl = (s.peek_batch(k) for k in ("z", "multipliers", "dz", "dmultipliers"))
        dw, gam = s.scalar_batch("delta_w"), s.scalar_batch("gamma")
        for b in range(B):
            dump[f"chunk{s.partitions()}_it{it}_b{b}"] = np.concatenate([[dw[b], gam[b], 0.0], z[b], lam[b], dz[b], dl[b]])
        s.release_state()
    np.savez_compressed(os.path.join(out, "step_dump_acrobot_T1000.npz"), nz=nz, nc=nc, **dump)
    import json
    with open(os.path.join(out, "linesearch_trials_per_tile.json"), "w") as f:
        json.dump(dict(tiles=tiles, T=T, note="k = log2(alpha_max / alpha): index of the accepted trial (99 / 9 = no step); "
                       "tile_hist: deepest trial among the running lanes of a tile", iterations=ls_hist), f)
    print("dumped", sorted(dump), file=sys.stderr)
    for h in ls_hist[::5]:
        print(h, file=sys.stderr)


if __name__ == "__main__":
    main()
