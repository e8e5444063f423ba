"""A/B of the two solver engines on the bench workload (acrobot, implicit midpoint, exact Hessians):
    python tools/im_vs_soa.py [--batch B] [--horizon T] [--iters K] [--engines soa,im] [--full]
Prints one JSON line per engine: iteration throughput over the first K iterations (every instance running), factorisations
per iteration, and with --full the wall time / throughput of complete solves (reference Options)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dto_amd
from dto_amd import problems as P
from bench import make_guesses_device


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=131072)
    ap.add_argument("--horizon", type=int, default=1000)
    ap.add_argument("--iters", type=int, default=25)
    ap.add_argument("--engines", default="soa,im")
    ap.add_argument("--full", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    T, B = a.horizon, a.batch
    os.environ["DTO_PLUGIN_IM"] = "1"      # the instance-major kernels are compiled into a plugin only on request
    p = P.build_acrobot(T=T, evaluate_hessian=True)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot")
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    z0 = make_guesses_device(s, p, B, 1000, dev)
    st = torch.cuda.current_stream().cuda_stream
    for eng in a.engines.split(","):
        s.set_engine(eng)
        s.options.max_iter = 1000
        s.begin_batch(z0.data_ptr(), B, nz, stream=st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.iterate_batch(a.iters, stream=st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        it = s.scalar_batch("iter")
        nf = s.scalar_batch("nfact")
        out = dict(engine=s.engine(), batch=B, horizon=T, iterations=a.iters, seconds=round(dt, 4),
                   its_per_sec=round(float(it.sum()) / dt, 1), ms_per_iteration=round(dt / a.iters * 1e3, 3),
                   factorizations_per_iteration=round(float(nf.sum() / max(1.0, it.sum())), 3),
                   hbm_free_gb=round(torch.cuda.mem_get_info(dev)[0] / 1e9, 1))
        if a.full:
            zo = torch.zeros((B, nz), device=dev, dtype=torch.float64)
            s.begin_batch(z0.data_ptr(), B, nz, stream=st)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            status, iters = s.run_batch(zo.data_ptr(), nz, stream=st)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out["full"] = dict(seconds=round(dt, 3), its_per_sec=round(float(iters.sum()) / dt, 1),
                               converged=int((status == 1).sum()), iteration_limit=int((status == 2).sum()),
                               other=int(((status != 1) & (status != 2)).sum()), iterations_median=float(np.median(iters)),
                               converged_solves_per_sec=round(float((status == 1).sum()) / dt, 1),
                               factorizations_per_iteration=round(float(s.scalar_batch("nfact").sum() / max(1.0, iters.sum())), 3))
            del zo
        print(json.dumps(out), flush=True)
        s.release_state()
    s.set_engine("auto")


if __name__ == "__main__":
    main()
