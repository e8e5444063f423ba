"""What one pass of iterative refinement (dto_options.kkt_refinement = 1) costs per iteration, by batch size (VERDICT r5 item 1:
"report its cost").  acrobot, exact Hessians; wall time of 30 iterations after 5 of warm-up, every instance running.

    python tools/refinement_cost.py > gpurun_out/refinement_cost.txt
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import dto_amd
    from dto_amd import problems as P
    from bench import make_guesses
    print(f"{'T':>5} {'instances':>9} {'chunks':>6} {'ms/iteration':>13} {'with one pass':>14} {'ratio':>6}")
    for T, B in ((101, 1), (1000, 1), (1000, 64), (1000, 1024), (1000, 8192), (1000, 32768), (1000, 131072)):
        row = []
        for passes in (0, 1):
            p = P.build_acrobot(T=T, evaluate_hessian=True)
            s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot",
                               options=dto_amd.Options(kkt_refinement=passes, tol=1e-30, max_iter=10000))
            nz = s.nlp.num_variables
            z0 = torch.tensor(make_guesses(s, p, B, seed=1000), device="cuda")
            s.begin_batch(z0.data_ptr(), B, nz)
            s.iterate_batch(5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.iterate_batch(30)
            torch.cuda.synchronize()
            row.append((time.perf_counter() - t0) / 30 * 1e3)
            chunks = s.partitions()
            s.close()
        print(f"{T:5d} {B:9d} {chunks:6d} {row[0]:13.4f} {row[1]:14.4f} {row[1] / row[0]:6.2f}", flush=True)


if __name__ == "__main__":
    main()
