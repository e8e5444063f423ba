"""GPU side of the extended-precision step study, time-partitioned path: the step of the acrobot T = 1000 bench state after `it`
iterations for several chunk counts P and batch sizes (B <= 4: separator system by cyclic reduction inside the tile's wavefront;
B > 4 with >= 16 chunks: cyclic reduction on one wavefront per instance; else lane-per-instance elimination).
Analysis on the CPU: tools/step_truth.py gpurun_out/step_dump_chunked_acrobot_T1000.npz

    python tools/dump_chunked_step.py
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
    T = 1000
    s, p = product_solver("acrobot", T)
    nz = s.nlp.num_variables
    dump = {}
    for B in (3, 8):
        Z = make_guesses(s, p, B, seed=1000)
        z0 = torch.tensor(Z, device="cuda")
        for P in (1, 2, 4, 8, 16, 32, 64):
            for it in (5, 14):
                s.set_partitions(P)
                try:
                    s.begin_batch(z0.data_ptr(), B, nz)
                    s.iterate_batch(it)
                    for op_name in ("eval", "conv", "factor_solve"):
                        s.launch_op(op_name)
                    torch.cuda.synchronize()
                    z, lam, dz, dl = (s.peek_batch(k) for k in ("z", "multipliers", "dz", "dmultipliers"))
                    dw, gam, st = s.scalar_batch("delta_w"), s.scalar_batch("gamma"), s.scalar_batch("status")
                    assert s.partitions() == P
                    for b in (0, B - 1):
                        if st[b] == 0:
                            dump[f"B{B}_P{P:02d}_it{it}_b{b}"] = np.concatenate([[dw[b], gam[b], 0.0], z[b], lam[b], dz[b], dl[b]])
                    s.release_state()
                finally:
                    s.set_partitions(0)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "step_dump_chunked_acrobot_T1000.npz"), nz=nz, nc=s.nlp.num_constraint, **dump)
    print("dumped", sorted(dump), file=sys.stderr)


if __name__ == "__main__":
    main()
