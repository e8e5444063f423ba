"""One round of the time-partitioned factor + solve split into its launches (k_kkt_fwd chunks, k_kkt_sep, k_kkt_bwd, k_kkt_post),
timed with events around Solver.launch_op on the state a running batch is in:  python tools/chunked_round_split.py [instances] [iteration]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import dto_amd
from dto_amd import problems as P
from bench import make_guesses_device

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 15
T = 1000
p = P.build_acrobot(T=T, evaluate_hessian=True)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot")
nz = s.nlp.num_variables
z0 = make_guesses_device(s, p, B, 1000, "cuda")
s.begin_batch(z0.data_ptr(), B, nz)
s.iterate_batch(IT)
torch.cuda.synchronize()


def timed(name, reps=1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        s.launch_op(name)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = dict(instances=B, iteration=IT, partitions=s.partitions(), ms={})
s.launch_op("eval"); s.launch_op("conv")
for name in ("kkt_fwd", "kkt_sep", "kkt_bwd", "kkt_post"):
    out["ms"][name] = round(timed(name), 4)
# the sequential form on the same state for comparison: one round, all tiles
out["whole_factor_solve_ms"] = round(timed("factor_solve"), 4)
s.set_partitions(1)
s.begin_batch(z0.data_ptr(), B, nz)
s.iterate_batch(IT)
torch.cuda.synchronize()
s.launch_op("eval"); s.launch_op("conv")
out["sequential_factor_solve_ms"] = round(timed("factor_solve"), 4)
nf = s.scalar_batch("nfact")
out["note"] = "kkt_fwd .. kkt_post: ONE round (every lane, first delta_w); factor_solve: all rounds of the iteration"
print(json.dumps(out))
