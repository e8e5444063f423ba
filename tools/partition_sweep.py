"""Iteration time against the number of time chunks P for batches around one residency of the sweeps:
    python tools/partition_sweep.py B1,B2,... P1,P2,...   (P = 0: the library's choice)"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dto_amd
from dto_amd import problems as P
from bench import make_guesses_device
dev = torch.device("cuda", 0)
p = P.build_acrobot(T=1000, evaluate_hessian=True)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot")
nz = s.nlp.num_variables
st = torch.cuda.current_stream().cuda_stream
for B in [int(x) for x in sys.argv[1].split(",")]:
    z0 = make_guesses_device(s, p, B, 1000, dev)
    for part in [int(x) for x in sys.argv[2].split(",")]:
        s.set_partitions(part)
        s.begin_batch(z0.data_ptr(), B, nz, stream=st)
        s.iterate_batch(5, stream=st)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); s.iterate_batch(20, stream=st); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(json.dumps(dict(batch=B, requested_P=part, P=s.partitions(), ms_per_iteration=round(dt / 20 * 1e3, 3),
                              M_it_per_s=round(B * 20 / dt / 1e6, 3))), flush=True)
    s.set_partitions(0)
    del z0
