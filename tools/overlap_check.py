"""Early back substitutions on a second stream (DTO_OVERLAP_SWEEPS, default on) against plain launches: bit-identity, time.
    python tools/overlap_check.py [B]"""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
z0 = make_guesses_device(s, p, B, 1000, dev)
st = torch.cuda.current_stream().cuda_stream
res = {}
for mode in ("0", "1", "0", "1"):
    os.environ["DTO_OVERLAP_SWEEPS"] = mode
    s.begin_batch(z0.data_ptr(), B, nz, stream=st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.iterate_batch(25, stream=st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res[mode] = {k: s.peek_batch(k)[:4096] for k in ("z", "multipliers", "dz")}
    for k in ("iter", "f", "nfact", "delta_w", "alpha"):
        res[mode][k] = s.scalar_batch(k)
    print(json.dumps(dict(overlap=int(mode), batch=B, ms_per_iteration=round(dt / 25 * 1e3, 3), nfact_mean=float(res[mode]["nfact"].mean()))), flush=True)
bad = [k for k in res["0"] if not np.array_equal(res["0"][k], res["1"][k])]
print("bit-identical" if not bad else f"DIFFERENT: {bad}")
