import os, sys, json
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
import dto_amd
from dto_amd import problems as P
from bench import make_guesses_device
dev = torch.device("cuda", 0)
p = P.build_acrobot(T=1000, evaluate_hessian=True)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot")
nz = s.nlp.num_variables
B = 66000
z0 = make_guesses_device(s, p, B, 1000, dev)
st = torch.cuda.current_stream().cuda_stream
res = {}
for mode in ("0", "1"):
    os.environ["DTO_FUSE_UPDATE"] = mode
    s.begin_batch(z0.data_ptr(), B, nz, stream=st)
    s.iterate_batch(13, stream=st)
    torch.cuda.synchronize()
    res[mode] = {k: s.peek_batch(k)[:4096] for k in ("z", "multipliers", "dz")}
    res[mode]["it"] = s.scalar_batch("iter"); res[mode]["f"] = s.scalar_batch("f")
for k in ("z", "multipliers", "dz", "it", "f"):
    print(k, "bit-identical" if np.array_equal(res["0"][k], res["1"][k]) else ("DIFF max %g" % np.max(np.abs(res["0"][k] - res["1"][k]))))
