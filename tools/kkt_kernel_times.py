"""Per-kernel durations of one solver iteration (HIP events), full-batch launches only: python tools/kkt_kernel_times.py [B] [T] [model]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dto_amd
from dto_amd import problems as P
from bench import make_guesses, event_time_ms
B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
model = sys.argv[3] if len(sys.argv) > 3 else "acrobot"
p = getattr(P, f"build_{model}")(T=T, evaluate_hessian=True)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name=model)
nz = s.nlp.num_variables
if model == "acrobot":
    Z = make_guesses(s, p, B, seed=1000)
else:
    xs, us = p["guess"](np.random.Generator(np.random.PCG64(0)))
    dto_amd.initialize_states(s, xs); dto_amd.initialize_controls(s, us)
    Z = np.tile(s._z0, (B, 1))
z0 = torch.tensor(Z, device="cuda")
st = torch.cuda.current_stream().cuda_stream
s.begin_batch(z0.data_ptr(), B, nz, stream=st)
s.iterate_batch(5, stream=st)
torch.cuda.synchronize()
out = {}
for rep in range(3):
    for o in ("eval", "conv"):
        out.setdefault(o, []).append(event_time_ms(lambda: s.launch_op(o, stream=st), 1))
    # first forward round: every lane factorises
    out.setdefault("kkt_fwd(round 1)", []).append(event_time_ms(lambda: s.launch_op("kkt_fwd", stream=st), 1))
    s.launch_op("kkt_sep", stream=st)
    for _ in range(9):
        s.launch_op("kkt_fwd", stream=st); s.launch_op("kkt_sep", stream=st)
    for o in ("kkt_bwd", "kkt_post", "linesearch", "ls_reduce", "update"):
        out.setdefault(o, []).append(event_time_ms(lambda: s.launch_op(o, stream=st), 1))
print(json.dumps(dict(B=B, T=T, model=model, partitions=s.partitions(), ms={k: round(float(np.median(v)), 3) for k, v in out.items()})))
