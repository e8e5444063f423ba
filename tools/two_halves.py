"""Would two independent half-batches on two streams fill each other's drains?  One solver with B instances against two
solvers with B/2 each whose iterations are enqueued on two streams.
    python tools/two_halves.py [B]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dto_amd
from dto_amd import problems as P
from bench import make_guesses_device
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
K = 25
p = P.build_acrobot(T=1000, evaluate_hessian=True)
def mk():
    return dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot")
s = mk()
nz = s.nlp.num_variables
z0 = make_guesses_device(s, p, B, 1000, dev)
st = torch.cuda.current_stream().cuda_stream
s.begin_batch(z0.data_ptr(), B, nz, stream=st)
torch.cuda.synchronize()
t0 = time.perf_counter(); s.iterate_batch(K, stream=st); torch.cuda.synchronize(); dt1 = time.perf_counter() - t0
f1 = s.scalar_batch("f").copy()
s.release_state()
print(json.dumps(dict(mode="one batch", batch=B, ms_per_iteration=round(dt1 / K * 1e3, 2))), flush=True)
for parts in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["2"])]:
    sol = [mk() for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    cuts = [(B * k) // parts // 64 * 64 for k in range(parts)] + [B]
    for k in range(parts):
        zk = z0[cuts[k]:cuts[k + 1]]
        sol[k].begin_batch(zk.data_ptr(), cuts[k + 1] - cuts[k], nz, stream=streams[k].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):     # interleaved enqueueing: no queue runs dry
        for k in range(parts):
            sol[k].iterate_batch(1, stream=streams[k].cuda_stream)
    torch.cuda.synchronize()
    dt2 = time.perf_counter() - t0
    f2 = np.concatenate([x.scalar_batch("f") for x in sol])
    print(json.dumps(dict(mode=f"{parts} sub-batches on {parts} streams", batch=B, ms_per_iteration=round(dt2 / K * 1e3, 2),
                          same_objectives=bool(np.array_equal(f1, f2)))), flush=True)
    for x in sol:
        x.release_state()
    del sol
