"""Is a small batch launch-bound?  One iteration of the solver captured into a HIP graph (torch.cuda.CUDAGraph on a side
stream; the library launches on the stream it is given) against the same iteration launched kernel by kernel.
    python tools/graph_latency.py [model T B]..."""
import os, sys, json, time
os.environ["DTO_FUSE_UPDATE"] = "0"     # the fused pass allocates at first use: not inside a capture
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dto_amd
from dto_amd import problems as P

cases = [("pendulum", 50, 1), ("acrobot", 101, 1), ("car", 51, 1), ("acrobot", 1000, 1), ("acrobot", 101, 1024), ("acrobot", 1000, 64)]
for model, T, B in cases:
    p = getattr(P, f"build_{model}")(T=T, evaluate_hessian=True)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name=model)
    rng = np.random.Generator(np.random.PCG64(0))
    Z = np.zeros((B, s.nlp.num_variables))
    for b in range(B):
        xs, us = p["guess"](rng)
        dto_amd.initialize_states(s, xs); dto_amd.initialize_controls(s, us)
        Z[b] = s._z0
    d = torch.tensor(Z, device="cuda")
    n = 40
    # kernel by kernel
    s.begin_batch(d.data_ptr(), B, Z.shape[1])
    s.iterate_batch(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); s.iterate_batch(n); torch.cuda.synchronize(); t_direct = (time.perf_counter() - t0) / n
    f_direct = s.scalar_batch("f").copy()
    # graph
    s.begin_batch(d.data_ptr(), B, Z.shape[1])
    s.iterate_batch(3)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        s.iterate_batch(1, stream=side.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / n
    f_graph = s.scalar_batch("f")
    print(json.dumps(dict(model=model, T=T, B=B, partitions=s.partitions(), ms_per_iteration_launches=round(t_direct * 1e3, 4),
                          ms_per_iteration_graph=round(t_graph * 1e3, 4), same_objective=bool(np.array_equal(f_direct, f_graph)))), flush=True)
    s.release_state()
