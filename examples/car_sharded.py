"""BASELINE configs[3]: car (Dubins) with an obstacle inequality at every knot, T = 500, 512 random seeds, sharded over
the GPUs of one node (one process per GPU, instances are independent: no collective inside the solve), converged
trajectories all-gathered over RCCL at the end (examples/car/car.jl:12-67 is the single-instance original, T = 51).

    python examples/car_sharded.py                                   # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        examples/car_sharded.py --batch 512
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512, help="instances in total (sharded over the ranks)")
    ap.add_argument("--horizon", type=int, default=500)
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    import dto_amd
    from dto_amd import problems as P
    from dto_amd.parallel import gather_trajectories, shard_range

    T = a.horizon
    p = P.build_car(T=T, evaluate_hessian=False)           # the example is written in the default mode
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], name="car")
    nz = s.nlp.num_variables
    lo, hi = shard_range(a.batch, rank, world)
    Z = np.zeros((hi - lo, nz))
    for k, seed in enumerate(range(lo, hi)):               # seed = global instance id: the shard layout does not matter
        xs, us = p["guess"](np.random.Generator(np.random.PCG64(seed)))
        dto_amd.initialize_states(s, xs)
        dto_amd.initialize_controls(s, us)
        Z[k] = s._z0
    z0 = torch.tensor(Z, device=dev)
    zo = torch.empty_like(z0)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    status, iters = s.solve_batch(z0.data_ptr(), hi - lo, nz, zo.data_ptr(), nz)
    torch.cuda.synchronize()
    gathered = gather_trajectories(zo, torch.tensor(status, device=dev, dtype=torch.float64), dist)
    dt = time.perf_counter() - t0
    tot = torch.tensor([float(np.sum(status == 1)), float(np.sum(iters)), dt], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(tot[:2], op=dist.ReduceOp.SUM)
        dist.all_reduce(tot[2:], op=dist.ReduceOp.MAX)
    if rank == 0:
        idx = s.nlp.indices
        xT = gathered[:, torch.tensor(np.array(idx.states[-1]) - 1, device=gathered.device)].cpu().numpy()
        print(json.dumps(dict(config="car obstacle, T=%d, %d seeds, %d GPU(s)" % (T, a.batch, world), converged=int(tot[0]),
                              iterations=int(tot[1]), seconds=round(float(tot[2]), 3), solves_per_sec=round(a.batch / float(tot[2]), 1),
                              gathered=list(gathered.shape), max_terminal_error=float(np.max(np.abs(xT - p["xT"]))))))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
