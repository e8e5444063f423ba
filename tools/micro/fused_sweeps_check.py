"""Both sweeps of a tile in ONE kernel (-DDTO_FUSE_SWEEPS=1, csrc/dto_kkt_kernels.hpp: k_kkt_fwdbwd_seq) against the product's
separate launches: same guesses, same iterations, every state vector compared bit by bit.  Round 3 saw deterministic wrong
results from the fused form when both bodies were inline and the forward prefetch was on; this is the
reproducer.   python tools/micro/fused_sweeps_check.py ["extra flags for the fused build"]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
os.environ["DTO_OVERLAP_SWEEPS"] = "0"
import dto_amd
from dto_amd import problems as P

extra = sys.argv[1] if len(sys.argv) > 1 else ""
T, B, ITERS = int(os.environ.get("FS_T", 101)), int(os.environ.get("FS_B", 300)), int(os.environ.get("FS_ITERS", 12))


def run(flags):
    os.environ["DTO_PLUGIN_CXXFLAGS"] = flags
    p = P.build_acrobot(T=T, evaluate_hessian=True)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot")
    nz = s.nlp.num_variables
    Z = np.zeros((B, nz))
    for b in range(B):
        xs, us = p["guess"](np.random.Generator(np.random.PCG64(b % 512)))
        dto_amd.initialize_states(s, xs); dto_amd.initialize_controls(s, us)
        Z[b] = s._z0
    d = torch.tensor(Z, device="cuda")
    s.set_partitions(1)
    s.begin_batch(d.data_ptr(), B, nz)
    out = []
    for k in range(ITERS):
        s.iterate_batch(1)
        out.append({n: s.peek_batch(n) for n in ("z", "multipliers", "dz", "dmultipliers")})
        out[-1]["nfact"] = s.scalar_batch("nfact").copy()
    s.close()
    return out


ref = run("")
fus = run((("" if os.environ.get("FS_NOFUSE") else "-DDTO_FUSE_SWEEPS=1 ") + extra).strip())
first = None
for k in range(ITERS):
    for n in ("dz", "dmultipliers", "z", "multipliers", "nfact"):
        if not np.array_equal(ref[k][n], fus[k][n]) and first is None:
            bad = np.flatnonzero(np.any(np.atleast_2d(ref[k][n] != fus[k][n]), axis=-1)) if ref[k][n].ndim > 1 else np.flatnonzero(ref[k][n] != fus[k][n])
            first = dict(iteration=k + 1, vector=n, instances_differing=int(len(bad)), first_instances=bad[:8].tolist(),
                         max_abs_diff=float(np.nanmax(np.abs(ref[k][n] - fus[k][n]))))
print(json.dumps(dict(flags=extra, identical=first is None, first_difference=first,
                      nfact_ref=float(ref[-1]["nfact"].mean()), nfact_fused=float(fus[-1]["nfact"].mean()))))
