"""How persistent is the number of factorisation attempts of an instance from one iteration to the next?  (decides whether
packing tiles by the previous iteration's attempts would cut the lock-step rounds of k_kkt_fwd_seq)
    python tools/attempt_persistence.py [--batch B] [--iters K] [--every E]
Prints per iteration: mean attempts per running instance, tile rounds (sum over tiles of the max over its 64 lanes) as laid
out, and the same if the instances had been sorted by their attempts E iterations earlier (or by delta_w of the last step)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dto_amd
from dto_amd import problems as P
from bench import make_guesses_device

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16384)
ap.add_argument("--horizon", type=int, default=1000)
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--every", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda", 0)
p = P.build_acrobot(T=a.horizon, evaluate_hessian=True)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot")
nz = s.nlp.num_variables
z0 = make_guesses_device(s, p, a.batch, 1000, dev)
st = torch.cuda.current_stream().cuda_stream
s.options.max_iter = 1000
s.begin_batch(z0.data_ptr(), a.batch, nz, stream=st)
prev_nf = s.scalar_batch("nfact").copy()
hist = []
B = a.batch


def rounds(att, order):
    x = att[order]
    x = np.concatenate([x, np.zeros((-len(x)) % 64, dtype=x.dtype)]).reshape(-1, 64)
    return int(x.max(axis=1).sum())


tot = dict(asis=0, ideal=0, prev1=0, prevE=0, dw=0, lanes=0)
perm_E = np.arange(B)
for k in range(a.iters):
    s.iterate_batch(1, stream=st)
    torch.cuda.synchronize()
    nf = s.scalar_batch("nfact")
    att = (nf - prev_nf).astype(np.int64)
    prev_nf = nf.copy()
    al, am = s.scalar_batch("alpha"), s.scalar_batch("alpha_pmax")
    kidx = np.where(al > 0, np.round(np.log2(np.maximum(am, 1e-300) / np.maximum(al, 1e-300))), 99).astype(int)
    ls_hist = np.bincount(np.clip(kidx, 0, 9), minlength=10)
    kt = np.concatenate([kidx, np.zeros((-len(kidx)) % 64, dtype=int)]).reshape(-1, 64).max(axis=1)
    hist.append(att)
    ident = np.arange(B)
    r = dict(it=k, mean=float(att.mean()), asis=rounds(att, ident), ideal=rounds(att, np.argsort(att, kind="stable")),
             ls_trial_hist=ls_hist.tolist(), tiles_needing_more_than_2_trials=float((kt >= 2).mean()),
             tiles_needing_more_than_1_trial=float((kt >= 1).mean()))
    if k > 0:
        r["prev1"] = rounds(att, np.argsort(hist[-2], kind="stable"))
        r["prevE"] = rounds(att, perm_E)
        for key in ("asis", "ideal", "prev1", "prevE"):
            tot[key] += r[key]
        tot["lanes"] += int(att.sum())
    if k % a.every == 0:
        perm_E = np.argsort(att, kind="stable")   # a repack every E iterations, by the attempts of that iteration
    print(json.dumps(r), flush=True)
n_t = (B + 63) // 64
print(json.dumps(dict(summary="tile rounds per iteration and tile (64 lanes)", **{k: round(v / (n_t * (a.iters - 1)), 3) for k, v in tot.items() if k != "lanes"},
                      attempts_per_lane=round(tot["lanes"] / (B * (a.iters - 1)), 3))))
