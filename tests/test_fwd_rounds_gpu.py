"""The sequential sweep does its inertia-correction rounds inside ONE launch (csrc/dto_kkt_kernels.hpp: kkt_fwd_body,
retry_update); DTO_FWD_ROUNDS caps the rounds per launch (1 = a launch per round, the round-1 structure).  The launch
structure must not change a single bit of the solve: acrobot T = 101, 130 seeded instances (three tiles, one of them
partly filled), sequential form forced with set_partitions(1), iterates / multipliers / status / iteration and
factorisation counts compared between DTO_FWD_ROUNDS = 0 (all rounds in the launch), 1 and 3.  The knob is read once per
process, hence the subprocesses."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = r"""
import hashlib, json, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import dto_amd
from dto_amd import problems as P
from bench import make_guesses
T, B = 101, 130
p = P.build_acrobot(T=T, evaluate_hessian=True)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot")
nz, nc = s.nlp.num_variables, s.nlp.num_constraint
z0 = torch.tensor(make_guesses(s, p, B, seed=7), device="cuda")
zo = torch.full((B, nz), float("nan"), device="cuda", dtype=torch.float64)
lo = torch.full((B, nc), float("nan"), device="cuda", dtype=torch.float64)
s.set_partitions(1)
status, iters = s.solve_batch(z0.data_ptr(), B, nz, zo.data_ptr(), nz, lo.data_ptr(), nc)
torch.cuda.synchronize()
assert s.partitions() == 1
nf = s.scalar_batch("nfact")
h = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
print(json.dumps(dict(z=h(zo.cpu().numpy()), lam=h(lo.cpu().numpy()), status=[int(v) for v in status], iters=[int(v) for v in iters],
                      nfact=float(np.sum(nf)), rounds=s.footprint()["factor_rounds"])))
"""


def run(rounds):
    env = dict(os.environ, DTO_FWD_ROUNDS=str(rounds))
    out = subprocess.run([sys.executable, "-c", SNIPPET % (ROOT, os.path.join(ROOT, "tests"))], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_rounds_per_launch_do_not_change_the_solve():
    ref = run(0)
    assert ref["rounds"] == 1                                   # one k_kkt_fwd_seq launch per iteration
    assert all(st == 1 for st in ref["status"]), ref["status"]  # every instance converges (T = 101: 1024/1024 in the bench)
    assert ref["nfact"] > sum(ref["iters"])                     # the inertia correction did retry
    for r in (1, 3):
        got = run(r)
        assert got["rounds"] == (10 + r - 1) // r
        for k in ("z", "lam", "status", "iters", "nfact"):
            assert got[k] == ref[k], (r, k)
