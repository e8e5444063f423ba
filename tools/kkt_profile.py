"""Cycle stamps inside k_kkt_fwd (workgroup 1 = tile 0, chunk 1): python tools/kkt_profile.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
prof = torch.zeros(32, dtype=torch.int64, device="cuda")
os.environ["DTO_KKT_PROF"] = hex(prof.data_ptr())
os.environ["DTO_PLUGIN_CXXFLAGS"] = "-DDTO_KKT_PROFILE=1"   # the stamps are compiled in on request only
import dto_amd
from dto_amd import problems as P
from bench import make_guesses
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
PART = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # 1: force the plain sequential sweeps (k_kkt_fwd_seq)
p = P.build_acrobot(T=1000, evaluate_hessian=True)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot")
nz = s.nlp.num_variables
z0 = torch.tensor(make_guesses(s, p, B, seed=1000), device="cuda")
s.set_partitions(PART)
s.begin_batch(z0.data_ptr(), B, nz)
s.iterate_batch(5); torch.cuda.synchronize(); prof.zero_()
s.iterate_batch(10); torch.cuda.synchronize()
c = prof.cpu().numpy()
n = max(1, c[7])
names = ["0 between stages (loop, prefetch, copy)", "1 loads + derivative code + scatter", "2 carry stores", "3 -", "4 LDL", "5 substitutions",
         "6 Schur complement"]
tot = sum(c[:7])
print("stages stamped", n, "cycles/stage in stamped parts", tot / n)
for i in range(7):
    print(f"{names[i]:42s} {c[i] / n:10.1f} {100.0 * c[i] / tot:5.1f}%")
