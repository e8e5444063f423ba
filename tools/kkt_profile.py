"""Cycle stamps inside k_kkt_fwd (workgroup 1 = tile 0, chunk 1): python tools/kkt_profile.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
prof = torch.zeros(32, dtype=torch.int64, device="cuda")
os.environ["DTO_KKT_PROF"] = hex(prof.data_ptr())
import dto_amd
from dto_amd import problems as P
from bench import make_guesses
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
p = P.build_acrobot(T=1000, evaluate_hessian=True)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot")
nz = s.nlp.num_variables
z0 = torch.tensor(make_guesses(s, p, B, seed=1000), device="cuda")
s.begin_batch(z0.data_ptr(), B, nz)
s.iterate_batch(5); torch.cuda.synchronize(); prof.zero_()
s.iterate_batch(10); torch.cuda.synchronize()
c = prof.cpu().numpy()
n = max(1, c[7])
names = ["0 wait for record DMA", "1 LDS reads + scatter + bounds", "2 lgkmcnt(0)", "3 DMA issue", "4 LDL", "5 substitutions", "6 -"]
tot = sum(c[:6])
print("stages stamped", n, "cycles/stage in stamped parts", tot / n)
for i in range(6):
    print(f"{names[i]:34s} {c[i] / n:10.1f} {100.0 * c[i] / tot:5.1f}%")
