"""Per-phase cycle breakdown of the wide-stage KKT kernel (workgroup 0): python tools/wide_profile.py [T] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DTO_PLUGIN_CXXFLAGS"] = (os.environ.get("DTO_PLUGIN_CXXFLAGS", "") + " -DDTO_WIDE_PROFILE=1").strip()   # stamps are compiled in on request only
import torch
prof = torch.zeros(32, dtype=torch.int64, device="cuda")
os.environ["DTO_WIDE_PROF"] = hex(prof.data_ptr())
import dto_amd
from dto_amd import problems as P
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
p = P.build_acrobot_padded(T=T)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot_padded")
nz, nc = s.nlp.num_variables, s.nlp.num_constraint
Z = torch.rand((B, nz), device="cuda", dtype=torch.float64); MU = torch.rand((B, nc), device="cuda", dtype=torch.float64)
dx = torch.empty_like(Z); dl = torch.empty_like(MU)
s.kkt_step_batch(Z.data_ptr(), B, nz, MU.data_ptr(), nc, 2.0, 1e-5, dx.data_ptr(), nz, dl.data_ptr(), nc)
torch.cuda.synchronize(); prof.zero_()
s.kkt_step_batch(Z.data_ptr(), B, nz, MU.data_ptr(), nc, 2.0, 1e-5, dx.data_ptr(), nz, dl.data_ptr(), nc)
torch.cuda.synchronize()
names = ["p0 load+const fill", "p1 model code", "p2 residual", "p3 scatter var/hess", "p3b dyn hess", "p4 gradient rhs", "p5 u-elim",
         "p6 LDL(A)", "p7 trsm F,V + trsv", "p8 M,E'' + store LA", "p9 LDL(M)", "p10 trsm E + trsv", "p11 P' + store", "p11b P' finish",
         "terminal", "bwd load", "bwd lam", "bwd x,u"]
# slots: 0..11 forward phases (slot 3 covers both scatter phases), 12 = P' finish of the last stage .. see kernel
c = prof.cpu().numpy()
tot = c.sum()
print("cycles per stage (clock64 @100MHz units if wall clock) total", tot / (T - 1))
for i in list(range(17)) + [20, 21, 22, 23, 24, 25, 26, 27]:
    print(f"slot {i:2d} {c[i] / (T - 1):10.1f}  {100.0 * c[i] / tot:5.1f}%")
