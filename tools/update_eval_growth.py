"""Does k_update_eval slow down because instances FINISH (tiles mixing finished and running lanes) or because of where the running
ones are?  Same batch twice under the kernel tracer, once with the reference tolerances and once with tol = 1e-30 (nothing
converges):  rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/update_eval_growth.py <tol> [B] [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dto_amd
from dto_amd import problems as P
from bench import make_guesses_device
tol = float(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 131072; K = int(sys.argv[3]) if len(sys.argv) > 3 else 34
p = P.build_acrobot(T=1000, evaluate_hessian=True)
o = dto_amd.Options(tol=tol, dual_inf_tol=tol if tol < 1e-6 else 1.0, constr_viol_tol=tol if tol < 1e-6 else 1e-3, acceptable_iter=0)
s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, options=o, name="acrobot")
nz = s.nlp.num_variables
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
z0 = make_guesses_device(s, p, B, 1000, dev)
s.begin_batch(z0.data_ptr(), B, nz, stream=st)
s.iterate_batch(K, stream=st)
torch.cuda.synchronize()
stt = s.scalar_batch("status")[:B]
print(f"tol={tol} B={B}: finished after {K} iterations: {int(np.sum(stt != 0))}", flush=True)
