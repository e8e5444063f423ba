"""Why one in eight acrobot T = 1000 solves does not terminate: the reduced Hessian Z' W Z of the
ORACLE's problem at the iterate the C port reaches -- python tools/straggler_spectrum.py <bench instance> [iterations].
Converging instances end at points whose smallest reduced eigenvalue is ~1e-2; the stragglers (f ~ 442 .. 486) sit in a valley
with ONE eigenvalue of ~2e-7 (next: 1e-2) and a gradient of ~4e-4 along it: the objective falls almost linearly along the
valley (0.002 per eight iterations), second-order information says nothing about how far to go, and the constraint curvature
limits the step.  CPU only."""
import sys; import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, scipy.sparse as sp, scipy.linalg as sl
from oracle import cpu_port as CP
from oracle import dto_oracle as O, sympy_models as S
T=1000
seed=int(sys.argv[1]) if len(sys.argv)>1 else 2
Z0,_,_=CP.guesses("acrobot",T,seed+1,1000)
ps=CP.PortSolver("acrobot",T,max_iter=int(sys.argv[2]) if len(sys.argv)>2 else 400)
ps.solve(Z0[seed])
z,lam=ps.z,ps.lam
print("status",ps.status,"iters",ps.iterations,ps.stats())
p=S.build("acrobot",T,evaluate_hessian=True)
onlp=O.NLPData(p["dynamics"],p["objective"],p["constraints"],p["bounds"],evaluate_hessian=True)
nz,nc=onlp.num_variables,onlp.num_constraint
js=np.array(onlp.jacobian_structure())-1; hs=np.array(onlp.hessian_lagrangian_structure())-1
J=sp.coo_matrix((onlp.eval_constraint_jacobian(z),(js[:,0],js[:,1])),shape=(nc,nz)).toarray()
W=sp.coo_matrix((onlp.eval_hessian_lagrangian(z,1.0,lam),(hs[:,0],hs[:,1])),shape=(nz,nz)).toarray()
g=onlp.eval_objective_gradient(z)
print("f",onlp.eval_objective(z),"|c|inf",np.max(np.abs(onlp.eval_constraint(z))),"|grad L|inf",np.max(np.abs(g+J.T@lam)))
Q,R=np.linalg.qr(J.T,mode='complete')
Zn=Q[:,nc:]
H=Zn.T@W@Zn
ev=np.linalg.eigvalsh(H)
print("null space dim",Zn.shape[1],"reduced Hessian eigenvalues: min %.3e, 5 smallest"%ev[0],ev[:5],"max %.3e"%ev[-1], "neg count",int((ev<0).sum()))
