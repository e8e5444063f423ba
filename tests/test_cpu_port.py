"""The oracle's serial C port (CPU baseline) on its own: it converges on the equality-constrained configs and
its final point satisfies the KKT conditions computed with the numpy/sympy oracle evaluator."""
import numpy as np

from oracle import dto_oracle as O, sympy_models as S
from oracle.cpu_port import PortSolver, acrobot_guesses


def test_port_converges_and_satisfies_kkt():
    T = 60
    Z, x1, xT = acrobot_guesses(T, 3, seed=5)
    p = S.build("acrobot", T, evaluate_hessian=False)
    onlp = O.NLPData(p["dynamics"], p["objective"], p["constraints"], p["bounds"])
    s = PortSolver("acrobot", T, x1, xT)
    for b in range(3):
        assert s.solve(Z[b]) == 1
        z, lam = s.z, s.lam
        c = onlp.eval_constraint(z)
        assert np.max(np.abs(c)) <= 1e-6
        J = np.zeros((onlp.num_constraint, onlp.num_variables))
        for (r, cc), v in zip(onlp.jacobian_structure(), onlp.eval_constraint_jacobian(z)):
            J[r - 1, cc - 1] = v
        g = onlp.eval_objective_gradient(z)
        assert np.max(np.abs(g + J.T @ lam)) <= 1e-5          # multipliers are in the reference order
        assert np.linalg.norm(z[:4] - x1) < 1e-3 and np.linalg.norm(z[-4:] - xT) < 1e-3   # test/solve.jl:136-137
