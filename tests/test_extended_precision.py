"""The extended-precision reference of the T = 1000 step tests (tests/extended_precision.py) pinned on the CPU: on the ORACLE's
K at the BASELINE size (acrobot T = 1000: dimension 9 003) it converges, reproduces a float64 sparse-LU solve to the accuracy
that solve has, drives the residual three orders below float64 rounding, and a system with a known solution (condition 1e6)
comes back to 1e-14 of it where its float64 LU solve is off by 1e-11."""
import numpy as np

from extended_precision import data_sensitivity, residual_extended, solve_extended


def test_longdouble_is_extended_on_this_platform():
    assert np.finfo(np.longdouble).eps < 1e-18


def test_refinement_recovers_a_known_solution_of_an_ill_conditioned_system():
    import scipy.sparse as sp
    from scipy.sparse.linalg import splu
    rng = np.random.default_rng(3)
    n = 400
    # symmetric indefinite matrix with a wide spectrum (1e-4 .. 1e2)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    d = np.concatenate([10.0 ** rng.uniform(-4, 2, n // 2), -(10.0 ** rng.uniform(-4, 2, n - n // 2))])
    K = sp.csc_matrix((Q * d) @ Q.T)
    x_true = rng.standard_normal(n)
    # right-hand side formed in extended precision so that x_true solves the float64 matrix to ~1e-19
    coo = K.tocoo()
    b_ld = np.zeros(n, dtype=np.longdouble)
    np.add.at(b_ld, coo.row, coo.data.astype(np.longdouble) * x_true.astype(np.longdouble)[coo.col])
    x, info = solve_extended(K, np.asarray(b_ld, dtype=np.float64), rhs_ld=b_ld)
    assert info["converged"] and info["iterations"] >= 2
    lu = splu(K).solve(np.asarray(b_ld, dtype=np.float64))
    err_lu = np.max(np.abs(lu - x_true)) / np.max(np.abs(x_true))
    err_ref = float(np.max(np.abs(x - x_true.astype(np.longdouble)))) / np.max(np.abs(x_true))
    print(err_lu, err_ref, info)
    assert err_lu > 30 * err_ref and err_ref < 1e-12, (err_lu, err_ref)     # the float64 solve is visibly off, the refined one is not


def test_on_the_oracles_kkt_matrix_at_the_baseline_size():
    from test_baseline_sizes_gpu import oracle_for, sparse_kkt
    from scipy.sparse.linalg import splu
    onlp = oracle_for("acrobot", 1000)
    rng = np.random.default_rng(0)
    z, mu = rng.random(onlp.num_variables), rng.random(onlp.num_constraint)
    K, rhs, _ = sparse_kkt(onlp, z, mu, 1e-2, 1e-8)
    assert K.shape == (9003, 9003)
    x, info = solve_extended(K, rhs)
    assert info["converged"]
    scale = float(np.max(np.abs(x)))
    lu = splu(K).solve(rhs)
    assert float(np.max(np.abs(lu - x))) <= 1e-11 * scale
    assert info["residual"] <= 1e-3 * residual_extended(K, lu, rhs) + 1e-300
    # half an ulp of noise in every entry of K and the right-hand side: what no float64 evaluation of the derivatives can avoid
    assert data_sensitivity(K, rhs, x, trials=1) <= 1e-12 * scale
