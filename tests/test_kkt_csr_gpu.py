"""dto_kkt_csr_structure / dto_kkt_csr_values_batch: the KKT matrix itself, CSR, reference ordering (the
north star's "scatters nonzeros into a CSR KKT with coalesced HBM writes").  Checked against the oracle's derivatives assembled
densely, and as an A/B of the block-tridiagonal solver: a sparse LU of the exported K must give dto_kkt_step_batch's step."""
import numpy as np
import pytest

from conftest import product_solver

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,T,B", [("pendulum", 6, 3), ("acrobot", 70, 5), ("car", 6, 2), ("cartpole", 5, 2)])
def test_csr_kkt_matches_the_oracle_and_the_block_solver(model, T, B):
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    import torch
    from test_solve_gpu import oracle_for
    s, p = product_solver(model, T)
    n = s.nlp
    nz, nc, nj, nh = n.num_variables, n.num_constraint, n.num_jacobian, int(n.sizes.nnz_hess_key)
    rp, ci = n.kkt_csr_structure()
    dim, nnz = nz + nc, len(ci)
    assert rp[0] == 1 and rp[-1] == nnz + 1 and len(rp) == dim + 1
    for i in range(dim):                                             # columns strictly increasing inside every row, diagonal present
        row = ci[rp[i] - 1:rp[i + 1] - 1]
        assert np.all(np.diff(row) > 0) and (i + 1) in row
    rng = np.random.default_rng(3)
    Z = rng.standard_normal((B, nz))
    MU = rng.standard_normal((B, nc))
    dw, dc = 0.37, 1e-5
    z, mu = torch.tensor(Z, device="cuda"), torch.tensor(MU, device="cuda")
    H = torch.empty((B, nh), device="cuda", dtype=torch.float64)
    J = torch.empty((B, nj), device="cuda", dtype=torch.float64)
    V = torch.full((B, nnz), float("nan"), device="cuda", dtype=torch.float64)
    n.eval_hessian_lagrangian_batch(z.data_ptr(), B, nz, 1.0, mu.data_ptr(), nc, H.data_ptr(), nh)
    n.eval_constraint_jacobian_batch(z.data_ptr(), B, nz, J.data_ptr(), nj)
    n.kkt_csr_values_batch(B, H.data_ptr(), nh, J.data_ptr(), nj, dw, dc, V.data_ptr(), nnz)
    torch.cuda.synchronize()
    Vh = V.cpu().numpy()
    onlp = oracle_for(model, T)
    dxs = torch.empty((B, nz), device="cuda", dtype=torch.float64)
    dls = torch.empty((B, nc), device="cuda", dtype=torch.float64)
    s.kkt_step_batch(z.data_ptr(), B, nz, mu.data_ptr(), nc, dw, dc, dxs.data_ptr(), nz, dls.data_ptr(), nc)
    torch.cuda.synchronize()
    for b in range(B):
        K = sp.csr_matrix((Vh[b], ci - 1, rp - 1), shape=(dim, dim))
        # oracle: dense K from its own Jacobian / Hessian values and structures
        Kd = np.zeros((dim, dim))
        for (r, c), v in zip(onlp.hessian_lagrangian_structure(), onlp.eval_hessian_lagrangian(Z[b], 1.0, MU[b])):
            Kd[r - 1, c - 1] += v
        for (r, c), v in zip(onlp.jacobian_structure(), onlp.eval_constraint_jacobian(Z[b])):
            Kd[nz + r - 1, c - 1] += v
            Kd[c - 1, nz + r - 1] += v
        Kd[np.arange(nz), np.arange(nz)] += dw
        Kd[nz + np.arange(nc), nz + np.arange(nc)] -= dc
        # entries: 1e-8 relative (the tolerance of every Jacobian / Hessian value test)
        assert np.max(np.abs(K.toarray() - Kd)) <= 1e-8 * max(1.0, np.max(np.abs(Kd)))
        # A/B: the exported matrix solved by a general sparse LU gives the step of the block-tridiagonal LDL^T
        rhs = -np.concatenate([onlp.eval_objective_gradient(Z[b]) + (Kd[nz:, :nz].T @ MU[b]), onlp.eval_constraint(Z[b])])
        sol = spla.splu(K.tocsc()).solve(rhs)
        scale = np.max(np.abs(sol))
        assert np.max(np.abs(dxs[b].cpu().numpy() - sol[:nz])) <= 1e-8 * scale
        assert np.max(np.abs(dls[b].cpu().numpy() - sol[nz:])) <= 1e-8 * scale


def test_csr_kkt_at_the_baseline_size_T1000():
    """The same export at BASELINE configs[2] (acrobot T = 1000, dim 9 003; VERDICT r3: "checked only up to T = 70"): every
    stored entry against the oracle's sparse K, the structure against the oracle's (both triangles, reference ordering), and
    the sparse-LU step of the exported matrix against the block-tridiagonal solver."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    import torch
    from test_baseline_sizes_gpu import oracle_for, sparse_kkt
    T, B = 1000, 2
    s, p = product_solver("acrobot", T)
    n = s.nlp
    nz, nc, nj, nh = n.num_variables, n.num_constraint, n.num_jacobian, int(n.sizes.nnz_hess_key)
    rp, ci = n.kkt_csr_structure()
    dim, nnz = nz + nc, len(ci)
    assert dim == 9003 and rp[0] == 1 and rp[-1] == nnz + 1 and len(rp) == dim + 1
    rng = np.random.default_rng(5)
    Z, MU = rng.random((B, nz)), rng.random((B, nc))
    dw, dc = 60.0, 1e-5
    z, mu = torch.tensor(Z, device="cuda"), torch.tensor(MU, device="cuda")
    H = torch.empty((B, nh), device="cuda", dtype=torch.float64)
    J = torch.empty((B, nj), device="cuda", dtype=torch.float64)
    V = torch.full((B, nnz), float("nan"), device="cuda", dtype=torch.float64)
    n.eval_hessian_lagrangian_batch(z.data_ptr(), B, nz, 1.0, mu.data_ptr(), nc, H.data_ptr(), nh)
    n.eval_constraint_jacobian_batch(z.data_ptr(), B, nz, J.data_ptr(), nj)
    n.kkt_csr_values_batch(B, H.data_ptr(), nh, J.data_ptr(), nj, dw, dc, V.data_ptr(), nnz)
    dxs = torch.empty((B, nz), device="cuda", dtype=torch.float64)
    dls = torch.empty((B, nc), device="cuda", dtype=torch.float64)
    s.kkt_step_batch(z.data_ptr(), B, nz, mu.data_ptr(), nc, dw, dc, dxs.data_ptr(), nz, dls.data_ptr(), nc)
    torch.cuda.synchronize()
    Vh = V.cpu().numpy()
    onlp = oracle_for("acrobot", T)
    for b in range(B):
        K = sp.csr_matrix((Vh[b], ci - 1, rp - 1), shape=(dim, dim))
        Ko, rhs, _ = sparse_kkt(onlp, Z[b], MU[b], dw, dc)
        D = (K - Ko.tocsr()).tocoo()
        assert (np.max(np.abs(D.data)) if D.nnz else 0.0) <= 1e-8 * max(1.0, abs(Ko).max())
        # the stored pattern contains the oracle's Hessian key and Jacobian pattern (both triangles) and the whole diagonal
        Po = (abs(Ko) + sp.identity(dim)).tocsr()
        Po.data[:] = 1.0
        Pk = K.copy()
        Pk.data[:] = 1.0
        assert (Po - Po.multiply(Pk)).count_nonzero() == 0
        sol = spla.splu(K.tocsc()).solve(rhs)
        scale = np.max(np.abs(sol))
        assert np.max(np.abs(dxs[b].cpu().numpy() - sol[:nz])) <= 1e-8 * scale
        assert np.max(np.abs(dls[b].cpu().numpy() - sol[nz:])) <= 1e-8 * scale
