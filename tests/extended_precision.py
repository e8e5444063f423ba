"""An extended-precision reference for a sparse KKT step (test infrastructure).

The float64 sparse-LU solve the T = 1000 step tests compared against is itself only as accurate as the conditioning of K lets
it be, so "GPU step vs LU step" could not tell whose error a 1e-7 difference was.  Here: LU of the float64 matrix as a
preconditioner + iterative refinement with the residual accumulated in np.longdouble (x86: 64-bit mantissa, eps = 1.1e-19)
until the correction is below 1e-13 of the step -- the solution of the float64 system (K, rhs) to ~1e-13 relative, i.e. exact
for the purposes of an 1e-8 bar.  `data_sensitivity` measures what half-ulp noise in the ENTRIES of K and rhs does to that
solution: the floor under any implementation that evaluates the derivatives in float64 with its own rounding.
"""
import numpy as np


def _spmv_ld(rows, cols, data_ld, x_ld, n):
    out = np.zeros(n, dtype=np.longdouble)
    np.add.at(out, rows, data_ld * x_ld[cols])
    return out


def solve_extended(K, rhs, max_iter=40, rtol=1e-13, data_ld=None, rhs_ld=None):
    """x (np.longdouble) with K x = rhs to `rtol` of max|x|; K a scipy sparse matrix (float64 entries, or their longdouble
    override `data_ld` in COO order of K.tocoo()), returns (x, info)."""
    from scipy.sparse.linalg import splu
    assert np.finfo(np.longdouble).eps < 1e-18, "np.longdouble is not extended precision on this platform"
    coo = K.tocoo()
    rows, cols = coo.row, coo.col
    d = coo.data.astype(np.longdouble) if data_ld is None else data_ld
    b = rhs.astype(np.longdouble) if rhs_ld is None else rhs_ld
    lu = splu(K.tocsc())
    x = lu.solve(np.asarray(b, dtype=np.float64)).astype(np.longdouble)
    n = K.shape[0]
    hist = []
    for it in range(max_iter):
        r = b - _spmv_ld(rows, cols, d, x, n)
        dx = lu.solve(np.asarray(r, dtype=np.float64)).astype(np.longdouble)
        x = x + dx
        rel = float(np.max(np.abs(dx)) / max(float(np.max(np.abs(x))), 1e-300))
        hist.append(rel)
        if rel <= rtol:
            break
    r = b - _spmv_ld(rows, cols, d, x, n)
    return x, dict(iterations=len(hist), corrections=hist, residual=float(np.max(np.abs(r))), converged=hist[-1] <= rtol)


def residual_extended(K, x, rhs):
    """max |rhs - K x| with the products accumulated in np.longdouble"""
    coo = K.tocoo()
    r = rhs.astype(np.longdouble) - _spmv_ld(coo.row, coo.col, coo.data.astype(np.longdouble), np.asarray(x).astype(np.longdouble), K.shape[0])
    return float(np.max(np.abs(r)))


def data_sensitivity(K, rhs, x_true, ulps=0.5, seed=0, trials=3):
    """max over `trials` of max|x(K', rhs') - x_true| where every entry of K and rhs is multiplied by (1 + ulps * 2^-52 * r),
    r uniform in [-1, 1] (symmetric: the (i, j) and (j, i) entries get the same factor) -- the solution's sensitivity to
    rounding-level noise in the data, in extended precision."""
    coo = K.tocoo()
    rng = np.random.default_rng(seed)
    worst = 0.0
    lo = np.minimum(coo.row, coo.col).astype(np.int64)
    hi = np.maximum(coo.row, coo.col).astype(np.int64)
    key = lo * K.shape[0] + hi
    uniq, inv = np.unique(key, return_inverse=True)
    for _ in range(trials):
        f = (1.0 + ulps * 2.0 ** -52 * rng.uniform(-1, 1, len(uniq))).astype(np.longdouble)[inv]
        fb = (1.0 + ulps * 2.0 ** -52 * rng.uniform(-1, 1, len(rhs))).astype(np.longdouble)
        xp, info = solve_extended(K, rhs, data_ld=coo.data.astype(np.longdouble) * f, rhs_ld=rhs.astype(np.longdouble) * fb)
        assert info["converged"], info
        worst = max(worst, float(np.max(np.abs(xp - x_true))))
    return worst
