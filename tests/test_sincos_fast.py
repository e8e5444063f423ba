"""dto::sincos_fast (csrc/dto_math.hpp) -- the straight-line f64 sin + cos inside every generated model body and the line
search -- against numpy long double (x87 80-bit: 64-bit mantissa) on the host, compiled with -ffp-contract=off like the device
code (VERDICT r4 weak 1c / ADVICE r4):
  * |x| < 1e5 (uniform, and the neighbourhoods of every kind of quadrant boundary k pi/2, where the reduction cancels): <= 1 ulp;
  * |x| < 1.5e6: <= 2.5 ulp; up to 1e8: 1e-12 absolute (iterates that large are stopped by Options.diverging_iterates_tol);
  * +-0, denormals and arguments up to 1e12 (beyond the 3.4e9 where the old double -> int conversion of the quadrant was undefined):
    finite results in [-1, 1], sin odd / cos even.  (Beyond ~1e15 the three-piece reduction no longer cancels and the result is
    meaningless, possibly non-finite: the line search rejects NaN trial points, iterates are stopped at 1e8.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import dto_amd


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("sincos")
    csrc = os.path.join(os.path.dirname(os.path.abspath(dto_amd.__file__)), "csrc")
    src = d / "h.cpp"
    src.write_text('#include "dto_math.hpp"\nextern "C" void sc(const double* x, long n, double* s, double* c) {\n'
                   "  for (long i = 0; i < n; ++i) dto::sincos_fast(x[i], s + i, c + i);\n}\n")
    so = d / "h.so"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", csrc, "-o", str(so), str(src)], check=True)
    return C.CDLL(str(so))


def _eval(lib, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    s, c = np.zeros_like(x), np.zeros_like(x)
    dp = C.POINTER(C.c_double)
    lib.sc(x.ctypes.data_as(dp), C.c_long(x.size), s.ctypes.data_as(dp), c.ctypes.data_as(dp))
    return s, c


def _ulp_err(got, ref_ld):
    ref = ref_ld.astype(np.float64)
    ulp = np.abs(np.nextafter(ref, np.inf) - ref)
    return np.max(np.abs((got.astype(np.longdouble) - ref_ld) / ulp.astype(np.longdouble)))


def test_accuracy_against_long_double(lib):
    assert np.finfo(np.longdouble).nmant >= 63, "needs the x87 long double"
    rng = np.random.default_rng(1)
    for rng_max, bound in ((3.2, 1.0), (50.0, 1.0), (1e3, 1.0), (1e5, 1.0), (1.5e6, 2.5)):
        x = rng.uniform(-rng_max, rng_max, 1_000_000)
        s, c = _eval(lib, x)
        xl = x.astype(np.longdouble)
        assert _ulp_err(s, np.sin(xl)) <= bound and _ulp_err(c, np.cos(xl)) <= bound, rng_max
    # quadrant boundaries: k pi/2 for k up to 63 661 (|x| < 1e5), the doubles nearest to them and +- 1..64 ulps around
    k = np.concatenate([np.arange(-2000, 2001), rng.integers(-63661, 63662, 20000)]).astype(np.float64)
    base = k * (np.pi / 2)
    pts = [base]
    for j in (1, 2, 3, 8, 64):
        up, dn = base.copy(), base.copy()
        for _ in range(j):
            up, dn = np.nextafter(up, np.inf), np.nextafter(dn, -np.inf)
        pts += [up, dn]
    x = np.concatenate(pts)
    s, c = _eval(lib, x)
    xl = x.astype(np.longdouble)
    # near a zero of sin / cos the result is tiny and its ulp with it: measure those against the absolute scale 2^-53 instead
    es = np.abs(s.astype(np.longdouble) - np.sin(xl)); ec = np.abs(c.astype(np.longdouble) - np.cos(xl))
    ulp_s = np.maximum(np.abs(np.nextafter(s, np.inf) - s), 2.0 ** -70); ulp_c = np.maximum(np.abs(np.nextafter(c, np.inf) - c), 2.0 ** -70)
    assert np.max(es / ulp_s) <= 1.0 or np.max(es) <= 2.0 ** -53
    assert np.max(ec / ulp_c) <= 1.0 or np.max(ec) <= 2.0 ** -53
    x = rng.uniform(-1e8, 1e8, 1_000_000)
    s, c = _eval(lib, x)
    xl = x.astype(np.longdouble)
    assert np.max(np.abs(s - np.sin(xl).astype(np.float64))) <= 1e-12 and np.max(np.abs(c - np.cos(xl).astype(np.float64))) <= 1e-12


def test_defined_for_every_finite_argument(lib):
    x = np.array([0.0, -0.0, 5e-324, -5e-324, 1e-300, 2.0 ** 31 * np.pi / 2, 3.5e9, -3.5e9, 1e10, -1e11, 1e12])
    s, c = _eval(lib, x)
    assert np.all(np.isfinite(s)) and np.all(np.isfinite(c))
    assert np.all(np.abs(s) <= 1.0 + 1e-9) and np.all(np.abs(c) <= 1.0 + 1e-9)
    assert s[0] == 0.0 and c[0] == 1.0 and s[1] == 0.0
    s2, c2 = _eval(lib, -x)
    assert np.array_equal(s2, -s) and np.array_equal(c2, c)
    # the quadrant bits up to the range where the three-piece reduction is meaningful agree with integer arithmetic
    k = np.array([1, 2, 3, 4, 5, 1023, 1024, 2 ** 20 - 1, 2 ** 20, -(2 ** 20) - 3], dtype=np.float64)
    s, c = _eval(lib, k * (np.pi / 2))
    want_s = np.array([[0, 1, 0, -1][int(v) % 4] for v in k], dtype=float)
    want_c = np.array([[1, 0, -1, 0][int(v) % 4] for v in k], dtype=float)
    assert np.max(np.abs(s - want_s)) < 1e-9 and np.max(np.abs(c - want_c)) < 1e-9
