"""The code generator emits several bodies for the same expressions (symbolic/codegen.py:emit_body): plain ones
(eval, jac, hess), fused ones that share subexpressions (eval_jac for k_stage_eval, jac_hess for the sweeps), and the
line-evaluation pair (trig_args + eval_trig) that lets k_linesearch produce sin / cos at its eight trial points by
recurrence.  The bodies are plain C: they are compiled with gcc here and compared numerically -- fused == separate,
eval_trig fed with sin / cos of trig_args == eval, and the kernel's recurrence (csrc/dto_kkt_kernels.hpp: k_linesearch)
restated in numpy reproduces eval at the trial points alpha_max 2^-k.  Rewrites inside the bodies (angle-addition
identity, constants of product chains multiplied out) move values in the last bits only: tolerance 1e-13 relative."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import dto_amd
from dto_amd import problems as P
from dto_amd.symbolic.codegen import emit_body, is_affine, trig_arguments


def _compile(tmp_path, name, funcs):
    # the bodies call DTO_SINCOS: compiled against csrc/dto_math.hpp itself, i.e. the same straight-line sin + cos the device
    # code uses (the header is host-compilable C++; -ffp-contract=off as on the device, where every fma is explicit)
    csrc = os.path.join(os.path.dirname(os.path.abspath(dto_amd.__file__)), "csrc")
    src = ["#include <cmath>", "#include <cstring>", '#include "dto_math.hpp"', 'extern "C" {']
    for fn, params, body in funcs:
        src.append(f"void {fn}({params}) {{\n{body}\n}}")
    src.append("}")
    c = tmp_path / f"{name}.cpp"
    c.write_text("\n".join(src))
    so = tmp_path / f"{name}.so"
    subprocess.run(["g++", "-O1", "-ffp-contract=off", "-shared", "-fPIC", "-I", csrc, "-o", str(so), str(c)], check=True)
    return C.CDLL(str(so))


def _call(lib, fn, ins, outs):
    dp = C.POINTER(C.c_double)
    arrs = [np.ascontiguousarray(a, dtype=float) for a in ins] + [np.zeros(n) for n in outs]
    getattr(lib, fn)(*[a.ctypes.data_as(dp) for a in arrs])
    return arrs[len(ins):]


@pytest.mark.parametrize("model", ["acrobot", "cartpole", "pendulum"])
def test_fused_and_line_bodies_agree_with_the_plain_ones(tmp_path, model):
    p = getattr(P, f"build_{model}")(T=5, evaluate_hessian=True)
    d = p["dynamics"][0]
    va = {"x": "x", "u": "u", "y": "y", "w": "w", "lam": "lam"}
    sig = "const double* x, const double* u, const double* y, const double* w, double* out"
    sigl = "const double* x, const double* u, const double* y, const double* w, const double* lam, double* out"
    ne, nj, nh = len(d.evaluate_expr), len(d.jacobian_expr), len(d.hessian_expr)
    targs = trig_arguments(d.evaluate_expr)
    affine = bool(targs) and all(is_affine(a) for a in targs)
    # implicit midpoint (acrobot): the angles enter as (x + y) / 2 -> affine; explicit Runge-Kutta stages (cartpole) are not,
    # and the plugin then keeps the plain per-trial evaluation (Dyn::NTRIG = 0)
    assert affine == (model == "acrobot") or model == "pendulum"
    funcs = [("eval", sig, emit_body(d.evaluate_expr, "out", va)),
             ("jac", sig, emit_body(d.jacobian_expr, "out", va)),
             ("hess", sigl, emit_body(d.hessian_expr, "out", va)),
             ("eval_jac", sig.replace("double* out", "double* eout, double* jout"),
              emit_body(list(d.evaluate_expr) + list(d.jacobian_expr), [("eout", ne), ("jout", nj)], va)),
             ("jac_hess", sigl.replace("double* out", "double* jout, double* hout"),
              emit_body(list(d.jacobian_expr) + list(d.hessian_expr), [("jout", nj), ("hout", nh)], va)),
             ]
    if affine:
        funcs += [("trig_args", sig, emit_body(targs, "out", va)),
                  ("eval_trig", sig.replace("double* out", "const double* sn, const double* cs, double* out"),
                   emit_body(d.evaluate_expr, "out", va, trig_override={a.id: j for j, a in enumerate(targs)}))]
    lib = _compile(tmp_path, model, funcs)
    rng = np.random.default_rng(3)
    nx, nu, ny = d.num_state, d.num_action, d.num_next_state
    w = np.zeros(max(1, d.num_parameter))
    close = lambda a, b: np.max(np.abs(a - b)) <= 1e-13 * max(1.0, np.max(np.abs(b)))
    for _ in range(20):
        x, u, y, lam = 3 * rng.standard_normal(nx), rng.standard_normal(max(1, nu)), 3 * rng.standard_normal(ny), rng.standard_normal(ny)
        (e,) = _call(lib, "eval", [x, u, y, w], [ne])
        (j,) = _call(lib, "jac", [x, u, y, w], [nj])
        (h,) = _call(lib, "hess", [x, u, y, w, lam], [nh])
        e2, j2 = _call(lib, "eval_jac", [x, u, y, w], [ne, nj])
        j3, h3 = _call(lib, "jac_hess", [x, u, y, w, lam], [nj, nh])
        assert close(e2, e) and close(j2, j) and close(j3, j) and close(h3, h)
        if not affine:
            continue
        (a0,) = _call(lib, "trig_args", [x, u, y, w], [len(targs)])
        (e3,) = _call(lib, "eval_trig", [x, u, y, w, np.sin(a0), np.cos(a0)], [ne])
        assert close(e3, e)
        # the line search: trial k at x + amax 2^-k dx; arguments a0 + 2^(7-k) da, sin / cos by angle addition + doubling
        dx, du, dy = rng.standard_normal(nx), rng.standard_normal(max(1, nu)), rng.standard_normal(ny)
        amax, K = 0.8, 8
        amin = amax / 2 ** (K - 1)
        (a1,) = _call(lib, "trig_args", [x + amin * dx, u + amin * du, y + amin * dy, w], [len(targs)])
        S0, C0, s, c = np.sin(a0), np.cos(a0), np.sin(a1 - a0), np.cos(a1 - a0)
        alpha = amin
        for kk in range(K):
            sn, cs = S0 * c + C0 * s, C0 * c - S0 * s
            xs, us, ys = x + alpha * dx, u + alpha * du, y + alpha * dy
            (et,) = _call(lib, "eval_trig", [xs, us, ys, w, sn, cs], [ne])
            (ex,) = _call(lib, "eval", [xs, us, ys, w], [ne])
            assert np.max(np.abs(et - ex)) <= 1e-12 * max(1.0, np.max(np.abs(ex))), (kk, np.max(np.abs(et - ex)))
            s, c = 2.0 * s * c, 1.0 - 2.0 * s * s
            alpha *= 2.0


def test_affinity_test_rejects_nonlinear_arguments():
    from dto_amd.symbolic import expr as E
    x = [E.var("x", i) for i in range(2)]
    assert is_affine(0.5 * x[0] + 3.0 * x[1] - 1.0)
    assert not is_affine(x[0] * x[1])
    assert not is_affine(E.sin(x[0]) + x[1])
    assert trig_arguments([E.sin(x[0] * x[1]) + E.cos(x[0])])[0].op == E.MUL
