"""ORACLE (test infrastructure only -- never imported by the product path).

Independent restatement, in sympy, of the model equations the reference examples/tests define, and
of what Symbolics.jl does with them in the reference constructors (Symbolics.jl 0.1.29-0.1.32 is a
third-party dependency that is NOT vendored under /root/reference; Project.toml:14,22):

    Symbolics.gradient / sparsejacobian / sparsehessian + build_function
        call sites: src/costs.jl:18-28, src/dynamics.jl:23-36, src/constraints.jl:27-41,
                    src/general_constraint.jl:23-37

Published algorithm restated: `sparsejacobian(f, vars)` returns the SparseMatrixCSC of df_i/dv_j
whose structural pattern holds (i, j) when v_j occurs in f_i; `sparsehessian(f, vars)` returns the
full symmetric sparse Hessian; `findnz` lists entries in CSC (column-major) order.  Here patterns
are computed from sympy itself: Jacobian by `free_symbols` occurrence (sympy folds `0*e` like
SymbolicUtils), Hessian by the *numerically non-vanishing* second derivatives at random
30-digit points -- a different route from the product's linearity propagation, so agreement of the
two is a real check.  PARITY NOTE: Symbolics' own structural rules on corner cases cannot be
exercised in this environment (no Julia); see DESIGN.md "parity status".

Model equations follow, line by line:
    pendulum  examples/pendulum/pendulum.jl:22-39      acrobot  examples/acrobot/acrobot.jl:19-91
    cartpole  examples/cartpole/cartpole.jl:19-56      car      examples/car/car.jl:19-26
    test pendulum / implicit Euler   test/dynamics.jl:8-19
    double integrator                test/solve.jl:149-183
"""
from __future__ import annotations

import math
import random

import mpmath
import numpy as np
import sympy as sp

mpmath.mp.dps = 30


def syms(name, n):
    return [sp.Symbol(f"{name}{i + 1}", real=True) for i in range(n)]


R = sp.Rational
F = sp.Float


def fl(v):
    """Exact float constant as sympy Float at full double precision (so 0.05 is the double 0.05)."""
    return sp.Float(float(v), 17)


# ------------------------------------------------------------------------------------ pendulum
def pendulum(x, u, w):
    mass, length_com, gravity, damping = fl(1.0), fl(0.5), fl(9.81), fl(0.1)
    return [x[1],
            u[0] / (mass * length_com * length_com) - gravity * sp.sin(x[0]) / length_com
            - damping * x[1] / (mass * length_com * length_com)]


def midpoint(f, h):
    def dyn(y, x, u, w):
        xm = [fl(0.5) * (a + b) for a, b in zip(x, y)]
        fx = f(xm, u, w)
        return [yi - (xi + fl(h) * fi) for yi, xi, fi in zip(y, x, fx)]
    return dyn


pendulum_midpoint = midpoint(pendulum, 0.05)


def pendulum_test(z, u, w):
    mass, lc, gravity, damping = fl(1.0), fl(1.0), fl(9.81), fl(0.1)
    return [z[1], u[0] / (mass * lc * lc) - gravity * sp.sin(z[0]) / lc - damping * z[1] / (mass * lc * lc)]


def euler_implicit_test(y, x, u, w):
    fy = pendulum_test(y, u, w)
    return [yi - (xi + fl(0.1) * fi) for yi, xi, fi in zip(y, x, fy)]


# ------------------------------------------------------------------------------------ cartpole
def cartpole(x, u, w):
    mc, mp, l, g = fl(1.0), fl(0.2), fl(0.5), fl(9.81)
    q = x[0:2]
    qd = x[2:4]
    s = sp.sin(q[1])
    c = sp.cos(q[1])
    H = sp.Matrix([[mc + mp, mp * l * c], [mp * l * c, mp * l ** 2]])
    Hinv = (1 / (H[0, 0] * H[1, 1] - H[0, 1] * H[1, 0])) * sp.Matrix([[H[1, 1], -H[0, 1]], [-H[1, 0], H[0, 0]]])
    Cm = sp.Matrix([[0, -mp * qd[1] * l * s], [0, 0]])
    G = sp.Matrix([0, mp * g * l * s])
    Bm = sp.Matrix([1, 0])
    qdd = -Hinv * (Cm * sp.Matrix(qd) + G - Bm * u[0])
    return [qd[0], qd[1], qdd[0], qdd[1]]


def cartpole_rk3_explicit(x, u, w):
    h = fl(0.05)
    k1 = [h * v for v in cartpole(x, u, w)]
    k2 = [h * v for v in cartpole([a + fl(0.5) * b for a, b in zip(x, k1)], u, w)]
    k3 = [h * v for v in cartpole([a - b + fl(2.0) * c for a, b, c in zip(x, k1, k2)], u, w)]
    return [a + (b + fl(4.0) * c + d) / fl(6.0) for a, b, c, d in zip(x, k1, k2, k3)]


def cartpole_rk3_implicit(y, x, u, w):
    return [a - b for a, b in zip(y, cartpole_rk3_explicit(x, u, w))]


# ------------------------------------------------------------------------------------ acrobot
def acrobot(x, u, w):
    mass1, inertia1, length1, lengthcom1 = fl(1.0), fl(0.33), fl(1.0), fl(0.5)
    mass2, inertia2, length2, lengthcom2 = fl(1.0), fl(0.33), fl(1.0), fl(0.5)
    gravity, friction1, friction2 = fl(9.81), fl(0.1), fl(0.1)

    def Minv(q):
        a = inertia1 + inertia2 + mass2 * length1 * length1 + fl(2.0) * mass2 * length1 * lengthcom2 * sp.cos(q[1])
        b = inertia2 + mass2 * length1 * lengthcom2 * sp.cos(q[1])
        c = inertia2
        return (1 / (a * c - b * b)) * sp.Matrix([[c, -b], [-b, a]])

    def tau(q):
        a = (-fl(1.0) * mass1 * gravity * lengthcom1 * sp.sin(q[0])
             - mass2 * gravity * (length1 * sp.sin(q[0]) + lengthcom2 * sp.sin(q[0] + q[1])))
        b = -fl(1.0) * mass2 * gravity * lengthcom2 * sp.sin(q[0] + q[1])
        return sp.Matrix([a, b])

    def Cmat(x):
        a = -fl(2.0) * mass2 * length1 * lengthcom2 * sp.sin(x[1]) * x[3]
        b = -fl(1.0) * mass2 * length1 * lengthcom2 * sp.sin(x[1]) * x[3]
        c = mass2 * length1 * lengthcom2 * sp.sin(x[1]) * x[2]
        return sp.Matrix([[a, b], [c, 0]])

    q = x[0:2]
    v = sp.Matrix(x[2:4])
    Bv = sp.Matrix([0, 1])
    fr = sp.Matrix([friction1 * v[0], friction2 * v[1]])
    qdd = Minv(q) * (-Cmat(x) * v + tau(q) + Bv * u[0] - fr)
    return [x[2], x[3], qdd[0], qdd[1]]


acrobot_midpoint = midpoint(acrobot, 0.05)


# ------------------------------------------------------------------------------------ car
def car(x, u, w):
    return [u[0] * sp.cos(x[2]), u[0] * sp.sin(x[2]), u[1]]


car_midpoint = midpoint(car, 0.1)


# ------------------------------------------------------------------------------------ double integrator
def double_integrator(y, x, u, w):
    return [y[0] - (x[0] + x[1]), y[1] - (x[1] + u[0])]


# ======================================================================================
# "Symbolics" stand-in
# ======================================================================================
def _rand_point(symbols, rng):
    return {s: mpmath.mpf(rng.uniform(0.1, 1.3)) for s in symbols}


def sparse_jacobian(f, wrt):
    """rows, cols (1-based, CSC order), symbolic values."""
    pat = []
    for i, fi in enumerate(f):
        fs = sp.sympify(fi).free_symbols
        for j, v in enumerate(wrt):
            if v in fs:
                pat.append((i, j))
    pat.sort(key=lambda rc: (rc[1], rc[0]))
    rows = [r + 1 for r, _ in pat]
    cols = [c + 1 for _, c in pat]
    vals = [sp.diff(f[r], wrt[c]) for r, c in pat]
    return rows, cols, vals


def sparse_hessian(L, wrt, extra_syms=()):
    """Full symmetric sparse Hessian: entries that do not vanish numerically, CSC order."""
    L = sp.sympify(L)
    n = len(wrt)
    grad = [sp.diff(L, v) for v in wrt]
    H = {}
    for i in range(n):
        for j in range(i, n):
            H[(i, j)] = sp.diff(grad[i], wrt[j])
    rng = random.Random(12345)
    allsyms = sorted(set(wrt) | set(extra_syms) | L.free_symbols, key=lambda s: s.name)
    nz = set()
    cand = [k for k, e in H.items() if e != 0]
    if cand:
        fns = {k: sp.lambdify(allsyms, H[k], modules="mpmath") for k in cand}
        for _ in range(3):
            pt = _rand_point(allsyms, rng)
            args = [pt[s] for s in allsyms]
            for k in cand:
                if abs(fns[k](*args)) > mpmath.mpf(10) ** (-22):
                    nz.add(k)
    pat = set()
    for (i, j) in nz:
        pat.add((i, j))
        pat.add((j, i))
    pat = sorted(pat, key=lambda rc: (rc[1], rc[0]))
    rows = [r + 1 for r, _ in pat]
    cols = [c + 1 for _, c in pat]
    vals = [H[(r, c) if r <= c else (c, r)] for r, c in pat]
    return rows, cols, vals


class _Fn:
    """Numeric closures of a symbolic vector: float64 (numpy) and 30-digit (mpmath)."""

    def __init__(self, args, exprs):
        self.n = len(exprs)
        self.args = args
        exprs = [sp.sympify(e) for e in exprs]
        self._np = sp.lambdify(args, exprs, modules="math", cse=True) if exprs else None
        self._mp = sp.lambdify(args, exprs, modules="mpmath") if exprs else None

    def __call__(self, *vals, hp=False):
        if self.n == 0:
            return np.zeros(0) if not hp else []
        flat = [v for group in vals for v in group]
        if hp:
            return [mpmath.mpf(v) for v in self._mp(*[mpmath.mpf(float(x)) if not isinstance(x, mpmath.mpf) else x for x in flat])]
        return np.array(self._np(*[float(x) for x in flat]), dtype=float)


class Cost:
    """src/costs.jl:13-45"""

    def __init__(self, f, num_state, num_action, num_parameter=0, evaluate_hessian=False):
        x, u, w = syms("x", num_state), syms("u", num_action), syms("w", num_parameter)
        self.f = f   # kept for oracle/cpu_port/gen_model_c.py (C restatement of the same closures)
        ev = sp.sympify(f(x, u, w))
        wrt = x + u
        args = x + u + w
        self.num_state, self.num_action, self.num_parameter = num_state, num_action, num_parameter
        self.num_gradient = num_state + num_action
        self.evaluate = _Fn(args, [ev])
        self.gradient = _Fn(args, [sp.diff(ev, v) for v in wrt])
        if evaluate_hessian:
            r, c, v = sparse_hessian(ev, wrt)
            self.sparsity = [r, c]
            self.hessian = _Fn(args, v)
        else:
            self.sparsity = [[], []]
            self.hessian = _Fn(args, [])
        self.num_hessian = len(self.sparsity[0])


class Dynamics:
    """src/dynamics.jl:18-57"""

    def __init__(self, f, num_next_state, num_state, num_action, num_parameter=0, evaluate_hessian=False):
        y, x, u, w = syms("y", num_next_state), syms("x", num_state), syms("u", num_action), syms("w", num_parameter)
        self.f = f
        ev = [sp.sympify(e) for e in f(y, x, u, w)]
        wrt = x + u + y
        args = y + x + u + w
        self.num_next_state, self.num_state, self.num_action, self.num_parameter = num_next_state, num_state, num_action, num_parameter
        self.evaluate = _Fn(args, ev)
        r, c, v = sparse_jacobian(ev, wrt)
        self.jacobian_sparsity = [r, c]
        self.jacobian = _Fn(args, v)
        self.num_jacobian = len(v)
        if evaluate_hessian:
            lam = syms("lam", num_next_state)
            L = sum(l * e for l, e in zip(lam, ev))
            r, c, v = sparse_hessian(L, wrt, lam)
            self.hessian_sparsity = [r, c]
            self.hessian = _Fn(args + lam, v)
        else:
            self.hessian_sparsity = [[], []]
            self.hessian = _Fn(args, [])
        self.num_hessian = len(self.hessian_sparsity[0])


class Constraint:
    """src/constraints.jl:21-78"""

    def __init__(self, f=None, num_state=0, num_action=0, num_parameter=0, indices_inequality=(), evaluate_hessian=False):
        self.num_state, self.num_action, self.num_parameter = num_state, num_action, num_parameter
        self.indices_inequality = list(indices_inequality)
        self.f = f
        if f is None:
            self.num_constraint = self.num_jacobian = self.num_hessian = 0
            self.jacobian_sparsity = [[], []]
            self.hessian_sparsity = [[], []]
            self.evaluate = self.jacobian = self.hessian = _Fn([], [])
            return
        x, u, w = syms("x", num_state), syms("u", num_action), syms("w", num_parameter)
        ev = [sp.sympify(e) for e in f(x, u, w)]
        wrt = x + u
        args = x + u + w
        self.num_constraint = len(ev)
        self.evaluate = _Fn(args, ev)
        r, c, v = sparse_jacobian(ev, wrt)
        self.jacobian_sparsity = [r, c]
        self.jacobian = _Fn(args, v)
        self.num_jacobian = len(v)
        if evaluate_hessian:
            lam = syms("lam", self.num_constraint)
            L = sum(l * e for l, e in zip(lam, ev))
            r, c, v = sparse_hessian(L, wrt, lam)
            self.hessian_sparsity = [r, c]
            self.hessian = _Fn(args + lam, v)
        else:
            self.hessian_sparsity = [[], []]
            self.hessian = _Fn(args, [])
        self.num_hessian = len(self.hessian_sparsity[0])


class GeneralConstraint:
    """src/general_constraint.jl:18-71"""

    def __init__(self, f=None, num_variables=0, num_parameter=0, indices_inequality=(), evaluate_hessian=False):
        self.num_variables, self.num_parameter = num_variables, num_parameter
        self.indices_inequality = list(indices_inequality)
        if f is None:
            self.num_constraint = self.num_jacobian = self.num_hessian = 0
            self.jacobian_sparsity = [[], []]
            self.hessian_sparsity = [[], []]
            self.evaluate = self.jacobian = self.hessian = _Fn([], [])
            return
        z, w = syms("z", num_variables), syms("w", num_parameter)
        ev = [sp.sympify(e) for e in f(z, w)]
        self.num_constraint = len(ev)
        self.evaluate = _Fn(z + w, ev)
        r, c, v = sparse_jacobian(ev, z)
        self.jacobian_sparsity = [r, c]
        self.jacobian = _Fn(z + w, v)
        self.num_jacobian = len(v)
        self.hessian_sparsity = [[], []]
        self.hessian = _Fn([], [])
        self.num_hessian = 0
        if evaluate_hessian:
            lam = syms("lam", self.num_constraint)
            L = sum(l * e for l, e in zip(lam, ev))
            r, c, v = sparse_hessian(L, z, lam)
            self.hessian_sparsity = [r, c]
            self.hessian = _Fn(z + w + lam, v)
            self.num_hessian = len(v)


class Bound:
    """src/bounds.jl:8-14"""

    def __init__(self, num_state=0, num_action=0, state_lower=None, state_upper=None, action_lower=None, action_upper=None):
        inf = math.inf
        self.state_lower = np.full(num_state, -inf) if state_lower is None else np.asarray(state_lower, float)
        self.state_upper = np.full(num_state, inf) if state_upper is None else np.asarray(state_upper, float)
        self.action_lower = np.full(num_action, -inf) if action_lower is None else np.asarray(action_lower, float)
        self.action_upper = np.full(num_action, inf) if action_upper is None else np.asarray(action_upper, float)


def dot(a, b):
    return sum(p * q for p, q in zip(a, b))


PI = math.pi


def build(name, T, evaluate_hessian=True):
    """Problem objects of the BASELINE configs, same argument meaning as the examples."""
    if name == "pendulum":
        n, m = 2, 1
        x1, xT = [0.0, 0.0], [PI, 0.0]
        dt = Dynamics(pendulum_midpoint, n, n, m, evaluate_hessian=evaluate_hessian)
        ct = Cost(lambda x, u, w: fl(0.1) * dot(x[0:2], x[0:2]) + fl(0.1) * dot(u, u), n, m, evaluate_hessian=evaluate_hessian)
        cT = Cost(lambda x, u, w: fl(0.1) * dot(x[0:2], x[0:2]), n, 0, evaluate_hessian=evaluate_hessian)
        con1 = Constraint(lambda x, u, w: [a - fl(b) for a, b in zip(x, x1)], n, m, evaluate_hessian=evaluate_hessian)
        conT = Constraint(lambda x, u, w: [a - fl(b) for a, b in zip(x, xT)], n, 0, evaluate_hessian=evaluate_hessian)
        cons = [con1] + [Constraint() for _ in range(T - 2)] + [conT]
        bnds = [Bound(n, m)] * (T - 1) + [Bound(n, 0)]
    elif name == "cartpole":
        n, m = 4, 1
        x1, xT = [0.0] * 4, [0.0, PI, 0.0, 0.0]
        Q, Rr, Qf = fl(1.0e-2), fl(1.0e-1), fl(1.0e2)
        dt = Dynamics(cartpole_rk3_implicit, n, n, m, evaluate_hessian=evaluate_hessian)
        dx = lambda x: [a - fl(b) for a, b in zip(x, xT)]
        ct = Cost(lambda x, u, w: fl(0.5) * Q * dot(dx(x), dx(x)) + fl(0.5) * Rr * dot(u, u), n, m, evaluate_hessian=evaluate_hessian)
        cT = Cost(lambda x, u, w: fl(0.5) * Qf * dot(dx(x), dx(x)), n, 0, evaluate_hessian=evaluate_hessian)
        con1 = Constraint(lambda x, u, w: [a - fl(b) for a, b in zip(x, x1)], n, m, evaluate_hessian=evaluate_hessian)
        conT = Constraint(lambda x, u, w: dx(x), n, 0, evaluate_hessian=evaluate_hessian)
        cons = [con1] + [Constraint() for _ in range(T - 2)] + [conT]
        bnd = Bound(n, m, action_lower=[-3.0], action_upper=[3.0])
        bnds = [bnd] * (T - 1) + [Bound(n, 0)]
    elif name in ("acrobot", "acrobot_bounds"):
        n, m = 4, 1
        x1 = [0.0] * 4
        dt = Dynamics(acrobot_midpoint, n, n, m, evaluate_hessian=evaluate_hessian)
        ct = Cost(lambda x, u, w: fl(0.1) * dot(x[2:4], x[2:4]) + fl(0.1) * dot(u, u), n, m, evaluate_hessian=evaluate_hessian)
        cT = Cost(lambda x, u, w: fl(0.1) * dot(x[2:4], x[2:4]), n, 0, evaluate_hessian=evaluate_hessian)
        if name == "acrobot":
            xT = [PI, 0.0, 0.0, 0.0]
            con1 = Constraint(lambda x, u, w: [a - fl(b) for a, b in zip(x, x1)], n, m, evaluate_hessian=evaluate_hessian)
            conT = Constraint(lambda x, u, w: [a - fl(b) for a, b in zip(x, xT)], n, 0, evaluate_hessian=evaluate_hessian)
            cons = [con1] + [Constraint() for _ in range(T - 2)] + [conT]
            bnds = [Bound(n, m)] * (T - 1) + [Bound(n, 0)]
        else:
            xT = [0.0, PI, 0.0, 0.0]
            cons = [Constraint() for _ in range(T)]
            bnds = [Bound(n, m, state_lower=x1, state_upper=x1)] + [Bound(n, m)] * (T - 2) + [Bound(n, 0, state_lower=xT, state_upper=xT)]
    elif name == "car":
        n, m = 3, 2
        x1, xT = [0.0] * 3, [1.0, 1.0, 0.0]
        dt = Dynamics(car_midpoint, n, n, m, evaluate_hessian=evaluate_hessian)
        dx = lambda x: [a - fl(b) for a, b in zip(x, xT)]
        ct = Cost(lambda x, u, w: fl(0.0) * dot(dx(x), dx(x)) + fl(1.0) * dot(u, u), n, m, evaluate_hessian=evaluate_hessian)
        cT = Cost(lambda x, u, w: fl(0.0) * dot(dx(x), dx(x)), n, 0, evaluate_hessian=evaluate_hessian)
        lo, hi = [-0.5] * m, [0.5] * m
        bnds = ([Bound(n, m, state_lower=x1, state_upper=x1, action_lower=lo, action_upper=hi)]
                + [Bound(n, m, action_lower=lo, action_upper=hi)] * (T - 2) + [Bound(n, 0, state_lower=xT, state_upper=xT)])
        p_obs, r_obs = [0.5, 0.5], 0.1

        def obs(x, u, w):
            e = [x[0] - fl(p_obs[0]), x[1] - fl(p_obs[1])]
            return [fl(r_obs) ** 2 - dot(e, e)]

        cont = Constraint(obs, n, m, indices_inequality=[1], evaluate_hessian=evaluate_hessian)
        conT = Constraint(obs, n, 0, indices_inequality=[1], evaluate_hessian=evaluate_hessian)
        cons = [cont] * (T - 1) + [conT]
    else:
        raise KeyError(name)
    return dict(dynamics=[dt] * (T - 1), objective=[ct] * (T - 1) + [cT], constraints=cons, bounds=bnds, T=T, n=n, m=m)


def build_coupled(name, T=None, total=None, inequality=True, u_max=None, evaluate_hessian=True, nonlinear=False):
    """Problems with a GeneralConstraint row that couples two knots (src/general_constraint.jl:18-59), restated from their
    descriptions for the tests of the bordered and the accumulator paths:
      "pendulum_coupled"     pendulum swing-up (build("pendulum")) + theta_15 + theta_35 - total (<= 0 | = 0), or with nonlinear=True
                             sin(theta_15) + theta_35^2 - total; optionally
                             |u| <= u_max at every knot (examples/cartpole/cartpole.jl:81-89 style bounds beside the general row);
      "ref_general_coupled"  test/solve.jl:227-296 (double integrator, T = 11, x1 fixed by bounds, rows z[end-1:end] - xT) + the row
                             x_4[1] + x_8[1] - total (default 0.9; <= 0 with inequality=True)."""
    if name == "pendulum_coupled":
        T = 50 if T is None else T
        p = build("pendulum", T, evaluate_hessian=evaluate_hessian)
        n, m = 2, 1
        nz = n * T + m * (T - 1)
        i15, i35 = 14 * (n + m), 34 * (n + m)
        tot = fl(1.0 if total is None else total)
        row = (lambda z, w: [sp.sin(z[i15]) + z[i35] ** 2 - tot]) if nonlinear else (lambda z, w: [z[i15] + z[i35] - tot])
        p["general_constraint"] = GeneralConstraint(row, nz, 0, indices_inequality=([1] if inequality else []),
                                                    evaluate_hessian=evaluate_hessian)
        if u_max is not None:
            p["bounds"] = [Bound(n, m, action_lower=[-u_max], action_upper=[u_max])] * (T - 1) + [Bound(n, 0)]
        return p
    if name == "ref_general_coupled":
        T, n, m = 11, 2, 1
        x1, xT = [0.0, 0.0], [1.0, 0.0]
        dt = Dynamics(double_integrator, n, n, m, evaluate_hessian=evaluate_hessian)
        ct = Cost(lambda x, u, w: fl(0.1) * dot(x, x) + fl(0.1) * dot(u, u), n, m, evaluate_hessian=evaluate_hessian)
        cT = Cost(lambda x, u, w: fl(0.1) * dot(x, x), n, 0, evaluate_hessian=evaluate_hessian)
        nz = n * T + m * (T - 1)
        i4, i8 = 3 * (n + m), 7 * (n + m)
        tot = fl(0.9 if total is None else total)
        ineq = total is not None and inequality
        gc = GeneralConstraint(lambda z, w: [z[nz - 2] - fl(xT[0]), z[nz - 1] - fl(xT[1]), z[i4] + z[i8] - tot], nz, 0,
                               indices_inequality=([3] if ineq else []), evaluate_hessian=evaluate_hessian)
        bnds = [Bound(n, m, state_lower=x1, state_upper=x1)] + [Bound(n, m)] * (T - 2) + [Bound(n, 0)]
        return dict(dynamics=[dt] * (T - 1), objective=[ct] * (T - 1) + [cT], constraints=[Constraint() for _ in range(T)], bounds=bnds,
                    general_constraint=gc, T=T, n=n, m=m)
    raise KeyError(name)
