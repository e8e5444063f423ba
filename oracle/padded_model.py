"""ORACLE (test infrastructure): the synthetic cfg5 model -- acrobot embedded in n = 64 states -- and its dense KKT system.

BASELINE.json configs[4] / SURVEY.md section 8(d) define the model only in words ("acrobot dynamics embedded in n=64 ...
padding rows y_i - x_i plus a small dense linear mixing so the Jacobian block is structurally dense"); the exact
definition used by this repository is

    d(y, x, u) = y - x - h * ( [acrobot(xm[0:4], u); 0] + eps * M [xm; u] ),   xm = (x + y) / 2,  h = 0.05,  eps = 0.05,
    M = PCG64(64).standard_normal((n, n + 1)) / sqrt(n + 1),
    cost_t = 0.1 |x[2:n]|^2 + 0.1 u^2,   cost_T = 0.1 |x[2:n]|^2.

Restatement strategy: the nonlinear part is the 4-state acrobot midpoint residual, whose value / Jacobian / Hessian come
from the pinned sympy oracle objects (oracle/sympy_models.py, checked against the reference's KATs in
tests/test_oracle_kat.py); the linear mixing is added in numpy.  Nothing here imports product code.

The KKT system is the one sketched in the reference at examples/pendulum/pendulum.jl:138-198.
"""
from __future__ import annotations

import numpy as np

from . import sympy_models as S

H_STEP, EPS, SEED = 0.05, 0.05, 64


def mixing(n=64, m=1):
    rng = np.random.Generator(np.random.PCG64(SEED))
    return rng.standard_normal((n, n + m)) / np.sqrt(n + m)


def torque(u):
    """m = 1: u[0] (cfg5).  m > 1 (tests of the action block only): sum_j 2^-j u_j + 0.1 u_0 u_{m-1}"""
    m = len(u)
    tau = u[0]
    for j in range(1, m):
        tau = tau + S.fl(0.5 ** j) * u[j]
    if m > 1:
        tau = tau + S.fl(0.1) * u[0] * u[m - 1]
    return tau


class PaddedAcrobot:
    def __init__(self, n=64, m=1, parameters=None):
        """parameters = (gain, weight) (tests of parameters on the tile path only): the torque is scaled by gain, the state cost by
        weight; None: cfg5 as BASELINE.json states it."""
        self.n, self.m = n, m
        self.M = mixing(n, m)
        self.gain, self.weight = (1.0, 1.0) if parameters is None else (float(parameters[0]), float(parameters[1]))
        g = S.fl(self.gain)
        tq = (lambda u: torque(u)) if parameters is None else (lambda u: torque(u) * g)
        self.phys = S.Dynamics(lambda y, x, u, w: S.acrobot_midpoint(y, x, [tq(u)], w), 4, 4, m, evaluate_hessian=True)
        # local variable order of the small object: [x(4); u(m); y(4)] -> positions in [x(n); u; y(n)]
        self.emb = np.array([0, 1, 2, 3] + [n + j for j in range(m)] + [n + m + k for k in range(4)])

    def _lin_blocks(self):
        n, M = self.n, self.M
        Fx = -np.eye(n) - H_STEP * EPS * 0.5 * M[:, :n]
        Fu = -H_STEP * EPS * M[:, n:]
        E = np.eye(n) - H_STEP * EPS * 0.5 * M[:, :n]
        return Fx, Fu, E

    def residual(self, x, u, y):
        n = self.n
        xm = 0.5 * (x + y)
        d = y - x - H_STEP * EPS * (self.M @ np.concatenate([xm, u]))
        # the small object returns y4 - x4 - h acrobot(xm4, u): take only its -h acrobot(...) part
        small = self.phys.evaluate(list(y[:4]), list(x[:4]), list(u), [])
        d[:4] += small - (y[:4] - x[:4])
        return d

    def jacobian(self, x, u, y):
        """dense n x (2n + 1) over [x; u; y]"""
        n = self.n
        Fx, Fu, E = self._lin_blocks()
        J = np.hstack([Fx, Fu, E])
        vals = self.phys.jacobian(list(y[:4]), list(x[:4]), list(u), [])
        m = self.m
        Js = np.zeros((4, 8 + m))
        for r, c, v in zip(self.phys.jacobian_sparsity[0], self.phys.jacobian_sparsity[1], vals):
            Js[r - 1, c - 1] = v
        # remove the y - x part of the small object (already in the linear blocks), keep -h d acrobot
        Js[:, 0:4] += np.eye(4)
        Js[:, 4 + m:8 + m] -= np.eye(4)
        J[:4][:, self.emb] += Js
        return J

    def hessian(self, x, u, y, lam):
        """dense (2n+m) x (2n+m) Hessian of lam' d over [x; u; y]"""
        n = self.n
        vals = self.phys.hessian(list(y[:4]), list(x[:4]), list(u), [], list(lam[:4]))
        H = np.zeros((2 * n + self.m, 2 * n + self.m))
        for r, c, v in zip(self.phys.hessian_sparsity[0], self.phys.hessian_sparsity[1], vals):
            H[self.emb[r - 1], self.emb[c - 1]] += v
        return H

    def cost_grad_hess(self, x, u):
        n = self.n
        g = np.zeros(n + len(u))
        g[2:n] = self.weight * 0.2 * x[2:n]
        W = np.zeros((n + len(u), n + len(u)))
        W[np.arange(2, n), np.arange(2, n)] = self.weight * 0.2
        m = len(u)
        for j in range(m):
            g[n + j] = 0.2 * u[j]
            W[n + j, n + j] = 0.2
        for j in range(m - 1):   # several actions: 0.05 u_j u_{j+1} + 0.02 u_{m-1} x_5
            g[n + j] += 0.05 * u[j + 1]; g[n + j + 1] += 0.05 * u[j]
            W[n + j, n + j + 1] += 0.05; W[n + j + 1, n + j] += 0.05
        if m > 1:
            g[n + m - 1] += 0.02 * x[5]; g[5] += 0.02 * u[m - 1]
            W[n + m - 1, 5] += 0.02; W[5, n + m - 1] += 0.02
        return g, W


def dense_kkt(model: PaddedAcrobot, T: int, z, mu, dw, dc):
    """K and right-hand side of the regularised Newton-KKT system at (z, mu); z = [x1; u1; ...; xT], mu = dynamics rows."""
    n, m = model.n, model.m
    nz, nc = (T - 1) * (n + m) + n, (T - 1) * n
    H = np.zeros((nz, nz)); J = np.zeros((nc, nz)); g = np.zeros(nz); c = np.zeros(nc)
    for t in range(T):
        o = t * (n + m)
        x = z[o:o + n]
        u = z[o + n:o + n + m] if t < T - 1 else np.zeros(0)
        gt, Wt = model.cost_grad_hess(x, u)
        npv = n + len(u)
        g[o:o + npv] += gt
        H[o:o + npv, o:o + npv] += Wt
        if t < T - 1:
            y = z[o + n + m:o + 2 * n + m]
            lam = mu[t * n:(t + 1) * n]
            c[t * n:(t + 1) * n] = model.residual(x, u, y)
            J[t * n:(t + 1) * n, o:o + 2 * n + m] = model.jacobian(x, u, y)
            H[o:o + 2 * n + m, o:o + 2 * n + m] += model.hessian(x, u, y, lam)
    K = np.block([[H + dw * np.eye(nz), J.T], [J, -dc * np.eye(nc)]])
    rhs = -np.concatenate([g + J.T @ mu, c])
    return K, rhs


def dense_derivatives(model: PaddedAcrobot, T: int, z, mu, sigma=1.0):
    """f, grad f, c, dense J, dense Hessian of sigma f + mu' c (what the five MOI methods return, src/moi.jl:1-120)."""
    n, m = model.n, model.m
    nz, nc = (T - 1) * (n + m) + n, (T - 1) * n
    H = np.zeros((nz, nz)); J = np.zeros((nc, nz)); g = np.zeros(nz); c = np.zeros(nc)
    f = 0.0
    for t in range(T):
        o = t * (n + m)
        x = z[o:o + n]
        u = z[o + n:o + n + m] if t < T - 1 else np.zeros(0)
        gt, Wt = model.cost_grad_hess(x, u)
        npv = n + len(u)
        f += model.weight * 0.1 * float(x[2:n] @ x[2:n]) + 0.1 * float(u @ u)
        if len(u) > 1:
            f += 0.05 * float(u[:-1] @ u[1:]) + 0.02 * float(u[-1] * x[5])
        g[o:o + npv] += gt
        H[o:o + npv, o:o + npv] += sigma * Wt
        if t < T - 1:
            y = z[o + n + m:o + 2 * n + m]
            c[t * n:(t + 1) * n] = model.residual(x, u, y)
            J[t * n:(t + 1) * n, o:o + 2 * n + m] = model.jacobian(x, u, y)
            H[o:o + 2 * n + m, o:o + 2 * n + m] += model.hessian(x, u, y, mu[t * n:(t + 1) * n])
    return f, g, c, J, H


def kkt_residual_blockwise(model: PaddedAcrobot, T: int, z, lam):
    """(c, grad f + J' lam) assembled stage by stage (no dense Jacobian: usable at the full horizon T = 2000 of configs[4])."""
    n, m = model.n, model.m
    nz, nc = (T - 1) * (n + m) + n, (T - 1) * n
    r = np.zeros(nz); c = np.zeros(nc)
    for t in range(T):
        o = t * (n + m)
        x = z[o:o + n]
        u = z[o + n:o + n + m] if t < T - 1 else np.zeros(0)
        gt, _ = model.cost_grad_hess(x, u)
        r[o:o + n + len(u)] += gt
        if t < T - 1:
            y = z[o + n + m:o + 2 * n + m]
            lt = lam[t * n:(t + 1) * n]
            c[t * n:(t + 1) * n] = model.residual(x, u, y)
            r[o:o + 2 * n + m] += model.jacobian(x, u, y).T @ lt
    return c, r


class PaddedStageRows:
    """ORACLE restatement of the stage constraints of problems.build_acrobot_padded(stage_constraints=(a, b, r)) (round 6: the
    reference's two uses of `Constraint` on a model with more than 16 states) -- rows in the reference order, knot by knot behind
    all dynamics rows (src/data.jl:68-69):
        knot 1:      x - x1 (n equality rows, examples/acrobot/acrobot.jl:114-116),  obstacle row
        knots 2..T-1: obstacle row   r^2 - (x_1 - a)^2 - (x_2 - b)^2 <= 0   (examples/car/car.jl:53-60)
        knot T:      x[1:4] - xT[1:4] (4 equality rows),  obstacle row
    Closed forms only: values, dense Jacobian, Hessian of nu' c."""

    def __init__(self, n, m, T, x1, xT, a, b, r):
        self.n, self.m, self.T = n, m, T
        self.x1, self.xT, self.a, self.b, self.r = np.asarray(x1, float), np.asarray(xT, float), float(a), float(b), float(r)
        self.rows_of = [n + 1] + [1] * (T - 2) + [5]
        self.off = np.concatenate([[0], np.cumsum(self.rows_of)])
        self.num = int(self.off[-1])
        ineq = np.zeros(self.num, dtype=bool)
        for t in range(T):
            ineq[self.off[t + 1] - 1] = True
        self.inequality = ineq

    def _x(self, z, t):
        o = t * (self.n + self.m)
        return o, z[o:o + self.n]

    def values(self, z):
        c = np.zeros(self.num)
        for t in range(self.T):
            _, x = self._x(z, t)
            k = self.off[t]
            if t == 0:
                c[k:k + self.n] = x - self.x1
                k += self.n
            if t == self.T - 1:
                c[k:k + 4] = x[:4] - self.xT[:4]
                k += 4
            c[k] = self.r ** 2 - (x[0] - self.a) ** 2 - (x[1] - self.b) ** 2
        return c

    def jacobian(self, z):
        J = np.zeros((self.num, len(z)))
        for t in range(self.T):
            o, x = self._x(z, t)
            k = self.off[t]
            if t == 0:
                J[k:k + self.n, o:o + self.n] = np.eye(self.n)
                k += self.n
            if t == self.T - 1:
                J[k:k + 4, o:o + 4] = np.eye(4)
                k += 4
            J[k, o] = -2.0 * (x[0] - self.a)
            J[k, o + 1] = -2.0 * (x[1] - self.b)
        return J

    def hessian(self, z, nu):
        H = np.zeros((len(z), len(z)))
        for t in range(self.T):
            o, _ = self._x(z, t)
            v = nu[self.off[t + 1] - 1]
            H[o, o] += -2.0 * v
            H[o + 1, o + 1] += -2.0 * v
        return H
