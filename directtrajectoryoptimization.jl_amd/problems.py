"""The model equations behind the BASELINE configs, written as plain numpy closures.

Each function restates one of the reference example scripts (constants matter for parity):

* pendulum  examples/pendulum/pendulum.jl:22-39   (implicit midpoint, h = 0.05)
* cartpole  examples/cartpole/cartpole.jl:19-56   (explicit RK3 written as y - rk3(x, u), h = 0.05)
* acrobot   examples/acrobot/acrobot.jl:19-91     (implicit midpoint, h = 0.05)
* car       examples/car/car.jl:19-26             (implicit midpoint, h = 0.1)

plus the small models the reference's unit tests use (test/dynamics.jl:8-19,
test/solve.jl:149-183).  The closures take numpy vectors of floats *or* of traced
`Expr` nodes; they are traced once per `Dynamics/Cost/Constraint` object.
`build_*` helpers assemble the full problem (objects + bounds + guesses) of a config.
"""
from __future__ import annotations

import math

import numpy as np

from .model import Bound, Constraint, Cost, Dynamics, linear_interpolation
from .symbolic.expr import dot

PI = math.pi


# ----------------------------------------------------------------------------- pendulum
def pendulum(x, u, w):
    mass, length_com, gravity, damping = 1.0, 0.5, 9.81, 0.1
    return np.array([
        x[1],
        u[0] / (mass * length_com * length_com)
        - gravity * np.sin(x[0]) / length_com
        - damping * x[1] / (mass * length_com * length_com),
    ], dtype=object)


def pendulum_midpoint(y, x, u, w, h=0.05):
    return y - (x + h * pendulum(0.5 * (x + y), u, w))


def pendulum_test(z, u, w):
    """test/dynamics.jl:8-14 (lc = 1)."""
    mass, lc, gravity, damping = 1.0, 1.0, 9.81, 0.1
    return np.array([z[1], u[0] / (mass * lc * lc) - gravity * np.sin(z[0]) / lc - damping * z[1] / (mass * lc * lc)],
                    dtype=object)


def euler_implicit_test(y, x, u, w, h=0.1):
    """test/dynamics.jl:16-19."""
    return y - (x + h * pendulum_test(y, u, w))


# ----------------------------------------------------------------------------- cartpole
def cartpole(x, u, w):
    mc, mp, l, g = 1.0, 0.2, 0.5, 9.81
    qd = x[2:4]
    s = np.sin(x[1])
    c = np.cos(x[1])
    H11, H12, H22 = mc + mp, mp * l * c, mp * l ** 2
    det = H11 * H22 - H12 * H12
    # C*qd + G - B*u  (C = [0 -mp*qd2*l*s; 0 0], G = [0, mp*g*l*s], B = [1, 0])
    r1 = -mp * qd[1] * l * s * qd[1] - u[0]
    r2 = mp * g * l * s
    # qdd = -Hinv * r, Hinv = 1/det * [H22 -H12; -H12 H11]
    qdd1 = -(1.0 / det) * (H22 * r1 - H12 * r2)
    qdd2 = -(1.0 / det) * (-H12 * r1 + H11 * r2)
    return np.array([qd[0], qd[1], qdd1, qdd2], dtype=object)


def cartpole_rk3_explicit(x, u, w, h=0.05):
    k1 = h * cartpole(x, u, w)
    k2 = h * cartpole(x + 0.5 * k1, u, w)
    k3 = h * cartpole(x - k1 + 2.0 * k2, u, w)
    return x + (k1 + 4.0 * k2 + k3) / 6.0


def cartpole_rk3_implicit(y, x, u, w):
    return y - cartpole_rk3_explicit(x, u, w)


# ----------------------------------------------------------------------------- acrobot
def acrobot(x, u, w):
    mass1, inertia1, length1, lengthcom1 = 1.0, 0.33, 1.0, 0.5
    mass2, inertia2, length2, lengthcom2 = 1.0, 0.33, 1.0, 0.5
    gravity, friction1, friction2 = 9.81, 0.1, 0.1
    q1, q2, v1, v2 = x[0], x[1], x[2], x[3]
    # Minv(q)
    a = inertia1 + inertia2 + mass2 * length1 * length1 + 2.0 * mass2 * length1 * lengthcom2 * np.cos(q2)
    b = inertia2 + mass2 * length1 * lengthcom2 * np.cos(q2)
    c = inertia2
    idet = 1.0 / (a * c - b * b)
    # tau(q)
    ta = (-1.0 * mass1 * gravity * lengthcom1 * np.sin(q1)
          - mass2 * gravity * (length1 * np.sin(q1) + lengthcom2 * np.sin(q1 + q2)))
    tb = -1.0 * mass2 * gravity * lengthcom2 * np.sin(q1 + q2)
    # C(x)
    Ca = -2.0 * mass2 * length1 * lengthcom2 * np.sin(q2) * v2
    Cb = -1.0 * mass2 * length1 * lengthcom2 * np.sin(q2) * v2
    Cc = mass2 * length1 * lengthcom2 * np.sin(q2) * v1
    # rhs = -C v + tau + B u - friction .* v ,  B = [0; 1]
    r1 = -1.0 * (Ca * v1 + Cb * v2) + ta - friction1 * v1
    r2 = -1.0 * (Cc * v1) + tb + u[0] - friction2 * v2
    qdd1 = idet * (c * r1 - b * r2)
    qdd2 = idet * (-b * r1 + a * r2)
    return np.array([v1, v2, qdd1, qdd2], dtype=object)


def acrobot_midpoint(y, x, u, w, h=0.05):
    return y - (x + h * acrobot(0.5 * (x + y), u, w))


# ----------------------------------------------------------------------------- acrobot embedded in n states (cfg5)
def padded_mixing(n=64, m=1, seed=64):
    """Dense (n x (n+m)) mixing matrix of the synthetic cfg5 model: fixed by the seed so that
    every rank, the oracle and the fixtures see the same numbers."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.standard_normal((n, n + m)) / np.sqrt(n + m)


def padded_torque(u):
    """Elbow torque of the padded acrobot with m actions: u[0] for m = 1 (cfg5 as BASELINE.json states it); for m > 1 the
    actions enter with weights 2^-j plus a bilinear term, so that the action block of the stage Hessian is dense."""
    m = len(u)
    tau = u[0]
    for j in range(1, m):
        tau = tau + 0.5 ** j * u[j]
    if m > 1:
        tau = tau + 0.1 * u[0] * u[m - 1]
    return tau


def acrobot_padded_midpoint(n=64, h=0.05, eps=0.05, seed=64, m=1, parametric=False):
    """y - x - h*f(0.5(x+y), u) with f = [acrobot(x[0:4], u); 0] + eps * M [xm; u]: the four physical states keep the
    acrobot dynamics, the padding states are a stable-ish dense linear system, and every residual row depends on every
    column of [x; u], so the stage Jacobian block is structurally dense (blocks of n + m + n = 129 for n = 64)."""
    M = padded_mixing(n, m, seed)

    def f(y, x, u, w):
        xm = 0.5 * (x + y)
        # parametric: the torque is scaled by the stage parameter w[0] (a test of parameters on the tile path)
        phys = acrobot(xm[0:4], [padded_torque(u) * w[0] if parametric else padded_torque(u)], w)
        lin = M @ np.concatenate([xm, u])
        rhs = np.array([(phys[i] if i < 4 else 0.0) + eps * lin[i] for i in range(n)], dtype=object)
        return y - (x + h * rhs)

    return f


# ----------------------------------------------------------------------------- car
def car(x, u, w):
    return np.array([u[0] * np.cos(x[2]), u[0] * np.sin(x[2]), u[1]], dtype=object)


def car_midpoint(y, x, u, w, h=0.1):
    return y - (x + h * car(0.5 * (x + y), u, w))


# ----------------------------------------------------------------------------- double integrator (test/solve.jl:149-183)
def double_integrator(y, x, u, w):
    A = np.array([[1.0, 1.0], [0.0, 1.0]])
    B = np.array([0.0, 1.0])
    return y - (A @ x + B * u[0])


def double_integrator_grad(y, x, u, w):
    A = np.array([[1.0, 1.0], [0.0, 1.0]])
    B = np.array([[0.0], [1.0]])
    return np.hstack([-A, -B, np.eye(2)])


# ============================================================================= config builders
def build_pendulum(T=50, evaluate_hessian=True):
    """cfg1: examples/pendulum/pendulum.jl:41-77 with T = 50."""
    n, m = 2, 1
    x1 = np.array([0.0, 0.0])
    xT = np.array([PI, 0.0])
    dt = Dynamics(pendulum_midpoint, n, n, m, evaluate_hessian=evaluate_hessian)
    ct = Cost(lambda x, u, w: 0.1 * dot(x[0:2], x[0:2]) + 0.1 * dot(u, u), n, m, evaluate_hessian=evaluate_hessian)
    cT = Cost(lambda x, u, w: 0.1 * dot(x[0:2], x[0:2]), n, 0, evaluate_hessian=evaluate_hessian)
    con1 = Constraint(lambda x, u, w: x - x1, n, m, evaluate_hessian=evaluate_hessian)
    # quirk 9: the reference builds conT with num_action = 1 although u_T is empty;
    # the closure never reads u, so num_action = 0 is equivalent and is what the layout needs.
    conT = Constraint(lambda x, u, w: x - xT, n, 0, evaluate_hessian=evaluate_hessian)
    return dict(
        dynamics=[dt] * (T - 1),
        objective=[ct] * (T - 1) + [cT],
        constraints=[con1] + [Constraint() for _ in range(T - 2)] + [conT],
        bounds=[Bound(n, m)] * (T - 1) + [Bound(n, 0)],
        x1=x1, xT=xT, T=T, n=n, m=m, evaluate_hessian=evaluate_hessian,
        guess=lambda rng: (linear_interpolation(x1, xT, T), [rng.standard_normal(m) for _ in range(T - 1)]),
    )


def build_cartpole(T=200, evaluate_hessian=False):
    """cfg2: examples/cartpole/cartpole.jl:58-106 with T = 200."""
    n, m = 4, 1
    x1 = np.zeros(4)
    xT = np.array([0.0, PI, 0.0, 0.0])
    Q, R, Qf = 1.0e-2, 1.0e-1, 1.0e2
    dt = Dynamics(cartpole_rk3_implicit, n, n, m, evaluate_hessian=evaluate_hessian)
    ct = Cost(lambda x, u, w: 0.5 * Q * dot(x - xT, x - xT) + 0.5 * R * dot(u, u), n, m, evaluate_hessian=evaluate_hessian)
    cT = Cost(lambda x, u, w: 0.5 * Qf * dot(x - xT, x - xT), n, 0, evaluate_hessian=evaluate_hessian)
    u_bnd = 3.0
    bnd = Bound(n, m, action_lower=[-u_bnd], action_upper=[u_bnd])
    con1 = Constraint(lambda x, u, w: x - x1, n, m, evaluate_hessian=evaluate_hessian)
    conT = Constraint(lambda x, u, w: x - xT, n, 0, evaluate_hessian=evaluate_hessian)

    def guess(rng):
        u_guess = [0.01 * np.ones(m) for _ in range(T - 1)]
        xs = [x1.copy()]
        for t in range(T - 1):
            xs.append(np.asarray(cartpole_rk3_explicit(xs[-1], u_guess[t], np.zeros(0)), dtype=float))
        return xs, u_guess

    return dict(
        dynamics=[dt] * (T - 1),
        objective=[ct] * (T - 1) + [cT],
        constraints=[con1] + [Constraint() for _ in range(T - 2)] + [conT],
        bounds=[bnd] * (T - 1) + [Bound(n, 0)],
        x1=x1, xT=xT, T=T, n=n, m=m, evaluate_hessian=evaluate_hessian, guess=guess,
    )


def build_acrobot(T=1000, evaluate_hessian=True, endpoint="constraints"):
    """cfg3: examples/acrobot/acrobot.jl:93-118 with T = 1000 (endpoint equality constraints,
    xT = [pi, 0, 0, 0]); endpoint="bounds" gives the variant of test/solve.jl:97-121
    (xT = [0, pi, 0, 0], endpoints fixed through equal bounds, no stage constraints)."""
    n, m = 4, 1
    x1 = np.zeros(4)
    dt = Dynamics(acrobot_midpoint, n, n, m, evaluate_hessian=evaluate_hessian)
    ct = Cost(lambda x, u, w: 0.1 * dot(x[2:4], x[2:4]) + 0.1 * dot(u, u), n, m, evaluate_hessian=evaluate_hessian)
    cT = Cost(lambda x, u, w: 0.1 * dot(x[2:4], x[2:4]), n, 0, evaluate_hessian=evaluate_hessian)
    if endpoint == "constraints":
        xT = np.array([PI, 0.0, 0.0, 0.0])
        con1 = Constraint(lambda x, u, w: x - x1, n, m, evaluate_hessian=evaluate_hessian)
        conT = Constraint(lambda x, u, w: x - xT, n, 0, evaluate_hessian=evaluate_hessian)
        constraints = [con1] + [Constraint() for _ in range(T - 2)] + [conT]
        bounds = [Bound(n, m)] * (T - 1) + [Bound(n, 0)]
    else:
        xT = np.array([0.0, PI, 0.0, 0.0])
        constraints = [Constraint() for _ in range(T)]
        bounds = ([Bound(n, m, state_lower=x1, state_upper=x1)] + [Bound(n, m)] * (T - 2)
                  + [Bound(n, 0, state_lower=xT, state_upper=xT)])
    return dict(
        dynamics=[dt] * (T - 1),
        objective=[ct] * (T - 1) + [cT],
        constraints=constraints, bounds=bounds,
        x1=x1, xT=xT, T=T, n=n, m=m, evaluate_hessian=evaluate_hessian,
        guess=lambda rng: (linear_interpolation(x1, xT, T), [rng.standard_normal(m) for _ in range(T - 1)]),
    )


def padded_action_cost(x, u):
    """0.1 |u|^2 for one action; several actions are coupled to each other and to one padding state."""
    m = len(u)
    c = 0.1 * dot(u, u)
    for j in range(m - 1):
        c = c + 0.05 * u[j] * u[j + 1]
    if m > 1:
        c = c + 0.02 * u[m - 1] * x[5]
    return c


def build_acrobot_padded(T=2000, n=64, evaluate_hessian=True, target=PI, terminal="full", u_max=None, m=1, parameters=None,
                         stage_constraints=None, general_row=None):
    """cfg5 (BASELINE.json configs[4]): acrobot swing-up with the state padded to n = 64 so that the per-stage KKT
    blocks are dense 129 x 129; endpoints fixed by equal bounds (as examples/car/car.jl:44-49 does).
    terminal="physical" fixes only the four acrobot states at the last knot (the padding states stay free): with one action a
    64-dimensional terminal state is reachable only for horizons of more than 64 knots.
    m > 1: several actions (padded_torque, padded_action_cost) -- not a BASELINE.json configuration, a test of the action block."""
    x1 = np.zeros(n)
    xT = np.zeros(n)
    xT[0] = target
    # parameters = (gain, weight): stage parameters w_t = [torque gain, weight of the state cost] (src/solver.jl:10 `parameters`;
    # not a BASELINE.json configuration: a test of shared / per-instance parameters on the tile path)
    par = parameters is not None
    nw = 2 if par else 0
    dt = Dynamics(acrobot_padded_midpoint(n, m=m, parametric=par), n, n, m, num_parameter=nw, evaluate_hessian=evaluate_hessian)
    ct = Cost(lambda x, u, w: (w[1] if par else 1.0) * 0.1 * dot(x[2:n], x[2:n]) + padded_action_cost(x, u), n, m,
              num_parameter=nw, evaluate_hessian=evaluate_hessian)
    cT = Cost(lambda x, u, w: (w[1] if par else 1.0) * 0.1 * dot(x[2:n], x[2:n]), n, 0, num_parameter=nw, evaluate_hessian=evaluate_hessian)
    # u_max: action bounds -u_max <= u <= u_max at every knot (examples/cartpole/cartpole.jl:81-89 style)
    ub = {} if u_max is None else dict(action_lower=-u_max * np.ones(m), action_upper=u_max * np.ones(m))
    b1 = Bound(n, m, state_lower=x1, state_upper=x1, **ub)
    bt = Bound(n, m, **ub)
    if terminal == "physical":
        lo, hi = np.full(n, -np.inf), np.full(n, np.inf)
        lo[:4] = hi[:4] = xT[:4]
        bT = Bound(n, 0, state_lower=lo, state_upper=hi)
    else:
        bT = Bound(n, 0, state_lower=xT, state_upper=xT)
    cons = [Constraint() for _ in range(T)]
    if stage_constraints is not None:
        # stage_constraints = (a, b, r): the reference's two uses of `Constraint` on a model with more than 16 states (round 6) --
        # the endpoints as EQUALITY ROWS instead of bounds (examples/acrobot/acrobot.jl:114-118: x - x1 at the first knot, the
        # physical states - xT at the last) and an obstacle-style INEQUALITY ROW at every knot (examples/car/car.jl:53-60):
        # r^2 - (q1 - a)^2 - (q2 - b)^2 <= 0 keeps the joint angles out of a disc the straight-line guess runs through
        a_, b_, r_ = (float(v) for v in stage_constraints)
        obs = lambda x: r_ ** 2.0 - (x[0] - a_) ** 2.0 - (x[1] - b_) ** 2.0
        con1 = Constraint(lambda x, u, w: np.array(list(x - x1) + [obs(x)], dtype=object), n, m, indices_inequality=[n + 1],
                          evaluate_hessian=evaluate_hessian)
        cont = Constraint(lambda x, u, w: np.array([obs(x)], dtype=object), n, m, indices_inequality=[1], evaluate_hessian=evaluate_hessian)
        conT = Constraint(lambda x, u, w: np.array(list(x[0:4] - xT[0:4]) + [obs(x)], dtype=object), n, 0, indices_inequality=[5],
                          evaluate_hessian=evaluate_hessian)
        cons = [con1] + [cont] * (T - 2) + [conT]
        b1 = Bound(n, m, **ub)
        bT = Bound(n, 0)
    gc = None
    if general_row is not None:
        # general_row = (ka, kb, total[, inequality]): one GeneralConstraint row coupling two knots (src/general_constraint.jl:18-59),
        # q1 at knot ka + q1 at knot kb - total (= 0 | <= 0) -- round 6: coupling rows on a model with more than 16 states
        from .model import GeneralConstraint
        ka, kb, tot = int(general_row[0]), int(general_row[1]), float(general_row[2])
        iq = bool(general_row[3]) if len(general_row) > 3 else False
        nz = n * T + m * (T - 1)
        ia, ib = (ka - 1) * (n + m), (kb - 1) * (n + m)
        gc = GeneralConstraint(lambda z, w: np.array([z[ia] + z[ib] - tot], dtype=object), nz, 0, indices_inequality=([1] if iq else []),
                               evaluate_hessian=evaluate_hessian)
    return dict(
        dynamics=[dt] * (T - 1),
        objective=[ct] * (T - 1) + [cT],
        constraints=cons,
        bounds=[b1] + [bt] * (T - 2) + [bT],
        x1=x1, xT=xT, T=T, n=n, m=m, evaluate_hessian=evaluate_hessian,
        parameters=[np.array(parameters, dtype=float) for _ in range(T)] if par else None,
        general_constraint=gc,
        guess=lambda rng: (linear_interpolation(x1, xT, T), [rng.standard_normal(m) for _ in range(T - 1)]),
    )


def build_car(T=500, evaluate_hessian=False):
    """cfg4 (one instance): examples/car/car.jl:28-67 with T = 500."""
    n, m = 3, 2
    x1 = np.zeros(3)
    xT = np.array([1.0, 1.0, 0.0])
    dt = Dynamics(car_midpoint, n, n, m, evaluate_hessian=evaluate_hessian)
    ct = Cost(lambda x, u, w: 0.0 * dot(x - xT, x - xT) + 1.0 * dot(u, u), n, m, evaluate_hessian=evaluate_hessian)
    cT = Cost(lambda x, u, w: 0.0 * dot(x - xT, x - xT), n, 0, evaluate_hessian=evaluate_hessian)
    lo, hi = -0.5 * np.ones(m), 0.5 * np.ones(m)
    bnd1 = Bound(n, m, state_lower=x1, state_upper=x1, action_lower=lo, action_upper=hi)
    bndt = Bound(n, m, action_lower=lo, action_upper=hi)
    bndT = Bound(n, 0, state_lower=xT, state_upper=xT)
    p_obs = np.array([0.5, 0.5])
    r_obs = 0.1

    def obs(x, u, w):
        e = x[0:2] - p_obs
        return np.array([r_obs ** 2.0 - dot(e, e)], dtype=object)

    cont = Constraint(obs, n, m, indices_inequality=[1], evaluate_hessian=evaluate_hessian)
    conT = Constraint(obs, n, 0, indices_inequality=[1], evaluate_hessian=evaluate_hessian)
    return dict(
        dynamics=[dt] * (T - 1),
        objective=[ct] * (T - 1) + [cT],
        constraints=[cont] * (T - 1) + [conT],
        bounds=[bnd1] + [bndt] * (T - 2) + [bndT],
        x1=x1, xT=xT, T=T, n=n, m=m, evaluate_hessian=evaluate_hessian,
        guess=lambda rng: (linear_interpolation(x1, xT, T), [0.001 * rng.standard_normal(m) for _ in range(T - 1)]),
    )


# ============================================================================= reference unit-test problems
def build_ref_objective():
    """test/objective.jl:1-20 (T = 3, n = 2, m = 1)."""
    T, n, m = 3, 2, 1
    ct = Cost(lambda x, u, w: dot(x, x) + 0.1 * dot(u, u), n, m)
    cT = Cost(lambda x, u, w: 10.0 * dot(x, x), n, 0)
    dyn = Dynamics(double_integrator, n, n, m)
    return dict(dynamics=[dyn] * (T - 1), objective=[ct] * (T - 1) + [cT], constraints=[Constraint() for _ in range(T)],
                bounds=[Bound(n, m)] * (T - 1) + [Bound(n, 0)], T=T, n=n, m=m, evaluate_hessian=False)


def build_ref_dynamics():
    """test/dynamics.jl:1-36 (pendulum, implicit Euler, T = 3)."""
    T, n, m = 3, 2, 1
    dt = Dynamics(euler_implicit_test, n, n, m)
    ct = Cost(lambda x, u, w: dot(x, x), n, m)
    cT = Cost(lambda x, u, w: dot(x, x), n, 0)
    return dict(dynamics=[dt] * (T - 1), objective=[ct] * (T - 1) + [cT], constraints=[Constraint() for _ in range(T)],
                bounds=[Bound(n, m)] * (T - 1) + [Bound(n, 0)], T=T, n=n, m=m, evaluate_hessian=False)


def build_ref_constraints():
    """test/constraints.jl:1-28 (T = 5; ct = [-1 - x; x - 1] all rows inequality, cT = x)."""
    T, n, m = 5, 2, 1
    cont = Constraint(lambda x, u, w: np.concatenate([-np.ones(n) - x, x - np.ones(n)]), n, m,
                      indices_inequality=list(range(1, 2 * n + 1)))
    conT = Constraint(lambda x, u, w: x, n, 0)
    dyn = Dynamics(double_integrator, n, n, m)
    ct = Cost(lambda x, u, w: dot(x, x), n, m)
    cT = Cost(lambda x, u, w: dot(x, x), n, 0)
    return dict(dynamics=[dyn] * (T - 1), objective=[ct] * (T - 1) + [cT], constraints=[cont] * (T - 1) + [conT],
                bounds=[Bound(n, m)] * (T - 1) + [Bound(n, 0)], T=T, n=n, m=m, evaluate_hessian=False)


def build_ref_hesslag():
    """test/hessian_lagrangian.jl:97-128 (acrobot midpoint, T = 3, nonlinear stage constraints, exact Hessians)."""
    T, n, m = 3, 4, 1
    dt = Dynamics(acrobot_midpoint, n, n, m, evaluate_hessian=True)
    objt = Cost(lambda x, u, w: 0.1 * dot(x[2:4], x[2:4]) + 0.1 * dot(u, u), n, m, evaluate_hessian=True)
    objT = Cost(lambda x, u, w: 0.1 * dot(x[2:4], x[2:4]), n, 0, evaluate_hessian=True)
    ctf = lambda x, u, w: np.concatenate([-5.0 * np.ones(m) - np.cos(u) * np.sum(x ** 2),
                                          np.cos(x) * np.tan(u) - 5.0 * np.ones(n)])
    cTf = lambda x, u, w: np.sin(x ** 3.0)
    cont = Constraint(ctf, n, m, indices_inequality=list(range(1, m + n + 1)), evaluate_hessian=True)
    conT = Constraint(cTf, n, 0, evaluate_hessian=True)
    return dict(dynamics=[dt] * 2, objective=[objt, objt, objT], constraints=[cont, cont, conT],
                bounds=[Bound(n, m)] * 2 + [Bound(n, 0)], T=T, n=n, m=m, evaluate_hessian=True)


def build_ref_general(user_jacobian=False):
    """test/solve.jl:140-296: double integrator, T = 11; (a) user-provided dense dynamics Jacobian
    (test/solve.jl:149-183), (b) GeneralConstraint z[end-1:end] - xT over the whole decision vector
    (test/solve.jl:273-274)."""
    from .model import GeneralConstraint
    T, n, m = 11, 2, 1
    x1 = np.array([0.0, 0.0])
    xT = np.array([1.0, 0.0])
    if user_jacobian:
        dt = Dynamics(double_integrator, double_integrator_grad, n, n, m)
        eh = False
    else:
        dt = Dynamics(double_integrator, n, n, m, evaluate_hessian=True)
        eh = True
    ct = Cost(lambda x, u, w: 0.1 * dot(x, x) + 0.1 * dot(u, u), n, m, evaluate_hessian=eh)
    cT = Cost(lambda x, u, w: 0.1 * dot(x, x), n, 0, evaluate_hessian=eh)
    nz = n * T + m * (T - 1)
    gc = GeneralConstraint(lambda z, w: z[nz - 2:nz] - xT, nz, 0, evaluate_hessian=eh)
    bounds = [Bound(n, m, state_lower=x1, state_upper=x1)] + [Bound(n, m)] * (T - 2) + [Bound(n, 0)]
    return dict(dynamics=[dt] * (T - 1), objective=[ct] * (T - 1) + [cT], constraints=[Constraint() for _ in range(T)],
                bounds=bounds, general_constraint=gc, x1=x1, xT=xT, T=T, n=n, m=m, evaluate_hessian=eh)


def build_ref_general_coupled(inequality=None):
    """test/solve.jl:227-296 extended with a row that couples two knots: GeneralConstraint
    (z, w) -> [z[end-1:end] - xT; x_4[1] + x_8[1] - 0.9] (positions at knots 4 and 8 add up to 0.9).  The terminal rows alone
    would be folded into a stage constraint; the coupling row cannot: the solver's bordered path (dto_solver.cpp:
    bordered_step / general_solve_batch) takes the whole GeneralConstraint as the border of the block-tridiagonal system."""
    from .model import GeneralConstraint
    T, n, m = 11, 2, 1
    x1 = np.array([0.0, 0.0])
    xT = np.array([1.0, 0.0])
    dt = Dynamics(double_integrator, n, n, m, evaluate_hessian=True)
    ct = Cost(lambda x, u, w: 0.1 * dot(x, x) + 0.1 * dot(u, u), n, m, evaluate_hessian=True)
    cT = Cost(lambda x, u, w: 0.1 * dot(x, x), n, 0, evaluate_hessian=True)
    nz = n * T + m * (T - 1)
    i4, i8 = 3 * (n + m), 7 * (n + m)          # 0-based offsets of x_4 and x_8 in z
    # inequality = total: the coupling row becomes x_4[1] + x_8[1] - total <= 0 (indices_inequality, src/general_constraint.jl:15-19)
    tot = 0.9 if inequality is None else float(inequality)
    gc = GeneralConstraint(lambda z, w: np.array([z[nz - 2] - xT[0], z[nz - 1] - xT[1], z[i4] + z[i8] - tot], dtype=object),
                           nz, 0, indices_inequality=([] if inequality is None else [3]), evaluate_hessian=True)
    bounds = [Bound(n, m, state_lower=x1, state_upper=x1)] + [Bound(n, m)] * (T - 2) + [Bound(n, 0)]
    return dict(dynamics=[dt] * (T - 1), objective=[ct] * (T - 1) + [cT], constraints=[Constraint() for _ in range(T)],
                bounds=bounds, general_constraint=gc, x1=x1, xT=xT, T=T, n=n, m=m, evaluate_hessian=True, coupling=(i4, i8, tot))


def build_pendulum_coupled(T=50, total=1.0, inequality=True, u_max=None, nonlinear=False, evaluate_hessian=True):
    """The pendulum swing-up (examples/pendulum/pendulum.jl, nonlinear dynamics) with one GeneralConstraint row that couples
    knots 15 and 35: theta_15 + theta_35 - total (<= 0 with inequality=True, = 0 otherwise).  Without the row the sum is 1.449.
    u_max: action bounds |u| <= u_max at every knot beside the general row (examples/cartpole/cartpole.jl:81-89 style).
    nonlinear: the row is sin(theta_15) + theta_35^2 - total instead (a sum of NONLINEAR one-knot terms)."""
    from .model import GeneralConstraint
    p = build_pendulum(T=T, evaluate_hessian=evaluate_hessian)
    n, m = 2, 1
    if u_max is not None:
        p["bounds"] = [Bound(n, m, action_lower=-u_max * np.ones(m), action_upper=u_max * np.ones(m))] * (T - 1) + [Bound(n, 0)]
    nz = n * T + m * (T - 1)
    i15, i35 = 14 * (n + m), 34 * (n + m)
    row = (lambda z, w: np.array([np.sin(z[i15]) + z[i35] ** 2.0 - total], dtype=object)) if nonlinear else \
          (lambda z, w: np.array([z[i15] + z[i35] - total], dtype=object))
    p["general_constraint"] = GeneralConstraint(row, nz, 0, indices_inequality=([1] if inequality else []),
                                                evaluate_hessian=evaluate_hessian and not nonlinear)   # src/general_constraint.jl:87 is broken
    p["coupling"] = (i15, i35, total)
    return p


def build_acrobot_coupled(T=8):
    """The acrobot (nonlinear dynamics, pinned endpoints) with two GeneralConstraint rows that couple knots: q1 at knot 3 minus
    q1 at knot 6 equals 0.2; q2 at knots 2 and 7 plus the action at knot 4 add up to zero.  Used for the bordered KKT step."""
    from .model import GeneralConstraint
    p = build_acrobot(T=T, evaluate_hessian=True)
    n, m = p["n"], p["m"]
    nz = n * T + m * (T - 1)
    off = lambda t: (t - 1) * (n + m)          # 0-based offset of x_t (t 1-based)
    gc = GeneralConstraint(lambda z, w: np.array([z[off(3)] - z[off(6)] - 0.2, z[off(2) + 1] + z[off(7) + 1] + z[off(4) + n]], dtype=object),
                           nz, 0, evaluate_hessian=True)
    p["general_constraint"] = gc
    return p


def build_ref_userjac():
    """test/solve.jl:140-226: double integrator, T = 11, user-provided dense dynamics Jacobian, both endpoints fixed by
    bounds, Solver(...) in its default mode (no exact Hessians)."""
    T, n, m = 11, 2, 1
    x1 = np.array([0.0, 0.0])
    xT = np.array([1.0, 0.0])
    dt = Dynamics(double_integrator, double_integrator_grad, n, n, m)
    ct = Cost(lambda x, u, w: 0.1 * dot(x, x) + 0.1 * dot(u, u), n, m)
    cT = Cost(lambda x, u, w: 0.1 * dot(x, x), n, 0)
    bounds = ([Bound(n, m, state_lower=x1, state_upper=x1)] + [Bound(n, m)] * (T - 2)
              + [Bound(n, 0, state_lower=xT, state_upper=xT)])
    return dict(dynamics=[dt] * (T - 1), objective=[ct] * (T - 1) + [cT], constraints=[Constraint() for _ in range(T)],
                bounds=bounds, x1=x1, xT=xT, T=T, n=n, m=m, evaluate_hessian=False)


def build_param_pendulum(T=8):
    """Per-stage parameters w_t (src/solver.jl:10 `parameters` kwarg): pendulum whose mass and goal angle are
    parameters; exercises num_parameter > 0 in Dynamics, Cost and Constraint."""
    n, m, nw = 2, 1, 2

    def pend(x, u, w):
        mass, length_com, gravity, damping = w[0], 0.5, 9.81, 0.1
        return np.array([x[1], u[0] / (mass * length_com * length_com) - gravity * np.sin(x[0]) / length_com
                         - damping * x[1] / (mass * length_com * length_com)], dtype=object)

    def dyn(y, x, u, w, h=0.05):
        return y - (x + h * pend(0.5 * (x + y), u, w))

    dt = Dynamics(dyn, n, n, m, num_parameter=nw, evaluate_hessian=True)
    ct = Cost(lambda x, u, w: 0.1 * (x[0] - w[1]) * (x[0] - w[1]) + 0.1 * x[1] * x[1] + 0.1 * dot(u, u), n, m,
              num_parameter=nw, evaluate_hessian=True)
    cT = Cost(lambda x, u, w: 10.0 * (x[0] - w[1]) * (x[0] - w[1]), n, 0, num_parameter=nw, evaluate_hessian=True)
    con = Constraint(lambda x, u, w: np.array([x[0] * x[0] + u[0] - w[0]], dtype=object), n, m, num_parameter=nw,
                     indices_inequality=[1], evaluate_hessian=True)
    conT = Constraint(lambda x, u, w: np.array([x[0] - w[1]], dtype=object), n, 0, num_parameter=nw, evaluate_hessian=True)
    rng = np.random.Generator(np.random.PCG64(99))
    params = [np.array([1.0 + 0.5 * rng.random(), 3.0 * rng.random()]) for _ in range(T)]
    return dict(dynamics=[dt] * (T - 1), objective=[ct] * (T - 1) + [cT], constraints=[con] * (T - 1) + [conT],
                bounds=[Bound(n, m)] * (T - 1) + [Bound(n, 0)], parameters=params, T=T, n=n, m=m, evaluate_hessian=True)


def build_mpc_pendulum(T=30, x1=(0.0, 0.0), goal=PI):
    """MPC-style instance family (BASELINE north star: "independent trajectory instances (MPC rollouts ...)"): pendulum
    swing-up whose initial state and goal angle are PARAMETERS w_t = [x1_0, x1_1, goal], so that instances of one batch can
    differ through dto_batch.params while sharing one compiled structure."""
    n, m, nw = 2, 1, 3
    dt = Dynamics(lambda y, x, u, w: pendulum_midpoint(y, x, u, w), n, n, m, num_parameter=nw, evaluate_hessian=True)
    ct = Cost(lambda x, u, w: 0.1 * dot(x, x) + 0.1 * dot(u, u), n, m, num_parameter=nw, evaluate_hessian=True)
    cT = Cost(lambda x, u, w: 0.1 * dot(x, x), n, 0, num_parameter=nw, evaluate_hessian=True)
    con1 = Constraint(lambda x, u, w: x - w[0:2], n, m, num_parameter=nw, evaluate_hessian=True)
    conT = Constraint(lambda x, u, w: np.array([x[0] - w[2], x[1]], dtype=object), n, 0, num_parameter=nw, evaluate_hessian=True)
    w = np.array([x1[0], x1[1], goal], dtype=float)
    return dict(dynamics=[dt] * (T - 1), objective=[ct] * (T - 1) + [cT],
                constraints=[con1] + [Constraint() for _ in range(T - 2)] + [conT],
                bounds=[Bound(n, m)] * (T - 1) + [Bound(n, 0)], parameters=[w.copy() for _ in range(T)],
                T=T, n=n, m=m, nw=nw, evaluate_hessian=True)
