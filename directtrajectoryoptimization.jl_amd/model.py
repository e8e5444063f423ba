"""Per-stage model objects: Cost, Dynamics, Constraint, GeneralConstraint, Bound.

Host-side mirror of the reference types (same names, argument order and meaning):

* `Cost`              src/costs.jl:1-45
* `Dynamics`          src/dynamics.jl:1-101 (symbolic ctor and user-Jacobian ctor)
* `Constraint`        src/constraints.jl:1-78
* `GeneralConstraint` src/general_constraint.jl:1-71
* `Bound`             src/bounds.jl:1-16

Where the reference stores `eval`'d Julia closures, these objects store the traced
expression DAGs; the HIP code for them is emitted when a `Solver` is built (plugin.py),
because only then are the stage classes and their neighbours known.  Local sparsity
patterns are kept exactly as the reference keeps them: `[rows, cols]`, 1-based, in the
CSC order `findnz` produces (src/dynamics.jl:29,35).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np

from .symbolic import expr as E
from .symbolic import diff as D


def _as_expr_list(v) -> List[E.Expr]:
    if isinstance(v, E.Expr) or np.isscalar(v):
        return [E.as_expr(v)]
    arr = np.asarray(v, dtype=object).ravel()
    return [E.as_expr(a) for a in arr]


def _one_based(idx: Sequence[int]) -> List[int]:
    return [int(i) + 1 for i in idx]


class Cost:
    """Cost(f, num_state, num_action; num_parameter=0, evaluate_hessian=false) -- src/costs.jl:13-45.

    f(x, u, w) -> scalar.  Gradient is dense over [x; u] (src/costs.jl:20-21); the Hessian is the
    full symmetric sparse one (src/costs.jl:25-28).
    """

    def __init__(self, f: Callable, num_state: int, num_action: int, num_parameter: int = 0,
                 evaluate_hessian: bool = False):
        x = E.variables("x", num_state)
        u = E.variables("u", num_action)
        w = E.variables("w", num_parameter)
        # `f`: the user closure, or (internal: solver-side clone with exact Hessians) ready expressions over x / u / w
        ev = [E.as_expr(e) for e in f] if isinstance(f, (list, tuple)) else _as_expr_list(f(x, u, w))
        if len(ev) != 1:
            raise ValueError("Cost function must return a scalar")
        wrt = list(x) + list(u)
        self.num_state = num_state
        self.num_action = num_action
        self.num_parameter = num_parameter
        self.evaluate_expr = ev
        self.gradient_expr = D.gradient(ev[0], wrt)
        self.num_gradient = num_state + num_action
        if evaluate_hessian:
            r, c, v = D.sparse_hessian(ev[0], wrt)
            self.hessian_expr = v
            self.sparsity = [_one_based(r), _one_based(c)]
            self.num_hessian = len(v)
        else:
            self.hessian_expr = []
            self.sparsity = [[], []]
            self.num_hessian = 0
        self.evaluate_hessian = evaluate_hessian
        # The GPU solver always has the objective Hessian: with evaluate_hessian=false (where the reference lets
        # Ipopt use a limited-memory Hessian) it is the starting point of the per-stage SR1
        # blocks and what remains in the Gauss-Newton inertia fallback.
        r, c, v = D.sparse_hessian(ev[0], wrt)
        self.solver_hessian_expr = v
        self.solver_sparsity = [_one_based(r), _one_based(c)]


class Dynamics:
    """Dynamics(f, num_next_state, num_state, num_action; num_parameter=0, evaluate_hessian=false)
    -- src/dynamics.jl:18-57; or Dynamics(constraint, constraint_jacobian, ny, nx, nu; num_parameter=0)
    -- the user-Jacobian ctor, src/dynamics.jl:59-101 (dense column-major pattern, no Hessian).

    f(y, x, u, w) -> residual of length num_next_state.  Jacobian columns are [x; u; y]
    (src/dynamics.jl:25).
    """

    def __init__(self, f: Callable, *args, num_parameter: int = 0, evaluate_hessian: bool = False):
        jac_f: Optional[Callable] = None
        if args and callable(args[0]):
            jac_f, args = args[0], args[1:]
        num_next_state, num_state, num_action = (int(a) for a in args)
        y = E.variables("y", num_next_state)
        x = E.variables("x", num_state)
        u = E.variables("u", num_action)
        w = E.variables("w", num_parameter)
        ev = [E.as_expr(e) for e in f] if isinstance(f, (list, tuple)) else _as_expr_list(f(y, x, u, w))
        if len(ev) != num_next_state:
            raise ValueError("dynamics residual has wrong length")
        wrt = list(x) + list(u) + list(y)
        self.num_next_state = num_next_state
        self.num_state = num_state
        self.num_action = num_action
        self.num_parameter = num_parameter
        self.evaluate_expr = ev
        if jac_f is None:
            r, c, v = D.sparse_jacobian(ev, wrt)
        else:
            nv = num_state + num_action + num_next_state
            J = np.asarray(jac_f(y, x, u, w), dtype=object).reshape(num_next_state, nv)
            r, c = D.dense_jacobian_pattern(num_next_state, nv)
            v = [E.as_expr(J[i, j]) for i, j in zip(r, c)]
            evaluate_hessian = False
        self.jacobian_expr = v
        self.jacobian_sparsity = [_one_based(r), _one_based(c)]
        self.num_jacobian = len(v)
        if evaluate_hessian:
            lam = E.variables("lam", num_next_state)
            lag = E.as_expr(E.dot(lam, ev))
            r, c, v = D.sparse_hessian(lag, wrt)
            self.hessian_expr = v
            self.hessian_sparsity = [_one_based(r), _one_based(c)]
            self.num_hessian = len(v)
        else:
            self.hessian_expr = []
            self.hessian_sparsity = [[], []]
            self.num_hessian = 0
        self.evaluate_hessian = evaluate_hessian
        self.user_jacobian = jac_f is not None


class Constraint:
    """Constraint(f, num_state, num_action; num_parameter=0, indices_inequality=[], evaluate_hessian=false)
    -- src/constraints.jl:21-64; Constraint() is the empty constraint, src/constraints.jl:66-78.

    f(x, u, w) -> vector; rows listed (1-based) in indices_inequality are `<= 0`, the rest `== 0`.
    """

    def __init__(self, f: Optional[Callable] = None, num_state: int = 0, num_action: int = 0,
                 num_parameter: int = 0, indices_inequality: Sequence[int] = (), evaluate_hessian: bool = False):
        self.num_state = num_state
        self.num_action = num_action
        self.num_parameter = num_parameter
        self.indices_inequality = [int(i) for i in indices_inequality]
        self.evaluate_hessian = evaluate_hessian
        if f is None:
            self.evaluate_expr, self.jacobian_expr, self.hessian_expr = [], [], []
            self.num_constraint = self.num_jacobian = self.num_hessian = 0
            self.jacobian_sparsity = [[], []]
            self.hessian_sparsity = [[], []]
            return
        x = E.variables("x", num_state)
        u = E.variables("u", num_action)
        w = E.variables("w", num_parameter)
        # `f` is the user closure, or (internal use: rows folded in from a stage-local GeneralConstraint) a ready list
        # of expressions over the x / u / w symbols
        ev = [E.as_expr(e) for e in f] if isinstance(f, (list, tuple)) else _as_expr_list(f(x, u, w))
        wrt = list(x) + list(u)
        self.evaluate_expr = ev
        self.num_constraint = len(ev)
        r, c, v = D.sparse_jacobian(ev, wrt)
        self.jacobian_expr = v
        self.jacobian_sparsity = [_one_based(r), _one_based(c)]
        self.num_jacobian = len(v)
        if evaluate_hessian:
            lam = E.variables("lam", self.num_constraint)
            lag = E.as_expr(E.dot(lam, ev))
            r, c, v = D.sparse_hessian(lag, wrt)
            self.hessian_expr = v
            self.hessian_sparsity = [_one_based(r), _one_based(c)]
            self.num_hessian = len(v)
        else:
            self.hessian_expr = []
            self.hessian_sparsity = [[], []]
            self.num_hessian = 0
        for i in self.indices_inequality:
            if not 1 <= i <= self.num_constraint:
                raise ValueError("indices_inequality out of range")


class GeneralConstraint:
    """GeneralConstraint(f, num_variables, num_parameter; indices_inequality=[], evaluate_hessian=false)
    -- src/general_constraint.jl:18-59; GeneralConstraint() is empty, src/general_constraint.jl:61-71.

    f(z, w) over the whole decision vector and the flattened parameter vector.
    """

    def __init__(self, f: Optional[Callable] = None, num_variables: int = 0, num_parameter: int = 0,
                 indices_inequality: Sequence[int] = (), evaluate_hessian: bool = False):
        self.num_variables = num_variables
        self.num_parameter = num_parameter
        self.indices_inequality = [int(i) for i in indices_inequality]
        self.evaluate_hessian = evaluate_hessian
        if f is None:
            self.evaluate_expr, self.jacobian_expr, self.hessian_expr = [], [], []
            self.num_constraint = self.num_jacobian = self.num_hessian = 0
            self.jacobian_sparsity = [[], []]
            self.hessian_sparsity = [[], []]
            return
        z = E.variables("z", num_variables)
        w = E.variables("w", num_parameter)
        ev = _as_expr_list(f(z, w))
        self.evaluate_expr = ev
        self.num_constraint = len(ev)
        r, c, v = D.sparse_jacobian(ev, list(z))
        self.jacobian_expr = v
        self.jacobian_sparsity = [_one_based(r), _one_based(c)]
        self.num_jacobian = len(v)
        if evaluate_hessian:
            lam = E.variables("lam", self.num_constraint)
            lag = E.as_expr(E.dot(lam, ev))
            r, c, v = D.sparse_hessian(lag, list(z))
            self.hessian_expr = v
            self.hessian_sparsity = [_one_based(r), _one_based(c)]
            self.num_hessian = len(v)
        else:
            self.hessian_expr = []
            self.hessian_sparsity = [[], []]
            self.num_hessian = 0


class Bound:
    """Bound(num_state=0, num_action=0; state_lower, state_upper, action_lower, action_upper)
    -- src/bounds.jl:8-14; defaults are -Inf/+Inf."""

    def __init__(self, num_state: int = 0, num_action: int = 0, state_lower=None, state_upper=None,
                 action_lower=None, action_upper=None):
        inf = float("inf")
        self.state_lower = np.full(num_state, -inf) if state_lower is None else np.asarray(state_lower, dtype=float)
        self.state_upper = np.full(num_state, inf) if state_upper is None else np.asarray(state_upper, dtype=float)
        self.action_lower = np.full(num_action, -inf) if action_lower is None else np.asarray(action_lower, dtype=float)
        self.action_upper = np.full(num_action, inf) if action_upper is None else np.asarray(action_upper, dtype=float)


def linear_interpolation(initial_state, final_state, horizon: int):
    """src/utils.jl:1-10."""
    x1 = np.asarray(initial_state, dtype=float)
    xT = np.asarray(final_state, dtype=float)
    return [(xT - x1) / (horizon - 1) * t + x1 for t in range(horizon)]
