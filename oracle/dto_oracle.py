"""ORACLE (test infrastructure only): CPU restatement of the reference's evaluator path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (directtrajectoryoptimization.jl_amd/) never does, and has no CPU path of its own.

Every function follows the reference function named in its docstring (file:line under
/root/reference).  The reference's serial stage loops, index algebra (including its O(T^2) prefix
sums, kept here as plain prefix sums) and `.+=` accumulation order are preserved; 1-based index
values are produced exactly as Julia would, and converted only at the point of numpy indexing.

PARITY STATUS: pinned against (a) the closed-form known answers in the reference's own tests
(test/objective.jl:24-34, test/dynamics.jl:37-59, test/constraints.jl:32-44,
test/hessian_lagrangian.jl:191-205 -- restated in tests/test_oracle_kat.py) and (b) 30-digit
mpmath evaluations of independently derived sympy derivatives (tests/golden/).  The reference itself
(Julia + Symbolics + Ipopt) cannot run in this environment, so iterate-level agreement with Ipopt is
"parity unpinned" (DESIGN.md).
"""
from __future__ import annotations

import numpy as np


# ---------------------------------------------------------------------------------------------------
# src/dynamics.jl
# ---------------------------------------------------------------------------------------------------
def dimensions(dynamics):
    """src/dynamics.jl:206-211"""
    states = [d.num_state for d in dynamics] + [dynamics[-1].num_next_state]
    actions = [d.num_action for d in dynamics] + [0]
    return states, actions


def dyn_sparsity_jacobian(dynamics, num_state, num_action, row_shift=0):
    """src/dynamics.jl:129-142"""
    out = []
    for t, con in enumerate(dynamics):
        col_shift = sum(num_state[:t]) + sum(num_action[:t])
        out += [(r + row_shift, c + col_shift) for r, c in zip(*con.jacobian_sparsity)]
        row_shift += con.num_next_state
    return out


def dyn_sparsity_hessian(dynamics, num_state, num_action):
    """src/dynamics.jl:144-155"""
    out = []
    for t, con in enumerate(dynamics):
        if len(con.hessian_sparsity[0]):
            shift = sum(num_state[:t]) + sum(num_action[:t])
            out += [(r + shift, c + shift) for r, c in zip(*con.hessian_sparsity)]
    return out


def dyn_constraint_indices(dynamics, shift=0):
    """src/dynamics.jl:162-165"""
    out, s = [], shift
    for d in dynamics:
        out.append([s + i for i in range(1, d.num_next_state + 1)])
        s += d.num_next_state
    return out


def dyn_jacobian_indices(dynamics, shift=0):
    """src/dynamics.jl:167-170"""
    out, s = [], shift
    for d in dynamics:
        out.append([s + i for i in range(1, d.num_jacobian + 1)])
        s += d.num_jacobian
    return out


def _hessian_indices(objs, sparsity_of, key, num_state, num_action):
    """src/dynamics.jl:172-186, src/costs.jl:88-104, src/constraints.jl:168-183: position of the first
    occurrence of each shifted (row, col) in the key (`findfirst`)."""
    first = {}
    for i, rc in enumerate(key):
        first.setdefault(rc, i + 1)
    out = []
    for t, o in enumerate(objs):
        sp = sparsity_of(o)
        if len(sp[0]):
            shift = sum(num_state[:t]) + sum(num_action[:t])
            out.append([first[(r + shift, c + shift)] for r, c in zip(*sp)])
        else:
            out.append([])
    return out


def state_indices(dynamics):
    """src/dynamics.jl:188-191"""
    out, s = [], 0
    for d in dynamics:
        out.append([s + i for i in range(1, d.num_state + 1)])
        s += d.num_state + d.num_action
    out.append([s + i for i in range(1, dynamics[-1].num_next_state + 1)])
    return out


def action_indices(dynamics):
    """src/dynamics.jl:193-195"""
    out, s = [], 0
    for d in dynamics:
        out.append([s + d.num_state + i for i in range(1, d.num_action + 1)])
        s += d.num_state + d.num_action
    return out


def state_action_indices(dynamics):
    """src/dynamics.jl:197-200"""
    out, s = [], 0
    for d in dynamics:
        out.append([s + i for i in range(1, d.num_state + d.num_action + 1)])
        s += d.num_state + d.num_action
    out.append([s + i for i in range(1, dynamics[-1].num_next_state + 1)])
    return out


def state_action_next_state_indices(dynamics):
    """src/dynamics.jl:202-204"""
    out, s = [], 0
    for d in dynamics:
        out.append([s + i for i in range(1, d.num_state + d.num_action + d.num_next_state + 1)])
        s += d.num_state + d.num_action
    return out


# ---------------------------------------------------------------------------------------------------
# src/constraints.jl, src/costs.jl, src/general_constraint.jl (layout parts)
# ---------------------------------------------------------------------------------------------------
def con_sparsity_jacobian(constraints, num_state, num_action, row_shift=0):
    """src/constraints.jl:106-120"""
    out = []
    for t, con in enumerate(constraints):
        col_shift = sum(num_state[:t]) + sum(num_action[:t])
        out += [(r + row_shift, c + col_shift) for r, c in zip(*con.jacobian_sparsity)]
        row_shift += con.num_constraint
    return out


def stagewise_sparsity_hessian(objs, sparsity_of, num_state, num_action):
    """src/costs.jl:75-86, src/constraints.jl:122-135"""
    out = []
    for t, o in enumerate(objs):
        sp = sparsity_of(o)
        if len(sp[0]):
            shift = sum(num_state[:t]) + sum(num_action[:t])
            out += [(r + shift, c + shift) for r, c in zip(*sp)]
    return out


def con_constraint_indices(constraints, shift=0):
    """src/constraints.jl:141-153"""
    out = []
    for con in constraints:
        out.append([shift + i for i in range(1, con.num_constraint + 1)])
        shift += con.num_constraint
    return out


def con_jacobian_indices(constraints, shift=0):
    """src/constraints.jl:155-166"""
    out = []
    for con in constraints:
        out.append([shift + i for i in range(1, con.num_jacobian + 1)])
        shift += con.num_jacobian
    return out


# ---------------------------------------------------------------------------------------------------
# src/data.jl: NLPData
# ---------------------------------------------------------------------------------------------------
class NLPData:
    """src/data.jl:150-220 (constructor) + src/moi.jl (methods)."""

    def __init__(self, dynamics, objective, constraints, bounds, evaluate_hessian=False, general_constraint=None,
                 parameters=None):
        self.dynamics, self.objective, self.constraints, self.bounds = dynamics, objective, constraints, bounds
        self.general = general_constraint
        g = general_constraint
        ns, na = dimensions(dynamics)
        self.state_dimensions, self.action_dimensions = ns, na
        T = len(ns)
        self.T = T
        self.parameters = parameters if parameters is not None else [np.zeros(0) for _ in range(T)]
        self.num_variables = sum(ns) + sum(na)
        num_dynamics = sum(d.num_next_state for d in dynamics)
        num_stage = sum(c.num_constraint for c in constraints)
        num_general = g.num_constraint if g else 0
        self.num_dynamics, self.num_stage, self.num_general = num_dynamics, num_stage, num_general
        self.num_constraint = num_dynamics + num_stage + num_general
        njd = sum(d.num_jacobian for d in dynamics)
        njs = sum(c.num_jacobian for c in constraints)
        njg = g.num_jacobian if g else 0
        self.num_jacobian = njd + njs + njg
        # Jacobian sparsity (src/data.jl:170-175)
        sp_dyn = dyn_sparsity_jacobian(dynamics, ns, na, row_shift=0)
        sp_con = con_sparsity_jacobian(constraints, ns, na, row_shift=num_dynamics)
        sp_gen = [(r + num_dynamics + num_stage, c) for r, c in zip(*g.jacobian_sparsity)] if g else []
        self.jacobian_sparsity = sp_dyn + sp_con + sp_gen
        # Hessian sparsity (src/data.jl:178-187)
        sp_obj_h = stagewise_sparsity_hessian(objective, lambda o: o.sparsity, ns, na)
        sp_dyn_h = dyn_sparsity_hessian(dynamics, ns, na)
        sp_con_h = stagewise_sparsity_hessian(constraints, lambda o: o.hessian_sparsity, ns, na)
        sp_gen_h = list(zip(*g.hessian_sparsity)) if (g and len(g.hessian_sparsity[0])) else []
        raw = sp_obj_h + sp_dyn_h + sp_con_h + sp_gen_h
        self.hessian_lagrangian_sparsity = sorted(set(raw))  # sort(unique(...)), row then column
        self.num_hessian_lagrangian = len(raw)
        key = self.hessian_lagrangian_sparsity
        # indices (src/data.jl:61-104)
        self.idx_dynamics_constraints = dyn_constraint_indices(dynamics, 0)
        self.idx_dynamics_jacobians = dyn_jacobian_indices(dynamics, 0)
        self.idx_stage_constraints = con_constraint_indices(constraints, num_dynamics)
        self.idx_stage_jacobians = con_jacobian_indices(constraints, njd)
        self.idx_general_constraint = [num_dynamics + num_stage + i for i in range(1, num_general + 1)]
        self.idx_general_jacobian = [njd + njs + i for i in range(1, njg + 1)]
        self.idx_objective_hessians = _hessian_indices(objective, lambda o: o.sparsity, key, ns, na)
        self.idx_dynamics_hessians = _hessian_indices(dynamics, lambda o: o.hessian_sparsity, key, ns, na)
        self.idx_stage_hessians = _hessian_indices(constraints, lambda o: o.hessian_sparsity, key, ns, na)
        self.idx_states = state_indices(dynamics)
        self.idx_actions = action_indices(dynamics)
        self.idx_state_action = state_action_indices(dynamics)
        self.idx_state_action_next_state = state_action_next_state_indices(dynamics)
        # bounds (src/data.jl:123-148)
        lo = -np.inf * np.ones(self.num_variables)
        hi = np.inf * np.ones(self.num_variables)
        for t, bnd in enumerate(bounds):
            if len(bnd.state_lower) > 0:
                lo[np.array(self.idx_states[t]) - 1] = bnd.state_lower
            if len(bnd.state_upper) > 0:
                hi[np.array(self.idx_states[t]) - 1] = bnd.state_upper
            if len(bnd.action_lower) > 0 and t < len(self.idx_actions):
                lo[np.array(self.idx_actions[t]) - 1] = bnd.action_lower
            if len(bnd.action_upper) > 0 and t < len(self.idx_actions):
                hi[np.array(self.idx_actions[t]) - 1] = bnd.action_upper
        self.variable_bounds = [lo, hi]
        clo, chi = np.zeros(self.num_constraint), np.zeros(self.num_constraint)
        for t, con in enumerate(constraints):
            for i in con.indices_inequality:
                clo[self.idx_stage_constraints[t][i - 1] - 1] = -np.inf
        if g:
            for i in g.indices_inequality:
                clo[num_dynamics + num_stage + i - 1] = -np.inf
        self.constraint_bounds = [clo, chi]
        self.hessian_lagrangian = evaluate_hessian

    # ---- src/data.jl:258-278
    def trajectory(self, z):
        z = np.asarray(z)
        xs = [z[np.array(i, dtype=int) - 1] for i in self.idx_states]
        us = [z[np.array(i, dtype=int) - 1] for i in self.idx_actions] + [np.zeros(0)]
        return xs, us

    def duals(self, mu):
        mu = np.asarray(mu)
        ld = [mu[np.array(i, dtype=int) - 1] for i in self.idx_dynamics_constraints]
        lc = [mu[np.array(i, dtype=int) - 1] if len(i) else np.zeros(0) for i in self.idx_stage_constraints]
        lg = mu[np.array(self.idx_general_constraint, dtype=int) - 1] if self.num_general else np.zeros(0)
        return ld, lc, lg

    # ---- src/moi.jl:1-13 + src/costs.jl:49-56
    def eval_objective(self, z, hp=False):
        xs, us = self.trajectory(z)
        J = 0.0
        for t, cost in enumerate(self.objective):
            J = J + cost.evaluate(xs[t], us[t], self.parameters[t], hp=hp)[0]
        return J

    # ---- src/moi.jl:15-30 + src/costs.jl:58-64
    def eval_objective_gradient(self, z, hp=False):
        g = [0.0] * self.num_variables if hp else np.zeros(self.num_variables)
        xs, us = self.trajectory(z)
        for t, cost in enumerate(self.objective):
            v = cost.gradient(xs[t], us[t], self.parameters[t], hp=hp)
            for k, i in enumerate(self.idx_state_action[t]):
                g[i - 1] = g[i - 1] + v[k]
        return g

    # ---- src/moi.jl:32-50 + src/dynamics.jl:103-109 + src/constraints.jl:80-86 + src/general_constraint.jl:73-77
    def eval_constraint(self, z, hp=False):
        c = [0.0] * self.num_constraint if hp else np.zeros(self.num_constraint)
        xs, us = self.trajectory(z)
        for t, con in enumerate(self.dynamics):
            v = con.evaluate(xs[t + 1], xs[t], us[t], self.parameters[t], hp=hp)
            for k, i in enumerate(self.idx_dynamics_constraints[t]):
                c[i - 1] = v[k]
        for t, con in enumerate(self.constraints):
            if con.num_constraint:
                v = con.evaluate(xs[t], us[t], self.parameters[t], hp=hp)
                for k, i in enumerate(self.idx_stage_constraints[t]):
                    c[i - 1] = v[k]
        if self.num_general:
            v = self.general.evaluate(list(z), np.concatenate(self.parameters) if len(self.parameters) else [], hp=hp)
            for k, i in enumerate(self.idx_general_constraint):
                c[i - 1] = v[k]
        return c

    # ---- src/moi.jl:52-70 + src/dynamics.jl:111-117 + src/constraints.jl:88-94 + src/general_constraint.jl:79-83
    def eval_constraint_jacobian(self, z, hp=False):
        J = [0.0] * self.num_jacobian if hp else np.zeros(self.num_jacobian)
        xs, us = self.trajectory(z)
        for t, con in enumerate(self.dynamics):
            v = con.jacobian(xs[t + 1], xs[t], us[t], self.parameters[t], hp=hp)
            for k, i in enumerate(self.idx_dynamics_jacobians[t]):
                J[i - 1] = v[k]
        for t, con in enumerate(self.constraints):
            if con.num_jacobian:
                v = con.jacobian(xs[t], us[t], self.parameters[t], hp=hp)
                for k, i in enumerate(self.idx_stage_jacobians[t]):
                    J[i - 1] = v[k]
        if self.num_general:
            v = self.general.jacobian(list(z), np.concatenate(self.parameters) if len(self.parameters) else [], hp=hp)
            for k, i in enumerate(self.idx_general_jacobian):
                J[i - 1] = v[k]
        return J

    # ---- src/moi.jl:72-120 + src/costs.jl:66-73 + src/dynamics.jl:119-127 + src/constraints.jl:96-104
    def eval_hessian_lagrangian(self, z, scaling, mu, hp=False):
        n = len(self.hessian_lagrangian_sparsity)
        H = [0.0] * n if hp else np.zeros(n)
        xs, us = self.trajectory(z)
        ld, lc, lg = self.duals(mu)
        for t, cost in enumerate(self.objective):
            if cost.num_hessian:
                v = cost.hessian(xs[t], us[t], self.parameters[t], hp=hp)
                for k, i in enumerate(self.idx_objective_hessians[t]):
                    H[i - 1] = H[i - 1] + v[k] * scaling
        for t, con in enumerate(self.dynamics):
            if con.num_hessian:
                v = con.hessian(xs[t + 1], xs[t], us[t], self.parameters[t], ld[t], hp=hp)
                for k, i in enumerate(self.idx_dynamics_hessians[t]):
                    H[i - 1] = H[i - 1] + v[k]
        for t, con in enumerate(self.constraints):
            if con.num_hessian:
                v = con.hessian(xs[t], us[t], self.parameters[t], lc[t], hp=hp)
                for k, i in enumerate(self.idx_stage_hessians[t]):
                    H[i - 1] = H[i - 1] + v[k]
        return H

    # ---- MOI.features_available etc. (src/moi.jl:122-125)
    def features_available(self):
        return ["Grad", "Jac", "Hess"] if self.hessian_lagrangian else ["Grad", "Jac"]

    def jacobian_structure(self):
        return self.jacobian_sparsity

    def hessian_lagrangian_structure(self):
        return self.hessian_lagrangian_sparsity
