"""Solver / NLPData mirror over the C-ABI.

* `NLPData`  -- the evaluator handed to Ipopt in the reference (src/data.jl:106-121,150-220) and its
  MOI methods (src/moi.jl:1-125).  Here it is a thin object around a `dto_problem*`; every method is
  one C-ABI call into libdto_hip.so, which runs the HIP kernels.  Nothing is evaluated in Python.
* `Solver`   -- src/solver.jl:1-47: same constructor arguments and helper functions
  (`initialize_states!` -> initialize_states, `initialize_controls!` -> initialize_controls,
  `solve!` -> solve, get_trajectory).
* `Options`  -- src/options.jl:6-36 (the tolerances that define convergence; print/file options are
  accepted and ignored).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import capi
from .model import Bound, Constraint, Cost, Dynamics, GeneralConstraint
from .plugin import Structure, build_plugin


@dataclass
class Options:
    """Base.@kwdef mutable struct Options (src/options.jl:6-36)."""
    tol: float = 1e-6
    s_max: float = 100.0
    max_iter: int = 1000
    max_cpu_time: float = 300.0
    dual_inf_tol: float = 1.0
    constr_viol_tol: float = 1.0e-3
    compl_inf_tol: float = 1.0e-3
    acceptable_tol: float = 1.0e-6
    acceptable_iter: int = 15
    acceptable_dual_inf_tol: float = 1.0e10
    acceptable_constr_viol_tol: float = 1.0e-2
    acceptable_compl_inf_tol: float = 1.0e-2
    acceptable_obj_change_tol: float = 1.0e-5
    diverging_iterates_tol: float = 1.0e8
    mu_target: float = 1.0e-4
    print_level: int = 5
    output_file: str = "output.txt"
    print_user_options: str = "no"
    print_info_string: str = "no"
    inf_pr_output: str = "original"
    print_frequency_iter: int = 1
    print_frequency_time: float = 0.0
    skip_finalize_solution_call: str = "no"
    # not a reference field: what the GPU solver uses for the Hessian of the Lagrangian (dto_options.hessian_approximation).
    # "auto": for a problem built with evaluate_hessian=false -- where the reference leaves Ipopt on its limited-memory Hessian --
    # compact L-BFGS ("lbfgs", round 5) on the lane-per-instance path; on the paths that have no such mode (17 .. 64 states,
    # multi-knot GeneralConstraint rows) exact second derivatives derived from the traced expressions with a one-time
    # HessianModeNotice.  "exact": those second derivatives, no notice (the fastest mode here: they cost nothing extra);
    # "lbfgs": also for a problem WITH Hessians; "sr1": per-stage SR1 blocks (Solver.hessian_mode reports what runs).
    hessian_approximation: str = "auto"
    # not reference fields either (dto_options.line_search / penalty_switch_theta, include/dto.h): "penalty-filter" chooses the
    # step size on the l1 exact-penalty function while max |c_i| > penalty_switch_theta and hands over to Ipopt's filter then;
    # "filter" is the filter line search from the first iteration (what Ipopt runs for the reference)
    line_search: str = "penalty-filter"
    penalty_switch_theta: float = 1.0
    # not a reference field (dto_options.kkt_refinement, ABI 4): passes of iterative refinement per KKT step.  0 for the batches
    # that fill the GPU (their sequential sweeps are within 1e-9 of an extended-precision solve); one pass brings the
    # time-partitioned sweeps of small batches from 2.5e-8 (5e-6 at delta_w = 0) to 1e-10 for one more factor + solve per iteration
    kkt_refinement: int = 0
    # not a reference field: how GeneralConstraint rows that couple several knots are solved.  "auto": rows that are sums of
    # one-knot terms ride accumulator states through the ordinary device loop (solver.py: accumulate_general_constraint) when the
    # state stays within the lane-per-instance kernels' 16, everything else takes the bordered path; "border": always the border
    general_rows: str = "auto"


class Indices:
    """TrajectoryOptimizationIndices (src/data.jl:44-59), 1-based vectors fetched lazily from the runtime."""

    _FIELDS = dict(states=capi.IDX_STATE, actions=capi.IDX_ACTION, state_action=capi.IDX_STATE_ACTION,
                   state_action_next_state=capi.IDX_STATE_ACTION_NEXT,
                   dynamics_constraints=capi.IDX_DYNAMICS_CONSTRAINT, dynamics_jacobians=capi.IDX_DYNAMICS_JACOBIAN,
                   dynamics_hessians=capi.IDX_DYNAMICS_HESSIAN, stage_constraints=capi.IDX_STAGE_CONSTRAINT,
                   stage_jacobians=capi.IDX_STAGE_JACOBIAN, stage_hessians=capi.IDX_STAGE_HESSIAN,
                   objective_hessians=capi.IDX_OBJECTIVE_HESSIAN)

    def __init__(self, nlp: "NLPData"):
        self._nlp = nlp
        self._cache = {}

    # general_constraint / general_jacobian / general_hessian (src/data.jl:52-54,70-73,80): plain ranges behind the dynamics
    # and stage blocks; the Hessian of a GeneralConstraint is not supported (the reference's own call is broken)
    @property
    def general_constraint(self):
        s = self._nlp.sizes
        lo = int(s.num_constraint_dynamics + s.num_constraint_stage)
        return list(range(lo + 1, int(s.num_constraint) + 1))

    @property
    def general_jacobian(self):
        s = self._nlp.sizes
        lo = int(s.num_jacobian_dynamics + s.num_jacobian_stage)
        return list(range(lo + 1, int(s.num_jacobian) + 1))

    @property
    def general_hessian(self):
        return []

    def __getattr__(self, name):
        if name.startswith("_") or name not in self._FIELDS:
            raise AttributeError(name)
        if name not in self._cache:
            T = self._nlp.T
            last = T - 1 if name in ("state_action_next_state", "dynamics_constraints", "dynamics_jacobians",
                                     "dynamics_hessians") else T
            if name == "actions":
                last = T - 1
            self._cache[name] = [self._nlp._stage_indices(self._FIELDS[name], t) for t in range(1, last + 1)]
        return self._cache[name]


class NLPData:
    """The NLP evaluator (src/data.jl:106-121) backed by a device-resident problem."""

    def __init__(self, dynamics: Sequence[Dynamics], objective: Sequence[Cost], constraints: Sequence[Constraint],
                 bounds: Sequence[Bound], evaluate_hessian: bool = False,
                 general_constraint: Optional[GeneralConstraint] = None, parameters=None, name: str = "model"):
        self.structure = Structure(dynamics, objective, constraints, general_constraint, evaluate_hessian)
        self.T = self.structure.T
        self.hessian_lagrangian = bool(evaluate_hessian)
        self.plugin_path = build_plugin(self.structure, name)
        lib = capi.lib()
        # variable bounds (primal_bounds, src/data.jl:123-133)
        st = self.structure
        nxs, nus = [], []
        for t in range(self.T):
            d, p, c, kc = st.kinds[st.stage_kind[t]]
            nxs.append(st.cost[c].num_state)
            nus.append(st.cost[c].num_action)
        nz = sum(nxs) + sum(nus)
        if len(bounds) != self.T:
            raise ValueError(f"bounds must hold one Bound per stage: got {len(bounds)}, horizon {self.T} (src/data.jl:123-133)")
        lo = np.full(nz, -np.inf)
        hi = np.full(nz, np.inf)
        off = 0
        for t, bnd in enumerate(bounds):
            nx, nu = nxs[t], nus[t]
            if len(bnd.state_lower) > 0:
                lo[off:off + nx] = bnd.state_lower
            if len(bnd.state_upper) > 0:
                hi[off:off + nx] = bnd.state_upper
            if len(bnd.action_lower) > 0 and nu > 0:
                lo[off + nx:off + nx + nu] = bnd.action_lower
            if len(bnd.action_upper) > 0 and nu > 0:
                hi[off + nx:off + nx + nu] = bnd.action_upper
            off += nx + nu
        par = None
        if parameters is not None:
            # src/solver.jl:10: one vector per stage (the reference default appends an empty one for stage T).  Stage t reads
            # w_t through its dynamics, cost and constraint; the flattened vector (src/data.jl:218) holds, per stage, as many
            # entries as the largest of the three asks for.  A vector that is too short is an error; extra entries are never
            # read by the closures and are dropped.
            nws = []
            for t in range(self.T):
                d, pd, c, kc = st.kinds[st.stage_kind[t]]
                nw = st.cost[c].num_parameter
                if d >= 0:
                    nw = max(nw, st.dyn[d].num_parameter)
                if kc >= 0:
                    nw = max(nw, st.con[kc].num_parameter)
                nws.append(nw)
            plist = list(parameters)
            if len(plist) == self.T - 1:
                plist.append(np.zeros(0))
            if len(plist) != self.T:
                raise ValueError(f"parameters must hold one vector per stage: got {len(parameters)}, horizon {self.T}")
            flat = []
            for t, w in enumerate(plist):
                w = np.asarray(w, dtype=float).ravel()
                if w.size < nws[t]:
                    raise ValueError(f"parameters[{t}] has {w.size} entries, stage {t + 1} reads {nws[t]}")
                flat.append(w[:nws[t]])
            par = np.concatenate(flat) if flat else np.zeros(0)
        kinds = np.asarray(st.stage_kind, dtype=np.int32)
        spec = capi.ProblemSpec()
        spec.abi_version = capi.DTO_ABI_VERSION
        spec.model_library = self.plugin_path.encode()
        spec.horizon = self.T
        spec.stage_kind = kinds.ctypes.data_as(capi.c_int32_p)
        spec.variable_lower = capi.dptr(lo)
        spec.variable_upper = capi.dptr(hi)
        spec.parameters = capi.dptr(par) if (par is not None and par.size) else None
        spec.num_parameters = int(par.size) if (par is not None and par.size) else 0
        spec.evaluate_hessian = 1 if evaluate_hessian else 0
        h = C.c_void_p()
        capi.check(lib.dto_problem_create(C.byref(spec), C.byref(h)))
        self._h = h
        self._lib = lib
        s = capi.Sizes()
        capi.check(lib.dto_sizes(self._h, C.byref(s)))
        self.sizes = s
        self.num_variables = int(s.num_variables)
        self.num_constraint = int(s.num_constraint)
        self.num_jacobian = int(s.num_jacobian)
        self.num_hessian_lagrangian = int(s.nnz_hess_raw)  # duplicate-counting, as src/data.jl:187
        self.num_parameters = int(s.num_parameters)
        self.state_dimensions = nxs
        self.action_dimensions = nus
        self.indices = Indices(self)
        self._jac_structure = None
        self._hess_structure = None

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._lib.dto_problem_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- structure queries
    def _stage_indices(self, which: int, t: int) -> List[int]:
        n = C.c_int64()
        capi.check(self._lib.dto_stage_indices(self._h, which, t, None, C.byref(n)))
        out = np.zeros(max(1, n.value), dtype=np.int64)
        capi.check(self._lib.dto_stage_indices(self._h, which, t, out.ctypes.data_as(capi.c_int64_p), C.byref(n)))
        return out[:n.value].tolist()

    def features_available(self) -> List[str]:
        bits = C.c_int()
        capi.check(self._lib.dto_features_available(self._h, C.byref(bits)))
        return ["Grad", "Jac"] + (["Hess"] if bits.value & 4 else [])

    def jacobian_structure(self):
        """MOI.jacobian_structure (src/moi.jl:124): list of 1-based (row, col)."""
        if self._jac_structure is None:
            r = np.zeros(max(1, self.num_jacobian), dtype=np.int64)
            c = np.zeros(max(1, self.num_jacobian), dtype=np.int64)
            capi.check(self._lib.dto_jacobian_structure(self._h, r.ctypes.data_as(capi.c_int64_p),
                                                        c.ctypes.data_as(capi.c_int64_p)))
            self._jac_structure = list(zip(r[:self.num_jacobian].tolist(), c[:self.num_jacobian].tolist()))
        return self._jac_structure

    def hessian_lagrangian_structure(self):
        """MOI.hessian_lagrangian_structure (src/moi.jl:125)."""
        if self._hess_structure is None:
            n = int(self.sizes.nnz_hess_key)
            r = np.zeros(max(1, n), dtype=np.int64)
            c = np.zeros(max(1, n), dtype=np.int64)
            capi.check(self._lib.dto_hessian_structure(self._h, r.ctypes.data_as(capi.c_int64_p),
                                                       c.ctypes.data_as(capi.c_int64_p)))
            self._hess_structure = list(zip(r[:n].tolist(), c[:n].tolist()))
        return self._hess_structure

    @property
    def jacobian_sparsity(self):
        return self.jacobian_structure()

    @property
    def hessian_lagrangian_sparsity(self):
        return self.hessian_lagrangian_structure()

    @property
    def variable_bounds(self):
        lo = np.zeros(self.num_variables)
        hi = np.zeros(self.num_variables)
        capi.check(self._lib.dto_variable_bounds(self._h, capi.dptr(lo), capi.dptr(hi)))
        return [lo, hi]

    @property
    def constraint_bounds(self):
        lo = np.zeros(max(1, self.num_constraint))
        hi = np.zeros(max(1, self.num_constraint))
        capi.check(self._lib.dto_constraint_bounds(self._h, capi.dptr(lo), capi.dptr(hi)))
        return [lo[:self.num_constraint], hi[:self.num_constraint]]

    # -- the five MOI evaluator methods (host vectors in, host vectors out; computed on the GPU)
    @staticmethod
    def _vec(x, n):
        a = np.ascontiguousarray(x, dtype=np.float64)
        if a.size != n:
            raise ValueError(f"expected a vector of length {n}, got {a.size}")
        return a

    def eval_objective(self, variables) -> float:
        x = self._vec(variables, self.num_variables)
        f = C.c_double()
        capi.check(self._lib.dto_eval_f(self._h, capi.dptr(x), C.byref(f)))
        return f.value

    def eval_objective_gradient(self, gradient, variables) -> None:
        x = self._vec(variables, self.num_variables)
        assert gradient.dtype == np.float64 and gradient.flags.c_contiguous and gradient.size == self.num_variables
        capi.check(self._lib.dto_eval_grad_f(self._h, capi.dptr(x), capi.dptr(gradient)))

    def eval_constraint(self, violations, variables) -> None:
        x = self._vec(variables, self.num_variables)
        assert violations.dtype == np.float64 and violations.flags.c_contiguous and violations.size == self.num_constraint
        capi.check(self._lib.dto_eval_g(self._h, capi.dptr(x), capi.dptr(violations)))

    def eval_constraint_jacobian(self, jacobian, variables) -> None:
        x = self._vec(variables, self.num_variables)
        assert jacobian.dtype == np.float64 and jacobian.flags.c_contiguous and jacobian.size == self.num_jacobian
        capi.check(self._lib.dto_eval_jac_g(self._h, capi.dptr(x), capi.dptr(jacobian)))

    def eval_hessian_lagrangian(self, hessian, variables, scaling, duals) -> None:
        x = self._vec(variables, self.num_variables)
        mu = self._vec(duals, self.num_constraint)
        assert hessian.dtype == np.float64 and hessian.flags.c_contiguous and hessian.size == int(self.sizes.nnz_hess_key)
        capi.check(self._lib.dto_eval_h(self._h, capi.dptr(x), float(scaling), capi.dptr(mu), capi.dptr(hessian)))

    # -- batched device-pointer forms (torch tensors or raw pointers)
    def _batch(self, x_ptr: int, B: int, ldx: int, stream: int = 0, params_ptr: int = 0, ldp: int = 0):
        b = capi.Batch()
        b.B, b.x, b.ldx, b.params, b.ldp, b.stream = B, x_ptr, ldx, params_ptr or None, ldp, stream or None
        return b

    def eval_objective_batch(self, x_ptr, B, ldx, out_ptr, stream=0):
        b = self._batch(x_ptr, B, ldx, stream)
        capi.check(self._lib.dto_eval_f_batch(self._h, C.byref(b), out_ptr))

    def eval_objective_gradient_batch(self, x_ptr, B, ldx, out_ptr, ldo, stream=0):
        b = self._batch(x_ptr, B, ldx, stream)
        capi.check(self._lib.dto_eval_grad_f_batch(self._h, C.byref(b), out_ptr, ldo))

    def eval_constraint_batch(self, x_ptr, B, ldx, out_ptr, ldo, stream=0):
        b = self._batch(x_ptr, B, ldx, stream)
        capi.check(self._lib.dto_eval_g_batch(self._h, C.byref(b), out_ptr, ldo))

    def eval_constraint_jacobian_batch(self, x_ptr, B, ldx, out_ptr, ldo, stream=0):
        b = self._batch(x_ptr, B, ldx, stream)
        capi.check(self._lib.dto_eval_jac_g_batch(self._h, C.byref(b), out_ptr, ldo))

    def eval_hessian_lagrangian_batch(self, x_ptr, B, ldx, sigma, mu_ptr, ldmu, out_ptr, ldo, stream=0):
        b = self._batch(x_ptr, B, ldx, stream)
        capi.check(self._lib.dto_eval_h_batch(self._h, C.byref(b), float(sigma), mu_ptr, ldmu, out_ptr, ldo))

    def kkt_csr_structure(self):
        """(row_ptr [dim + 1], col_ind [nnz]) of K = [H + dw I, J'; J, -dc I] in the reference ordering, 1-based
        (include/dto.h: dto_kkt_csr_structure; the matrix of examples/pendulum/pendulum.jl:138-198)."""
        dim, nnz = C.c_int64(0), C.c_int64(0)
        capi.check(self._lib.dto_kkt_csr_structure(self._h, None, None, C.byref(dim), C.byref(nnz)))
        rp = np.zeros(dim.value + 1, dtype=np.int64)
        ci = np.zeros(max(1, nnz.value), dtype=np.int64)
        capi.check(self._lib.dto_kkt_csr_structure(self._h, rp.ctypes.data_as(capi.c_int64_p), ci.ctypes.data_as(capi.c_int64_p), None, None))
        return rp, ci[:nnz.value]

    def kkt_csr_values_batch(self, B, h_ptr, ldh, j_ptr, ldj, delta_w, delta_c, out_ptr, ldo, stream=0):
        capi.check(self._lib.dto_kkt_csr_values_batch(self._h, int(B), h_ptr, ldh, j_ptr, ldj, float(delta_w), float(delta_c),
                                                      out_ptr, ldo, stream or None))


def _c_options(o: "Options", check_every: int = 10, lbfgs: bool = False) -> "capi.COptions":
    c = capi.COptions()
    capi.check(capi.lib().dto_options_default(C.byref(c)))
    c.tol, c.s_max, c.max_iter = o.tol, o.s_max, int(o.max_iter)
    c.dual_inf_tol, c.constr_viol_tol, c.compl_inf_tol = o.dual_inf_tol, o.constr_viol_tol, o.compl_inf_tol
    c.check_every = check_every
    c.max_cpu_time = float(o.max_cpu_time)
    c.acceptable_tol, c.acceptable_iter = float(o.acceptable_tol), int(o.acceptable_iter)
    c.acceptable_dual_inf_tol, c.acceptable_constr_viol_tol = float(o.acceptable_dual_inf_tol), float(o.acceptable_constr_viol_tol)
    c.acceptable_compl_inf_tol, c.acceptable_obj_change_tol = float(o.acceptable_compl_inf_tol), float(o.acceptable_obj_change_tol)
    c.diverging_iterates_tol, c.mu_target = float(o.diverging_iterates_tol), float(o.mu_target)
    if o.line_search not in ("penalty-filter", "filter"):
        raise ValueError(f"Options.line_search must be 'penalty-filter' or 'filter', not {o.line_search!r}")
    c.line_search = capi.DTO_LS_PENALTY_FILTER if o.line_search == "penalty-filter" else capi.DTO_LS_FILTER
    c.penalty_switch_theta = float(o.penalty_switch_theta)
    c.hessian_approximation = capi.DTO_HESSIAN_LBFGS if lbfgs else capi.DTO_HESSIAN_EXACT
    c.kkt_refinement = int(o.kkt_refinement)
    return c


class HessianModeNotice(UserWarning):
    """Issued once per process when a problem built with evaluate_hessian=false is solved with second derivatives anyway."""


_NOTICED = False


def _notice_default_mode():
    """The reference's default (evaluate_hessian=false, src/solver.jl:7) leaves Ipopt on its limited-memory quasi-Newton Hessian.
    Since round 5 so does this solver on the lane-per-instance path (compact L-BFGS).  On the paths without that mode -- 17 .. 64
    states (tile kernels), multi-knot GeneralConstraint rows (bordered system) -- the traced expressions are differentiated twice
    and the exact Hessian of the Lagrangian is used instead: said out loud, once, because it is NOT what the reference does."""
    global _NOTICED
    if not _NOTICED:
        _NOTICED = True
        import warnings
        warnings.warn("evaluate_hessian=false on a problem with more than 16 states or with multi-knot GeneralConstraint rows: the GPU "
                      "solver differentiates the traced expressions twice and iterates with the exact Hessian of the Lagrangian (the "
                      "reference leaves Ipopt on its limited-memory quasi-Newton Hessian here, and so does this solver on smaller "
                      "problems); Options(hessian_approximation='exact') silences this notice",
                      HessianModeNotice, stacklevel=3)


class Solver:
    """Solver(dynamics, objective, constraints, bounds; evaluate_hessian=false,
    general_constraint=GeneralConstraint(), options=Options(), parameters=...) -- src/solver.jl:6-21."""

    def __init__(self, dynamics, objective, constraints, bounds, evaluate_hessian: bool = False,
                 general_constraint: Optional[GeneralConstraint] = None, options: Optional[Options] = None,
                 parameters=None, name: str = "model"):
        self.options = options or Options()
        self.nlp = NLPData(dynamics, objective, constraints, bounds, evaluate_hessian=evaluate_hessian,
                           general_constraint=general_constraint, parameters=parameters, name=name)
        self._z0 = np.zeros(self.nlp.num_variables)
        self._solution = None
        # The block-tridiagonal solver has no border for dense coupling rows.  A GeneralConstraint whose rows each
        # touch the variables of ONE knot (the reference's own use, test/solve.jl:273: z[end-1:end] - xT) is folded
        # into that knot's stage constraint for the solve; the evaluator callbacks keep the reference layout and the
        # multipliers are mapped back to it ([dynamics; stage; general], src/data.jl:64-75).
        self._solve_nlp = self.nlp
        self._mu_to_reference = None
        self.general_rows_path = None        # "accumulators" | "border" | "folded" | None: how GeneralConstraint rows are solved
        s_dyn, s_obj, s_con, s_eh, changed = list(dynamics), list(objective), list(constraints), bool(evaluate_hessian), False
        ha = self.options.hessian_approximation
        if ha not in ("auto", "exact", "sr1", "lbfgs"):
            raise ValueError("Options.hessian_approximation must be 'auto', 'exact', 'lbfgs' or 'sr1'")
        # How the solver gets its Hessian of the Lagrangian (reported: Solver.hessian_mode).  evaluate_hessian=False is the
        # reference's default (src/solver.jl:7) and leaves Ipopt on its limited-memory BFGS: "auto" does the same ("lbfgs",
        # dto_options.hessian_approximation = DTO_HESSIAN_LBFGS) wherever the lane-per-instance solver path runs the problem; the
        # tile path (17 .. 64 states), multi-knot GeneralConstraint rows and user-Jacobian dynamics keep round 4's substitutes
        # (exact second derivatives of the traced expressions, announced once; per-stage SR1 blocks).
        want_lbfgs = (ha == "lbfgs") or (ha == "auto" and not s_eh)
        if ha == "lbfgs" and s_eh:
            pass   # asked for explicitly on a problem that has Hessians: the approximation is used all the same
        self.hessian_mode = "exact" if (s_eh and not want_lbfgs) else ("sr1" if ha == "sr1" else ("lbfgs" if want_lbfgs else "exact-from-trace"))
        if not s_eh and ha != "sr1":
            # Default mode: the expressions are there, so the solver differentiates them twice itself; the MOI surface of
            # self.nlp still reports [:Grad, :Jac] exactly like the reference (src/moi.jl:122).
            up = _with_exact_hessians(s_dyn, s_obj, s_con)
            if up is not None:
                s_dyn, s_obj, s_con = up
                s_eh, changed = True, True
        gen = general_constraint if (general_constraint is not None and general_constraint.num_constraint > 0) else None
        from .plugin import WIDE_MIN_STATE as _WMIN, WIDE_STATE as _WST
        if self.options.general_rows not in ("auto", "border"):
            raise ValueError("Options.general_rows must be 'auto' or 'border'")
        n_max = max(d.num_state for d in s_dyn)
        self._pad = None
        s_bounds = bounds
        # ---- 1. GeneralConstraint rows become stage structure where they can (both transformations feed the steps below):
        #      rows of one knot each join that knot's stage constraint; coupling rows that are sums of one-knot terms ride
        #      accumulator states (Options.general_rows = "border" keeps those on the bordered path)
        if gen is not None:
            folded = fold_general_constraint(s_dyn, s_obj, s_con, gen, s_eh)
            if folded is not None:
                s_con, self._mu_to_reference = folded
                gen, changed = None, True
        if gen is not None and s_eh and self.options.general_rows == "auto":
            acc = accumulate_general_constraint(s_dyn, s_obj, s_con, s_bounds, gen, s_eh, max_state=16 if n_max < _WMIN else _WST - 1)
            if acc is not None:
                s_dyn, s_obj, s_con, s_bounds, zmap, mumap, musign = acc
                self._pad = (zmap, mumap, musign)
                self.general_rows_path = "accumulators"
                gen, changed = None, True
        # ---- 2. 17 .. 63 states: embedded in the 64 states of the tile kernels (padding states fixed at zero, stage constraints as
        #      auxiliary states); solve() / get_trajectory() map between the layouts, the batched entry points take the solver's
        #      (pad_batch / unpad_batch).  After step 1 the maps compose.
        n_max = max(d.num_state for d in s_dyn)
        wide_embedded = False
        self._pins = None
        if gen is None and s_eh:
            padded = pad_to_wide(s_dyn, s_obj, s_con, s_bounds, s_eh)
            if padded is None and _WMIN <= n_max < _WST and any(c.num_constraint for c in s_con):
                # more rows in one stage than padding states: rows that pin ONE variable (the reference's endpoint rows) become
                # variable bounds, only the rest needs auxiliary states; their multipliers are recovered after the solve
                pb = pins_to_bounds(s_con, s_bounds, s_eh)
                if pb is not None:
                    padded = pad_to_wide(s_dyn, s_obj, pb[0], pb[1], s_eh)
                    if padded is not None:
                        _, _, _, _, zm, mm, ms = padded
                        nd_ = sum(d.num_next_state for d in s_dyn)
                        n_rows_pad = sum(d.num_next_state for d in padded[0])
                        full_m, full_s = np.zeros(nd_ + len(pb[2]) + len(pb[3]), dtype=np.int64), np.ones(nd_ + len(pb[2]) + len(pb[3]))
                        full_m[:nd_], full_s[:nd_] = mm[:nd_], ms[:nd_]
                        for k, pos in enumerate(pb[3]):
                            full_m[nd_ + pos], full_s[nd_ + pos] = mm[nd_ + k], ms[nd_ + k]
                        nvar_t = [o.num_state + o.num_action for o in s_obj]
                        voff = np.concatenate([[0], np.cumsum(nvar_t)])
                        self._pins = []
                        for k, (pos, t, col, a) in enumerate(pb[2]):
                            full_m[nd_ + pos] = n_rows_pad + k          # (unpad_batch appends the recovered multipliers)
                            self._pins.append((int(zm[voff[t] + col]), float(a)))
                        padded = padded[:5] + (full_m, full_s)
            if padded is not None:
                s_dyn, s_obj, s_con, s_bounds, zmap, mumap, musign = padded
                if self._pad is not None:
                    za, ma, sa = self._pad
                    zmap, mumap, musign = zmap[za], mumap[ma], musign[ma] * sa
                self._pad = (zmap, mumap, musign)
                changed = wide_embedded = True
        # 17 .. 63 states that the embedding cannot take: the problem gets evaluator callbacks (a tile-family plugin of its own
        # size) but no solver -- say so here, not as "plugin has no KKT kernels" at the first solve
        if _WMIN <= n_max < _WST and not wide_embedded:
            why = ("GeneralConstraint rows that are not sums of one-knot terms" if gen is not None else
                   "the per-stage SR1 mode (no second derivatives to embed)" if not s_eh else
                   "more than four actions, varying dimensions, user-Jacobian dynamics, or more stage-constraint rows in one stage than "
                   "there are padding states (rows affine in one variable are restated as bounds and do not count; rows of the last knot count "
                   "with the last stage's; parameters in them are not supported)")
            self.solve_unsupported = (f"problems with {_WMIN} .. {_WST - 1} states are solved through the 64-state embedding of the tile "
                                      f"kernels, which does not take {why}: the MOI callbacks of this Solver work, solve!/solve_batch do not")
        else:
            self.solve_unsupported = None
        if self.hessian_mode == "lbfgs":
            traced = s_eh                        # (user-Jacobian dynamics cannot be differentiated: their plugin carries SR1 blocks)
            if not traced:
                # an explicit request that cannot be honoured is an error like the other ones below; only "auto" falls back
                if ha == "lbfgs":
                    raise ValueError("Options(hessian_approximation='lbfgs'): dynamics with a user-provided Jacobian (src/dynamics.jl:59-101) "
                                     "have no traced expression to build the limited-memory border from; use 'sr1' or 'auto'")
                self.hessian_mode = "sr1"
            elif gen is not None or wide_embedded or max(d.num_state for d in s_dyn) >= 17:
                if ha == "lbfgs":
                    raise ValueError("Options(hessian_approximation='lbfgs'): the limited-memory mode runs on the lane-per-instance "
                                     "solver path (at most 16 states, no GeneralConstraint rows over several knots)")
                self.hessian_mode = "exact-from-trace"
        if self.hessian_mode == "exact-from-trace" and ha == "auto":
            _notice_default_mode()
        if gen is not None:
            self.general_rows_path = "border"
        elif self._mu_to_reference is not None:
            self.general_rows_path = "folded"
        if changed:
            self._solve_nlp = NLPData(s_dyn, s_obj, s_con, s_bounds, evaluate_hessian=s_eh, general_constraint=gen,
                                      parameters=parameters, name=name if gen is None and self._mu_to_reference is None else name + "_folded")

    @property
    def num_variables(self):
        return self.nlp.num_variables

    # ---- batched device entry points (torch tensors / raw device pointers)
    def close(self):
        """Give the device memory of this solver back (the problem handles and the batch state they own)."""
        if self._solve_nlp is not self.nlp:
            self._solve_nlp.close()
        self.nlp.close()

    def kkt_step_batch(self, x_ptr, B, ldx, mu_ptr, ldmu, delta_w, delta_c, dx_ptr, lddx, dmu_ptr, lddmu, stream=0,
                       params_ptr=0, ldp=0):
        """One regularised Newton-KKT step (include/dto.h: dto_kkt_step_batch). Returns inertia_ok."""
        b = self._solve_nlp._batch(x_ptr, B, ldx, stream, params_ptr, ldp)
        ok = C.c_int(1)
        capi.check(self._solve_nlp._lib.dto_kkt_step_batch(self._solve_nlp._h, C.byref(b), mu_ptr, ldmu, float(delta_w), float(delta_c),
                                                    dx_ptr, lddx, dmu_ptr, lddmu, C.byref(ok)))
        return bool(ok.value)

    def pad_batch(self, Z):
        """Host array [..., num_variables] in the problem's layout -> the solver's layout (identity unless the problem was embedded
        in the 64 states of the tile kernels: the padding states are zero)."""
        Z = np.asarray(Z, dtype=float)
        if self._pad is None:
            return Z
        out = np.zeros(Z.shape[:-1] + (self._solve_nlp.num_variables,))
        out[..., self._pad[0]] = Z
        return out

    def unpad_batch(self, Z, multipliers=False, solution=None):
        """The solver's layout -> the problem's.  multipliers=True: Z holds constraint multipliers (solver row order); when stage
        rows were restated as variable bounds (pins_to_bounds) their multipliers are recovered from stationarity, which needs the
        solver-layout `solution` the multipliers belong to."""
        Z = np.asarray(Z)
        if self._pad is None:
            return Z
        if not multipliers:
            return Z[..., self._pad[0]]
        if self._pins:
            if solution is None:
                raise ValueError("unpad_batch(multipliers=True): this problem's endpoint rows were restated as variable bounds; pass "
                                 "solution= (the solver-layout iterates the multipliers belong to) to recover their multipliers")
            Z = np.concatenate([Z, self._pin_multipliers(np.asarray(solution), Z)], axis=-1)
        # (a stage row that the embedding carries as an auxiliary dynamics row comes back with the opposite sign)
        return Z[..., self._pad[1]] * self._pad[2]

    def _pin_multipliers(self, Zs, Mu):
        """Multipliers of the stage rows pins_to_bounds turned into bounds: lam = -(grad f + J^T mu)_v / a, evaluated with the batched
        callbacks of the solver-layout problem (the rows that are left, auxiliary rows included, carry their multipliers in Mu)."""
        import torch
        n = self._solve_nlp
        Zs2, Mu2 = np.atleast_2d(Zs), np.atleast_2d(Mu)
        B = Zs2.shape[0]
        nj = int(n.num_jacobian)
        if getattr(self, "_pin_entries", None) is None:
            # (the raw index arrays of the C ABI: no Python list of (T - 1) x 64 x 129 pairs at long horizons)
            rows, cols = np.zeros(max(1, nj), dtype=np.int64), np.zeros(max(1, nj), dtype=np.int64)
            capi.check(n._lib.dto_jacobian_structure(n._h, rows.ctypes.data_as(capi.c_int64_p), cols.ctypes.data_as(capi.c_int64_p)))
            rows, cols = rows[:nj] - 1, cols[:nj] - 1
            order = np.argsort(cols, kind="stable")
            sc = cols[order]
            self._pin_entries = []
            for p, _ in self._pins:
                lo, hi = np.searchsorted(sc, p, "left"), np.searchsorted(sc, p, "right")
                ent = np.sort(order[lo:hi])
                self._pin_entries.append((ent, rows[ent]))
        z = torch.tensor(Zs2, device="cuda", dtype=torch.float64).contiguous()
        g = torch.empty_like(z)
        J = torch.empty((B, max(1, nj)), device="cuda", dtype=torch.float64)
        n.eval_objective_gradient_batch(z.data_ptr(), B, z.shape[1], g.data_ptr(), g.shape[1])
        n.eval_constraint_jacobian_batch(z.data_ptr(), B, z.shape[1], J.data_ptr(), J.shape[1])
        torch.cuda.synchronize()
        out = np.zeros((B, len(self._pins)))
        for k, ((p, a), (ent, rws)) in enumerate(zip(self._pins, self._pin_entries)):
            gk = g[:, p].cpu().numpy()
            if len(ent):
                gk = gk + np.sum(J[:, torch.as_tensor(ent, device="cuda")].cpu().numpy() * Mu2[:, rws], axis=1)
            out[:, k] = -gk / a
        return out.reshape(np.asarray(Mu).shape[:-1] + (len(self._pins),))

    def multipliers_to_reference(self, mu):
        """Multipliers of the batched entry points come back in the SOLVER's row order; when a stage-local GeneralConstraint
        was folded into stage constraints this maps a host array [..., num_constraint] to the reference order
        [dynamics; stage; general] (src/data.jl:64-75).  solve() applies it by itself."""
        mu = np.asarray(mu)
        if self._mu_to_reference is None:
            return mu
        out = np.zeros(mu.shape[:-1] + (self.nlp.num_constraint,))
        out[..., self._mu_to_reference] = mu[..., :len(self._mu_to_reference)]
        return out

    def solve_batch(self, x0_ptr, B, ldx, x_out_ptr, ldxo, mu_out_ptr=0, ldmuo=0, stream=0, check_every=10,
                    params_ptr=0, ldp=0):
        """Solve B instances resident on the device; returns (status[B], iterations[B]) numpy int32 arrays.
        params_ptr: optional DEVICE [B][ldp] per-instance parameter vectors (flattened w_1..w_T) replacing the shared ones.
        mu_out: solver row order, see multipliers_to_reference."""
        if self.solve_unsupported:
            raise ValueError(self.solve_unsupported)
        b = self._solve_nlp._batch(x0_ptr, B, ldx, stream, params_ptr, ldp)
        co = _c_options(self.options, check_every, lbfgs=self.hessian_mode == "lbfgs")
        self._B = B
        status = np.zeros(B, dtype=np.int32)
        iters = np.zeros(B, dtype=np.int32)
        capi.check(self._solve_nlp._lib.dto_solve_batch(self._solve_nlp._h, C.byref(co), C.byref(b), x_out_ptr, ldxo,
                                                 mu_out_ptr or None, ldmuo, status.ctypes.data_as(capi.c_int32_p),
                                                 iters.ctypes.data_as(capi.c_int32_p)))
        return status, iters

    def begin_batch(self, x0_ptr, B, ldx, stream=0, params_ptr=0, ldp=0):
        if self.solve_unsupported:
            raise ValueError(self.solve_unsupported)
        b = self._solve_nlp._batch(x0_ptr, B, ldx, stream, params_ptr, ldp)
        co = _c_options(self.options, lbfgs=self.hessian_mode == "lbfgs")
        capi.check(self._solve_nlp._lib.dto_solver_begin(self._solve_nlp._h, C.byref(co), C.byref(b)))
        self._B = B

    def begin_warm_batch(self, B, x0_ptr=0, ldx=0, stream=0, params_ptr=0, ldp=0, mu0=0.0):
        """dto_solver_begin_warm: re-solve from the device-resident state of the previous solve (receding-horizon MPC).
        x0_ptr = 0 keeps the final iterate; mu0 <= 0 keeps the barrier parameter."""
        if self.solve_unsupported:
            raise ValueError(self.solve_unsupported)
        b = self._solve_nlp._batch(x0_ptr, B, ldx or self._solve_nlp.num_variables, stream, params_ptr, ldp)
        co = _c_options(self.options, lbfgs=self.hessian_mode == "lbfgs")
        capi.check(self._solve_nlp._lib.dto_solver_begin_warm(self._solve_nlp._h, C.byref(co), C.byref(b), float(mu0)))
        self._B = B

    def shift_batch(self, knots: int = 1, stream=0):
        """dto_solver_shift: move the device-resident iterate `knots` knots forward (receding-horizon warm start)."""
        capi.check(self._solve_nlp._lib.dto_solver_shift(self._solve_nlp._h, int(knots), stream or None))

    def repack_batch(self, stream=0) -> int:
        """dto_solver_repack: close the gaps finished instances leave in the tiles; returns the number still running."""
        n = C.c_int(0)
        capi.check(self._solve_nlp._lib.dto_solver_repack(self._solve_nlp._h, C.byref(n), stream or None))
        return n.value

    def run_batch(self, x_out_ptr, ldxo, mu_out_ptr=0, ldmuo=0, stream=0):
        """dto_solver_run: iterate the begun batch to termination; returns (status[B], iterations[B])."""
        B = self._B
        status, iters = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        capi.check(self._solve_nlp._lib.dto_solver_run(self._solve_nlp._h, x_out_ptr, ldxo, mu_out_ptr or None, ldmuo,
                                                       status.ctypes.data_as(capi.c_int32_p), iters.ctypes.data_as(capi.c_int32_p),
                                                       stream or None))
        return status, iters

    # ---- the linear solver alone (include/dto.h: dto_kkt_assemble / dto_kkt_factor / dto_kkt_solve)
    def kkt_assemble(self, x_ptr, B, ldx, mu_ptr, ldmu, delta_w, delta_c, sigma_x_ptr=0, ldsx=0, sigma_c_ptr=0, ldsc=0, stream=0,
                     params_ptr=0, ldp=0):
        b = self._solve_nlp._batch(x_ptr, B, ldx, stream, params_ptr, ldp)
        sysd = capi.KktSystem()
        sysd.mu, sysd.ldmu = mu_ptr, ldmu
        sysd.sigma_x, sysd.ldsx = sigma_x_ptr or None, ldsx
        sysd.sigma_c, sysd.ldsc = sigma_c_ptr or None, ldsc
        sysd.delta_w, sysd.delta_c = float(delta_w), float(delta_c)
        capi.check(self._solve_nlp._lib.dto_kkt_assemble(self._solve_nlp._h, C.byref(b), C.byref(sysd)))
        self._B = B

    def kkt_factor(self, stream=0):
        """Returns (inertia_ok[B], num_negative[B])."""
        ok, neg = np.zeros(self._B, dtype=np.int32), np.zeros(self._B, dtype=np.int32)
        capi.check(self._solve_nlp._lib.dto_kkt_factor(self._solve_nlp._h, ok.ctypes.data_as(capi.c_int32_p),
                                                       neg.ctypes.data_as(capi.c_int32_p), stream or None))
        return ok, neg

    def kkt_solve(self, rhs_x_ptr, ldrx, rhs_c_ptr, ldrc, sol_x_ptr, ldsx, sol_c_ptr, ldsc, stream=0):
        capi.check(self._solve_nlp._lib.dto_kkt_solve(self._solve_nlp._h, rhs_x_ptr, ldrx, rhs_c_ptr, ldrc, sol_x_ptr, ldsx,
                                                      sol_c_ptr, ldsc, stream or None))

    def iterate_batch(self, n, stream=0):
        capi.check(self._solve_nlp._lib.dto_solver_iterate(self._solve_nlp._h, int(n), stream or None))

    def stats_batch(self):
        B = self._B
        st, it = np.zeros(B, np.int32), np.zeros(B, np.int32)
        arrs = [np.zeros(B) for _ in range(6)]
        capi.check(self._solve_nlp._lib.dto_solver_stats(self._solve_nlp._h, st.ctypes.data_as(capi.c_int32_p), it.ctypes.data_as(capi.c_int32_p),
                                                  *[capi.dptr(a) for a in arrs]))
        names = ["objective", "constr_viol", "dual_inf", "mu", "delta_w", "alpha"]
        return dict(status=st, iterations=it, **dict(zip(names, arrs)))

    KKT_OPS = dict(eval=3, conv=4, factor_solve=5, linesearch=6, ls_reduce=7, update=8, kkt_fwd=9, kkt_sep=10, kkt_bwd=11,
                   kkt_post=12, update_eval=15, kkt_refine=25)

    def set_partitions(self, partitions: int):
        """Chunks of the time-partitioned factorisation (0 = automatic, 1 = sequential)."""
        capi.check(self._solve_nlp._lib.dto_solver_set_partitions(self._solve_nlp._h, int(partitions)))

    ENGINES = dict(auto=0, soa=1, im=2)

    def set_engine(self, engine):
        """Engine of the batched solver entry points (include/dto.h: dto_solver_set_engine): 'auto', 'soa' (SoA tiles) or
        'im' (instance-major stage records + work lists); takes effect at the next begin_batch / solve_batch."""
        capi.check(self._solve_nlp._lib.dto_solver_set_engine(self._solve_nlp._h, self.ENGINES.get(engine, engine)))

    def engine(self) -> str:
        v = C.c_int(0)
        capi.check(self._solve_nlp._lib.dto_solver_engine(self._solve_nlp._h, C.byref(v)))
        return {1: "soa", 2: "im"}[v.value]

    def release_state(self):
        """Free the device state of the batched solver entry points (dto_solver_release)."""
        capi.check(self._solve_nlp._lib.dto_solver_release(self._solve_nlp._h))

    IM_OPS = dict(eval=103, conv=104, fwd=105, bwd=106, linesearch=107, ls_reduce=108, update=109)

    def partitions(self) -> int:
        v = C.c_int(0)
        capi.check(self._solve_nlp._lib.dto_solver_partitions(self._solve_nlp._h, C.byref(v)))
        return v.value

    def fused_update(self) -> bool:
        """True if iterate_batch runs UPDATE + EVAL as one pass for the begun batch (include/dto.h: dto_solver_fused_update)."""
        v = C.c_int(0)
        capi.check(self._solve_nlp._lib.dto_solver_fused_update(self._solve_nlp._h, C.byref(v)))
        return bool(v.value)

    def launch_op(self, name: str, stream=0):
        """Launch one kernel of the iteration (diagnostic / timing).  `name`: a key of KKT_OPS (SoA engine) or 'im_' + a key of
        IM_OPS (instance-major engine: on the work list its last pass built)."""
        op = self.IM_OPS[name[3:]] if name.startswith("im_") else self.KKT_OPS[name]
        capi.check(self._solve_nlp._lib.dto_solver_launch_op(self._solve_nlp._h, op, stream or None))

    def footprint(self):
        r, f, n, k = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
        capi.check(self._solve_nlp._lib.dto_solver_footprint(self._solve_nlp._h, C.byref(r), C.byref(f), C.byref(n), C.byref(k)))
        return dict(record_doubles=r.value, factor_doubles=f.value, num_slacks=n.value, factor_rounds=k.value)

    def trace(self, on: bool = True):
        """Switch the launch trace of the solver entry points on (clearing it) or off (include/dto.h: dto_solver_trace)."""
        capi.check(self._solve_nlp._lib.dto_solver_trace(self._solve_nlp._h, 1 if on else 0))

    def read_trace(self):
        """Launch by launch in issue order: (op name, iteration, start_ms, duration_ms) as numpy arrays (dto_solver_trace_read)."""
        n = C.c_int64(0)
        capi.check(self._solve_nlp._lib.dto_solver_trace_read(self._solve_nlp._h, None, None, None, None, 0, C.byref(n)))
        cnt = int(n.value)
        op, it = np.zeros(max(cnt, 1), np.int32), np.zeros(max(cnt, 1), np.int32)
        t0, dt = np.zeros(max(cnt, 1)), np.zeros(max(cnt, 1))
        capi.check(self._solve_nlp._lib.dto_solver_trace_read(self._solve_nlp._h, op.ctypes.data_as(capi.c_int32_p), it.ctypes.data_as(capi.c_int32_p),
                                                       capi.dptr(t0), capi.dptr(dt), cnt, C.byref(n)))
        names = {v: k for k, v in self.KKT_OPS.items()}
        names.update({16: "kkt_bwd_early", 17: "kkt_bwd_rest", 18: "kkt_bwd_gate", 25: "kkt_refine"})
        return dict(op=op[:cnt], name=[names.get(int(o), str(int(o))) for o in op[:cnt]], iteration=it[:cnt], start_ms=t0[:cnt], duration_ms=dt[:cnt])

    def hessian_mode_last(self) -> str:
        """What stood in for the Hessian of the Lagrangian in the solve / batch begun last (dto_solver_hessian_mode)."""
        v = C.c_int(-1)
        capi.check(self._solve_nlp._lib.dto_solver_hessian_mode(self._solve_nlp._h, C.byref(v)))
        return {-1: "none", 0: "exact", 1: "lbfgs", 2: "sr1"}[v.value]

    def scalar_batch(self, name: str):
        out = np.zeros(self._B)
        capi.check(self._solve_nlp._lib.dto_solver_scalar(self._solve_nlp._h, capi.SCALARS.index(name), capi.dptr(out)))
        return out

    PEEK = dict(z=0, multipliers=1, dz=2, dmultipliers=3, z_lower=5, z_upper=6, slack=7, slack_multipliers=8, dslack=9)

    def peek_batch(self, name: str):
        """One vector of the solver's device state as a numpy array [B, n] (include/dto.h: dto_solver_peek)."""
        import torch
        which = self.PEEK[name]
        n = self._solve_nlp.num_variables if which in (0, 2, 5, 6) else (
            self._solve_nlp.num_constraint if which in (1, 3) else self.footprint()["num_slacks"])
        out = torch.zeros((self._B, max(1, n)), device="cuda", dtype=torch.float64)
        capi.check(self._solve_nlp._lib.dto_solver_peek(self._solve_nlp._h, which, out.data_ptr(), max(1, n), None))
        torch.cuda.synchronize()
        return out.cpu().numpy()[:, :n]

    def end_batch(self, x_out_ptr, ldxo, mu_out_ptr=0, ldmuo=0, stream=0):
        capi.check(self._solve_nlp._lib.dto_solver_end(self._solve_nlp._h, x_out_ptr, ldxo, mu_out_ptr or None, ldmuo, stream or None))


def _with_exact_hessians(dynamics, objective, constraints):
    """Clones of the stage objects with Hessians, built from their traced expressions (object sharing preserved).
    None if some dynamics came with a user-provided Jacobian (src/dynamics.jl:59-101: no expression to differentiate twice
    is promised there)."""
    cache = {}

    def clone(o):
        if id(o) in cache:
            return cache[id(o)]
        if isinstance(o, Dynamics):
            if o.user_jacobian:
                return None
            n = o if o.evaluate_hessian else Dynamics(list(o.evaluate_expr), o.num_next_state, o.num_state, o.num_action,
                                                      num_parameter=o.num_parameter, evaluate_hessian=True)
        elif isinstance(o, Cost):
            n = o if o.evaluate_hessian else Cost(list(o.evaluate_expr), o.num_state, o.num_action,
                                                  num_parameter=o.num_parameter, evaluate_hessian=True)
        else:
            n = o if (o.num_constraint == 0 or o.evaluate_hessian) else Constraint(
                list(o.evaluate_expr), o.num_state, o.num_action, num_parameter=o.num_parameter,
                indices_inequality=o.indices_inequality, evaluate_hessian=True)
        cache[id(o)] = n
        return n

    out = [[clone(o) for o in lst] for lst in (dynamics, objective, constraints)]
    if any(o is None for lst in out for o in lst):
        return None
    return out


def pad_to_wide(dynamics, objective, constraints, bounds, evaluate_hessian):
    """Stage objects of a problem with 17 .. 63 states embedded in the 64 states the tile (MFMA) kernels are built for
    (csrc/dto_wide_kernels.hpp; the reference allows any dimensions, src/dynamics.jl:206-211): the padding states follow
    y_k - x_k = 0, cost nothing and are fixed at zero by equal bounds at every knot, so the kernels treat them as identity
    rows.

    Stage constraints (round 6; src/constraints.jl:21-64, e.g. the endpoint rows of examples/acrobot/acrobot.jl:114-118 or the
    obstacle row of examples/car/car.jl:53-60 on a model with more than 16 states): the tile kernels have dynamics rows and
    variable bounds, no stage rows -- so a row c_j(x_t, u_t) becomes an AUXILIARY STATE a_{t+1, j} of the embedding,
        y_{n + j} - c_j(x, u, w) = 0      as one more dynamics row of stage t,
        a_{t+1, j} = 0 (equal bounds: a fixed variable) for an equality row,   a_{t+1, j} <= 0 for an inequality row
    (Ipopt's own slack formulation, the slack being a bounded state here: primal-dual barrier on the tile path).  The rows of
    the LAST knot, c(x_T), ride on the last dynamics stage as functions of its next state, y_{n + q + j} - c_j(y).  The multiplier
    of row j is minus the multiplier of its dynamics row (L = ... + lam (a - c) against ... + nu c).  Needs n + the largest number
    of rows that meet in one stage <= 64; rows of the last knot must not use parameters (they are evaluated with the last
    stage's).

    Returns (dynamics, objective, constraints, bounds, zmap, mumap, musign) -- zmap / mumap: positions of the padded problem's
    variables / constraint rows that belong to the original problem, in the original order [dynamics rows; stage rows]
    (src/data.jl:64-75), musign: +1 / -1 per original row -- or None if the problem does not fit the tile path either way (more
    than four actions, varying dimensions, user Jacobians, too many rows)."""
    from .plugin import WIDE_MAX_ACTION, WIDE_MIN_STATE, WIDE_STATE
    from .symbolic import expr as E
    T = len(objective)
    n = dynamics[0].num_state
    if not (WIDE_MIN_STATE <= n < WIDE_STATE):
        return None
    nu = dynamics[0].num_action
    if not (1 <= nu <= WIDE_MAX_ACTION):
        return None
    if any(d.num_state != n or d.num_next_state != n or d.num_action != nu or d.user_jacobian for d in dynamics):
        return None
    N = WIDE_STATE
    Q = [c.num_constraint for c in constraints]
    if any(Q):
        if T < 2 or len(constraints) != T:
            return None
        for t, c in enumerate(constraints):
            if c.num_constraint and (c.num_state != n or c.num_action != (nu if t < T - 1 else 0)):
                return None
        if Q[T - 1] and constraints[T - 1].num_parameter > 0:
            return None
    term = constraints[T - 1] if (any(Q) and Q[T - 1] > 0) else None
    slots = [Q[t] + (Q[T - 1] if t == T - 2 else 0) for t in range(T - 1)] if any(Q) else [0]
    QS = max(slots)
    if n + QS > N:
        return None
    x, y = E.variables("x", N), E.variables("y", N)
    cache = {}

    def pad_cost(o):
        if id(o) not in cache:
            cache[id(o)] = Cost(list(o.evaluate_expr), N, o.num_action, num_parameter=o.num_parameter, evaluate_hessian=evaluate_hessian)
        return cache[id(o)]

    def pad_dyn(d, c, ct):
        key = (id(d), id(c) if (c is not None and c.num_constraint) else None, id(ct) if ct is not None else None)
        if key not in cache:
            rows = list(d.evaluate_expr)
            k, nw = n, d.num_parameter
            if key[1] is not None:
                rows += [y[k + j] - e for j, e in enumerate(c.evaluate_expr)]
                k += c.num_constraint
                nw = max(nw, c.num_parameter)
            if ct is not None:
                at_y = E.substitute(list(ct.evaluate_expr), {x[i]: y[i] for i in range(n)})
                rows += [y[k + j] - e for j, e in enumerate(at_y)]
                k += ct.num_constraint
            rows += [y[q] + 0.0 for q in range(k, n + QS)]             # aux slots no row of this stage feeds: a = 0
            rows += [y[q] - x[q] for q in range(n + QS, N)]
            cache[key] = Dynamics(rows, N, N, nu, num_parameter=nw, evaluate_hessian=evaluate_hessian)
        return cache[key]

    dyn2 = [pad_dyn(dynamics[t], constraints[t] if any(Q) else None, term if t == T - 2 else None) for t in range(T - 1)]
    obj2 = [pad_cost(c) for c in objective]
    inf = float("inf")
    bnd2 = []
    for t, b in enumerate(bounds):
        alo, ahi = np.zeros(N - n), np.zeros(N - n)
        if t >= 1 and any(Q):
            feed = [(constraints[t - 1], 0)] + ([(term, Q[t - 1])] if (t - 1 == T - 2 and term is not None) else [])
            for c, off in feed:
                for j1 in c.indices_inequality:
                    alo[off + j1 - 1] = -inf
        bnd2.append(Bound(N, len(b.action_lower), state_lower=np.concatenate([b.state_lower, alo]), state_upper=np.concatenate([b.state_upper, ahi]),
                          action_lower=b.action_lower, action_upper=b.action_upper))
    zmap, mumap, musign = [], [], []
    for t in range(T):
        base = t * (N + nu)
        zmap += list(range(base, base + n)) + (list(range(base + N, base + N + nu)) if t < T - 1 else [])
        if t < T - 1:
            mumap += list(range(t * N, t * N + n))
            musign += [1.0] * n
    for t in range(T):
        if not (any(Q) and Q[t]):
            continue
        first = (t * N + n) if t < T - 1 else ((T - 2) * N + n + Q[T - 2])
        mumap += list(range(first, first + Q[t]))
        musign += [-1.0] * Q[t]
    return (dyn2, obj2, [Constraint() for _ in range(T)], bnd2, np.asarray(zmap, dtype=np.int64), np.asarray(mumap, dtype=np.int64),
            np.asarray(musign))


def pins_to_bounds(constraints, bounds, evaluate_hessian):
    """Stage-constraint rows that are affine in ONE state or action -- the endpoint rows `x - x1`, `x - xT` of the reference's
    examples (examples/acrobot/acrobot.jl:114-118), a box written as rows -- restated as variable bounds (src/bounds.jl:1-24):
        a v + b  = 0   ->   v fixed at -b / a  (states only: a fixed action has no interior for the barrier),
        a v + b <= 0   ->   v <= -b / a  (a > 0)   or   v >= -b / a  (a < 0)  where that side has no bound yet.
    Used by the 64-state embedding when a problem has more stage rows than padding states (pad_to_wide: every remaining row
    needs an auxiliary state).  The multiplier of a restated row follows from stationarity in its variable,
    lam = -(grad f + J^T mu)_v / a  over the rows that are left (Solver._pin_multipliers).

    Returns (constraints, bounds, pins, keep) -- pins: [(position among the original stage rows, knot, position of the variable in
    the knot's [x; u], a)], keep: positions among the original stage rows of the rows that are left, in order -- or None when no row
    qualifies."""
    from .symbolic import diff as D, expr as E
    T = len(constraints)
    inf = float("inf")
    new_con, new_bnd, pins, keep = [], [], [], []
    base = 0
    for t, (c, b) in enumerate(zip(constraints, bounds)):
        q = c.num_constraint
        if q == 0:
            new_con.append(c); new_bnd.append(b); continue
        nx, nu = c.num_state, c.num_action
        slo, shi = np.array(b.state_lower, dtype=float), np.array(b.state_upper, dtype=float)
        alo, ahi = np.array(b.action_lower, dtype=float), np.array(b.action_upper, dtype=float)
        if len(slo) != nx or len(alo) != nu:
            new_con.append(c); new_bnd.append(b); keep += list(range(base, base + q)); base += q; continue
        jr, jc = c.jacobian_sparsity
        per_row = {}
        for k, (r1, c1) in enumerate(zip(jr, jc)):
            per_row.setdefault(r1 - 1, []).append((c1 - 1, c.jacobian_expr[k]))
        taken, left = set(), []
        for j in range(q):
            ent = per_row.get(j, [])
            e = c.evaluate_expr[j]
            ok = len(ent) == 1 and ent[0][1].op == E.CONST and ent[0][1].value != 0.0 and ent[0][0] not in taken \
                and not any(n.op == E.VAR and n.name == "w" for n in E.topo_order([e]))
            if ok:
                col, a = ent[0][0], float(ent[0][1].value)
                v = E.var("x", col) if col < nx else E.var("u", col - nx)
                at0 = E.substitute([e], {v: E.const(0.0)})[0]
                ok = at0.op == E.CONST
            if ok:
                val = -float(at0.value) / a
                lo, hi, i = (slo, shi, col) if col < nx else (alo, ahi, col - nx)
                if (j + 1) in c.indices_inequality:
                    if a > 0 and hi[i] == inf:
                        hi[i] = val
                    elif a < 0 and lo[i] == -inf:
                        lo[i] = val
                    else:
                        ok = False
                elif col < nx and lo[i] <= val <= hi[i]:
                    lo[i] = hi[i] = val
                else:
                    ok = False
            if ok:
                taken.add(col)
                pins.append((base + j, t, col, a))
            else:
                left.append(j)
        if len(left) == q:
            new_con.append(c); new_bnd.append(b)
        else:
            if left:
                ineq = [k + 1 for k, j in enumerate(left) if (j + 1) in c.indices_inequality]
                new_con.append(Constraint([c.evaluate_expr[j] for j in left], nx, nu, num_parameter=c.num_parameter, indices_inequality=ineq,
                                          evaluate_hessian=evaluate_hessian))
            else:
                new_con.append(Constraint())
            new_bnd.append(Bound(nx, nu, state_lower=slo, state_upper=shi, action_lower=alo, action_upper=ahi))
        keep += [base + j for j in left]
        base += q
    if not pins:
        return None
    return new_con, new_bnd, pins, keep


def fold_general_constraint(dynamics, objective, constraints, general, evaluate_hessian):
    """Stage constraints equivalent to `constraints` + `general` when every general row depends on one knot only.

    Returns (new_constraints, mu_map) with mu_map[i] = 0-based position in the reference multiplier vector of the
    solver-internal constraint row i, or None if some row couples several knots or reads parameters."""
    from .symbolic import expr as E
    T = len(objective)
    nxs = [c.num_state for c in objective]
    nus = [c.num_action for c in objective]
    zoff = np.concatenate([[0], np.cumsum([nxs[t] + nus[t] for t in range(T)])])
    nz = int(zoff[-1])
    if general.num_variables != nz or general.num_parameter != 0:
        return None
    stage_of = np.zeros(nz, dtype=int)
    for t in range(T):
        stage_of[zoff[t]:zoff[t + 1]] = t
    rows_of_stage = {t: [] for t in range(T)}
    for r, e in enumerate(general.evaluate_expr):
        stages = {int(stage_of[n.index]) for n in E.topo_order([e]) if n.op == E.VAR and n.name == "z"}
        if any(n.op == E.VAR and n.name == "w" for n in E.topo_order([e])) or len(stages) > 1:
            return None
        rows_of_stage[stages.pop() if stages else T - 1].append(r)
    z = E.variables("z", nz)
    new_constraints, stage_rows = [], []   # stage_rows[t] = list of ("s", j) / ("g", r) in internal row order
    for t in range(T):
        con, gr = constraints[t], rows_of_stage[t]
        if not gr:
            new_constraints.append(con)
            stage_rows.append([("s", j) for j in range(con.num_constraint)])
            continue
        x, u = E.variables("x", nxs[t]), E.variables("u", nus[t])
        mapping = {z[zoff[t] + i]: x[i] for i in range(nxs[t])}
        mapping.update({z[zoff[t] + nxs[t] + j]: u[j] for j in range(nus[t])})
        exprs = list(con.evaluate_expr) + E.substitute([general.evaluate_expr[r] for r in gr], mapping)
        ineq = list(con.indices_inequality) + [con.num_constraint + k + 1 for k, r in enumerate(gr)
                                               if (r + 1) in general.indices_inequality]
        new_constraints.append(Constraint(exprs, nxs[t], nus[t], num_parameter=con.num_parameter if con.num_constraint else
                                          objective[t].num_parameter, indices_inequality=ineq, evaluate_hessian=evaluate_hessian))
        stage_rows.append([("s", j) for j in range(con.num_constraint)] + [("g", r) for r in gr])
    n_dyn = sum(d.num_next_state for d in dynamics)
    n_stage = sum(c.num_constraint for c in constraints)
    stage_base = np.concatenate([[0], np.cumsum([c.num_constraint for c in constraints])])
    mu_map = list(range(n_dyn))
    for t in range(T):
        for kind, j in stage_rows[t]:
            mu_map.append(n_dyn + int(stage_base[t]) + j if kind == "s" else n_dyn + n_stage + j)
    return new_constraints, np.asarray(mu_map, dtype=np.int64)


def accumulate_general_constraint(dynamics, objective, constraints, bounds, general, evaluate_hessian, max_state=16):
    """GeneralConstraint rows that couple SEVERAL knots (src/general_constraint.jl:18-59) as ordinary stage structure, when every
    row is a sum of terms of one knot each, g_r(z) = sum_t e_{r,t}(x_t, u_t) + const -- which is what coupling rows look like in
    practice ("theta_15 + theta_35 <= total", "x_4[1] + x_8[1] = 0.9" are sums): one ACCUMULATOR STATE per coupling row,
        s_1 = 0 (equal bounds),   s_{t+1} = s_t + e_{r,t}(x_t, u_t)  as one more dynamics row of stage t,
        s_T + e_{r,T}(x_T) + const  (= | <=) 0   as one more stage-constraint row of the last knot;
    rows that touch one knot only (the reference's own use, test/solve.jl:273) join that knot's stage constraint as
    fold_general_constraint does.  The problem then has no general rows at all: it runs the lane-per-instance solver loop on the
    device at full speed -- no border, no host-driven filter loop --, with variable bounds, stage inequalities and the
    limited-memory mode beside the rows, none of which the bordered path (csrc/dto_solver.cpp: general_solve_batch) has.  The
    multiplier of a coupling row is the multiplier of its last-knot row.

    Returns (dynamics, objective, constraints, bounds, zmap, mumap, musign) in the convention of pad_to_wide, or None when a row is
    not additively separable over the knots (a product of variables of two knots, ...), reads parameters, or the state would
    exceed `max_state` (the lane-per-instance kernels)."""
    from .symbolic import expr as E
    from .symbolic import diff as D
    T = len(objective)
    ng = general.num_constraint
    nxs = [c.num_state for c in objective]
    nus = [c.num_action for c in objective]
    if len(set(nxs)) != 1 or any(d.user_jacobian for d in dynamics) or T < 2:
        return None
    n = nxs[0]
    zoff = np.concatenate([[0], np.cumsum([nxs[t] + nus[t] for t in range(T)])])
    nz = int(zoff[-1])
    if general.num_variables != nz or general.num_parameter != 0:
        return None
    stage_of = np.zeros(nz, dtype=int)
    for t in range(T):
        stage_of[zoff[t]:zoff[t + 1]] = t
    z = E.variables("z", nz)
    zero = E.const(0.0)
    # per general row: the knots it touches and its expression
    knots_of, used_of = [], []
    for r, e in enumerate(general.evaluate_expr):
        nodes = E.topo_order([e])
        if any(nd.op == E.VAR and nd.name == "w" for nd in nodes):
            return None
        used = sorted({nd.index for nd in nodes if nd.op == E.VAR and nd.name == "z"})
        used_of.append(used)
        knots_of.append(sorted({int(stage_of[i]) for i in used}))
    acc_rows = [r for r in range(ng) if len(knots_of[r]) > 1]           # coupling rows: one accumulator state each
    na = len(acc_rows)
    N = n + na
    if N > max_state or na == 0:
        return None
    x = E.variables("x", N)
    y = E.variables("y", N)

    def localise(exprs, t, drop=()):
        """expressions over z -> over the local x / u symbols of knot t; variables in `drop` are set to zero first"""
        local = {z[zoff[t] + i]: x[i] for i in range(n)}
        local.update({z[zoff[t] + n + j]: E.variables("u", nus[t])[j] for j in range(nus[t])})
        if drop:
            exprs = E.substitute(exprs, {z[i]: zero for i in drop})
        return E.substitute(exprs, local)

    terms = {r: [None] * T for r in acc_rows}     # terms[r][t]: e_{r,t} over the local symbols of knot t, or None
    consts = {}
    for r in acc_rows:
        e, used = general.evaluate_expr[r], used_of[r]
        # additively separable over the knots <=> no second derivative couples two knots
        rr, cc, _ = D.sparse_hessian(e, [z[i] for i in used])
        if any(stage_of[used[a]] != stage_of[used[b]] for a, b in zip(rr, cc)):
            return None
        e0 = E.substitute([e], {z[i]: zero for i in used})[0]
        if not e0.is_const:
            return None
        consts[r] = e0
        for t in knots_of[r]:
            et = localise([e], t, drop=[i for i in used if stage_of[i] != t])[0] - e0
            if not (et.is_const and float(et.value) == 0.0):
                terms[r][t] = et
    # extra stage rows per knot: single-knot general rows where they belong, the accumulator rows at the last knot
    extra = [[] for _ in range(T)]                # (general row, local expression)
    for r in range(ng):
        if r in terms:
            continue
        t = knots_of[r][0] if knots_of[r] else T - 1
        extra[t].append((r, localise([general.evaluate_expr[r]], t)[0]))
    for k, r in enumerate(acc_rows):
        row = x[n + k] + consts[r]
        extra[T - 1].append((r, row + terms[r][T - 1] if terms[r][T - 1] is not None else row))
    cache = {}

    def wide_cost(o):
        if id(o) not in cache:
            cache[id(o)] = Cost(list(o.evaluate_expr), N, o.num_action, num_parameter=o.num_parameter, evaluate_hessian=evaluate_hessian)
        return cache[id(o)]

    def wide_con(o, t):
        if not extra[t]:
            if o.num_constraint == 0:
                return o
            if id(o) not in cache:
                cache[id(o)] = Constraint(list(o.evaluate_expr), N, o.num_action, num_parameter=o.num_parameter,
                                          indices_inequality=o.indices_inequality, evaluate_hessian=evaluate_hessian)
            return cache[id(o)]
        rows = list(o.evaluate_expr) + [e for _, e in extra[t]]
        ineq = list(o.indices_inequality) + [o.num_constraint + k + 1 for k, (r, _) in enumerate(extra[t]) if (r + 1) in general.indices_inequality]
        return Constraint(rows, N, nus[t], num_parameter=o.num_parameter if o.num_constraint else objective[t].num_parameter,
                          indices_inequality=ineq, evaluate_hessian=evaluate_hessian)

    def wide_dyn(d, t):
        key = (id(d), tuple(id(terms[r][t]) if terms[r][t] is not None else 0 for r in acc_rows))
        if key not in cache:
            rows = list(d.evaluate_expr)
            for k, r in enumerate(acc_rows):
                acc = y[n + k] - x[n + k]
                rows.append(acc - terms[r][t] if terms[r][t] is not None else acc)
            cache[key] = Dynamics(rows, N, N, d.num_action, num_parameter=d.num_parameter, evaluate_hessian=evaluate_hessian)
        return cache[key]

    dyn2 = [wide_dyn(dynamics[t], t) for t in range(T - 1)]
    obj2 = [wide_cost(c) for c in objective]
    con2 = [wide_con(constraints[t], t) for t in range(T)]
    inf = float("inf")
    bnd2 = []
    for t, b in enumerate(bounds):
        alo = np.zeros(na) if t == 0 else np.full(na, -inf)
        ahi = np.zeros(na) if t == 0 else np.full(na, inf)
        bnd2.append(Bound(N, len(b.action_lower), state_lower=np.concatenate([b.state_lower, alo]), state_upper=np.concatenate([b.state_upper, ahi]),
                          action_lower=b.action_lower, action_upper=b.action_upper))
    zmap, mumap = [], []
    for t in range(T):
        base = t * N + sum(nus[:t])
        zmap += list(range(base, base + n)) + list(range(base + N, base + N + nus[t]))
        if t < T - 1:
            mumap += list(range(t * N, t * N + n))
    off = (T - 1) * N
    gen_pos = {}
    for t in range(T):
        q = constraints[t].num_constraint
        mumap += list(range(off, off + q))
        for k, (r, _) in enumerate(extra[t]):
            gen_pos[r] = off + q + k
        off += con2[t].num_constraint
    mumap += [gen_pos[r] for r in range(ng)]
    return dyn2, obj2, con2, bnd2, np.asarray(zmap, dtype=np.int64), np.asarray(mumap, dtype=np.int64), np.ones(len(mumap))


def solve(solver: Solver):
    """solve!(solver) -- src/solver.jl:45-47: run the GPU interior-point solve from the initial guess."""
    if solver.solve_unsupported:
        raise ValueError(solver.solve_unsupported)
    n = solver._solve_nlp
    x = np.zeros(n.num_variables)
    mu = np.zeros(max(1, n.num_constraint))
    status, iters = C.c_int32(0), C.c_int32(0)
    co = _c_options(solver.options, lbfgs=solver.hessian_mode == "lbfgs")
    capi.check(n._lib.dto_solve(n._h, C.byref(co), capi.dptr(np.ascontiguousarray(solver.pad_batch(solver._z0))), capi.dptr(x), capi.dptr(mu),
                                C.byref(status), C.byref(iters)))
    solver._solution = solver.unpad_batch(x)
    solver._duals = solver.multipliers_to_reference(solver.unpad_batch(mu[:n.num_constraint], multipliers=True, solution=x))
    solver.status, solver.iterations = int(status.value), int(iters.value)
    if solver.options.print_level >= 5:
        # the reference prints Ipopt's iteration log at this level (src/options.jl:23); here: one summary line
        names = {0: "running", 6: "cut off (max_cpu_time)", 1: "converged", 2: "maximum iterations reached", 3: "failed (non-finite iterate)"}
        try:
            f = solver.nlp.eval_objective(solver._solution)
        except Exception:
            f = float("nan")
        print(f"dto_amd: {names.get(solver.status, solver.status)} after {solver.iterations} iterations, objective {f:.10e}")
    return solver.status


def get_trajectory(solver: Solver):
    """get_trajectory(solver) -- src/solver.jl:41-43: (states[1..T], actions[1..T-1]) of the accepted final iterate."""
    if solver._solution is None:
        raise RuntimeError("solve(solver) has not been called")
    z = solver._solution
    idx = solver.nlp.indices
    xs = [z[np.array(i) - 1] for i in idx.states]
    us = [z[np.array(i) - 1] for i in idx.actions]
    return xs, us


def initialize_states(solver: Solver, states) -> None:
    """initialize_states!(solver, states) -- src/solver.jl:23-30."""
    idx = solver.nlp.indices.states
    for t, xt in enumerate(states):
        xt = np.asarray(xt, dtype=float)
        for i in range(len(xt)):
            solver._z0[idx[t][i] - 1] = xt[i]


def initialize_controls(solver: Solver, actions) -> None:
    """initialize_controls!(solver, actions) -- src/solver.jl:32-39."""
    idx = solver.nlp.indices.actions
    for t, ut in enumerate(actions):
        ut = np.asarray(ut, dtype=float)
        for j in range(len(ut)):
            solver._z0[idx[t][j] - 1] = ut[j]
