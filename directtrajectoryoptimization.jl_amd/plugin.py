"""Model plugin generation: traced stage objects -> HIP source -> gfx950 shared object.

This is the build-time half of what `eval(Symbolics.build_function(...))` does in the
reference constructors (src/dynamics.jl:26-27,34; src/costs.jl:22-26;
src/constraints.jl:30-31,39; src/general_constraint.jl:26-27,35): turning the symbolic
value / Jacobian-nonzero / Hessian-nonzero vectors into executable code.  Here the code is
a set of `__device__` functions that the hand-written stage kernels in
csrc/dto_eval_kernels.hpp (and the KKT kernels) inline; the result is compiled with
`hipcc --offload-arch=gfx950` into one plugin .so per distinct problem structure and cached
by content hash under _plugins/.

Stage classes: objects are classified by identity, exactly as the reference shares one
`Dynamics` object between stages (`[dt for t = 1:T-1]`, examples/acrobot/acrobot.jl:95).
A stage *kind* is the tuple (dynamics class, previous dynamics class, cost class,
constraint class) that meets at a knot; kernels dispatch on it at compile time.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
from typing import Dict, List, Sequence, Tuple

from .model import Constraint, Cost, Dynamics, GeneralConstraint
from .symbolic.codegen import emit_body, is_affine, trig_arguments
from .symbolic import expr as E

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
PLUGIN_DIR = os.path.join(_HERE, "_plugins")
GENERATOR_VERSION = "9"
WIDE_MIN_STATE = 17   # above this the lane-per-instance register kernels give way to the tile (MFMA) kernels
WIDE_STATE = 64       # the state dimension the tile kernels are built for
WIDE_MAX_ACTION = 4   # actions per knot on the tile path (LDS budget of one workgroup)


class Structure:
    """Classes, kinds and per-stage kind ids of one problem."""

    def __init__(self, dynamics: Sequence[Dynamics], objective: Sequence[Cost],
                 constraints: Sequence[Constraint], general: GeneralConstraint | None,
                 evaluate_hessian: bool):
        T = len(objective)
        if len(dynamics) != T - 1 or len(constraints) != T:
            raise ValueError("need T-1 dynamics, T costs and T constraints")
        self.T = T
        self.evaluate_hessian = bool(evaluate_hessian)
        self.dyn: List[Dynamics] = []
        self.cost: List[Cost] = []
        self.con: List[Constraint] = []
        self.general = general if (general is not None and general.num_constraint > 0) else None

        def cls(lst, obj):
            for i, o in enumerate(lst):
                if o is obj:
                    return i
            lst.append(obj)
            return len(lst) - 1

        self.kinds: List[Tuple[int, int, int, int]] = []
        self.stage_kind: List[int] = []
        prev = -1
        for t in range(T):
            d = cls(self.dyn, dynamics[t]) if t < T - 1 else -1
            c = cls(self.cost, objective[t])
            k = constraints[t]
            kc = cls(self.con, k) if k.num_constraint > 0 else -1
            kind = (d, prev, c, kc)
            if kind not in self.kinds:
                self.kinds.append(kind)
            self.stage_kind.append(self.kinds.index(kind))
            prev = d
        # consistency the reference leaves implicit (index vectors would mismatch otherwise)
        for t in range(T):
            d, p, c, kc = self.kinds[self.stage_kind[t]]
            nx = self.dyn[d].num_state if d >= 0 else self.dyn[p].num_next_state
            nu = self.dyn[d].num_action if d >= 0 else 0
            if p >= 0 and self.dyn[p].num_next_state != nx:
                raise ValueError(f"stage {t + 1}: next-state dim of previous dynamics != state dim")
            if (self.cost[c].num_state, self.cost[c].num_action) != (nx, nu):
                raise ValueError(f"stage {t + 1}: cost dims {self.cost[c].num_state, self.cost[c].num_action} != {(nx, nu)}")
            if kc >= 0 and (self.con[kc].num_state, self.con[kc].num_action) != (nx, nu):
                raise ValueError(f"stage {t + 1}: constraint dims != {(nx, nu)}")
        max_nx = max([d.num_state for d in self.dyn] + [c.num_state for c in self.cost])
        self.wide = max_nx >= WIDE_MIN_STATE
        self.wide_n = WIDE_STATE
        self.wide_nu = 1
        if self.wide:
            # the tile (MFMA) KKT kernels are built for exactly WIDE_STATE states; the evaluator callbacks of the same family
            # take any uniform dimension up to it: a problem with 17 .. 63 states gets its callbacks from a plugin of its own
            # size and its solves from the 64-state embedding (solver.py: pad_to_wide)
            n0 = self.dyn[0].num_state if self.dyn else max_nx
            nu0 = self.dyn[0].num_action if self.dyn else 1
            ok = (WIDE_MIN_STATE <= n0 <= WIDE_STATE
                  and all(d.num_state == n0 and d.num_next_state == n0 and d.num_action == nu0 for d in self.dyn)
                  and 1 <= nu0 <= WIDE_MAX_ACTION
                  and all(c.num_state == n0 for c in self.cost))
            if not ok:
                raise ValueError(f"stages with more than {WIDE_MIN_STATE - 1} states use the tile kernels, which are built for "
                                 f"one uniform state dimension up to {WIDE_STATE} and one to {WIDE_MAX_ACTION} actions (the same number at "
                                 f"every knot)")
            self.wide_n = n0
            self.wide_nu = nu0
        # (the tile KKT kernels have dynamics rows and variable bounds, no stage rows: problems with stage constraints are solved
        #  through the embedding of solver.py: pad_to_wide, which turns the rows into auxiliary states -- their own plugin, the
        #  one with the Constraint objects, carries the evaluator callbacks only)
        self.wide_solver = self.wide and self.wide_n == WIDE_STATE and not self.con and self.general is None
        if self.evaluate_hessian:
            # all objects must agree on the flag
            for o in list(self.dyn) + list(self.cost) + list(self.con):
                if not o.evaluate_hessian and not getattr(o, "user_jacobian", False):
                    raise ValueError("evaluate_hessian=True needs every object built with evaluate_hessian=True")

    # number of Hessian key slots owned by the rows of a stage of kind k (exact, for LDS sizing)
    def key_slots(self, kind: Tuple[int, int, int, int]) -> int:
        d, p, c, kc = kind
        slots = set()
        if self.evaluate_hessian:
            cc = self.cost[c]
            for r, q in zip(*cc.sparsity):
                slots.add((r - 1, q - 1))
            if d >= 0:
                dd = self.dyn[d]
                npd = dd.num_state + dd.num_action
                for r, q in zip(*dd.hessian_sparsity):
                    if r - 1 < npd:
                        slots.add((r - 1, q - 1))
            if p >= 0:
                pp = self.dyn[p]
                npp = pp.num_state + pp.num_action
                for r, q in zip(*pp.hessian_sparsity):
                    if r - 1 >= npp:
                        slots.add((r - 1 - npp, q - 1 - npp))
            if kc >= 0:
                kk = self.con[kc]
                for r, q in zip(*kk.hessian_sparsity):
                    slots.add((r - 1, q - 1))
        return len(slots)


def _int_array(name: str, vals: Sequence[int]) -> str:
    body = ", ".join(str(int(v)) for v in vals) if len(vals) else "0"
    return f"static const int {name}[] = {{{body}}};"


def _fn(name: str, params: str, body: str) -> str:
    return f"  static __device__ __forceinline__ void {name}({params}) {{\n{body}\n  }}\n"


def _tri(i: int, j: int) -> int:
    """index of (i, j), i >= j, in a packed lower triangle"""
    return i * (i + 1) // 2 + j


def _scatter_dyn(d: Dynamics, h: bool) -> str:
    """Straight-line scatter of a dynamics class's local nonzeros into the dense stage blocks the KKT
    kernels use (all indices are literals, so everything stays in registers):
      jac  -> F = d(d_t)/d[x_t;u_t] (NY x NP, row-major), E = d(d_t)/d x_{t+1} (NY x NY)
      hess -> W (pp block, packed lower), V (NP x NY cross block), YY (packed lower)
      jtlam: rp += F' lam ; etlam: rx += E' lam
    """
    npd, ny = d.num_state + d.num_action, d.num_next_state
    out = []
    body = []
    for k, (r, c) in enumerate(zip(*d.jacobian_sparsity)):
        r, c = r - 1, c - 1
        body.append(f"    F[{r * npd + c}] = jv[{k}];" if c < npd else f"    E[{r * ny + (c - npd)}] = jv[{k}];")
    out.append(_fn("scatter_jac", "const double* jv, double* F, double* E", "\n".join(body) or "    (void)jv;"))
    body = [f"    rp[{c - 1}] += jv[{k}] * lam[{r - 1}];" for k, (r, c) in enumerate(zip(*d.jacobian_sparsity)) if c - 1 < npd]
    out.append(_fn("jtlam", "const double* jv, const double* lam, double* rp", "\n".join(body) or "    (void)jv;"))
    body = [f"    rx[{c - 1 - npd}] += jv[{k}] * lam[{r - 1}];" for k, (r, c) in enumerate(zip(*d.jacobian_sparsity)) if c - 1 >= npd]
    out.append(_fn("etlam", "const double* jv, const double* lam, double* rx", "\n".join(body) or "    (void)jv;"))
    body = []
    if h:
        for k, (r, c) in enumerate(zip(*d.hessian_sparsity)):
            r, c = r - 1, c - 1
            if r < c:
                continue
            if r < npd:
                body.append(f"    W[{_tri(r, c)}] += hv[{k}];")
            elif c < npd:
                body.append(f"    V[{c * ny + (r - npd)}] += hv[{k}];")
            else:
                body.append(f"    YY[{_tri(r - npd, c - npd)}] += hv[{k}];")
    out.append(_fn("scatter_hess", "const double* hv, double* W, double* V, double* YY", "\n".join(body) or "    (void)hv;"))
    # lower-triangle packing used by the solver's stage records (structural nonzeros only)
    pack, scat = [], []
    j = 0
    if h:
        for k, (r, c) in enumerate(zip(*d.hessian_sparsity)):
            r, c = r - 1, c - 1
            if r < c:
                continue
            pack.append(f"    hl[{j}] = hv[{k}];")
            if r < npd:
                scat.append(f"    W[{_tri(r, c)}] += gam * hl[{j}];")
            elif c < npd:
                scat.append(f"    V[{c * ny + (r - npd)}] += gam * hl[{j}];")
            else:
                scat.append(f"    YY[{_tri(r - npd, c - npd)}] += gam * hl[{j}];")
            j += 1
    out.append(f"  static constexpr int NHL = {j};\n")
    out.append(_fn("pack_hess_lower", "const double* hv, double* hl", "\n".join(pack) or "    (void)hv;"))
    out.append(_fn("scatter_hess_lower", "const double* hl, double gam, double* W, double* V, double* YY",
                   "\n".join(scat) or "    (void)hl;"))
    return "".join(out)


def _scatter_cost(c: Cost, h: bool) -> str:
    body = []
    if h:
        for k, (r, q) in enumerate(zip(*c.sparsity)):
            if r >= q:
                body.append(f"    W[{_tri(r - 1, q - 1)}] += hv[{k}];")
    out = [_fn("scatter_hess", "const double* hv, double* W", "\n".join(body) or "    (void)hv;")]
    # lower-triangle packing of the SOLVER's objective Hessian (always generated)
    pack, scat = [], []
    j = 0
    for k, (r, q) in enumerate(zip(*c.solver_sparsity)):
        if r >= q:
            pack.append(f"    hl[{j}] = hv[{k}];")
            scat.append(f"    W[{_tri(r - 1, q - 1)}] += hl[{j}];")
            j += 1
    out.append(f"  static constexpr int NHL = {j};\n")
    out.append(_fn("pack_hess_lower", "const double* hv, double* hl", "\n".join(pack) or "    (void)hv;"))
    out.append(_fn("scatter_hess_lower", "const double* hl, double* W", "\n".join(scat) or "    (void)hl;"))
    return "".join(out)


def _scatter_con(c: Constraint, h: bool) -> str:
    npd = c.num_state + c.num_action
    out = []
    body = [f"    G[{(r - 1) * npd + (q - 1)}] = jv[{k}];" for k, (r, q) in enumerate(zip(*c.jacobian_sparsity))]
    out.append(_fn("scatter_jac", "const double* jv, double* G", "\n".join(body) or "    (void)jv;"))
    body = [f"    rp[{q - 1}] += jv[{k}] * lam[{r - 1}];" for k, (r, q) in enumerate(zip(*c.jacobian_sparsity))]
    out.append(_fn("jtlam", "const double* jv, const double* lam, double* rp", "\n".join(body) or "    (void)jv;"))
    body = []
    if h:
        for k, (r, q) in enumerate(zip(*c.hessian_sparsity)):
            if r >= q:
                body.append(f"    W[{_tri(r - 1, q - 1)}] += hv[{k}];")
    out.append(_fn("scatter_hess", "const double* hv, double* W", "\n".join(body) or "    (void)hv;"))
    pack, scat = [], []
    j = 0
    if h:
        for k, (r, q) in enumerate(zip(*c.hessian_sparsity)):
            if r >= q:
                pack.append(f"    hl[{j}] = hv[{k}];")
                scat.append(f"    W[{_tri(r - 1, q - 1)}] += gam * hl[{j}];")
                j += 1
    out.append(f"  static constexpr int NHL = {j};\n")
    out.append(_fn("pack_hess_lower", "const double* hv, double* hl", "\n".join(pack) or "    (void)hv;"))
    out.append(_fn("scatter_hess_lower", "const double* hl, double gam, double* W", "\n".join(scat) or "    (void)hl;"))
    return "".join(out)


def _dev_int_array(name: str, vals: Sequence[int]) -> str:
    body = ", ".join(str(int(v)) for v in vals) if len(vals) else "0"
    return f"__device__ const int {name}[] = {{{body}}};"


def _lookup(fn: str, table: str) -> str:
    return f"  static __device__ __forceinline__ int {fn}(int k) {{ return {table}[k]; }}\n"


def generate_wide_source(st: Structure, name: str) -> str:
    """Plugin for wide stages (csrc/dto_wide_kernels.hpp): the stage Jacobian is split into a constant dense table and
    the few state-dependent entries; the residual is the constant part times [x; u; y] plus a nonlinear remainder."""
    from .symbolic.codegen import _lit
    out: List[str] = []
    out.append(f"// generated by directtrajectoryoptimization.jl_amd/plugin.py (v{GENERATOR_VERSION}, wide) -- do not edit")
    out.append("#include <type_traits>")
    out.append('#include "dto_wide_kernels.hpp"')
    out.append("namespace {")
    wkinds: List[Tuple[int, int, int]] = []
    wk_of_kind = []
    for (d, p, c, kc) in st.kinds:
        if (d, c, kc) not in wkinds:
            wkinds.append((d, c, kc))
        wk_of_kind.append(wkinds.index((d, c, kc)))
    dev_tables: List[str] = []
    host_tables: List[str] = []
    classes: List[str] = []
    max_njv, max_nh, max_snh = 1, 1, 1
    for i, d in enumerate(st.dyn):
        nx, nu, ny = d.num_state, d.num_action, d.num_next_state
        ncol = nx + nu + ny
        x = E.variables("x", nx); u = E.variables("u", nu); y = E.variables("y", ny)
        wrt = list(x) + list(u) + list(y)
        fe = [0.0] * (ny * ncol)
        var_idx = []
        const_cols: Dict[int, List[int]] = {r: [] for r in range(ny)}
        for k, (r1, c1, e) in enumerate(zip(d.jacobian_sparsity[0], d.jacobian_sparsity[1], d.jacobian_expr)):
            r, c = r1 - 1, c1 - 1
            if e.is_const:
                fe[r * ncol + c] = float(e.value)
                const_cols[r].append(c)
            else:
                var_idx.append(k)
        # nonlinear remainder of row r: the residual with every constant-coefficient variable set to zero
        rem = []
        for r in range(ny):
            rem.append(E.substitute([d.evaluate_expr[r]], {wrt[c]: E.const(0.0) for c in const_cols[r]})[0])
        nl_rows = [r for r in range(ny) if not rem[r].is_zero()]
        va = {"x": "x", "u": "u", "y": "y", "w": "w", "lam": "lam"}
        sig = "const double* x, const double* u, const double* y, const double* w, double* out"
        sigh = "const double* x, const double* u, const double* y, const double* w, const double* lam, double* out"
        nh = d.num_hessian if st.evaluate_hessian else 0
        max_njv, max_nh = max(max_njv, len(var_idx)), max(max_nh, nh)
        cl = [f"template <> struct Model::Dyn<{i}> {{"]
        cl.append(f"  static constexpr int NX = {nx}, NU = {nu}, NY = {ny}, NW = {d.num_parameter}, NJ = {d.num_jacobian}, "
                  f"NH = {nh}, NJV = {len(var_idx)}, NNL = {len(nl_rows)};")
        cl.append(f"  static __device__ __forceinline__ const double* fe_const() {{ return dyn{i}_fe; }}")
        cl.append(f"  static __device__ __forceinline__ const double* jc_const() {{ return dyn{i}_jcv; }}")
        cl.append(_fn("eval_nl", sig, emit_body([rem[r] for r in nl_rows], "out", va) if nl_rows else "    (void)x;"))
        cl.append(_fn("jac_var", sig, emit_body([d.jacobian_expr[k] for k in var_idx], "out", va) if var_idx else "    (void)x;"))
        if nh:
            cl.append(_fn("hess", sigh, emit_body(d.hessian_expr, "out", va)))
        cl.append(_lookup("jv_k", f"dyn{i}_jvk"))
        cl.append(_lookup("nl_row", f"dyn{i}_nlr") + _lookup("jv_row", f"dyn{i}_jvr") + _lookup("jv_col", f"dyn{i}_jvc")
                  + _lookup("h_row", f"dyn{i}_hr0") + _lookup("h_col", f"dyn{i}_hc0"))
        cl.append("};")
        classes.append("\n".join(cl))
        dev_tables.append(f"__device__ const double dyn{i}_fe[] = {{" + ", ".join(_lit(v) for v in fe) + "};")
        jc = [float(e.value) if e.is_const else 0.0 for e in d.jacobian_expr]
        dev_tables.append(f"__device__ const double dyn{i}_jcv[] = {{" + ", ".join(_lit(v) for v in jc) + "};")
        dev_tables.append(_dev_int_array(f"dyn{i}_jvk", var_idx))
        dev_tables.append(_dev_int_array(f"dyn{i}_nlr", nl_rows))
        dev_tables.append(_dev_int_array(f"dyn{i}_jvr", [d.jacobian_sparsity[0][k] - 1 for k in var_idx]))
        dev_tables.append(_dev_int_array(f"dyn{i}_jvc", [d.jacobian_sparsity[1][k] - 1 for k in var_idx]))
        dev_tables.append(_dev_int_array(f"dyn{i}_hr0", [r - 1 for r in d.hessian_sparsity[0]] if nh else []))
        dev_tables.append(_dev_int_array(f"dyn{i}_hc0", [c - 1 for c in d.hessian_sparsity[1]] if nh else []))
        host_tables.append(_int_array(f"dyn{i}_jr", d.jacobian_sparsity[0]))
        host_tables.append(_int_array(f"dyn{i}_jc", d.jacobian_sparsity[1]))
        host_tables.append(_int_array(f"dyn{i}_hr", d.hessian_sparsity[0] if nh else []))
        host_tables.append(_int_array(f"dyn{i}_hc", d.hessian_sparsity[1] if nh else []))
    for i, c in enumerate(st.cost):
        va = {"x": "x", "u": "u", "w": "w"}
        sig = "const double* x, const double* u, const double* w, double* out"
        snh = len(c.solver_hessian_expr)
        max_snh = max(max_snh, snh)
        cl = [f"template <> struct Model::Cost<{i}> {{"]
        cl.append(f"  static constexpr int NX = {c.num_state}, NU = {c.num_action}, NW = {c.num_parameter}, "
                  f"NH = {c.num_hessian if st.evaluate_hessian else 0}, SNH = {snh};")
        cl.append(_fn("eval", sig, emit_body(c.evaluate_expr, "out", va)))
        cl.append(_fn("grad", sig, emit_body(c.gradient_expr, "out", va)))
        cl.append(_fn("shess", sig, emit_body(c.solver_hessian_expr, "out", va) if snh else "    (void)x;"))
        cl.append(_lookup("sh_row", f"cost{i}_sr0") + _lookup("sh_col", f"cost{i}_sc0"))
        cl.append("};")
        classes.append("\n".join(cl))
        dev_tables.append(_dev_int_array(f"cost{i}_sr0", [r - 1 for r in c.solver_sparsity[0]]))
        dev_tables.append(_dev_int_array(f"cost{i}_sc0", [q - 1 for q in c.solver_sparsity[1]]))
        host_tables.append(_int_array(f"cost{i}_hr", c.sparsity[0] if st.evaluate_hessian else []))
        host_tables.append(_int_array(f"cost{i}_hc", c.sparsity[1] if st.evaluate_hessian else []))
    # stage constraints (round 6): evaluator callbacks only (k_wide_eval) -- values, Jacobian nonzeros, Hessian of nu' c
    max_con = 1
    for i, c in enumerate(st.con):
        va = {"x": "x", "u": "u", "w": "w", "lam": "lam"}
        nh = c.num_hessian if st.evaluate_hessian else 0
        max_con = max(max_con, c.num_constraint, c.num_jacobian, nh)
        sig = "const double* x, const double* u, const double* w, double* out"
        cl = [f"template <> struct Model::Con<{i}> {{"]
        cl.append(f"  static constexpr int NX = {c.num_state}, NU = {c.num_action}, NW = {c.num_parameter}, "
                  f"NC = {c.num_constraint}, NJ = {c.num_jacobian}, NH = {nh};")
        cl.append(_fn("eval", sig, emit_body(c.evaluate_expr, "out", va)))
        cl.append(_fn("jac", sig, emit_body(c.jacobian_expr, "out", va) if c.num_jacobian else "    (void)x;"))
        if nh:
            sigh = "const double* x, const double* u, const double* w, const double* lam, double* out"
            cl.append(_fn("hess", sigh, emit_body(c.hessian_expr, "out", va)))
        cl.append("};")
        classes.append("\n".join(cl))
        host_tables.append(_int_array(f"con{i}_jr", c.jacobian_sparsity[0]))
        host_tables.append(_int_array(f"con{i}_jc", c.jacobian_sparsity[1]))
        host_tables.append(_int_array(f"con{i}_hr", c.hessian_sparsity[0] if nh else []))
        host_tables.append(_int_array(f"con{i}_hc", c.hessian_sparsity[1] if nh else []))
        host_tables.append(_int_array(f"con{i}_iq", sorted(c.indices_inequality)))
    # GeneralConstraint (round 6): evaluator callbacks only; the solver reaches such rows through solver.py's transformations
    g = st.general
    if g is not None:
        va = {"z": "z", "w": "w", "lam": "lam"}
        sigg = "const double* z, const double* w, double* out"
        cl = ["struct Model::General {", f"  static constexpr int NC = {g.num_constraint}, NJ = {g.num_jacobian};"]
        cl.append(_fn("eval", sigg, emit_body(g.evaluate_expr, "out", va)))
        cl.append(_fn("jac", sigg, emit_body(g.jacobian_expr, "out", va) if g.num_jacobian else "    (void)z;"))
        cl.append("};")
        classes.append("\n".join(cl))
        host_tables.append(_int_array("gen_jr", g.jacobian_sparsity[0]))
        host_tables.append(_int_array("gen_jc", g.jacobian_sparsity[1]))
        host_tables.append(_int_array("gen_hr", g.hessian_sparsity[0] if st.evaluate_hessian else []))
        host_tables.append(_int_array("gen_hc", g.hessian_sparsity[1] if st.evaluate_hessian else []))
        host_tables.append(_int_array("gen_iq", sorted(g.indices_inequality)))
    else:
        classes.append("struct Model::General { static constexpr int NC = 0, NJ = 0; };")
    out.extend(dev_tables)
    out.append(_dev_int_array("k_wk_of_kind", wk_of_kind))
    dev_extra: List[str] = []
    dev_extra_at = len(out)
    out.append("struct Model {")
    out.append(f"  static constexpr int WIDE_N = {st.wide_n}, WIDE_NU = {st.wide_nu}, N_KIND = {len(st.kinds)}, N_WKIND = {len(wkinds)};")
    max_key = max([1] + [st.key_slots(k) for k in st.kinds])
    if st.wide_solver:
        # csrc/dto_wide_kernels.hpp: StepLds -- the stage data of one instance must fit the LDS of one workgroup
        n, nu = st.wide_n, st.wide_nu
        lds = 8 * (4 * n * (n + 1) + (n // 16) * 16 * 17 + (15 + 3 * nu) * n + 8 + max_nh + max_snh + max_njv
                   + ((3 * nu + nu * nu + 7) & ~7) + 4 + 16 + 11 * n + 8 * nu)
        if lds > 160 * 1024:
            raise ValueError(f"wide stages: {lds} bytes of stage data per instance exceed the 160 KB of LDS of one workgroup "
                             f"({nu} actions, {max_nh} dynamics-Hessian / {max_snh} cost-Hessian / {max_njv} variable Jacobian entries)")
    out.append(f"  static constexpr int MAX_NH = {max_nh}, MAX_SNH = {max_snh}, MAX_NJV = {max_njv}, EVALUATE_HESSIAN = {1 if st.evaluate_hessian else 0}, MAX_KEY = {max_key};")
    out.append(f"  static constexpr int N_DYN = {len(st.dyn)};")
    out.append(f"  static constexpr bool HAS_GENERAL = {'true' if st.general is not None else 'false'};")
    out.append("  struct General;")
    out.append(f"  static constexpr int N_CON = {len(st.con)}, MAX_CON = {max_con};   // stage-constraint classes; largest of their value / Jacobian / Hessian counts")
    out.append("  template <int K> struct WKind;")
    out.append("  template <int C> struct Dyn;")
    out.append("  template <int C> struct Cost;")
    out.append("  template <int C> struct Con;")
    out.append("  static __device__ __forceinline__ int wk_of_kind(int k) { return k_wk_of_kind[k]; }")
    out.append("  template <class F> static __device__ __forceinline__ void dispatch_wk(int wk, F&& f) {")
    for i in range(len(wkinds)):
        out.append(f"    if (wk == {i}) {{ f(std::integral_constant<int, {i}>{{}}); return; }}")
    out.append("  }")
    out.append("};")
    for i, (d, c, kc) in enumerate(wkinds):
        # states that the actions couple to through second derivatives: rows of A_xu (x-u entries of the cost and of lam' d'')
        # and of V_u (u-y entries of lam' d'').  The rank-one terms of the action elimination touch only these rows / columns of
        # the stage matrices (csrc/dto_wide_kernels.hpp, phase 5): for the acrobot embedding 1 + 1 of 64 + 64.  (Several actions:
        # the union over the actions -- the elimination inside the action block mixes their rows.)
        au_s, vu_s = set(), set()
        if d >= 0:
            dd = st.dyn[d]
            nx, nu = dd.num_state, dd.num_action
            if st.evaluate_hessian:
                for r1, c1 in zip(dd.hessian_sparsity[0], dd.hessian_sparsity[1]):
                    r, cc = r1 - 1, c1 - 1
                    for a_, b_ in ((r, cc), (cc, r)):
                        if a_ < nx and nx <= b_ < nx + nu:
                            au_s.add(a_)
                        if nx <= a_ < nx + nu and b_ >= nx + nu:
                            vu_s.add(b_ - nx - nu)
            cc_ = st.cost[c]
            for r1, c1 in zip(cc_.solver_sparsity[0], cc_.solver_sparsity[1]):
                r, q = r1 - 1, c1 - 1
                for a_, b_ in ((r, q), (q, r)):
                    if a_ < nx and b_ >= nx:
                        au_s.add(a_)
        au_l, vu_l = sorted(au_s), sorted(vu_s)
        dev_extra.append(_dev_int_array(f"wk{i}_aus", au_l))
        dev_extra.append(_dev_int_array(f"wk{i}_vus", vu_l))
        out.append(f"template <> struct Model::WKind<{i}> {{ static constexpr int DYN = {d}, COST = {c}, CON = {kc}, AU_N = {len(au_l)}, VU_N = {len(vu_l)};")
        out.append(_lookup("au_s", f"wk{i}_aus") + _lookup("vu_s", f"wk{i}_vus"))
        out.append("};")
    out[dev_extra_at:dev_extra_at] = dev_extra
    out.extend(classes)
    out.extend(host_tables)
    rows = []
    for i, d in enumerate(st.dyn):
        rows.append(f"  {{{d.num_next_state}, {d.num_state}, {d.num_action}, {d.num_parameter}, {d.num_jacobian}, "
                    f"{d.num_hessian if st.evaluate_hessian else 0}, dyn{i}_jr, dyn{i}_jc, dyn{i}_hr, dyn{i}_hc}}")
    out.append("static const dto_dyn_class k_dyn[] = {\n" + ",\n".join(rows) + "\n};")
    rows = [f"  {{{c.num_state}, {c.num_action}, {c.num_parameter}, {c.num_hessian if st.evaluate_hessian else 0}, cost{i}_hr, cost{i}_hc}}"
            for i, c in enumerate(st.cost)]
    out.append("static const dto_cost_class k_cost[] = {\n" + ",\n".join(rows) + "\n};")
    rows = []
    for i, c in enumerate(st.con):
        nh = c.num_hessian if st.evaluate_hessian else 0
        rows.append(f"  {{{c.num_state}, {c.num_action}, {c.num_parameter}, {c.num_constraint}, {c.num_jacobian}, {nh}, "
                    f"con{i}_jr, con{i}_jc, con{i}_hr, con{i}_hc, {len(c.indices_inequality)}, con{i}_iq}}")
    out.append("static const dto_con_class k_con[] = {\n" + (",\n".join(rows) if rows else "  {0}") + "\n};")
    rows = [f"  {{{d}, {p}, {c}, {kc}}}" for (d, p, c, kc) in st.kinds]
    out.append("static const dto_kind k_kinds[] = {\n" + ",\n".join(rows) + "\n};")
    if g is not None:
        nhg = g.num_hessian if st.evaluate_hessian else 0
        out.append(f"static const dto_general_class k_general = {{{g.num_variables}, {g.num_parameter}, {g.num_constraint}, "
                   f"{g.num_jacobian}, {nhg}, gen_jr, gen_jc, gen_hr, gen_hc, {len(g.indices_inequality)}, gen_iq}};")
    out.append("static int launch(int op, const dto_eval_args* a, void* s) { return dto::wide::launch_wide_eval<Model>(op, a, s); }")
    if st.wide_solver:
        out.append("static int launch_wide(int op, const dto_wide_args* a, void* s) { return dto::wide::launch_wide<Model>(op, a, s); }")
    out.append("static const dto_model_vtable k_vtable = {")
    out.append(f'  DTO_PLUGIN_ABI, "{name}", {len(st.dyn)}, {len(st.cost)}, {len(st.con)}, {len(st.kinds)},')
    out.append(f"  k_dyn, k_cost, k_con, k_kinds, {'&k_general' if g is not None else 'nullptr'}, {1 if st.evaluate_hessian else 0},")
    out.append(f"  {max_key}, launch, nullptr, nullptr, "
               + ("launch_wide, dto::wide::wide_info<Model>" if st.wide_solver else "nullptr, nullptr") + ", nullptr, nullptr")
    out.append("};")
    out.append("}  // namespace")
    out.append('extern "C" const dto_model_vtable* dto_model_get(void) { return &k_vtable; }')
    return "\n".join(out) + "\n"


def with_im_engine() -> bool:
    """The instance-major engine (csrc/dto_im_kernels.hpp) is an opt-in experiment measured 1.7x slower than the SoA tiles
    (DESIGN.md section 5): it is compiled into a plugin only when DTO_PLUGIN_IM=1 is set while the plugin is generated (its own
    cache key).  Default plugins carry no instance-major kernels: dto_solver_set_engine(h, 2) then reports DTO_ERR_UNSUPPORTED."""
    return os.environ.get("DTO_PLUGIN_IM", "0") == "1"


def generate_source(st: Structure, name: str) -> str:
    if st.wide:
        return generate_wide_source(st, name)
    h = st.evaluate_hessian
    out: List[str] = []
    out.append(f"// generated by directtrajectoryoptimization.jl_amd/plugin.py (v{GENERATOR_VERSION}) -- do not edit")
    out.append('#include "dto_eval_kernels.hpp"')
    out.append('#include "dto_kkt_kernels.hpp"')
    with_im = with_im_engine()
    if with_im:
        out.append('#include "dto_im_kernels.hpp"')
    out.append("namespace {")
    mx = lambda xs: max([0] + [int(x) for x in xs])
    max_nx = mx([d.num_state for d in st.dyn] + [d.num_next_state for d in st.dyn] + [c.num_state for c in st.cost])
    max_nu = mx([c.num_action for c in st.cost])
    max_nxu = mx([c.num_state + c.num_action for c in st.cost])
    consts = dict(
        N_KIND=len(st.kinds), N_DYN=len(st.dyn), N_COST=len(st.cost), N_CON=len(st.con),
        MAX_NX=max(1, max_nx), MAX_NU=max_nu, MAX_NXU=max(1, max_nxu),
        MAX_NY=max(1, mx(d.num_next_state for d in st.dyn)),
        MAX_NW=mx([d.num_parameter for d in st.dyn] + [c.num_parameter for c in st.cost] + [c.num_parameter for c in st.con]),
        MAX_DYN_NC=max(1, mx(d.num_next_state for d in st.dyn)),
        MAX_DYN_NJ=max(1, mx(d.num_jacobian for d in st.dyn)),
        MAX_DYN_NH=max(1, mx(d.num_hessian for d in st.dyn) if h else 0),
        MAX_CON_NC=max(1, mx(c.num_constraint for c in st.con)),
        MAX_CON_NJ=max(1, mx(c.num_jacobian for c in st.con)),
        MAX_CON_NH=max(1, mx(c.num_hessian for c in st.con) if h else 0),
        MAX_COST_NH=max(1, mx(c.num_hessian for c in st.cost) if h else 0),
        MAX_KEY=max(1, mx(st.key_slots(k) for k in st.kinds)),
        EVALUATE_HESSIAN=1 if h else 0,
    )
    out.append("struct Model {")
    for k, v in consts.items():
        out.append(f"  static constexpr int {k} = {v};")
    out.append(f"  static constexpr bool HAS_GENERAL = {'true' if st.general is not None else 'false'};")
    out.append("  template <int K> struct Kind;")
    out.append("  template <int C> struct Dyn;")
    out.append("  template <int C> struct Cost;")
    out.append("  template <int C> struct Con;")
    out.append("  struct General;")
    out.append("};")
    for i, (d, p, c, kc) in enumerate(st.kinds):
        out.append(f"template <> struct Model::Kind<{i}> {{ static constexpr int DYN = {d}, PREV = {p}, COST = {c}, CON = {kc}; }};")

    tables: List[str] = []
    # ---- dynamics classes
    for i, d in enumerate(st.dyn):
        va = {"x": "x", "u": "u", "y": "y", "w": "w", "lam": "lam"}
        nh = d.num_hessian if h else 0
        out.append(f"template <> struct Model::Dyn<{i}> {{")
        out.append(f"  static constexpr int NX = {d.num_state}, NU = {d.num_action}, NY = {d.num_next_state}, "
                   f"NW = {d.num_parameter}, NJ = {d.num_jacobian}, NH = {nh};")
        sig = "const double* x, const double* u, const double* y, const double* w, double* out"
        out.append(_fn("eval", sig, emit_body(d.evaluate_expr, "out", va)))
        out.append(_fn("jac", sig, emit_body(d.jacobian_expr, "out", va)))
        # Line evaluation (k_linesearch evaluates the residual at 8 points x + alpha_k dx): when every sin / cos argument is an
        # affine function of (x, u, y) the kernel gets the arguments (trig_args) and a residual that takes the sin / cos values
        # from the caller (eval_trig), and produces them for all trial points from two sincos per argument
        targs = trig_arguments(d.evaluate_expr)
        memo: Dict[int, bool] = {}
        if targs and len(targs) <= 8 and all(is_affine(a_, memo) for a_ in targs):
            out.append(f"  static constexpr int NTRIG = {len(targs)};")
            out.append(_fn("trig_args", sig, emit_body(targs, "out", va)))
            sigt = "const double* x, const double* u, const double* y, const double* w, const double* sn, const double* cs, double* out"
            out.append(_fn("eval_trig", sigt, emit_body(d.evaluate_expr, "out", va, trig_override={a_.id: j for j, a_ in enumerate(targs)})))
        else:
            out.append("  static constexpr int NTRIG = 0;")
        # residual and Jacobian from one body (k_stage_eval needs both at every stage)
        sigej = "const double* x, const double* u, const double* y, const double* w, double* eout, double* jout"
        out.append(_fn("eval_jac", sigej, emit_body(list(d.evaluate_expr) + list(d.jacobian_expr),
                                                    [("eout", len(d.evaluate_expr)), ("jout", len(d.jacobian_expr))], va)))
        if nh:
            sigh = "const double* x, const double* u, const double* y, const double* w, const double* lam, double* out"
            out.append(_fn("hess", sigh, emit_body(d.hessian_expr, "out", va)))
            # Jacobian and Hessian from ONE body: the solver's sweeps need both at every stage, and they share most of their
            # work (the sin/cos pairs, the inverse mass matrix, ...)
            sigjh = "const double* x, const double* u, const double* y, const double* w, const double* lam, double* jout, double* hout"
            out.append(_fn("jac_hess", sigjh, emit_body(list(d.jacobian_expr) + list(d.hessian_expr),
                                                        [("jout", len(d.jacobian_expr)), ("hout", len(d.hessian_expr))], va)))
            # Hessian nonzero i belongs to the rows of its own stage (row in [x; u]) or of the next one (row in y): the
            # same rule as csrc/dto_layout.hpp:build_hmaps, as a literal table so that k_hess's deposit loops fold it
            own = ", ".join("1" if r <= d.num_state + d.num_action else "0" for r in d.hessian_sparsity[0])
            out.append(f"  static constexpr __host__ __device__ bool hess_row_own(int i) {{ constexpr bool own[] = {{{own}}}; return own[i]; }}")
        else:
            out.append("  static constexpr __host__ __device__ bool hess_row_own(int) { return true; }")
        out.append(_scatter_dyn(d, h))
        out.append("};")
        tables.append(_int_array(f"dyn{i}_jr", d.jacobian_sparsity[0]))
        tables.append(_int_array(f"dyn{i}_jc", d.jacobian_sparsity[1]))
        tables.append(_int_array(f"dyn{i}_hr", d.hessian_sparsity[0] if h else []))
        tables.append(_int_array(f"dyn{i}_hc", d.hessian_sparsity[1] if h else []))
    # ---- cost classes
    for i, c in enumerate(st.cost):
        va = {"x": "x", "u": "u", "w": "w"}
        nh = c.num_hessian if h else 0
        out.append(f"template <> struct Model::Cost<{i}> {{")
        out.append(f"  static constexpr int NX = {c.num_state}, NU = {c.num_action}, NW = {c.num_parameter}, NH = {nh};")
        sig = "const double* x, const double* u, const double* w, double* out"
        out.append(_fn("eval", sig, emit_body(c.evaluate_expr, "out", va)))
        out.append(_fn("grad", sig, emit_body(c.gradient_expr, "out", va)))
        if nh:
            out.append(_fn("hess", sig, emit_body(c.hessian_expr, "out", va)))
        # objective Hessian for the solver (always available: Gauss-Newton mode when evaluate_hessian=false)
        out.append(f"  static constexpr int SNH = {len(c.solver_hessian_expr)};\n")
        out.append(_fn("shess", sig, emit_body(c.solver_hessian_expr, "out", va) if c.solver_hessian_expr else "    (void)x;"))
        out.append(_scatter_cost(c, h))
        out.append("};")
        tables.append(_int_array(f"cost{i}_hr", c.sparsity[0] if h else []))
        tables.append(_int_array(f"cost{i}_hc", c.sparsity[1] if h else []))
    # ---- constraint classes
    for i, c in enumerate(st.con):
        va = {"x": "x", "u": "u", "w": "w", "lam": "lam"}
        nh = c.num_hessian if h else 0
        out.append(f"template <> struct Model::Con<{i}> {{")
        out.append(f"  static constexpr int NX = {c.num_state}, NU = {c.num_action}, NW = {c.num_parameter}, "
                   f"NC = {c.num_constraint}, NJ = {c.num_jacobian}, NH = {nh};")
        sig = "const double* x, const double* u, const double* w, double* out"
        out.append(_fn("eval", sig, emit_body(c.evaluate_expr, "out", va)))
        out.append(_fn("jac", sig, emit_body(c.jacobian_expr, "out", va)))
        if nh:
            sigh = "const double* x, const double* u, const double* w, const double* lam, double* out"
            out.append(_fn("hess", sigh, emit_body(c.hessian_expr, "out", va)))
        ineq = sorted(c.indices_inequality)
        flags = ["true" if (r + 1) in ineq else "false" for r in range(c.num_constraint)]
        cond = " || ".join(f"j == {r - 1}" for r in ineq) or "false"
        out.append(f"  static constexpr __host__ __device__ bool ineq(int j) {{ return {cond}; }}")
        slk = " : ".join(f"j == {r - 1} ? {k}" for k, r in enumerate(ineq))
        out.append(f"  static constexpr __host__ __device__ int slack(int j) {{ return {slk + ' : -1' if ineq else '-1'}; }}")
        hr = c.hessian_sparsity[0] if h else []
        hc = c.hessian_sparsity[1] if h else []
        out.append(f"  static constexpr int NI = {len(ineq)};")
        out.append(_scatter_con(c, h))
        out.append("};")
        tables.append(_int_array(f"con{i}_jr", c.jacobian_sparsity[0]))
        tables.append(_int_array(f"con{i}_jc", c.jacobian_sparsity[1]))
        tables.append(_int_array(f"con{i}_hr", hr))
        tables.append(_int_array(f"con{i}_hc", hc))
        tables.append(_int_array(f"con{i}_iq", ineq))
    # ---- general constraint
    g = st.general
    if g is not None:
        va = {"z": "z", "w": "w", "lam": "lam"}
        out.append("struct Model::General {")
        out.append(f"  static constexpr int NC = {g.num_constraint}, NJ = {g.num_jacobian};")
        sig = "const double* z, const double* w, double* out"
        out.append(_fn("eval", sig, emit_body(g.evaluate_expr, "out", va)))
        out.append(_fn("jac", sig, emit_body(g.jacobian_expr, "out", va)))
        out.append("};")
        tables.append(_int_array("gen_jr", g.jacobian_sparsity[0]))
        tables.append(_int_array("gen_jc", g.jacobian_sparsity[1]))
        tables.append(_int_array("gen_hr", g.hessian_sparsity[0] if h else []))
        tables.append(_int_array("gen_hc", g.hessian_sparsity[1] if h else []))
        tables.append(_int_array("gen_iq", sorted(g.indices_inequality)))
    else:
        out.append("struct Model::General { static constexpr int NC = 0, NJ = 0; };")
    out.extend(tables)
    # ---- class tables
    rows = []
    for i, d in enumerate(st.dyn):
        nh = d.num_hessian if h else 0
        rows.append(f"  {{{d.num_next_state}, {d.num_state}, {d.num_action}, {d.num_parameter}, {d.num_jacobian}, {nh}, "
                    f"dyn{i}_jr, dyn{i}_jc, dyn{i}_hr, dyn{i}_hc}}")
    out.append("static const dto_dyn_class k_dyn[] = {\n" + (",\n".join(rows) if rows else "  {0}") + "\n};")
    rows = []
    for i, c in enumerate(st.cost):
        nh = c.num_hessian if h else 0
        rows.append(f"  {{{c.num_state}, {c.num_action}, {c.num_parameter}, {nh}, cost{i}_hr, cost{i}_hc}}")
    out.append("static const dto_cost_class k_cost[] = {\n" + ",\n".join(rows) + "\n};")
    rows = []
    for i, c in enumerate(st.con):
        nh = c.num_hessian if h else 0
        rows.append(f"  {{{c.num_state}, {c.num_action}, {c.num_parameter}, {c.num_constraint}, {c.num_jacobian}, {nh}, "
                    f"con{i}_jr, con{i}_jc, con{i}_hr, con{i}_hc, {len(c.indices_inequality)}, con{i}_iq}}")
    out.append("static const dto_con_class k_con[] = {\n" + (",\n".join(rows) if rows else "  {0}") + "\n};")
    rows = [f"  {{{d}, {p}, {c}, {kc}}}" for (d, p, c, kc) in st.kinds]
    out.append("static const dto_kind k_kinds[] = {\n" + ",\n".join(rows) + "\n};")
    if g is not None:
        nh = g.num_hessian if h else 0
        out.append(f"static const dto_general_class k_general = {{{g.num_variables}, {g.num_parameter}, {g.num_constraint}, "
                   f"{g.num_jacobian}, {nh}, gen_jr, gen_jc, gen_hr, gen_hc, {len(g.indices_inequality)}, gen_iq}};")
    out.append("static int launch(int op, const dto_eval_args* a, void* s) { return dto::launch_eval<Model>(op, a, s); }")
    out.append("static int launch_kkt(int op, const dto_kkt_args* a, void* s) { return dto::launch_kkt<Model>(op, a, s); }")
    if with_im:
        out.append("static int launch_im(int op, const dto_im_args* a, void* s) { return dto::im::launch_im<Model>(op, a, s); }")
    out.append("static const dto_model_vtable k_vtable = {")
    out.append(f'  DTO_PLUGIN_ABI, "{name}", {len(st.dyn)}, {len(st.cost)}, {len(st.con)}, {len(st.kinds)},')
    out.append(f"  k_dyn, k_cost, k_con, k_kinds, {'&k_general' if g is not None else 'nullptr'}, {1 if h else 0},")
    out.append(f"  Model::MAX_KEY, launch, launch_kkt, dto::kkt_info<Model>, nullptr, nullptr, "
               + ("launch_im, dto::im::im_info<Model>" if with_im else "nullptr, nullptr"))
    out.append("};")
    out.append("}  // namespace")
    out.append('extern "C" const dto_model_vtable* dto_model_get(void) { return &k_vtable; }')
    return "\n".join(out) + "\n"


def _kernel_headers_digest() -> str:
    """Everything besides the model that decides what a plugin contains: the hand-written kernel headers and the code
    generator itself (the structural key replaces expression bodies by fingerprints, so a change in HOW an expression is
    emitted would otherwise keep serving stale plugins)."""
    hsh = hashlib.sha256()
    for fn in sorted(os.listdir(CSRC)):
        # (dto_problem.hpp / dto_layout.hpp belong to the runtime library only: no plugin includes them)
        if fn.endswith((".hpp", ".h")) and fn not in ("dto_problem.hpp", "dto_layout.hpp"):
            with open(os.path.join(CSRC, fn), "rb") as f:
                hsh.update(f.read())
    here = os.path.dirname(os.path.abspath(__file__))
    for fn in ("plugin.py", os.path.join("symbolic", "codegen.py"), os.path.join("symbolic", "expr.py"),
               os.path.join("symbolic", "diff.py")):
        with open(os.path.join(here, fn), "rb") as f:
            hsh.update(f.read())
    return hsh.hexdigest()


def _extra_flags() -> List[str]:
    """Extra compiler flags of a plugin build (measurement variants: -DDTO_KKT_PROFILE=1, -DDTO_SEQ_FWD_OCC=2, ...); part of the
    cache key."""
    return os.environ.get("DTO_PLUGIN_CXXFLAGS", "").split()


# tile (MFMA) plugins: keep the f64 MFMA accumulators in VGPRs.  In AGPR form (the compiler's choice at this register pressure)
# the 32 accumulator registers of a product loop were copied VGPR -> AGPR at the top and AGPR -> VGPR at the bottom of every
# pass behind an `s_nop 15` that drains the matrix pipe (csrc/dto_wide_kernels.hpp: mm_row4)
WIDE_CXXFLAGS = ["-mllvm", "-amdgpu-mfma-vgpr-form"]

# Every device build: keep the exec-mask restore of EVERY divergent region (round 5, DESIGN.md section 4.3 "root cause").
# AMD clang 22 (ROCm 7.2) drops the restore of an inner `if` whose end coincides with the end of the enclosing divergent region
# (`s_and_saveexec` becomes `s_and_b64 exec, exec, cond`, SILowerControlFlow's redundant-endcf removal, before register
# allocation); the register allocator may then place a reload of a split live range in the merge block between the two ends,
# where it executes with the INNER mask: the lanes outside it keep a stale register.  That was the wrong-result mode of the
# fused sweeps (round 3/4: a never-taken `if (prof && tid == 0)` inside the stage loop) and of the solver-mode use of
# k_wide_step (`if (w == 0) { ...; if (a.stats) { ...; if (l == 0) ...; } }` around a register whose value the pass over the
# fixed states needs in every lane).  tools/check_exec_merge.py finds the pattern in the ISA; tests/test_exec_merge_guard.py
# pins it.
BASE_CXXFLAGS = ["-mllvm", "-amdgpu-remove-redundant-endcf=0"]


def _prepare_plugin(st: Structure, name: str):
    """(path of the plugin .so, compile command or None if it is already built).  Not thread-safe (the structural-key
    switch of the code generator is a module global): call it serially, compile in parallel."""
    # cache key: the source with every expression body replaced by a structural (id-independent) fingerprint
    from .symbolic import codegen as _cg
    _cg.STRUCTURAL_KEYS = True
    try:
        key_src = generate_source(st, name)
    finally:
        _cg.STRUCTURAL_KEYS = False
    flags = BASE_CXXFLAGS + (WIDE_CXXFLAGS if st.wide else []) + _extra_flags()
    digest = hashlib.sha256((key_src + _kernel_headers_digest() + GENERATOR_VERSION + " ".join(flags)).encode()).hexdigest()[:16]
    os.makedirs(PLUGIN_DIR, exist_ok=True)
    base = os.path.join(PLUGIN_DIR, f"{name}_{digest}")
    so = base + ".so"
    if os.path.exists(so):
        return so, None
    src = generate_source(st, name)
    hip_src = base + ".hip"
    with open(hip_src, "w") as f:
        f.write(src)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tmp = so + f".tmp{os.getpid()}"
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-I", CSRC,
           "-Wno-unused-value"] + flags + ["-o", tmp, hip_src]
    return so, cmd


COMPILED: List[str] = []   # plugins compiled by this process (tests/conftest.py reports it: a GPU box should compile none)


def _compile_plugin(name: str, so: str, cmd, verbose: bool = False) -> str:
    if cmd is None:
        return so
    COMPILED.append(os.path.basename(so))
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed for plugin {name}:\n{res.stderr[-4000:]}")
    os.replace(cmd[cmd.index("-o") + 1], so)
    return so


def build_plugin(st: Structure, name: str = "model", verbose: bool = False) -> str:
    """Generate + compile (if not cached) and return the plugin path."""
    so, cmd = _prepare_plugin(st, name)
    return _compile_plugin(name, so, cmd, verbose)


def build_plugins(items, verbose: bool = False, jobs: int = 0):
    """Build several plugins: sources are generated serially, hipcc runs `jobs` at a time (default: one per core, at
    most 8 -- a plugin compile peaks at a few GB of host memory)."""
    from concurrent.futures import ThreadPoolExecutor
    prepared, seen = [], set()
    for name, st in items:
        so, cmd = _prepare_plugin(st, name)
        if so in seen:
            cmd = None
        seen.add(so)
        prepared.append((name, so, cmd))
    jobs = jobs or max(1, min(8, os.cpu_count() or 1))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        return list(ex.map(lambda a: _compile_plugin(a[0], a[1], a[2], verbose), prepared))
