"""Symbolic differentiation and structural sparsity on the Expr DAG.

Stands in for `Symbolics.gradient`, `Symbolics.sparsejacobian` and
`Symbolics.sparsehessian` as called at src/costs.jl:20,25, src/dynamics.jl:25,33,
src/constraints.jl:29,38 and src/general_constraint.jl:25,34 of the reference.

Pattern rules (SURVEY.md Appendix A.4 -- Symbolics' own rules are recalled, not
verifiable in this environment, so they are restated here explicitly):

* Jacobian: entry (i, j) is structural iff variable j occurs in expression i after
  the construction-time folding done in expr.py.
* Hessian: linearity propagation.  Every expression carries a set of monomial-like
  *terms* (variable -> degree capped at 2).  `+`/`-` unite term sets, `*` multiplies
  them pairwise, `/` multiplies by the nonlinear image of the denominator, a
  nonlinear unary function maps its argument to one term holding every variable of
  the argument at degree 2.  (i, i) is structural iff some term has degree 2 in i,
  (i, j) iff some term holds both.  The full symmetric pattern (both triangles) is
  returned, as the reference does (test/hessian_lagrangian.jl:196-201).

Both patterns are produced in CSC order (column-major: sorted by column, then row),
the order of `findnz` on a SparseMatrixCSC (src/dynamics.jl:29,35).
"""
from __future__ import annotations

from typing import Dict, FrozenSet, List, Sequence, Tuple

from . import expr as E
from .expr import Expr

VarKey = Tuple[str, int]


def _key(v: Expr) -> VarKey:
    return (v.name, v.index)


# ----------------------------------------------------------------------------
# occurrence
# ----------------------------------------------------------------------------
_occ_cache: Dict[int, FrozenSet[VarKey]] = {}


def occurs(e: Expr) -> FrozenSet[VarKey]:
    hit = _occ_cache.get(e.id)
    if hit is not None:
        return hit
    for n in E.topo_order([e]):
        if n.id in _occ_cache:
            continue
        if n.op == E.CONST:
            s = frozenset()
        elif n.op == E.VAR:
            s = frozenset([_key(n)])
        elif len(n.args) == 1:
            s = _occ_cache[n.args[0].id]
        else:
            s = _occ_cache[n.args[0].id]
            for a in n.args[1:]:  # ifelse: the variables of the condition occur too (occurrence rule)
                s = s | _occ_cache[a.id]
        _occ_cache[n.id] = s
    return _occ_cache[e.id]


# ----------------------------------------------------------------------------
# derivative
# ----------------------------------------------------------------------------
_diff_cache: Dict[Tuple[int, VarKey], Expr] = {}


def diff(e: Expr, v: Expr) -> Expr:
    """d e / d v for a VAR node v (forward symbolic derivative, memoised on the DAG)."""
    vk = _key(v)
    hit = _diff_cache.get((e.id, vk))
    if hit is not None:
        return hit
    for n in E.topo_order([e]):
        ck = (n.id, vk)
        if ck in _diff_cache:
            continue
        if vk not in occurs(n):
            _diff_cache[ck] = E.ZERO
            continue
        d = lambda a: _diff_cache[(a.id, vk)]
        op = n.op
        if op == E.VAR:
            r = E.ONE
        elif op == E.ADD:
            r = E.add(d(n.args[0]), d(n.args[1]))
        elif op == E.SUB:
            r = E.sub(d(n.args[0]), d(n.args[1]))
        elif op == E.NEG:
            r = E.neg(d(n.args[0]))
        elif op == E.MUL:
            a, b = n.args
            r = E.add(E.mul(d(a), b), E.mul(a, d(b)))
        elif op == E.DIV:
            a, b = n.args
            # (a/b)' = (a' - (a/b) b') / b ; reuses the quotient node itself
            r = E.div(E.sub(d(a), E.mul(n, d(b))), b)
        elif op == E.POWI:
            a = n.args[0]
            k = n.value
            r = E.mul(E.mul(E.const(float(k)), E.power(a, k - 1)), d(a))
        elif op == E.POW:
            a, b = n.args
            if b.op == E.CONST:
                r = E.mul(E.mul(b, E.power(a, b.value - 1.0)), d(a))
            else:
                r = E.mul(n, E.add(E.mul(d(b), E.func("log", a)), E.div(E.mul(b, d(a)), a)))
        elif op == E.IFELSE:
            # the branches are differentiated, the condition is kept (IfElse.ifelse under Symbolics.derivative)
            l, rr, a, b = n.args
            r = E.ifelse(E.Cond(n.fn, l, rr), d(a), d(b))
        else:  # FUNC
            a = n.args[0]
            fn = n.fn
            if fn == "sin":
                g = E.func("cos", a)
            elif fn == "cos":
                g = E.neg(E.func("sin", a))
            elif fn == "tan":
                g = E.add(E.ONE, E.mul(n, n))
            elif fn == "exp":
                g = n
            elif fn == "log":
                g = E.div(E.ONE, a)
            elif fn == "sqrt":
                g = E.div(E.const(0.5), n)
            elif fn == "tanh":
                g = E.sub(E.ONE, E.mul(n, n))
            elif fn == "atan":
                g = E.div(E.ONE, E.add(E.ONE, E.mul(a, a)))
            elif fn == "asin":
                g = E.div(E.ONE, E.func("sqrt", E.sub(E.ONE, E.mul(a, a))))
            elif fn == "acos":
                g = E.neg(E.div(E.ONE, E.func("sqrt", E.sub(E.ONE, E.mul(a, a)))))
            elif fn == "sinh":
                g = E.func("cosh", a)
            elif fn == "cosh":
                g = E.func("sinh", a)
            elif fn == "abs":
                g = E.div(a, n)
            else:
                raise NotImplementedError(fn)
            r = E.mul(g, d(a))
        _diff_cache[ck] = r
    return _diff_cache[(e.id, vk)]


def gradient(e: Expr, wrt: Sequence[Expr]) -> List[Expr]:
    """Dense gradient (Symbolics.gradient, src/costs.jl:20)."""
    return [diff(e, v) for v in wrt]


# ----------------------------------------------------------------------------
# Jacobian
# ----------------------------------------------------------------------------
def jacobian_sparsity(f: Sequence[Expr], wrt: Sequence[Expr]) -> List[Tuple[int, int]]:
    """Structural (row, col) pairs, 0-based, CSC order."""
    col_of = {_key(v): j for j, v in enumerate(wrt)}
    pat = []
    for i, fi in enumerate(f):
        for k in occurs(fi):
            j = col_of.get(k)
            if j is not None:
                pat.append((i, j))
    pat.sort(key=lambda rc: (rc[1], rc[0]))
    return pat


def sparse_jacobian(f: Sequence[Expr], wrt: Sequence[Expr]):
    """(rows, cols, values) with 0-based indices in CSC order (src/dynamics.jl:25-29)."""
    pat = jacobian_sparsity(f, wrt)
    rows = [r for r, _ in pat]
    cols = [c for _, c in pat]
    vals = [diff(f[r], wrt[c]) for r, c in pat]
    return rows, cols, vals


def dense_jacobian_pattern(nrow: int, ncol: int):
    """Dense column-major pattern of the user-Jacobian ctor (src/dynamics.jl:72-79)."""
    rows, cols = [], []
    for j in range(ncol):
        for i in range(nrow):
            rows.append(i)
            cols.append(j)
    return rows, cols


# ----------------------------------------------------------------------------
# Hessian: linearity propagation
# ----------------------------------------------------------------------------
Term = Tuple[Tuple[int, int], ...]  # sorted ((col, degree), ...)


def _prune(terms: FrozenSet[Term]) -> FrozenSet[Term]:
    """Drop terms dominated by another term (same pairs and diagonals implied)."""
    ts = sorted(terms, key=lambda t: -len(t))
    kept: List[Dict[int, int]] = []
    out = []
    for t in ts:
        dt = dict(t)
        dom = False
        for k in kept:
            if all(k.get(c, 0) >= dg for c, dg in dt.items()):
                dom = True
                break
        if not dom:
            kept.append(dt)
            out.append(t)
    return frozenset(out)


def _tmul(a: FrozenSet[Term], b: FrozenSet[Term]) -> FrozenSet[Term]:
    if not a:
        return b
    if not b:
        return a
    out = set()
    for s in a:
        ds = dict(s)
        for t in b:
            m = dict(ds)
            for c, dg in t:
                m[c] = min(2, m.get(c, 0) + dg)
            out.add(tuple(sorted(m.items())))
    return _prune(frozenset(out))


def _nonlinear(a: FrozenSet[Term]) -> FrozenSet[Term]:
    cols = set()
    for t in a:
        for c, _ in t:
            cols.add(c)
    if not cols:
        return frozenset()
    return frozenset([tuple((c, 2) for c in sorted(cols))])


def linearity_terms(e: Expr, col_of: Dict[VarKey, int]) -> FrozenSet[Term]:
    cache: Dict[int, FrozenSet[Term]] = {}
    for n in E.topo_order([e]):
        op = n.op
        if op == E.CONST:
            t = frozenset()
        elif op == E.VAR:
            j = col_of.get(_key(n))
            t = frozenset() if j is None else frozenset([((j, 1),)])
        elif op in (E.ADD, E.SUB):
            t = _prune(cache[n.args[0].id] | cache[n.args[1].id])
        elif op == E.NEG:
            t = cache[n.args[0].id]
        elif op == E.MUL:
            t = _tmul(cache[n.args[0].id], cache[n.args[1].id])
        elif op == E.DIV:
            t = _tmul(cache[n.args[0].id], _nonlinear(cache[n.args[1].id]))
            if not cache[n.args[0].id]:
                t = _nonlinear(cache[n.args[1].id])
        elif op == E.POWI:
            a = cache[n.args[0].id]
            t = _tmul(a, a)  # degree caps at 2, so k >= 2 all look alike
        elif op == E.POW:
            t = _nonlinear(cache[n.args[0].id] | cache[n.args[1].id])
        elif op == E.IFELSE:
            # second derivatives come from the branches only: union of their terms, the condition contributes nothing
            t = _prune(cache[n.args[2].id] | cache[n.args[3].id])
        else:
            t = _nonlinear(cache[n.args[0].id])
        cache[n.id] = t
    return cache[e.id]


def hessian_sparsity(e: Expr, wrt: Sequence[Expr]) -> List[Tuple[int, int]]:
    """Full symmetric structural pattern, 0-based (row, col), CSC order."""
    col_of = {_key(v): j for j, v in enumerate(wrt)}
    pat = set()
    for t in linearity_terms(e, col_of):
        cols = [c for c, _ in t]
        for c, dg in t:
            if dg >= 2:
                pat.add((c, c))
        for a in cols:
            for b in cols:
                if a != b:
                    pat.add((a, b))
    return sorted(pat, key=lambda rc: (rc[1], rc[0]))


def sparse_hessian(e: Expr, wrt: Sequence[Expr]):
    """(rows, cols, values), both triangles, CSC order (src/dynamics.jl:33-35)."""
    pat = hessian_sparsity(e, wrt)
    grad: Dict[int, Expr] = {}
    vals_by: Dict[Tuple[int, int], Expr] = {}
    for r, c in pat:
        lo, hi = (r, c) if r <= c else (c, r)
        if (lo, hi) not in vals_by:
            if lo not in grad:
                grad[lo] = diff(e, wrt[lo])
            vals_by[(lo, hi)] = diff(grad[lo], wrt[hi])
    rows = [r for r, _ in pat]
    cols = [c for _, c in pat]
    vals = [vals_by[(r, c) if r <= c else (c, r)] for r, c in pat]
    return rows, cols, vals
