"""Hash-consed expression DAG used to trace user model closures.

This is the construction-time front end that takes the place Symbolics.jl has in
the reference (`@variables` + tracing the closure with symbolic arrays:
src/dynamics.jl:23-24, src/costs.jl:18-19, src/constraints.jl:27-28,
src/general_constraint.jl:23-24).  It is deliberately small: scalars only,
vectors/matrices are numpy object arrays holding `Expr` nodes, so a user closure
written with numpy (`np.sin`, `@`, slicing) traces unchanged.

Construction-time folding (needed for the sparsity contract):
constants are folded, `0*e -> 0`, `1*e -> e`, `e+0 -> e`, `e-e -> 0`, `--e -> e`.
Nodes are interned, so structurally equal sub-expressions are one object: CSE in
the emitted kernels falls out of the representation.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

# op tags
CONST, VAR, ADD, SUB, MUL, DIV, NEG, POWI, POW, FUNC, IFELSE = range(11)

_FUNCS = ("sin", "cos", "tan", "exp", "log", "sqrt", "tanh", "atan", "asin", "acos", "sinh", "cosh", "abs")


class Expr:
    """One DAG node. Do not construct directly: use `const`, `var` and operators."""

    __slots__ = ("op", "args", "value", "name", "index", "fn", "id", "__weakref__")
    _table: Dict[tuple, "Expr"] = {}
    _count = 0

    def __new__(cls, op, args=(), value=None, name=None, index=None, fn=None):
        if op == CONST:
            key = (CONST, float(value).hex())
        elif op == VAR:
            key = (VAR, name, index)
        elif op == FUNC:
            key = (FUNC, fn, args[0].id)
        elif op == POWI:
            key = (POWI, args[0].id, value)
        elif op == IFELSE:
            key = (IFELSE, fn) + tuple(a.id for a in args)
        else:
            key = (op,) + tuple(a.id for a in args)
        hit = cls._table.get(key)
        if hit is not None:
            return hit
        self = object.__new__(cls)
        self.op = op
        self.args = tuple(args)
        self.value = value
        self.name = name
        self.index = index
        self.fn = fn
        self.id = cls._count
        cls._count += 1
        cls._table[key] = self
        return self

    def __hash__(self):
        return self.id

    def __eq__(self, other):  # identity: nodes are interned
        return self is other

    # ---- classification helpers
    @property
    def is_const(self):
        return self.op == CONST

    def is_zero(self):
        return self.op == CONST and self.value == 0.0

    def is_one(self):
        return self.op == CONST and self.value == 1.0

    # ---- arithmetic
    def __add__(self, o):
        if isinstance(o, np.ndarray) and o.ndim:
            return NotImplemented
        return add(self, as_expr(o))

    def __radd__(self, o):
        if isinstance(o, np.ndarray) and o.ndim:
            return NotImplemented
        return add(as_expr(o), self)

    def __sub__(self, o):
        if isinstance(o, np.ndarray) and o.ndim:
            return NotImplemented
        return sub(self, as_expr(o))

    def __rsub__(self, o):
        if isinstance(o, np.ndarray) and o.ndim:
            return NotImplemented
        return sub(as_expr(o), self)

    def __mul__(self, o):
        if isinstance(o, np.ndarray) and o.ndim:
            return NotImplemented
        return mul(self, as_expr(o))

    def __rmul__(self, o):
        if isinstance(o, np.ndarray) and o.ndim:
            return NotImplemented
        return mul(as_expr(o), self)

    def __truediv__(self, o):
        if isinstance(o, np.ndarray) and o.ndim:
            return NotImplemented
        return div(self, as_expr(o))

    def __rtruediv__(self, o):
        if isinstance(o, np.ndarray) and o.ndim:
            return NotImplemented
        return div(as_expr(o), self)

    def __neg__(self):
        return neg(self)

    def __pos__(self):
        return self

    def __pow__(self, o):
        return power(self, o)

    def __rpow__(self, o):
        return power(as_expr(o), self)

    # numpy ufuncs on object arrays dispatch to these methods
    def sin(self):
        return func("sin", self)

    def cos(self):
        return func("cos", self)

    def tan(self):
        return func("tan", self)

    def exp(self):
        return func("exp", self)

    def log(self):
        return func("log", self)

    def sqrt(self):
        return func("sqrt", self)

    def tanh(self):
        return func("tanh", self)

    def arctan(self):
        return func("atan", self)

    def arcsin(self):
        return func("asin", self)

    def arccos(self):
        return func("acos", self)

    def sinh(self):
        return func("sinh", self)

    def cosh(self):
        return func("cosh", self)

    def __abs__(self):
        return func("abs", self)

    # ---- comparisons build a condition for `ifelse` (IfElse.ifelse in the reference, src/DirectTrajectoryOptimization.jl:5)
    def __lt__(self, o):
        return Cond("lt", self, as_expr(o))

    def __le__(self, o):
        return Cond("le", self, as_expr(o))

    def __gt__(self, o):
        return Cond("lt", as_expr(o), self)

    def __ge__(self, o):
        return Cond("le", as_expr(o), self)

    def __repr__(self):
        return to_str(self)

    def __float__(self):
        if self.op == CONST:
            return float(self.value)
        raise TypeError("symbolic expression has no float value")


class Cond:
    """`lhs < rhs` or `lhs <= rhs` between symbolic scalars; only meaningful as the first argument of `ifelse`."""

    __slots__ = ("fn", "lhs", "rhs")

    def __init__(self, fn, lhs, rhs):
        self.fn, self.lhs, self.rhs = fn, lhs, rhs

    def __bool__(self):
        raise TypeError("a comparison of symbolic values has no truth value while the model is traced: write "
                        "ifelse(a < b, x, y) / minimum(a, b) / maximum(a, b) instead of a Python `if`, min() or max()")


def as_expr(v) -> Expr:
    if isinstance(v, Expr):
        return v
    if isinstance(v, (int, float, np.integer, np.floating)):
        return const(float(v))
    if isinstance(v, np.ndarray) and v.ndim == 0:
        return as_expr(v.item())
    raise TypeError(f"cannot convert {type(v)} to Expr")


def const(v: float) -> Expr:
    v = float(v)
    if v == 0.0:
        v = 0.0  # fold -0.0
    return Expr(CONST, value=v)


ZERO = const(0.0)
ONE = const(1.0)


def var(name: str, index: int) -> Expr:
    return Expr(VAR, name=name, index=int(index))


def variables(name: str, n: int) -> np.ndarray:
    """`@variables name[1:n]` (src/dynamics.jl:23): a numpy object vector of VAR nodes."""
    out = np.empty(n, dtype=object)
    for i in range(n):
        out[i] = var(name, i)
    return out


def add(a: Expr, b: Expr) -> Expr:
    if a.op == CONST and b.op == CONST:
        return const(a.value + b.value)
    if a.is_zero():
        return b
    if b.is_zero():
        return a
    if b.op == NEG:
        return sub(a, b.args[0])
    if a.op == NEG:
        return sub(b, a.args[0])
    if a.id > b.id:
        a, b = b, a
    return Expr(ADD, (a, b))


def sub(a: Expr, b: Expr) -> Expr:
    if a.op == CONST and b.op == CONST:
        return const(a.value - b.value)
    if b.is_zero():
        return a
    if a.is_zero():
        return neg(b)
    if a is b:
        return ZERO
    if b.op == NEG:
        return add(a, b.args[0])
    return Expr(SUB, (a, b))


def neg(a: Expr) -> Expr:
    if a.op == CONST:
        return const(-a.value)
    if a.op == NEG:
        return a.args[0]
    if a.op == SUB:
        return sub(a.args[1], a.args[0])
    return Expr(NEG, (a,))


def mul(a: Expr, b: Expr) -> Expr:
    if a.op == CONST and b.op == CONST:
        return const(a.value * b.value)
    if a.is_zero() or b.is_zero():
        return ZERO
    if a.is_one():
        return b
    if b.is_one():
        return a
    if a.op == CONST and a.value == -1.0:
        return neg(b)
    if b.op == CONST and b.value == -1.0:
        return neg(a)
    if a.op == NEG and b.op == NEG:
        return mul(a.args[0], b.args[0])
    if a.op == NEG:
        return neg(mul(a.args[0], b))
    if b.op == NEG:
        return neg(mul(a, b.args[0]))
    if a is b:
        return Expr(POWI, (a,), value=2)
    if a.id > b.id:
        a, b = b, a
    return Expr(MUL, (a, b))


def div(a: Expr, b: Expr) -> Expr:
    if b.op == CONST:
        if b.value == 1.0:
            return a
        if a.op == CONST:
            return const(a.value / b.value)
        if b.value == -1.0:
            return neg(a)
    if a.is_zero():
        return ZERO
    if a.op == NEG:
        return neg(div(a.args[0], b))
    if b.op == NEG:
        return neg(div(a, b.args[0]))
    return Expr(DIV, (a, b))


def power(a: Expr, p) -> Expr:
    if isinstance(p, Expr) and p.op == CONST:
        p = p.value
    if isinstance(p, (int, float, np.integer, np.floating)):
        pf = float(p)
        if a.op == CONST:
            return const(a.value ** pf)
        if pf == int(pf) and abs(pf) <= 64:
            k = int(pf)
            if k == 0:
                return ONE
            if k == 1:
                return a
            if k < 0:
                return div(ONE, power(a, -k))
            if a.op == POWI:
                return Expr(POWI, (a.args[0],), value=a.value * k)
            return Expr(POWI, (a,), value=k)
        if pf == 0.5:
            return func("sqrt", a)
        return Expr(POW, (a, const(pf)))
    return Expr(POW, (a, as_expr(p)))


_FOLD = {
    "sin": math.sin, "cos": math.cos, "tan": math.tan, "exp": math.exp, "log": math.log,
    "sqrt": math.sqrt, "tanh": math.tanh, "atan": math.atan, "asin": math.asin,
    "acos": math.acos, "sinh": math.sinh, "cosh": math.cosh, "abs": abs,
}


def func(fn: str, a: Expr) -> Expr:
    a = as_expr(a)
    if a.op == CONST:
        return const(_FOLD[fn](a.value))
    return Expr(FUNC, (a,), fn=fn)


def ifelse(cond, a, b):
    """IfElse.ifelse(cond, a, b): `a` where cond holds, else `b`.  cond is a comparison of symbolic scalars (or a bool).
    Derivatives differentiate the branches and keep the condition (what Symbolics does for ifelse)."""
    if isinstance(cond, (bool, np.bool_)):
        return a if cond else b
    if not isinstance(cond, Cond):
        raise TypeError("ifelse: the condition must be a comparison such as x[0] < 0.0")
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        a2, b2 = np.broadcast_arrays(np.asarray(a, dtype=object), np.asarray(b, dtype=object))
        out = np.empty(a2.shape, dtype=object)
        for i in np.ndindex(a2.shape):
            out[i] = ifelse(cond, a2[i], b2[i])
        return out
    a, b = as_expr(a), as_expr(b)
    l, r = cond.lhs, cond.rhs
    if l.op == CONST and r.op == CONST:
        return a if ((l.value < r.value) if cond.fn == "lt" else (l.value <= r.value)) else b
    if a is b:
        return a
    return Expr(IFELSE, (l, r, a, b), fn=cond.fn)


def minimum(a, b):
    """min(a, b) on symbolic scalars (elementwise on arrays)."""
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        a2, b2 = np.broadcast_arrays(np.asarray(a, dtype=object), np.asarray(b, dtype=object))
        out = np.empty(a2.shape, dtype=object)
        for i in np.ndindex(a2.shape):
            out[i] = minimum(a2[i], b2[i])
        return out
    a, b = as_expr(a), as_expr(b)
    return ifelse(Cond("lt", a, b), a, b)


def maximum(a, b):
    """max(a, b) on symbolic scalars (elementwise on arrays)."""
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        a2, b2 = np.broadcast_arrays(np.asarray(a, dtype=object), np.asarray(b, dtype=object))
        out = np.empty(a2.shape, dtype=object)
        for i in np.ndindex(a2.shape):
            out[i] = maximum(a2[i], b2[i])
        return out
    a, b = as_expr(a), as_expr(b)
    return ifelse(Cond("lt", b, a), a, b)


# convenience module-level functions (mirror of Base.sin etc. on symbolic scalars)
def sin(a):
    return np.sin(a) if isinstance(a, np.ndarray) else (func("sin", a) if isinstance(a, Expr) else math.sin(a))


def cos(a):
    return np.cos(a) if isinstance(a, np.ndarray) else (func("cos", a) if isinstance(a, Expr) else math.cos(a))


def tan(a):
    return np.tan(a) if isinstance(a, np.ndarray) else (func("tan", a) if isinstance(a, Expr) else math.tan(a))


def dot(a, b):
    """LinearAlgebra.dot for symbolic or numeric vectors."""
    a = np.asarray(a, dtype=object).ravel()
    b = np.asarray(b, dtype=object).ravel()
    if len(a) != len(b):
        raise ValueError("dot: length mismatch")
    acc = 0.0
    for p, q in zip(a, b):
        acc = acc + p * q
    return acc


def to_str(e: Expr, depth: int = 0) -> str:
    if depth > 6:
        return "..."
    if e.op == CONST:
        return repr(e.value)
    if e.op == VAR:
        return f"{e.name}[{e.index}]"
    if e.op == NEG:
        return f"(-{to_str(e.args[0], depth + 1)})"
    if e.op == FUNC:
        return f"{e.fn}({to_str(e.args[0], depth + 1)})"
    if e.op == POWI:
        return f"{to_str(e.args[0], depth + 1)}^{e.value}"
    if e.op == IFELSE:
        sym = "<" if e.fn == "lt" else "<="
        return (f"ifelse({to_str(e.args[0], depth + 1)} {sym} {to_str(e.args[1], depth + 1)}, "
                f"{to_str(e.args[2], depth + 1)}, {to_str(e.args[3], depth + 1)})")
    sym = {ADD: "+", SUB: "-", MUL: "*", DIV: "/", POW: "^"}[e.op]
    return f"({to_str(e.args[0], depth + 1)} {sym} {to_str(e.args[1], depth + 1)})"


def topo_order(roots: Iterable[Expr]) -> List[Expr]:
    """Nodes reachable from `roots`, children before parents (iterative DFS)."""
    seen = set()
    out: List[Expr] = []
    for r in roots:
        if r.id in seen:
            continue
        stack: List[Tuple[Expr, int]] = [(r, 0)]
        while stack:
            node, i = stack.pop()
            if i == 0 and node.id in seen:
                continue
            if i < len(node.args):
                stack.append((node, i + 1))
                child = node.args[i]
                if child.id not in seen:
                    stack.append((child, 0))
            else:
                if node.id not in seen:
                    seen.add(node.id)
                    out.append(node)
    return out


def substitute(roots: Sequence[Expr], mapping: Dict[Expr, Expr]) -> List[Expr]:
    """Rebuild `roots` with the nodes in `mapping` replaced (folding rules re-applied on the way up)."""
    memo: Dict[int, Expr] = {}
    for node in topo_order(roots):
        if node in mapping:
            new = mapping[node]
        elif node.op in (CONST, VAR):
            new = node
        else:
            args = [memo[a.id] for a in node.args]
            if node.op == ADD:
                new = add(args[0], args[1])
            elif node.op == SUB:
                new = sub(args[0], args[1])
            elif node.op == MUL:
                new = mul(args[0], args[1])
            elif node.op == DIV:
                new = div(args[0], args[1])
            elif node.op == NEG:
                new = neg(args[0])
            elif node.op == POWI:
                new = power(args[0], node.value)
            elif node.op == POW:
                new = power(args[0], args[1])
            elif node.op == IFELSE:
                new = ifelse(Cond(node.fn, args[0], args[1]), args[2], args[3])
            else:
                new = func(node.fn, args[0])
        memo[node.id] = new
    return [memo[r.id] for r in roots]
