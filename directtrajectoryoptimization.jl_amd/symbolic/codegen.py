"""Straight-line C/HIP emission for lists of Expr outputs.

Takes the place of `Symbolics.build_function(...)[2]` + `eval` (src/dynamics.jl:26-27,34;
src/costs.jl:22-26; src/constraints.jl:30-31,39) -- but the target is a `__device__`
function body that the hand-written stage kernels in csrc/ inline, not a Julia closure.

The emitted body is pure straight-line code over `double` temporaries: every DAG node is
computed once (the DAG is interned, so this is global CSE across *all* outputs of the
function -- the reference's generated closures recompute shared sub-expressions per entry),
sin/cos of one argument are paired into a single `DTO_SINCOS` (csrc/dto_math.hpp: straight-line f64 sin + cos), integer powers are expanded into
multiplications.  No indexing is dynamic, so after inlining everything lives in VGPRs.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

from . import expr as E
from .expr import Expr


def _lit(v: float) -> str:
    if v != v:
        return "(0.0/0.0)"
    if v in (float("inf"), float("-inf")):
        return "(1.0/0.0)" if v > 0 else "(-1.0/0.0)"
    s = repr(float(v))
    if "." not in s and "e" not in s and "n" not in s:
        s += ".0"
    return s if v >= 0 else f"({s})"


def _powi(a: str, k: int) -> str:
    if k == 1:
        return a
    if k % 2 == 0:
        h = _powi(a, k // 2)
        return f"({h}*{h})"
    return f"({_powi(a, k - 1)}*{a})"


_CFN = {"sin": "sin", "cos": "cos", "tan": "tan", "exp": "exp", "log": "log", "sqrt": "sqrt",
        "tanh": "tanh", "atan": "atan", "asin": "asin", "acos": "acos", "sinh": "sinh",
        "cosh": "cosh", "abs": "fabs"}


# When set, emit_body returns a structural fingerprint of its outputs instead of C statements.  The statements name
# temporaries after node ids and order commutative operands by id, and ids depend on everything traced earlier in the
# process -- the same model would otherwise get a different source text (and plugin cache key) in every process.
STRUCTURAL_KEYS = False


def structural_hashes(outputs: Sequence[Expr]) -> List[str]:
    """Merkle hash of every output: independent of node ids, commutative operands (ADD, MUL) sorted."""
    import hashlib
    memo: Dict[int, str] = {}
    for n in E.topo_order(outputs):
        if n.op == E.CONST:
            key = ("c", repr(float(n.value)))
        elif n.op == E.VAR:
            key = ("v", n.name, int(n.index))
        else:
            kids = [memo[a.id] for a in n.args]
            if n.op in (E.ADD, E.MUL):
                kids.sort()
            key = ("o", int(n.op), n.fn if n.op in (E.FUNC, E.IFELSE) else "", repr(n.value) if n.op == E.POWI else "", tuple(kids))
        memo[n.id] = hashlib.blake2b(repr(key).encode(), digest_size=12).hexdigest()
    return [memo[o.id] for o in outputs]


def is_affine(e: Expr, memo: Dict[int, bool] | None = None) -> bool:
    """e is an affine function of the VAR leaves (constants and parameters 'w' count as coefficients)."""
    memo = {} if memo is None else memo

    def is_coef(x: Expr) -> bool:      # free of x / u / y / z variables
        return all(n.op != E.VAR or n.name == "w" for n in E.topo_order([x]))

    def rec(x: Expr) -> bool:
        if x.id in memo:
            return memo[x.id]
        if x.op in (E.CONST, E.VAR):
            r = True
        elif x.op in (E.ADD, E.SUB):
            r = rec(x.args[0]) and rec(x.args[1])
        elif x.op == E.NEG:
            r = rec(x.args[0])
        elif x.op == E.MUL:
            r = (is_coef(x.args[0]) and rec(x.args[1])) or (is_coef(x.args[1]) and rec(x.args[0]))
        elif x.op == E.DIV:
            r = is_coef(x.args[1]) and rec(x.args[0])
        else:
            r = is_coef(x)
        memo[x.id] = r
        return r

    return rec(e)


def trig_arguments(outputs: Sequence[Expr]) -> List[Expr]:
    """Distinct argument nodes of the sin / cos calls in `outputs`, in first-use order."""
    seen, args = set(), []
    for n in E.topo_order(outputs):
        if n.op == E.FUNC and n.fn in ("sin", "cos") and n.args[0].id not in seen:
            seen.add(n.args[0].id)
            args.append(n.args[0])
    return args


def emit_body(outputs: Sequence[Expr], out_name, var_arrays: Dict[str, str],
              tmp_prefix: str = "t", indent: str = "    ", scale: str | None = None,
              trig_override: Dict[int, int] | None = None) -> str:
    """C statements assigning `out_name[k] = outputs[k]` for all k.

    var_arrays maps a VAR family name ('x', 'u', 'y', 'w', 'lam', 'z') to the C array it is read from.
    out_name may be a list of (array name, count) pairs: `outputs` is then the concatenation of several output lists that
    share ONE body (common subexpressions, sincos pairs) and are written to their own arrays.
    """
    groups = [(out_name, len(outputs))] if isinstance(out_name, str) else list(out_name)
    assert sum(c for _, c in groups) == len(outputs)
    out_name = "+".join(f"{nm}:{c}" for nm, c in groups) if len(groups) > 1 else groups[0][0]
    if STRUCTURAL_KEYS:
        va = ",".join(f"{k}={v}" for k, v in sorted(var_arrays.items()))
        return f"{indent}// {out_name} {tmp_prefix} {scale} {'trig' if trig_override is not None else ''} [{va}] " + " ".join(structural_hashes(outputs))
    order = E.topo_order(outputs)
    name: Dict[int, str] = {}
    lines: List[str] = []
    # which arguments have both sin and cos taken
    sin_of: Dict[int, Expr] = {}
    cos_of: Dict[int, Expr] = {}
    for n in order:
        if n.op == E.FUNC and n.fn == "sin":
            sin_of[n.args[0].id] = n
        elif n.op == E.FUNC and n.fn == "cos":
            cos_of[n.args[0].id] = n
    paired = {aid for aid in sin_of if aid in cos_of}
    done_pairs = set()
    uses: Dict[int, int] = {}
    for n in order:
        for a in n.args:
            uses[a.id] = uses.get(a.id, 0) + 1
    for o in outputs:
        uses[o.id] = uses.get(o.id, 0) + 1

    def ref(a: Expr) -> str:
        return name[a.id]

    def ensure_pair(a: Expr):
        """names of (sin a, cos a); the sincos call is emitted where the pair is first needed"""
        s, c = f"{tmp_prefix}{sin_of[a.id].id}", f"{tmp_prefix}{cos_of[a.id].id}"
        if a.id not in done_pairs:
            done_pairs.add(a.id)
            lines.append(f"{indent}double {s}, {c}; DTO_SINCOS({ref(a)}, &{s}, &{c});")
        return s, c

    for n in order:
        op = n.op
        if op == E.CONST:
            name[n.id] = _lit(n.value)
            continue
        if op == E.VAR:
            name[n.id] = f"{var_arrays[n.name]}[{n.index}]"
            continue
        if op == E.NEG:
            name[n.id] = f"(-{ref(n.args[0])})"
            continue
        t = f"{tmp_prefix}{n.id}"
        if op == E.MUL:
            # constants of a chain of single-use products are multiplied out here: c1 * (c2 * e) -> (c1 c2) * e (the chain rule
            # through m = (x + y) / 2 leaves many 0.5 * (0.5 * e); the inner temporaries become dead code)
            coef, factors, folded = 1.0, [], 0
            stack = list(n.args)
            while stack:
                a = stack.pop()
                if a.op == E.CONST:
                    coef *= a.value
                    folded += 1
                elif a.op == E.MUL and uses[a.id] == 1 and any(k.op == E.CONST for k in a.args):
                    stack.extend(a.args)
                else:
                    factors.append(a)
            if folded >= 2 and factors:
                factors.sort(key=lambda e: e.id)
                rhs = " * ".join([_lit(coef)] + [ref(f) for f in factors])
                lines.append(f"{indent}const double {t} = {rhs};")
                name[n.id] = t
                continue
        if op == E.ADD:
            rhs = f"{ref(n.args[0])} + {ref(n.args[1])}"
        elif op == E.SUB:
            rhs = f"{ref(n.args[0])} - {ref(n.args[1])}"
        elif op == E.MUL:
            rhs = f"{ref(n.args[0])} * {ref(n.args[1])}"
        elif op == E.DIV:
            rhs = f"{ref(n.args[0])} / {ref(n.args[1])}"
        elif op == E.POWI:
            rhs = _powi(ref(n.args[0]), n.value)
        elif op == E.POW:
            rhs = f"pow({ref(n.args[0])}, {ref(n.args[1])})"
        elif op == E.IFELSE:
            # a select, not a branch: both sides are straight-line temporaries already (v_cndmask on the GPU)
            rhs = f"({ref(n.args[0])} {'<' if n.fn == 'lt' else '<='} {ref(n.args[1])}) ? {ref(n.args[2])} : {ref(n.args[3])}"
        else:
            aid = n.args[0].id
            arg = n.args[0]
            if trig_override is not None and n.fn in ("sin", "cos") and aid in trig_override:
                # the caller supplies sin / cos of this argument (line evaluation: csrc k_linesearch)
                name[n.id] = f"{'sn' if n.fn == 'sin' else 'cs'}[{trig_override[aid]}]"
                continue
            if n.fn in ("sin", "cos") and aid in done_pairs:   # pair already produced by a sincos call (ensure_pair)
                name[n.id] = t
                continue
            if (n.fn in ("sin", "cos") and arg.op in (E.ADD, E.SUB)
                    and all(k.id in paired and k.op != E.CONST for k in arg.args)):
                # sin / cos of a sum whose terms have their own sincos pairs in this body (acrobot: q1, q2, q1 + q2): the
                # angle-addition identity costs two multiplications and an FMA instead of a ~100-instruction f64 sincos.
                # Absolute accuracy ~1e-16, which is what dynamics values need (parity tolerance 1e-8).
                (sa, ca), (sb, cb) = ensure_pair(arg.args[0]), ensure_pair(arg.args[1])
                plus = arg.op == E.ADD
                if n.fn == "sin":
                    rhs = f"{sa} * {cb} {'+' if plus else '-'} {ca} * {sb}"
                else:
                    rhs = f"{ca} * {cb} {'-' if plus else '+'} {sa} * {sb}"
                lines.append(f"{indent}const double {t} = {rhs};")
                name[n.id] = t
                continue
            if n.fn in ("sin", "cos") and aid in paired:
                ensure_pair(arg)
                name[n.id] = t
                continue
            rhs = f"{_CFN[n.fn]}({ref(n.args[0])})"
        lines.append(f"{indent}const double {t} = {rhs};")
        name[n.id] = t
    k0 = 0
    for nm, cnt in groups:
        for k in range(cnt):
            v = name[outputs[k0 + k].id]
            if scale is not None:
                v = f"{scale} * ({v})"
            lines.append(f"{indent}{nm}[{k}] = {v};")
        k0 += cnt
    return "\n".join(lines)


def op_count(outputs: Sequence[Expr]) -> Dict[str, int]:
    """Arithmetic census of the emitted body (for DESIGN.md roofline arithmetic)."""
    c = {"addsub": 0, "mul": 0, "div": 0, "trans": 0, "nodes": 0}
    for n in E.topo_order(outputs):
        if n.op in (E.ADD, E.SUB, E.IFELSE):
            c["addsub"] += 1
        elif n.op == E.MUL:
            c["mul"] += 1
        elif n.op == E.POWI:
            c["mul"] += max(1, n.value.bit_length())
        elif n.op == E.DIV:
            c["div"] += 1
        elif n.op in (E.FUNC, E.POW):
            c["trans"] += 1
        if n.op not in (E.CONST, E.VAR):
            c["nodes"] += 1
    return c
