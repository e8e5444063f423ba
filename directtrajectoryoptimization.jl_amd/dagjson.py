"""Expression-DAG exchange file ("dto-dag-v1"): how a host language that traces the model closures itself hands the traced
expressions to the plugin generator.

In the reference the per-stage closures are traced by Symbolics.jl inside the constructors (src/dynamics.jl:23-36,
src/costs.jl:18-28, src/constraints.jl:27-41); nothing but those expressions defines a model.  `julia/emit_plugin.jl` walks
exactly those Symbolics expressions and writes this file; `python -m dto_amd.dagjson model.json` (or `load_problem`) rebuilds
the stage objects from it through the normal constructors (so folding, differentiation, sparsity and code generation are the
ones every other model goes through) and compiles the plugin.  `export_problem` writes the same file from objects traced in
Python -- that is how the committed fixture tests/golden/acrobot_T5_dag.json was made and what the round-trip test checks:
a problem rebuilt from its file produces the bit-identical plugin source.

File layout (JSON):
  {"format": "dto-dag-v1", "name": ..., "T": T, "evaluate_hessian": bool,
   "dynamics":    {"classes": [ {num_next_state, num_state, num_action, num_parameter, nodes, outputs} ], "stages": [T-1 class ids]},
   "objective":   {"classes": [ {num_state, num_action, num_parameter, nodes, outputs(1)} ],              "stages": [T class ids]},
   "constraints": {"classes": [ {num_state, num_action, num_parameter, indices_inequality(1-based), nodes, outputs} ],
                   "stages": [T class ids, -1 = Constraint()]},
   "bounds": [T x {state_lower, state_upper, action_lower, action_upper}]  (null = +-Inf),
   "parameters": [T x [..]]}
`nodes` is a topologically ordered list, a node refers to earlier nodes by index:
  {"op": "const", "value": v} | {"op": "var", "name": "x"|"u"|"y"|"w", "index": i (0-based)} |
  {"op": "add"|"mul", "args": [i, j, ...]} (n-ary) | {"op": "sub"|"div"|"pow", "args": [i, j]} | {"op": "neg", "args": [i]} |
  {"op": "call", "fn": "sin"|..., "args": [i]} | {"op": "ifelse", "cmp": "lt"|"le", "args": [lhs, rhs, then, else]}
"""
from __future__ import annotations

import json
from typing import Dict, List, Sequence

import numpy as np

from .model import Bound, Constraint, Cost, Dynamics
from .symbolic import expr as E

FORMAT = "dto-dag-v1"


# ------------------------------------------------------------------------------------------------ export
def _dump_exprs(outputs: Sequence[E.Expr]):
    nodes: List[dict] = []
    index: Dict[int, int] = {}

    def visit(e: E.Expr) -> int:
        if e.id in index:
            return index[e.id]
        args = [visit(a) for a in e.args]
        if e.op == E.CONST:
            n = {"op": "const", "value": float(e.value)}
        elif e.op == E.VAR:
            n = {"op": "var", "name": e.name, "index": int(e.index)}
        elif e.op in (E.ADD, E.SUB, E.MUL, E.DIV):
            n = {"op": {E.ADD: "add", E.SUB: "sub", E.MUL: "mul", E.DIV: "div"}[e.op], "args": args}
        elif e.op == E.NEG:
            n = {"op": "neg", "args": args}
        elif e.op == E.POWI:
            nodes.append({"op": "const", "value": float(e.value)})
            n = {"op": "pow", "args": [args[0], len(nodes) - 1]}
        elif e.op == E.POW:
            n = {"op": "pow", "args": args}
        elif e.op == E.FUNC:
            n = {"op": "call", "fn": e.fn, "args": args}
        elif e.op == E.IFELSE:
            n = {"op": "ifelse", "cmp": e.fn, "args": args}
        else:
            raise ValueError(f"unknown expression op {e.op}")
        nodes.append(n)
        index[e.id] = len(nodes) - 1
        return index[e.id]

    outs = [visit(E.as_expr(e)) for e in outputs]
    return nodes, outs


def _classes(objs):
    """(distinct objects in order of first appearance, class id per stage)"""
    cls, ids = [], []
    for o in objs:
        for i, c in enumerate(cls):
            if c is o:
                ids.append(i)
                break
        else:
            cls.append(o)
            ids.append(len(cls) - 1)
    return cls, ids


def _lim(v):
    return [None if not np.isfinite(x) else float(x) for x in np.asarray(v, dtype=float)]


def export_problem(dynamics, objective, constraints, bounds=None, parameters=None, evaluate_hessian=True, name="model") -> dict:
    dcls, dst = _classes(dynamics)
    ocls, ost = _classes(objective)
    nonempty = [c for c in constraints if c.num_constraint > 0]
    ccls, _ = _classes(nonempty)
    cst = [(-1 if c.num_constraint == 0 else [i for i, o in enumerate(ccls) if o is c][0]) for c in constraints]
    out = {"format": FORMAT, "name": name, "T": len(objective), "evaluate_hessian": bool(evaluate_hessian)}
    dd = []
    for d in dcls:
        if getattr(d, "user_jacobian", False):
            raise ValueError("user-Jacobian dynamics have no expression to export")
        nodes, outs = _dump_exprs(d.evaluate_expr)
        dd.append(dict(num_next_state=d.num_next_state, num_state=d.num_state, num_action=d.num_action,
                       num_parameter=d.num_parameter, nodes=nodes, outputs=outs))
    out["dynamics"] = {"classes": dd, "stages": dst}
    oo = []
    for c in ocls:
        nodes, outs = _dump_exprs(c.evaluate_expr)
        oo.append(dict(num_state=c.num_state, num_action=c.num_action, num_parameter=c.num_parameter, nodes=nodes, outputs=outs))
    out["objective"] = {"classes": oo, "stages": ost}
    cc = []
    for c in ccls:
        nodes, outs = _dump_exprs(c.evaluate_expr)
        cc.append(dict(num_state=c.num_state, num_action=c.num_action, num_parameter=c.num_parameter,
                       indices_inequality=[int(i) for i in c.indices_inequality], nodes=nodes, outputs=outs))
    out["constraints"] = {"classes": cc, "stages": cst}
    if bounds is not None:
        out["bounds"] = [dict(state_lower=_lim(b.state_lower), state_upper=_lim(b.state_upper),
                              action_lower=_lim(b.action_lower), action_upper=_lim(b.action_upper)) for b in bounds]
    if parameters is not None:
        out["parameters"] = [[float(v) for v in np.asarray(p, dtype=float).ravel()] for p in parameters]
    return out


# ------------------------------------------------------------------------------------------------ import
def _build_exprs(nodes, outputs, dims):
    """Rebuild the expressions through the public constructors of symbolic.expr (same folding as a traced closure)."""
    vars_ = {k: E.variables(k, n) for k, n in dims.items()}
    built: List[object] = []
    for n in nodes:
        op = n["op"]
        a = [built[i] for i in n.get("args", [])]
        if op == "const":
            v = E.const(float(n["value"]))
        elif op == "var":
            v = vars_[n["name"]][int(n["index"])]
        elif op == "add":
            v = a[0]
            for t in a[1:]:
                v = v + t
        elif op == "mul":
            v = a[0]
            for t in a[1:]:
                v = v * t
        elif op == "sub":
            v = a[0] - a[1]
        elif op == "div":
            v = a[0] / a[1]
        elif op == "neg":
            v = -a[0]
        elif op == "pow":
            e = a[1]
            if e.op == E.CONST and float(e.value).is_integer() and abs(e.value) <= 64:
                v = a[0] ** int(e.value)
            else:
                v = E.power(a[0], e)
        elif op == "call":
            v = E.func(n["fn"], a[0])
        elif op == "ifelse":
            v = E.ifelse(E.Cond(n["cmp"], a[0], a[1]), a[2], a[3])
        else:
            raise ValueError(f"dto-dag: unknown node op {op!r}")
        built.append(E.as_expr(v))
    return [built[i] for i in outputs]


def load_problem(doc) -> dict:
    """dict(dynamics, objective, constraints, bounds, parameters, evaluate_hessian, T, name) from a dto-dag-v1 document
    (a dict or a path)."""
    if isinstance(doc, str):
        with open(doc) as f:
            doc = json.load(f)
    if doc.get("format") != FORMAT:
        raise ValueError(f"not a {FORMAT} document")
    T, h = int(doc["T"]), bool(doc["evaluate_hessian"])
    dcl = []
    for d in doc["dynamics"]["classes"]:
        ny, nx, nu, nw = d["num_next_state"], d["num_state"], d["num_action"], d.get("num_parameter", 0)
        ev = _build_exprs(d["nodes"], d["outputs"], dict(y=ny, x=nx, u=nu, w=nw))
        dcl.append(Dynamics(ev, ny, nx, nu, num_parameter=nw, evaluate_hessian=h))
    ocl = []
    for c in doc["objective"]["classes"]:
        nx, nu, nw = c["num_state"], c["num_action"], c.get("num_parameter", 0)
        ev = _build_exprs(c["nodes"], c["outputs"], dict(x=nx, u=nu, w=nw))
        ocl.append(Cost(ev, nx, nu, num_parameter=nw, evaluate_hessian=h))
    ccl = []
    for c in doc["constraints"]["classes"]:
        nx, nu, nw = c["num_state"], c["num_action"], c.get("num_parameter", 0)
        ev = _build_exprs(c["nodes"], c["outputs"], dict(x=nx, u=nu, w=nw))
        ccl.append(Constraint(ev, nx, nu, num_parameter=nw, indices_inequality=c.get("indices_inequality", []), evaluate_hessian=h))
    dyn = [dcl[i] for i in doc["dynamics"]["stages"]]
    obj = [ocl[i] for i in doc["objective"]["stages"]]
    empty = Constraint()
    con = [empty if i < 0 else ccl[i] for i in doc["constraints"]["stages"]]
    if len(dyn) != T - 1 or len(obj) != T or len(con) != T:
        raise ValueError("dto-dag: stage lists do not match T")
    inf = float("inf")
    un = lambda v, s: np.array([(s * inf if x is None else x) for x in v], dtype=float)
    bounds = None
    if doc.get("bounds") is not None:
        bounds = [Bound(len(b["state_lower"]), len(b["action_lower"]), un(b["state_lower"], -1), un(b["state_upper"], 1),
                        un(b["action_lower"], -1), un(b["action_upper"], 1)) for b in doc["bounds"]]
    params = [np.array(p, dtype=float) for p in doc["parameters"]] if doc.get("parameters") is not None else None
    return dict(dynamics=dyn, objective=obj, constraints=con, bounds=bounds, parameters=params, evaluate_hessian=h, T=T,
                name=doc.get("name", "model"))


def build_plugin_from_file(path: str, verbose: bool = False) -> str:
    """Compile (or find in the cache) the gfx950 plugin of the model in `path`; returns the .so path."""
    from .plugin import Structure, build_plugin
    p = load_problem(path)
    return build_plugin(Structure(p["dynamics"], p["objective"], p["constraints"], None, p["evaluate_hessian"]), p["name"], verbose=verbose)


if __name__ == "__main__":
    import sys
    if len(sys.argv) != 2:
        sys.exit("usage: python -m dto_amd.dagjson model.json   (writes the plugin into the package's _plugins/ cache)")
    print(build_plugin_from_file(sys.argv[1], verbose=True))
