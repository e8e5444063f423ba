"""Test helper: evaluate an Expr DAG with python floats (host-side check of the symbolic front end).

Lives under tests/ on purpose: the product package has no CPU evaluation path.
"""
from typing import Dict, List, Sequence, Tuple

import dto_amd  # noqa: F401
from dto_amd.symbolic import expr as E
from dto_amd.symbolic.expr import Expr, topo_order

_FOLD = E._FOLD
CONST, VAR, ADD, SUB, MUL, DIV, NEG, POWI, POW, FUNC = (E.CONST, E.VAR, E.ADD, E.SUB, E.MUL, E.DIV, E.NEG, E.POWI, E.POW, E.FUNC)


def evaluate(roots: Sequence[Expr], env: Dict[Tuple[str, int], float]) -> List[float]:
    """Numeric evaluation with python floats (used by host-side tests only)."""
    val: Dict[int, float] = {}
    for n in topo_order(roots):
        if n.op == CONST:
            v = n.value
        elif n.op == VAR:
            v = env[(n.name, n.index)]
        elif n.op == ADD:
            v = val[n.args[0].id] + val[n.args[1].id]
        elif n.op == SUB:
            v = val[n.args[0].id] - val[n.args[1].id]
        elif n.op == MUL:
            v = val[n.args[0].id] * val[n.args[1].id]
        elif n.op == DIV:
            v = val[n.args[0].id] / val[n.args[1].id]
        elif n.op == NEG:
            v = -val[n.args[0].id]
        elif n.op == POWI:
            v = val[n.args[0].id] ** n.value
        elif n.op == POW:
            v = val[n.args[0].id] ** val[n.args[1].id]
        elif n.op == E.IFELSE:
            l, r = val[n.args[0].id], val[n.args[1].id]
            v = val[n.args[2].id] if ((l < r) if n.fn == "lt" else (l <= r)) else val[n.args[3].id]
        else:
            v = _FOLD[n.fn](val[n.args[0].id])
        val[n.id] = v
    return [val[r.id] for r in roots]
