"""The expression-DAG exchange file (dto-dag-v1) that a host language tracing the closures itself -- julia/emit_plugin.jl --
hands to the plugin generator ("Julia can build its own plugin").  CPU tests of the consumer."""
import hashlib
import json
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN

import dto_amd  # noqa: F401
from dto_amd import dagjson, problems as P
from dto_amd.plugin import Structure, generate_source
from _dag_eval import evaluate


def _source(p, name):
    return generate_source(Structure(p["dynamics"], p["objective"], p["constraints"], None, p["evaluate_hessian"]), name)


def test_committed_acrobot_file_gives_the_traced_plugin_bit_for_bit():
    """tests/golden/acrobot_T5_dag.json -> stage objects -> plugin source: identical (SHA-256) to the source generated from the
    closures traced in this process, i.e. the file carries everything the generator needs."""
    q = dagjson.load_problem(os.path.join(GOLDEN, "acrobot_T5_dag.json"))
    p = P.build_acrobot(T=5, evaluate_hessian=True)
    a, b = _source(p, "acrobot"), _source(q, "acrobot")
    assert hashlib.sha256(a.encode()).hexdigest() == hashlib.sha256(b.encode()).hexdigest()
    assert q["T"] == 5 and len(q["bounds"]) == 5 and q["evaluate_hessian"]


@pytest.mark.parametrize("name,builder,kw", [("pendulum", P.build_pendulum, dict(T=6)), ("cartpole", P.build_cartpole, dict(T=5)),
                                             ("car", P.build_car, dict(T=6))])
def test_export_import_round_trip(name, builder, kw):
    """Bounds, inequality rows and per-class structure survive the file; the rebuilt problem generates the same source."""
    p = builder(evaluate_hessian=True, **kw)
    doc = json.loads(json.dumps(dagjson.export_problem(p["dynamics"], p["objective"], p["constraints"], p["bounds"], None, True, name)))
    q = dagjson.load_problem(doc)
    assert _source(p, name) == _source(q, name)
    for b0, b1 in zip(p["bounds"], q["bounds"]):
        assert np.array_equal(b0.state_lower, b1.state_lower) and np.array_equal(b0.action_upper, b1.action_upper)
    assert [c.indices_inequality for c in p["constraints"]] == [c.indices_inequality for c in q["constraints"]]


def test_symbolics_style_expressions_are_accepted():
    """What Symbolics.jl actually prints differs in shape from a Python trace: n-ary + and *, powers with constant exponents
    (x^2, x^-1, x^0.5), unary minus as (-1)*x.  A hand-written class in that style must evaluate like the formula it encodes."""
    # d(y, x, u) = y - (x + 0.05 * [x2, u - 9.81 * sin(x1) / 0.5 - 0.1 * x2^2 + sqrt-free x1^-1 + x2^0.5])   (2 states, 1 action)
    nodes = [
        {"op": "var", "name": "x", "index": 0}, {"op": "var", "name": "x", "index": 1},      # 0, 1
        {"op": "var", "name": "u", "index": 0}, {"op": "var", "name": "y", "index": 0},      # 2, 3
        {"op": "var", "name": "y", "index": 1},                                              # 4
        {"op": "const", "value": 0.05}, {"op": "const", "value": -1.0}, {"op": "const", "value": 2.0},   # 5, 6, 7
        {"op": "const", "value": 0.5}, {"op": "const", "value": -19.62}, {"op": "const", "value": -0.1},  # 8, 9, 10
        {"op": "call", "fn": "sin", "args": [0]},                                            # 11
        {"op": "mul", "args": [9, 11]},                                                      # 12: -19.62 sin(x1)
        {"op": "pow", "args": [1, 7]},                                                       # 13: x2^2
        {"op": "mul", "args": [10, 13]},                                                     # 14
        {"op": "pow", "args": [0, 6]},                                                       # 15: x1^-1
        {"op": "pow", "args": [1, 8]},                                                       # 16: x2^0.5
        {"op": "add", "args": [2, 12, 14, 15, 16]},                                          # 17: n-ary sum
        {"op": "mul", "args": [5, 1]},                                                       # 18: 0.05 x2
        {"op": "mul", "args": [5, 17]},                                                      # 19
        {"op": "mul", "args": [6, 0]}, {"op": "mul", "args": [6, 1]},                        # 20, 21: -x1, -x2 as (-1)*x
        {"op": "mul", "args": [6, 18]}, {"op": "mul", "args": [6, 19]},                      # 22, 23
        {"op": "add", "args": [3, 20, 22]}, {"op": "add", "args": [4, 21, 23]},              # 24, 25
    ]
    doc = {"format": "dto-dag-v1", "name": "toy", "T": 2, "evaluate_hessian": True,
           "dynamics": {"classes": [dict(num_next_state=2, num_state=2, num_action=1, num_parameter=0, nodes=nodes, outputs=[24, 25])], "stages": [0]},
           "objective": {"classes": [dict(num_state=2, num_action=1, num_parameter=0,
                                          nodes=[{"op": "var", "name": "u", "index": 0}, {"op": "const", "value": 2.0}, {"op": "pow", "args": [0, 1]}], outputs=[2]),
                                     dict(num_state=2, num_action=0, num_parameter=0,
                                          nodes=[{"op": "var", "name": "x", "index": 0}, {"op": "const", "value": 2.0}, {"op": "pow", "args": [0, 1]}], outputs=[2])],
                         "stages": [0, 1]},
           "constraints": {"classes": [], "stages": [-1, -1]}}
    q = dagjson.load_problem(doc)
    d = q["dynamics"][0]
    x1, x2, u, y1, y2 = 0.7, 1.3, -0.4, 0.2, 0.9
    env = {("x", 0): x1, ("x", 1): x2, ("u", 0): u, ("y", 0): y1, ("y", 1): y2}
    got = evaluate(d.evaluate_expr, env)
    acc = u - 19.62 * math.sin(x1) - 0.1 * x2 ** 2 + 1.0 / x1 + math.sqrt(x2)
    assert abs(got[0] - (y1 - x1 - 0.05 * x2)) <= 1e-15 and abs(got[1] - (y2 - x2 - 0.05 * acc)) <= 1e-15
    # the constructors differentiated it: Jacobian pattern = occurrence (row 1: x1, x2, y1; row 2: x1, x2, u, y2)
    assert sorted(zip(*d.jacobian_sparsity)) == sorted([(1, 1), (1, 2), (1, 4), (2, 1), (2, 2), (2, 3), (2, 5)])
    assert generate_source(Structure(q["dynamics"], q["objective"], q["constraints"], None, True), "toy")   # emits


def test_unknown_format_or_op_is_rejected():
    with pytest.raises(ValueError, match="dto-dag-v1"):
        dagjson.load_problem({"format": "something"})
    doc = json.load(open(os.path.join(GOLDEN, "acrobot_T5_dag.json")))
    doc["dynamics"]["classes"][0]["nodes"].append({"op": "besselj", "args": [0]})
    doc["dynamics"]["classes"][0]["outputs"][0] = len(doc["dynamics"]["classes"][0]["nodes"]) - 1
    with pytest.raises(ValueError, match="unknown node op"):
        dagjson.load_problem(doc)
