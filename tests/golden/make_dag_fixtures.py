"""Writes tests/golden/acrobot_T5_dag.json: the acrobot of the reference (examples/acrobot/acrobot.jl:19-118, T = 5) traced by
the Python front end and exported in the dto-dag-v1 exchange format (directtrajectoryoptimization.jl_amd/dagjson.py) -- the
file julia/emit_plugin.jl would write for the same closures.  python tests/golden/make_dag_fixtures.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dto_amd  # noqa: F401
from dto_amd import dagjson, problems as P

p = P.build_acrobot(T=5, evaluate_hessian=True)
doc = dagjson.export_problem(p["dynamics"], p["objective"], p["constraints"], p["bounds"], None, True, "acrobot")
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "acrobot_T5_dag.json"), "w") as f:
    json.dump(doc, f)
