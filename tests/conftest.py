import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter):
    """How many model plugins this session had to compile: 0 on a box that received the prebuilt ones
    (python __graft_entry__.py builds them all; a nonzero count on the GPU box means prebuild.py misses a model)."""
    try:
        from dto_amd import plugin
        terminalreporter.write_line(f"[dto] model plugins compiled in this session: {len(plugin.COMPILED)} {plugin.COMPILED}")
    except Exception:
        pass


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


_PRODUCT_CACHE = {}


def product_solver(model, T, evaluate_hessian=True):
    """Build (once per session) the product-side Solver of a BASELINE model through the public API."""
    import dto_amd
    from dto_amd import problems as P
    key = (model, T, evaluate_hessian)
    if key not in _PRODUCT_CACHE:
        if model == "ref_general":
            p = P.build_ref_general(user_jacobian=not evaluate_hessian)
        elif model == "param_pendulum":
            p = P.build_param_pendulum(T)
        elif model.startswith("ref_"):
            p = getattr(P, f"build_{model}")()
            assert p["T"] == T and p["evaluate_hessian"] == evaluate_hessian
        elif model == "acrobot_bounds":
            p = P.build_acrobot(T=T, evaluate_hessian=evaluate_hessian, endpoint="bounds")
        else:
            p = getattr(P, f"build_{model}")(T=T, evaluate_hessian=evaluate_hessian)
        s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"],
                           evaluate_hessian=evaluate_hessian, general_constraint=p.get("general_constraint"),
                           parameters=p.get("parameters"), name=model)
        _PRODUCT_CACHE[key] = (s, p)
    return _PRODUCT_CACHE[key]


@pytest.fixture(scope="session")
def golden_cases():
    return ["pendulum_T6.json", "cartpole_T5.json", "acrobot_T5.json", "car_T6.json", "acrobot_bounds_T4.json",
            "acrobot_T70.json"]
