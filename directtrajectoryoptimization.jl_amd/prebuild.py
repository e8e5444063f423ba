"""Build every model plugin the tests, smoke() and bench.py use, so they travel prebuilt to the GPU box."""
from __future__ import annotations

from . import problems as P
from .plugin import Structure, build_plugin, build_plugins


def _entries():
    return [
        ("pendulum", P.build_pendulum, dict(T=6, evaluate_hessian=True)),
        ("pendulum", P.build_pendulum, dict(T=2, evaluate_hessian=True)),
        ("cartpole", P.build_cartpole, dict(T=5, evaluate_hessian=True)),
        ("acrobot", P.build_acrobot, dict(T=5, evaluate_hessian=True)),
        ("acrobot_bounds", P.build_acrobot, dict(T=4, evaluate_hessian=True, endpoint="bounds")),
        ("car", P.build_car, dict(T=6, evaluate_hessian=True)),
        ("car", P.build_car, dict(T=6, evaluate_hessian=False)),
        ("cartpole", P.build_cartpole, dict(T=5, evaluate_hessian=False)),
        ("pendulum", P.build_pendulum, dict(T=6, evaluate_hessian=False)),
        ("ref_objective", P.build_ref_objective, {}),
        ("ref_dynamics", P.build_ref_dynamics, {}),
        ("ref_constraints", P.build_ref_constraints, {}),
        ("ref_hesslag", P.build_ref_hesslag, {}),
        ("ref_general", P.build_ref_general, dict(user_jacobian=False)),
        ("ref_general", P.build_ref_general, dict(user_jacobian=True)),
        ("ref_general_coupled", P.build_ref_general_coupled, {}),
        ("acrobot_coupled", P.build_acrobot_coupled, dict(T=8)),
        ("param_pendulum", P.build_param_pendulum, dict(T=8)),
        ("ref_userjac", P.build_ref_userjac, {}),
        ("mpc_pendulum", P.build_mpc_pendulum, dict(T=5)),
        ("acrobot_padded", P.build_acrobot_padded, dict(T=3)),
        ("acrobot_padded", P.build_acrobot_padded, dict(T=2)),
        ("acrobot_padded", P.build_acrobot_padded, dict(T=3, evaluate_hessian=False)),
        ("acrobot", P.build_acrobot, dict(T=5, evaluate_hessian=False)),
    ]


def all_structures():
    out = []
    for name, builder, kw in _entries():
        p = builder(**kw)
        out.append((name, Structure(p["dynamics"], p["objective"], p["constraints"], p.get("general_constraint"),
                                    p["evaluate_hessian"])))
        if p.get("general_constraint") is not None:
            # the solver's internal form of a stage-local general constraint (solver.py:fold_general_constraint)
            from .solver import fold_general_constraint
            folded = fold_general_constraint(p["dynamics"], p["objective"], p["constraints"], p["general_constraint"],
                                             p["evaluate_hessian"])
            if folded is not None:
                out.append((name + "_folded", Structure(p["dynamics"], p["objective"], folded[0], None, p["evaluate_hessian"])))
    return out


def build_all(verbose: bool = False):
    items = list(all_structures())
    # the solver-internal exact-Hessian clones of the evaluate_hessian=false models are compiled in the same parallel batch
    from .solver import _with_exact_hessians, fold_general_constraint
    for name, builder, kw in _entries():
        p = builder(**kw)
        if p["evaluate_hessian"] or p.get("general_constraint") is not None:
            continue
        up = _with_exact_hessians(list(p["dynamics"]), list(p["objective"]), list(p["constraints"]))
        if up is not None:
            items.append((name, Structure(up[0], up[1], up[2], None, True)))
    paths = build_plugins(items, verbose=verbose)
    # the solver-internal forms (exact-Hessian clones of evaluate_hessian=false models, solver.py:_with_exact_hessians):
    # constructing the Solver builds whatever it will load; no device is needed for that
    from .solver import Solver
    for name, builder, kw in _entries():
        p = builder(**kw)
        if p["evaluate_hessian"]:
            continue
        try:
            Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=False,
                   general_constraint=p.get("general_constraint"), parameters=p.get("parameters"), name=name)
        except Exception as e:  # a model the solver does not take (callbacks only) still has its callback plugin above
            if verbose:
                print(f"prebuild: no solver form for {name}: {e}")
    return paths
