"""Generate tests/golden/*.json with the ORACLE (oracle/sympy_models.py + oracle/dto_oracle.py).

The reference ships no data files and cannot run here (Julia absent), so the vectors are produced by
the independent sympy derivation evaluated with mpmath at 30 digits and rounded once to float64.
Run from the repo root:  python tests/golden/make_golden.py
Inputs are seeded (numpy PCG64), U(0,1) like the reference tests draw them
(test/hessian_lagrangian.jl:167); everything needed to replay a case is stored in the fixture.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import dto_oracle as O  # noqa: E402
from oracle import sympy_models as S  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def f64(v):
    return [float(x) for x in v]


def structure_digest(pairs):
    h = hashlib.sha256()
    h.update(np.asarray(pairs, dtype=np.int64).tobytes())
    return h.hexdigest()


def case(name, T, seed):
    p = S.build(name, T, evaluate_hessian=True)
    nlp = O.NLPData(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True)
    rng = np.random.Generator(np.random.PCG64(seed))
    z = rng.random(nlp.num_variables)
    mu = rng.random(nlp.num_constraint)
    sigma = 0.37
    d = p["dynamics"][0]
    out = dict(
        model=name, T=T, seed=seed, z=f64(z), mu=f64(mu), sigma=sigma,
        num_variables=nlp.num_variables, num_constraint=nlp.num_constraint, num_jacobian=nlp.num_jacobian,
        num_hessian_lagrangian_raw=nlp.num_hessian_lagrangian, num_hessian_key=len(nlp.hessian_lagrangian_sparsity),
        dynamics_jacobian_sparsity=[list(map(int, d.jacobian_sparsity[0])), list(map(int, d.jacobian_sparsity[1]))],
        dynamics_hessian_sparsity=[list(map(int, d.hessian_sparsity[0])), list(map(int, d.hessian_sparsity[1]))],
        jacobian_structure=[list(map(int, rc)) for rc in nlp.jacobian_structure()],
        hessian_structure=[list(map(int, rc)) for rc in nlp.hessian_lagrangian_structure()],
        variable_lower=[float(v) if np.isfinite(v) else (None if v < 0 else "inf") for v in nlp.variable_bounds[0]],
        variable_upper=[float(v) if np.isfinite(v) else ("inf" if v > 0 else None) for v in nlp.variable_bounds[1]],
        constraint_lower_is_minus_inf=[bool(np.isneginf(v)) for v in nlp.constraint_bounds[0]],
        idx_states=nlp.idx_states, idx_actions=nlp.idx_actions,
        idx_dynamics_hessians=nlp.idx_dynamics_hessians, idx_objective_hessians=nlp.idx_objective_hessians,
        idx_stage_hessians=nlp.idx_stage_hessians,
        objective=float(nlp.eval_objective(z, hp=True)),
        gradient=f64(nlp.eval_objective_gradient(z, hp=True)),
        constraint=f64(nlp.eval_constraint(z, hp=True)),
        jacobian=f64(nlp.eval_constraint_jacobian(z, hp=True)),
        hessian_sigma=f64(nlp.eval_hessian_lagrangian(z, sigma, mu, hp=True)),
        hessian_one=f64(nlp.eval_hessian_lagrangian(z, 1.0, mu, hp=True)),
    )
    # float64 oracle against its own 30-digit evaluation (sanity of the fixture)
    err = np.max(np.abs(np.array(out["jacobian"]) - nlp.eval_constraint_jacobian(z)))
    assert err < 1e-10, err
    return out


def full_size(name, T):
    """Sizes + structure digests at the BASELINE sizes (too large to store entry by entry)."""
    p = S.build(name, T, evaluate_hessian=True)
    nlp = O.NLPData(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True)
    return dict(model=name, T=T, num_variables=nlp.num_variables, num_constraint=nlp.num_constraint,
                num_jacobian=nlp.num_jacobian, num_hessian_lagrangian_raw=nlp.num_hessian_lagrangian,
                num_hessian_key=len(nlp.hessian_lagrangian_sparsity),
                jacobian_structure_sha256=structure_digest(nlp.jacobian_structure()),
                hessian_structure_sha256=structure_digest(nlp.hessian_lagrangian_structure()))


def main():
    small = [("pendulum", 6, 1), ("cartpole", 5, 2), ("acrobot", 5, 3), ("car", 6, 4), ("acrobot_bounds", 4, 5),
             ("acrobot", 70, 6)]
    for name, T, seed in small:
        fx = case(name, T, seed)
        with open(os.path.join(OUT, f"{name}_T{T}.json"), "w") as f:
            json.dump(fx, f)
        print("wrote", name, T)
    sizes = [full_size(n, T) for n, T in [("pendulum", 50), ("cartpole", 200), ("acrobot", 1000), ("car", 500),
                                          ("acrobot", 2000), ("acrobot_bounds", 101)]]
    with open(os.path.join(OUT, "full_size_structure.json"), "w") as f:
        json.dump(sizes, f, indent=1)
    print("wrote full_size_structure.json")


if __name__ == "__main__":
    main()
