"""The reference's default is evaluate_hessian=false (src/solver.jl:7): Ipopt then runs its limited-memory quasi-Newton Hessian.
The GPU solver differentiates the traced expressions twice instead -- it must SAY so ("exact Hessians silently
substituted"), report the mode, keep the MOI surface at [:Grad, :Jac], and offer the mode that evaluates no second
derivatives.  CPU only: nothing is launched."""
import warnings

import pytest

import dto_amd
from dto_amd import problems as P
from dto_amd import solver as S


def _build(**opts):
    p = P.build_pendulum(T=6, evaluate_hessian=False)
    return dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=False,
                          options=dto_amd.Options(**opts), name="pendulum")


def test_default_mode_is_announced_once_and_reported():
    S._NOTICED = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        s = _build()
        s2 = _build()
    notes = [x for x in w if issubclass(x.category, S.HessianModeNotice)]
    assert len(notes) == 1 and "sr1" in str(notes[0].message)
    assert s.hessian_mode == "exact-from-trace" and s2.hessian_mode == "exact-from-trace"
    assert s.nlp.features_available() == ["Grad", "Jac"]            # src/moi.jl:122 with hessian_lagrangian = false
    assert s._solve_nlp is not s.nlp                                  # the solver's own exact-Hessian clone


def test_explicit_modes():
    S._NOTICED = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert _build(hessian_approximation="exact").hessian_mode == "exact-from-trace"
        q = _build(hessian_approximation="sr1")
    assert not [x for x in w if issubclass(x.category, S.HessianModeNotice)]
    assert q.hessian_mode == "sr1" and q._solve_nlp is q.nlp          # no second derivatives anywhere
    with pytest.raises(ValueError):
        _build(hessian_approximation="lbfgs")
    p = P.build_pendulum(T=6, evaluate_hessian=True)
    e = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="pendulum")
    assert e.hessian_mode == "exact" and e.nlp.features_available() == ["Grad", "Jac", "Hess"]
