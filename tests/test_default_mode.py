"""The reference's default is evaluate_hessian=false (src/solver.jl:7): Ipopt then runs its limited-memory quasi-Newton Hessian.
Since round 5 the GPU solver does the same wherever its lane-per-instance path runs the problem (compact L-BFGS, history 6:
dto_options.hessian_approximation = DTO_HESSIAN_LBFGS); where it cannot (the tile path: more than 16 states) it differentiates the
traced expressions twice instead and must SAY so ("exact Hessians silently substituted").  The mode is reported
(Solver.hessian_mode), the MOI surface stays at [:Grad, :Jac], the other modes can be asked for.  CPU only: nothing is launched."""
import warnings

import pytest

import dto_amd
from dto_amd import capi
from dto_amd import problems as P
from dto_amd import solver as S


def _build(**opts):
    p = P.build_pendulum(T=6, evaluate_hessian=False)
    return dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=False,
                          options=dto_amd.Options(**opts), name="pendulum")


def test_default_mode_is_limited_memory_bfgs_without_a_notice():
    S._NOTICED = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        s = _build()
    assert not [x for x in w if issubclass(x.category, S.HessianModeNotice)]
    assert s.hessian_mode == "lbfgs"
    assert s.nlp.features_available() == ["Grad", "Jac"]            # src/moi.jl:122 with hessian_lagrangian = false
    co = S._c_options(s.options, lbfgs=s.hessian_mode == "lbfgs")
    assert co.hessian_approximation == capi.DTO_HESSIAN_LBFGS and co.line_search == capi.DTO_LS_PENALTY_FILTER
    d = capi.COptions()
    capi.check(capi.lib().dto_options_default(d))
    assert d.hessian_approximation == capi.DTO_HESSIAN_EXACT and d.penalty_switch_theta == 1.0


def test_tile_path_problems_keep_the_announced_substitute():
    S._NOTICED = False
    p = P.build_acrobot_padded(T=3, evaluate_hessian=False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=False, name="acrobot_padded")
        s2 = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=False, name="acrobot_padded")
    notes = [x for x in w if issubclass(x.category, S.HessianModeNotice)]
    assert len(notes) == 1 and "more than 16 states" in str(notes[0].message) and "'exact'" in str(notes[0].message)
    assert s.hessian_mode == "exact-from-trace" and s2.hessian_mode == "exact-from-trace"
    with pytest.raises(ValueError):
        dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=False,
                       options=dto_amd.Options(hessian_approximation="lbfgs"), name="acrobot_padded")


def test_explicit_modes():
    S._NOTICED = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert _build(hessian_approximation="exact").hessian_mode == "exact-from-trace"
        assert _build(hessian_approximation="lbfgs").hessian_mode == "lbfgs"
        q = _build(hessian_approximation="sr1")
    assert not [x for x in w if issubclass(x.category, S.HessianModeNotice)]
    assert q.hessian_mode == "sr1" and q._solve_nlp is q.nlp          # no second derivatives anywhere
    with pytest.raises(ValueError):
        _build(hessian_approximation="bfgs")
    p = P.build_pendulum(T=6, evaluate_hessian=True)
    e = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="pendulum")
    assert e.hessian_mode == "exact" and e.nlp.features_available() == ["Grad", "Jac", "Hess"]
    el = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True,
                        options=dto_amd.Options(hessian_approximation="lbfgs"), name="pendulum")
    assert el.hessian_mode == "lbfgs"
