"""Guard against the code-generation fault that produced the wrong-result modes of rounds 3-4 ("root
cause"): AMD clang 22 / ROCm 7.2 removes the exec-mask restore of an inner divergent `if` whose end coincides with the end of the
enclosing divergent region, and the register allocator may then put a reload into the merge block, where it runs with the inner
mask.  tools/check_exec_merge.py finds that shape in gfx950 ISA.

  * the scanner flags the committed excerpt of the faulty build (tests/golden/exec_merge_fault_excerpt.s: compiler output of
    k_wide_step for the 24-state / two-action embedding, the failing case of tests/test_wide_gpu.py) and accepts the same
    region with the restore in place;
  * every device build of the product carries the workaround (-mllvm -amdgpu-remove-redundant-endcf=0): the plugin cache
    key, the compile command, the runtime library;
  * the ISA of the product's plugins of both kernel families, compiled HERE with the product's flags (hipcc cross-compiles
    without a GPU), contains no unsaved exec narrowing at all.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _scanner():
    import check_exec_merge as C
    return C


def test_scanner_flags_the_faulty_build_and_accepts_the_restored_one():
    C = _scanner()
    with open(os.path.join(ROOT, "tests", "golden", "exec_merge_fault_excerpt.s")) as f:
        bad = f.read()
    hits = C.scan(bad)
    assert len(hits) == 1
    fn, line, label, instrs = hits[0]
    assert "k_wide_step" in fn and label == ".LBB10_282" and instrs[0][1] == "v_accvgpr_read_b32 v52, a34"
    # the same region as the compiler emits it with -amdgpu-remove-redundant-endcf=0: inner mask saved and restored
    good = bad.replace("\ts_and_b64 exec, exec, vcc\n", "\ts_and_saveexec_b64 s[2:3], vcc\n") \
              .replace(".LBB10_282:\n", ".LBB10_282:\n\ts_or_b64 exec, exec, s[2:3]\n.LBB10_282b:\n")
    assert good != bad and C.scan(good) == []
    # lane operations ignore exec: not an instance
    lane = bad.replace("v_accvgpr_read_b32 v52, a34", "v_readlane_b32 s36, v248, 16")
    assert C.scan(lane) == []


def test_every_device_build_carries_the_workaround():
    from dto_amd import plugin as PL, build as BL, problems as P
    assert PL.BASE_CXXFLAGS == ["-mllvm", "-amdgpu-remove-redundant-endcf=0"]
    p = P.build_pendulum(T=5, evaluate_hessian=True)
    st = PL.Structure(p["dynamics"], p["objective"], p["constraints"], None, True)
    os.environ["DTO_PLUGIN_CXXFLAGS"] = "-DDTO_GUARD_TEST_NEVER_BUILT=1"     # a key nobody has built: the command is returned
    try:
        so, cmd = PL._prepare_plugin(st, "pendulum")
    finally:
        del os.environ["DTO_PLUGIN_CXXFLAGS"]
    assert cmd is not None and "-amdgpu-remove-redundant-endcf=0" in cmd
    hip_src = cmd[-1]
    if os.path.exists(hip_src):
        os.remove(hip_src)
    import inspect
    assert "-amdgpu-remove-redundant-endcf=0" in inspect.getsource(BL.build_runtime)


@pytest.mark.parametrize("family", ["lane", "tile"])
def test_product_plugin_isa_has_no_unsaved_exec_narrowing(family):
    """acrobot (lane-per-instance sweeps: the round-3/4 fused-sweeps fault lived there) and the 24-state two-action embedding
    (tile path: the round-4/5 solver-mode fault), product flags."""
    from dto_amd import plugin as PL, problems as P
    C = _scanner()
    if family == "lane":
        p = P.build_acrobot(T=5, evaluate_hessian=True)
        st = PL.Structure(p["dynamics"], p["objective"], p["constraints"], None, True)
        name = "acrobot"
    else:
        import dto_amd
        p = P.build_acrobot_padded(T=4, n=24, m=2, target=0.4, terminal="physical", parameters=(1.2, 0.8))
        s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True,
                           parameters=p["parameters"], name="acrobot24u2")
        st = s._solve_nlp._structure if hasattr(s._solve_nlp, "_structure") else None
        name = "acrobot24u2"
    if st is not None:
        src = PL.generate_source(st, name)
        path = os.path.join(PL.PLUGIN_DIR, f"_guard_{family}.hip")
        os.makedirs(PL.PLUGIN_DIR, exist_ok=True)
        with open(path, "w") as f:
            f.write(src)
    else:   # the embedding's structure is not kept on the Solver: take the generated source of the plugin it loaded
        cands = sorted((os.path.getmtime(os.path.join(PL.PLUGIN_DIR, f)), f) for f in os.listdir(PL.PLUGIN_DIR)
                       if f.startswith("acrobot24u2_") and f.endswith(".hip"))
        path = None
        for _, f in reversed(cands):
            with open(os.path.join(PL.PLUGIN_DIR, f)) as fh:
                if "WIDE_N = 64, WIDE_NU = 2" in fh.read():
                    path = os.path.join(PL.PLUGIN_DIR, f)
                    break
        assert path, "no generated source of the 64-state embedding found"
    flags = PL.BASE_CXXFLAGS + (PL.WIDE_CXXFLAGS if family == "tile" else [])
    isa = C.compile_to_isa(path, flags)
    assert "k_wide_step" in isa if family == "tile" else "k_kkt_fwd_seq" in isa
    assert C.scan(isa) == []
    assert sum(1 for ln in isa.split("\n") if C.NARROW.match(ln)) == 0
    if family == "lane":
        os.remove(path)


def test_standalone_reproducer_of_the_dropped_exec_restore():
    """tools/micro/endcf_unsaved_narrowing.hip (65 lines, no project headers): with the compiler's defaults the innermost region's
    exec restore is dropped (one unsaved narrowing in the ISA -- the precondition of the fault), with the product's flag it is
    kept.  What an upstream report would carry next to tests/golden/exec_merge_fault_excerpt.s."""
    C = _scanner()
    src = os.path.join(ROOT, "tools", "micro", "endcf_unsaved_narrowing.hip")
    with open(src) as f:
        assert len(f.read().split("\n")) <= 150
    default = C.compile_to_isa(src, [])
    fixed = C.compile_to_isa(src, ["-mllvm", "-amdgpu-remove-redundant-endcf=0"])
    n_default = sum(1 for ln in default.split("\n") if C.NARROW.match(ln))
    n_fixed = sum(1 for ln in fixed.split("\n") if C.NARROW.match(ln))
    assert n_default >= 1 and n_fixed == 0, (n_default, n_fixed)
