"""Debug harness for the solver-mode use of k_wide_step (NaN steps at -O3 under semantics-preserving build
changes).  Builds the 64-state plugin under several flag sets, runs the same solver-mode solve with each (own process: the
flag set is part of the plugin cache key), dumps everything the first launches produce (DTO_WIDE_DUMP: factor records, step,
statistics, flags) and reports, per variant, the first (launch, instance, stage, record field) that differs from the reference
variant.

    python tools/wide_debug.py prebuild            # here (no GPU): compile every variant, 8 at a time
    python tools/wide_debug.py all [--out DIR]     # on the GPU box: run + diff every variant, summary under gpurun_out/
    python tools/wide_debug.py run NAME            # one variant (DTO_PLUGIN_CXXFLAGS already in the environment)
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {
    # name: extra compiler flags of the plugin
    "stamps_in": "-DDTO_WIDE_PROFILE=1",     # the product's shape until round 4 (all solver tests passed): the reference of the diffs
    "prod": "",
    "sync": "-DDTO_WIDE_DBG_SYNC=1",
    "nsa": "-fno-strict-aliasing",
    "nsa_sync": "-fno-strict-aliasing -DDTO_WIDE_DBG_SYNC=1",
    "nsa_poison": "-fno-strict-aliasing -DDTO_WIDE_DBG_POISON=1",
    "poison": "-DDTO_WIDE_DBG_POISON=1",
    "O1": "-O1",
    "O1_jitter": "-O1 -DDTO_WIDE_DBG_JITTER=1",
    "jitter": "-DDTO_WIDE_DBG_JITTER=1",
    "nsa_O1": "-fno-strict-aliasing -O1",
    "nsa_ldl_inline": "-fno-strict-aliasing -DDTO_WIDE_LDL_INLINE=",
    "nsa_sgpr_mem": "-fno-strict-aliasing -mllvm -amdgpu-spill-sgpr-to-vgpr=0",
    "stamps_in_nsa": "-DDTO_WIDE_PROFILE=1 -fno-strict-aliasing",
}
DUMP_DIR = "/tmp/wide_dbg"
CASES = {
    # name: (T, B, NU): pad64 = tests/test_wide_gpu.py::test_wide_solve_converges_to_a_kkt_point[24-0.3-physical];
    # emb24u2 = the first solve of ::test_24_state_two_action_parametric_problem_through_the_embedding (24 states, two actions,
    # parameters; every knot has fixed -- padding -- states)
    "pad64": (24, 3, 1),
    "emb24u2": (25, 1, 2),
}


def _problem(case):
    import dto_amd
    from dto_amd import problems as P
    T, B, NU = CASES[case]
    if case == "pad64":
        p = P.build_acrobot_padded(T=T, target=0.3, terminal="physical")
        s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot_padded")
    else:
        p = P.build_acrobot_padded(T=T, n=24, m=2, target=0.4, terminal="physical", parameters=(1.2, 0.8))
        s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True,
                           parameters=p["parameters"], name="acrobot24u2")
    return s, p


def prebuild(names, case):
    """every variant's plugins, eight compiler processes at a time (constructing the Solver compiles what is missing)"""
    from concurrent.futures import ThreadPoolExecutor

    def one(name):
        env = dict(os.environ, DTO_PLUGIN_CXXFLAGS=VARIANTS[name])
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "build1", name, "--case", case], env=env, capture_output=True, text=True)
        return name, r.returncode, (r.stdout + r.stderr)[-300:].replace("\n", " | ")
    with ThreadPoolExecutor(max_workers=8) as ex:
        for name, rc, tail in ex.map(one, names):
            print(f"{name:16s} rc={rc} {tail if rc else ''}", flush=True)


def run(name, case):
    import torch
    import dto_amd
    from dto_amd.plugin import COMPILED
    T, B, NU = CASES[case]
    os.makedirs(DUMP_DIR, exist_ok=True)
    os.environ["DTO_WIDE_DUMP"] = os.path.join(DUMP_DIR, f"{case}_{name}")
    os.environ.setdefault("DTO_WIDE_DUMP_MAX", "4")
    s, p = _problem(case)
    if case == "pad64":
        nz, nc = s.nlp.num_variables, s.nlp.num_constraint
        Z = np.zeros((B, nz))
        for b in range(B):
            xs, us = p["guess"](np.random.Generator(np.random.PCG64(b)))
            dto_amd.initialize_states(s, xs)
            dto_amd.initialize_controls(s, [0.1 * u for u in us])
            Z[b] = s._z0
        z0 = torch.tensor(Z, device="cuda")
        zo = torch.full((B, nz), float("nan"), device="cuda", dtype=torch.float64)
        lo = torch.full((B, nc), float("nan"), device="cuda", dtype=torch.float64)
        status, iters = s.solve_batch(z0.data_ptr(), B, nz, zo.data_ptr(), nz, lo.data_ptr(), nc)
        torch.cuda.synchronize()
        status, iters, znan = status.tolist(), iters.tolist(), int(torch.isnan(zo).sum())
    else:
        xs, us = p["guess"](np.random.Generator(np.random.PCG64(0)))
        dto_amd.initialize_states(s, xs); dto_amd.initialize_controls(s, [0.1 * u for u in us])
        st = dto_amd.solve(s)
        status, iters, znan = [int(st)], [int(s.iterations)], int(np.isnan(s._solution).sum())
    print(json.dumps(dict(variant=name, case=case, flags=VARIANTS.get(name, "?"), status=status, iterations=iters, z_nan=znan,
                          compiled_here=list(COMPILED))), flush=True)


# ---- factor record layout (csrc/dto_wide_kernels.hpp: Dims<64, 1>)
N, LD = 64, 65
MAT, PKL = 64 * 65, 10 * 256
FIELDS = [("L_A(tiles)", 0, PKL, False), ("F~", PKL, MAT, True), ("V~", PKL + MAT, MAT, True), ("L_M(tiles)", PKL + 2 * MAT, PKL, False),
          ("E~", 2 * PKL + 2 * MAT, MAT, True)]
FV = 2 * PKL + 3 * MAT


def vecs(NU):
    v_au = 4 * N; v_fu = v_au + NU * N; v_vu = v_fu + NU * N; v_gc = v_vu + NU * N; v_sc = v_gc + N + 8
    return [("1/D_A", 0, 64), ("1/D_M", 64, 64), ("bx~", 128, 64), ("bd^", 192, 64), ("A_xu", v_au, NU * N), ("F_u", v_fu, NU * N),
            ("V_u", v_vu, NU * N), ("grad cost", v_gc, N + NU), ("1/piv_u", v_sc, NU), ("bu", v_sc + NU, NU)] + \
           ([("L_u", v_sc + 2 * NU, NU * NU)] if NU > 1 else [])


def _cmp(a, b, tol=1e-9):
    """indices where a and b differ: NaN pattern, or relative difference above tol"""
    na, nb = np.isnan(a), np.isnan(b)
    sc = np.maximum(np.maximum(np.abs(np.nan_to_num(a)), np.abs(np.nan_to_num(b))), 1e-30)
    bad = (na != nb) | (~na & ~nb & (np.abs(np.nan_to_num(a) - np.nan_to_num(b)) > tol * np.maximum(sc, np.max(sc) * 1e-6)))
    return np.nonzero(bad)[0]


def diff(ref, name, out, case):
    T, B, NU = CASES[case]
    VECS = vecs(NU)

    def load(v, k, what, dt=np.float64):
        fn = os.path.join(DUMP_DIR, f"{case}_{v}_L{k}_{what}.bin")
        return np.fromfile(fn, dtype=dt) if os.path.exists(fn) else None
    for k in range(4):   # line-search tables (phi, theta_1 at the eight trial steps) of the first iterations
        fn_a, fn_b = (os.path.join(DUMP_DIR, f"{case}_{v}_M{k}_merit.bin") for v in (ref, name))
        if os.path.exists(fn_a) and os.path.exists(fn_b):
            ma, mb = np.fromfile(fn_a).reshape(B, -1), np.fromfile(fn_b).reshape(B, -1)
            d = _cmp(ma.ravel(), mb.ravel())
            out.append(f"  merit table {k}: {d.size} of {ma.size} differ" + ("" if not d.size else "; instance 0 ref " +
                       np.array2string(ma[0], precision=5, max_line_width=400) + " this " + np.array2string(mb[0], precision=5, max_line_width=400)))
    for k in range(8):
        fa, fb = load(ref, k, "fac"), load(name, k, "fac")
        if fa is None or fb is None:
            out.append(f"  launch {k}: no dump ({'ref' if fa is None else name}) -- stop")
            return
        fs = fa.size // (B * T)
        fa, fb = fa.reshape(B, T, fs), fb.reshape(B, T, fs)
        line = [f"  launch {k}:"]
        for what, dt in (("z", np.float64), ("lam", np.float64), ("dw", np.float64), ("active", np.int32)):
            xa, xb = load(ref, k, what, dt), load(name, k, what, dt)
            if xa is not None and xb is not None and not np.array_equal(xa, xb, equal_nan=(dt == np.float64)):
                nd = int(np.sum(xa != xb))
                line.append(f"INPUT {what} differs in {nd} places (max {np.nanmax(np.abs(xa.astype(float) - xb.astype(float))):.3e});")
        sa, sb = load(ref, k, "stats").reshape(B, -1), load(name, k, "stats").reshape(B, -1)
        fl_a, fl_b = load(ref, k, "flags", np.int32), load(name, k, "flags", np.int32)
        line.append(f"flags ref {fl_a.tolist()} this {fl_b.tolist()};")
        ds = _cmp(sa.ravel(), sb.ravel())
        if ds.size:
            line.append("stats differ at (inst, slot): " + ", ".join(f"({i // sa.shape[1]},{i % sa.shape[1]}): {sa.ravel()[i]:.6g} vs {sb.ravel()[i]:.6g}"
                                                                     for i in ds[:8]) + ";")
        first = None
        for b in range(B):
            for t in range(T - 1):
                ra, rb = fa[b, t], fb[b, t]
                hits = []
                for fname, off, ln, padded in FIELDS:
                    xa, xb = ra[off:off + ln], rb[off:off + ln]
                    if padded:
                        keep = (np.arange(ln) % LD) != N
                        xa, xb = xa[keep], xb[keep]
                    d = _cmp(xa, xb)
                    if d.size:
                        hits.append(f"{fname}: {d.size} of {xa.size} (first idx {int(d[0])}: {xa[d[0]]:.6g} vs {xb[d[0]]:.6g}; NaN this: {int(np.isnan(xb).sum())})")
                for vname, off, ln in VECS:
                    xa, xb = ra[FV + off:FV + off + ln], rb[FV + off:FV + off + ln]
                    d = _cmp(xa, xb)
                    if d.size:
                        hits.append(f"{vname}: {d.size} of {ln} (first idx {int(d[0])}: {xa[d[0]]:.6g} vs {xb[d[0]]:.6g})")
                if hits:
                    first = (b, t, hits)
                    break
            if first:
                break
        if first:
            line.append(f"FIRST RECORD DIFFERENCE instance {first[0]} stage {first[1]}: " + " | ".join(first[2]))
        else:
            line.append("factor records equal (1e-9);")
        for what in ("dz", "dlam"):
            xa, xb = load(ref, k, what), load(name, k, what)
            d = _cmp(xa, xb, 1e-7)
            line.append(f"{what}: {d.size} differ, NaN ref {int(np.isnan(xa).sum())} this {int(np.isnan(xb).sum())};")
        out.append(" ".join(line))


def run_all(out_dir, names, case):
    os.makedirs(out_dir, exist_ok=True)
    lines = []
    for name in names:
        env = dict(os.environ, DTO_PLUGIN_CXXFLAGS=VARIANTS[name])
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "run", name, "--case", case], env=env, capture_output=True, text=True, timeout=600)
            rc, so, se = r.returncode, r.stdout, r.stderr
        except subprocess.TimeoutExpired as e:
            rc, so, se = -9, "", "TIMEOUT " + str(e)
        res = [ln for ln in so.splitlines() if ln.startswith("{")]
        lines.append(f"== {case} / {name}  [{VARIANTS[name]}]  rc={rc}")
        lines.append("  " + (res[-1] if res else "NO RESULT: " + se[-600:].replace("\n", " | ")))
        if name != names[0]:
            try:
                diff(names[0], name, lines, case)
            except Exception as e:   # keep going: the other variants still tell something
                lines.append(f"  diff failed: {e!r}")
        with open(os.path.join(out_dir, f"wide_debug_summary_{case}.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    argv = sys.argv[1:]
    opt = {}
    for key in ("--out", "--case"):
        if key in argv:
            i = argv.index(key); opt[key] = argv[i + 1]; del argv[i:i + 2]
    cmd = argv[0] if argv else "all"
    names = argv[1:] or list(VARIANTS)
    case = opt.get("--case", "pad64")
    if cmd == "prebuild":
        prebuild(names, case)
    elif cmd == "build1":
        _problem(case)
    elif cmd == "run":
        run(names[0], case)
    elif cmd == "all":
        run_all(opt.get("--out", os.path.join(ROOT, "gpurun_out", "wide_debug")), names, case)
