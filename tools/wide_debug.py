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
    "prod": "",
    "stamps_in": "-DDTO_WIDE_PROFILE=1",
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
T_RUN, B_RUN, TARGET = 24, 3, 0.3
DUMP_DIR = "/tmp/wide_dbg"


def _solver(T=T_RUN):
    import dto_amd
    from dto_amd import problems as P
    p = P.build_acrobot_padded(T=T, target=TARGET, terminal="physical")
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot_padded")
    return s, p


def prebuild(names):
    from dto_amd import problems as P
    from dto_amd.plugin import Structure, _prepare_plugin, _compile_plugin
    from concurrent.futures import ThreadPoolExecutor
    jobs = []
    for name in names:
        os.environ["DTO_PLUGIN_CXXFLAGS"] = VARIANTS[name]
        p = P.build_acrobot_padded(T=T_RUN, target=TARGET, terminal="physical")
        st = Structure(p["dynamics"], p["objective"], p["constraints"], None, True)
        so, cmd = _prepare_plugin(st, "acrobot_padded")
        jobs.append((name, so, cmd))
    os.environ.pop("DTO_PLUGIN_CXXFLAGS", None)
    with ThreadPoolExecutor(max_workers=8) as ex:
        for name, so in zip(names, ex.map(lambda j: _compile_plugin(j[0], j[1], j[2], False), jobs)):
            print(f"{name:16s} {os.path.basename(so)}", flush=True)


def run(name):
    import torch
    import dto_amd
    from dto_amd.plugin import COMPILED
    os.makedirs(DUMP_DIR, exist_ok=True)
    os.environ["DTO_WIDE_DUMP"] = os.path.join(DUMP_DIR, name)
    os.environ.setdefault("DTO_WIDE_DUMP_MAX", "4")
    s, p = _solver()
    nz, nc = s.nlp.num_variables, s.nlp.num_constraint
    Z = np.zeros((B_RUN, nz))
    for b in range(B_RUN):
        xs, us = p["guess"](np.random.Generator(np.random.PCG64(b)))
        dto_amd.initialize_states(s, xs)
        dto_amd.initialize_controls(s, [0.1 * u for u in us])
        Z[b] = s._z0
    z0 = torch.tensor(Z, device="cuda")
    zo = torch.full((B_RUN, nz), float("nan"), device="cuda", dtype=torch.float64)
    lo = torch.full((B_RUN, nc), float("nan"), device="cuda", dtype=torch.float64)
    status, iters = s.solve_batch(z0.data_ptr(), B_RUN, nz, zo.data_ptr(), nz, lo.data_ptr(), nc)
    torch.cuda.synchronize()
    # the plain KKT step of the same build at a random point (the use that stays correct)
    rng = np.random.default_rng(5)
    Zr, Mr = torch.tensor(rng.random((B_RUN, nz)), device="cuda"), torch.tensor(rng.random((B_RUN, nc)), device="cuda")
    dx, dl = torch.empty_like(Zr), torch.empty_like(Mr)
    ok = s.kkt_step_batch(Zr.data_ptr(), B_RUN, nz, Mr.data_ptr(), nc, 2.0, 1e-5, dx.data_ptr(), nz, dl.data_ptr(), nc)
    torch.cuda.synchronize()
    print(json.dumps(dict(variant=name, flags=VARIANTS.get(name, "?"), status=status.tolist(), iterations=iters.tolist(),
                          z_nan=int(torch.isnan(zo).sum()), plain_step_ok=bool(ok), plain_step_nan=int(torch.isnan(dx).sum()),
                          plain_step_norm=float(dx.abs().max()), compiled_here=list(COMPILED))), flush=True)


# ---- factor record layout (csrc/dto_wide_kernels.hpp: Dims<64, 1>)
N, NU, LD = 64, 1, 65
MAT, PKL = 64 * 65, 10 * 256
FIELDS = [("L_A(tiles)", 0, PKL, False), ("F~", PKL, MAT, True), ("V~", PKL + MAT, MAT, True), ("L_M(tiles)", PKL + 2 * MAT, PKL, False),
          ("E~", 2 * PKL + 2 * MAT, MAT, True)]
FV = 2 * PKL + 3 * MAT
VECS = [("1/D_A", 0, 64), ("1/D_M", 64, 64), ("bx~", 128, 64), ("bd^", 192, 64), ("A_xu", 256, 64), ("F_u", 320, 64), ("V_u", 384, 64),
        ("grad cost", 448, 65), ("1/piv_u", 520, 1), ("bu", 521, 1)]


def _cmp(a, b, tol=1e-9):
    """indices where a and b differ: NaN pattern, or relative difference above tol"""
    na, nb = np.isnan(a), np.isnan(b)
    sc = np.maximum(np.maximum(np.abs(np.nan_to_num(a)), np.abs(np.nan_to_num(b))), 1e-30)
    bad = (na != nb) | (~na & ~nb & (np.abs(np.nan_to_num(a) - np.nan_to_num(b)) > tol * np.maximum(sc, np.max(sc) * 1e-6)))
    return np.nonzero(bad)[0]


def diff(ref, name, out):
    def load(v, k, what, dt=np.float64):
        fn = os.path.join(DUMP_DIR, f"{v}_L{k}_{what}.bin")
        return np.fromfile(fn, dtype=dt) if os.path.exists(fn) else None
    for k in range(8):
        fa, fb = load(ref, k, "fac"), load(name, k, "fac")
        if fa is None or fb is None:
            out.append(f"  launch {k}: no dump ({'ref' if fa is None else name}) -- stop")
            return
        B, T = B_RUN, T_RUN
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


def run_all(out_dir, names):
    os.makedirs(out_dir, exist_ok=True)
    lines = []
    for name in names:
        env = dict(os.environ, DTO_PLUGIN_CXXFLAGS=VARIANTS[name])
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "run", name], env=env, capture_output=True, text=True, timeout=900)
        res = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        lines.append(f"== {name}  [{VARIANTS[name]}]  rc={r.returncode}")
        lines.append("  " + (res[-1] if res else "NO RESULT: " + r.stderr[-600:].replace("\n", " | ")))
        if name != names[0]:
            try:
                diff(names[0], name, lines)
            except Exception as e:   # keep going: the other variants still tell something
                lines.append(f"  diff failed: {e!r}")
        with open(os.path.join(out_dir, "wide_debug_summary.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "all"
    names = [a for a in sys.argv[2:] if not a.startswith("--")] or list(VARIANTS)
    if cmd == "prebuild":
        prebuild(names)
    elif cmd == "run":
        run(sys.argv[2])
    elif cmd == "all":
        out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(ROOT, "gpurun_out", "wide_debug")
        names = [n for n in names if n != out]
        run_all(out, names)
