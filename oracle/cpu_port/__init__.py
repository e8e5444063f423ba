"""ORACLE / CPU baseline: ctypes wrapper of the serial C port (oracle/cpu_port/solver_port.c).

Test infrastructure only: imported by tests/ and by bench.py's cpu_baseline leg, never by the product.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(os.path.dirname(_HERE), "_build", "libdto_cpu_port.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.run(["make", "-C", os.path.dirname(_HERE), "-s"], check=True)
        L = C.CDLL(_LIB)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.port_create.restype = C.c_void_p
        L.port_create.argtypes = [C.c_char_p, C.c_int, ip, dp, dp, C.c_int]
        L.port_destroy.argtypes = [C.c_void_p]
        L.port_begin.argtypes = [C.c_void_p, dp]
        L.port_iterate.argtypes = [C.c_void_p]
        L.port_iterate.restype = C.c_int
        L.port_set_int.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.port_set_double.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        for name in ("port_status", "port_iterations", "port_nfact", "port_nsoc", "port_ls_kind", "port_num_variables",
                     "port_num_constraint", "port_num_slacks", "port_qn_pairs"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_int
        for name in ("port_objective", "port_constr_viol", "port_dual_inf", "port_alpha", "port_delta_w", "port_mu", "port_qn_sigma"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_double
        for name in ("port_z", "port_lam"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = dp
        L.port_run_batch.restype = C.c_long
        L.port_run_batch.argtypes = [C.c_char_p, C.c_int, ip, dp, dp, C.c_int, C.c_int, C.c_int, dp, ip, ip, C.POINTER(C.c_long)]
        _lib = L
    return _lib


_DESC = {}


def problem_arrays(model: str, T: int):
    """(constraint class per stage, lower bounds, upper bounds) of a BASELINE model at horizon T, from the oracle's
    restatement of the examples (oracle/sympy_models.py:build).  Class k = k-th distinct non-empty stage-constraint object
    in order of first appearance -- the same rule gen_model_c.py uses when it emits <model>_con<k>."""
    key = (model, T)
    if key not in _DESC:
        from oracle import sympy_models as S
        # the structure (which object sits at which stage, which bounds) does not depend on T beyond first/interior/last:
        # build a short instance and stretch it (sympy construction at T = 1000 would take seconds for nothing)
        Ts = min(T, 4)
        p = S.build(model, Ts, evaluate_hessian=False)
        classes, con_s = [], []
        for c in p["constraints"]:
            if c.num_constraint == 0:
                con_s.append(-1)
                continue
            if all(c is not o for o in classes):
                classes.append(c)
            con_s.append([i for i, o in enumerate(classes) if o is c][0])
        stage = lambda t: 0 if t == 0 else (Ts - 1 if t == T - 1 else min(1, Ts - 2))
        con = np.array([con_s[stage(t)] for t in range(T)], dtype=np.int32)
        lo, hi = [], []
        for t in range(T):
            b = p["bounds"][stage(t)]
            lo += list(b.state_lower) + (list(b.action_lower) if t < T - 1 else [])
            hi += list(b.state_upper) + (list(b.action_upper) if t < T - 1 else [])
        _DESC[key] = (con, np.array(lo, dtype=float), np.array(hi, dtype=float))
    return _DESC[key]


class PortSolver:
    def __init__(self, model: str, T: int, x1=None, xT=None, max_iter: int = 1000, **options):
        """x1 / xT are accepted for compatibility; the end states are part of the generated model (as in the examples)."""
        con, lo, hi = problem_arrays(model, T)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        self._h = lib().port_create(model.encode(), T, con.ctypes.data_as(ip), lo.ctypes.data_as(dp), hi.ctypes.data_as(dp), max_iter)
        if not self._h:
            raise ValueError(f"the CPU port has no model {model!r}")
        self.model, self.T = model, T
        self.nz = lib().port_num_variables(self._h)
        self.nc = lib().port_num_constraint(self._h)
        for k, v in options.items():
            self.set(k, v)

    def set(self, name, value):
        if isinstance(value, (int, np.integer)) and name in ("max_soc", "max_iter", "watchdog_trigger", "watchdog_trials", "acceptable_iter", "ls_penalty", "pen_gn", "lbfgs"):
            lib().port_set_int(self._h, name.encode(), int(value))
        else:
            lib().port_set_double(self._h, name.encode(), float(value))

    def begin(self, z0):
        z0 = np.ascontiguousarray(z0, dtype=float)
        assert z0.size == self.nz
        lib().port_begin(self._h, z0.ctypes.data_as(C.POINTER(C.c_double)))

    def iterate(self) -> int:
        return lib().port_iterate(self._h)

    def solve(self, z0, max_iter=1000):
        self.begin(z0)
        while self.iterate():
            pass
        return self.status

    @property
    def status(self):
        return lib().port_status(self._h)

    @property
    def iterations(self):
        return lib().port_iterations(self._h)

    @property
    def nfact(self):
        return lib().port_nfact(self._h)

    @property
    def qn_sigma(self):
        return lib().port_qn_sigma(self._h)

    @property
    def qn_pairs(self):
        return lib().port_qn_pairs(self._h)

    @property
    def nsoc(self):
        return lib().port_nsoc(self._h)

    def stats(self):
        L = lib()
        return dict(objective=L.port_objective(self._h), constr_viol=L.port_constr_viol(self._h),
                    dual_inf=L.port_dual_inf(self._h), alpha=L.port_alpha(self._h), delta_w=L.port_delta_w(self._h),
                    mu=L.port_mu(self._h), ls_kind=L.port_ls_kind(self._h))

    @property
    def z(self):
        return np.ctypeslib.as_array(lib().port_z(self._h), shape=(self.nz,)).copy()

    @property
    def lam(self):
        return np.ctypeslib.as_array(lib().port_lam(self._h), shape=(self.nc,)).copy()

    def close(self):
        if self._h:
            lib().port_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_batch(model, T, Z0, max_iter=1000, iters_per_instance=0, threads=0):
    """Solve (or iterate a fixed number of times) the rows of Z0, OpenMP over instances (threads = 0: all cores).
    Returns (total iterations, seconds, iterations[B], status[B], factorizations)."""
    con, lo, hi = problem_arrays(model, T)
    Z0 = np.ascontiguousarray(Z0, dtype=float)
    B = Z0.shape[0]
    it, st = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
    nf = C.c_long(0)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    old = os.environ.get("OMP_NUM_THREADS")
    if threads:
        try:
            omp = C.CDLL("libgomp.so.1")
            omp.omp_set_num_threads(int(threads))
        except OSError:
            pass
    t0 = time.perf_counter()
    total = lib().port_run_batch(model.encode(), T, con.ctypes.data_as(ip), lo.ctypes.data_as(dp), hi.ctypes.data_as(dp), max_iter,
                                 iters_per_instance, B, Z0.ctypes.data_as(dp), it.ctypes.data_as(ip), st.ctypes.data_as(ip), C.byref(nf))
    dt = time.perf_counter() - t0
    return int(total), dt, it, st, int(nf.value)


def acrobot_guesses(T, B, seed):
    """Same guesses as bench.py: linear interpolation 0 -> [pi,0,0,0], u ~ N(0,1) from PCG64(seed)."""
    n, m = 4, 1
    x1, xT = np.zeros(4), np.array([np.pi, 0.0, 0.0, 0.0])
    rng = np.random.Generator(np.random.PCG64(seed))
    U = rng.standard_normal((B, T - 1, m))
    Z = np.zeros((B, T * (n + m) - m))
    for t in range(T):
        o = t * (n + m)
        Z[:, o:o + n] = (xT - x1) / (T - 1) * t + x1
        if t < T - 1:
            Z[:, o + n:o + n + m] = U[:, t]
    return Z, x1, xT


def guesses(model, T, B, seed):
    """Seeded guesses of a port model, restating the examples: acrobot = the bench workload's; cartpole = rollout guess
    u = 0.01 (examples/cartpole/cartpole.jl:102-106: deterministic); car = linear interpolation + 0.001 N(0,1)
    (examples/car/car.jl:62-67), one PCG64 stream per instance seeded with the instance id."""
    if model == "acrobot":
        return acrobot_guesses(T, B, seed)
    if model == "car":
        n, m = 3, 2
        x1, xT = np.zeros(3), np.array([1.0, 1.0, 0.0])
        Z = np.zeros((B, T * (n + m) - m))
        for b in range(B):
            rng = np.random.Generator(np.random.PCG64(b))
            for t in range(T):
                o = t * (n + m)
                Z[b, o:o + n] = (xT - x1) / (T - 1) * t + x1
            for t in range(T - 1):
                o = t * (n + m)
                Z[b, o + n:o + n + m] = 0.001 * rng.standard_normal(m)
        return Z, x1, xT
    if model == "cartpole":
        import sympy as sp
        from oracle import sympy_models as S
        n, m = 4, 1
        xs, us = S.syms("x", n), S.syms("u", m)
        step = sp.lambdify(xs + us, S.cartpole_rk3_explicit(xs, us, []), modules="math")
        x1, xT = np.zeros(4), np.array([0.0, np.pi, 0.0, 0.0])
        z = np.zeros(T * (n + m) - m)
        x = x1.copy()
        for t in range(T):
            o = t * (n + m)
            z[o:o + n] = x
            if t < T - 1:
                z[o + n] = 0.01
                x = np.array(step(*x, 0.01), dtype=float)
        return np.tile(z, (B, 1)), x1, xT
    if model == "pendulum":
        n, m = 2, 1
        x1, xT = np.zeros(2), np.array([np.pi, 0.0])
        rng = np.random.Generator(np.random.PCG64(seed))
        U = rng.standard_normal((B, T - 1, m))
        Z = np.zeros((B, T * (n + m) - m))
        for t in range(T):
            o = t * (n + m)
            Z[:, o:o + n] = (xT - x1) / (T - 1) * t + x1
            if t < T - 1:
                Z[:, o + n:o + n + m] = U[:, t]
        return Z, x1, xT
    raise ValueError(f"no guess generator for {model!r}")


def cpu_baseline(T=1000, seed=1000, seconds=12.0, batch=4096, iters_per_instance=23, model="acrobot"):
    """SQP iterations/s of the C port on the host for a bounded sample of the bench workload: the first instances of rank
    0's batch, each run for the same number of iterations the GPU bench executes per instance (iters_per_instance <= 0:
    solved to the reference tolerances).  Two figures: all host cores (OpenMP over instances, `value` / `cores`) and one
    core (`value_1core`), each on about `seconds` of wall time."""
    Z, x1, xT = guesses(model, T, batch, seed)
    ncores = os.cpu_count() or 1
    # pilot runs size the samples to the time budget: one instance per thread, on one core and on all of them (the
    # all-cores rate per thread is lower: shared caches, memory bandwidth, SMT)
    _, p1, _, _, _ = run_batch(model, T, Z[:1], iters_per_instance=iters_per_instance, threads=1)
    n1 = int(max(1, min(batch, seconds / max(p1, 1e-9))))
    it1, dt1, _, st1, nf1 = run_batch(model, T, Z[:n1], iters_per_instance=iters_per_instance, threads=1)
    _, pa, _, _, _ = run_batch(model, T, Z[:min(batch, ncores)], iters_per_instance=iters_per_instance, threads=ncores)
    nall = int(max(ncores, min(batch, ncores * max(1.0, (seconds - pa) / max(pa, 1e-9)))))
    nall = min(batch, (nall // ncores) * ncores if nall >= ncores else nall)
    ita, dta, _, sta, nfa = run_batch(model, T, Z[:nall], iters_per_instance=iters_per_instance, threads=ncores)
    n1b = n1
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    what = f"{iters_per_instance} iterations" if iters_per_instance > 0 else "full solves (tol 1e-6, max_iter 1000)"
    return dict(value=ita / dta, unit="SQP iterations/s", cores=ncores, kind="port",
                sample=f"{nall} instances x {what} of {model} T={T} (same guesses as GPU rank 0), {dta:.1f} s on {ncores} "
                       f"threads (OpenMP over instances), {nfa / max(ita, 1):.2f} factorizations/iteration",
                value_1core=it1 / dt1, sample_1core=f"{n1b} instances, {dt1:.1f} s on 1 core",
                converged_fraction=float(np.mean(sta == 1)) if iters_per_instance <= 0 else None,
                host_cpu=cpu_model, host_cores_available=ncores)
