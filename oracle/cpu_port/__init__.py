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
        L.port_create_named.restype = C.c_void_p
        L.port_create_named.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
        L.port_destroy.argtypes = [C.c_void_p]
        L.port_begin.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.port_iterate.argtypes = [C.c_void_p]
        L.port_iterate.restype = C.c_int
        for name in ("port_status", "port_iterations", "port_nfact", "port_num_variables", "port_num_constraint"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_int
        for name in ("port_objective", "port_constr_viol", "port_dual_inf", "port_alpha", "port_delta_w"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_double
        for name in ("port_z", "port_lam"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.POINTER(C.c_double)
        _lib = L
    return _lib


class PortSolver:
    def __init__(self, model: str, T: int, x1, xT, max_iter: int = 1000):
        x1 = np.ascontiguousarray(x1, dtype=float)
        xT = np.ascontiguousarray(xT, dtype=float)
        self._h = lib().port_create_named(model.encode(), T, x1.ctypes.data_as(C.POINTER(C.c_double)),
                                          xT.ctypes.data_as(C.POINTER(C.c_double)), max_iter)
        if not self._h:
            raise ValueError(f"the CPU port has no model {model!r}")
        self.nz = lib().port_num_variables(self._h)
        self.nc = lib().port_num_constraint(self._h)

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

    def stats(self):
        L = lib()
        return dict(objective=L.port_objective(self._h), constr_viol=L.port_constr_viol(self._h),
                    dual_inf=L.port_dual_inf(self._h), alpha=L.port_alpha(self._h), delta_w=L.port_delta_w(self._h))

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
    """Seeded guesses of a port model (acrobot: the bench workload's)."""
    if model == "acrobot":
        return acrobot_guesses(T, B, seed)
    raise ValueError(f"no guess generator for {model!r}")


def cpu_baseline(T=1000, seed=1000, seconds=12.0, batch=4096, iters_per_instance=23):
    """SQP iterations/s of the C port on one host core for a bounded sample of the bench workload:
    the first instances of rank 0's batch, each run for the same number of iterations the GPU bench
    executes per instance (warmup + steps), until about `seconds` of CPU time have been spent."""
    Z, x1, xT = acrobot_guesses(T, batch, seed)
    s = PortSolver("acrobot", T, x1, xT)
    done_iters, done_inst, nfact = 0, 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds and done_inst < batch:
        s.begin(Z[done_inst])
        k = 0
        while k < iters_per_instance and s.iterate():
            k += 1
        done_iters += k
        nfact += s.nfact
        done_inst += 1
    dt = time.perf_counter() - t0
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return dict(value=done_iters / dt, unit="SQP iterations/s", cores=1, kind="port",
                sample=f"{done_inst} instances x {iters_per_instance} iterations of acrobot T={T} "
                       f"(same guesses as GPU rank 0), {dt:.1f} s on 1 core, {nfact / max(done_iters, 1):.2f} factorizations/iteration",
                host_cpu=cpu_model, host_cores_available=os.cpu_count())
