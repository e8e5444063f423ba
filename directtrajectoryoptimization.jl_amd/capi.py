"""ctypes binding of include/dto.h (the C-ABI boundary).  No compute happens in Python."""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB, build_runtime

c_double_p = C.POINTER(C.c_double)
c_int64_p = C.POINTER(C.c_int64)
c_int32_p = C.POINTER(C.c_int32)

DTO_ABI_VERSION = 4
DTO_OK = 0
STATUS_NAMES = {0: "DTO_OK", 1: "DTO_ERR_INVALID", 2: "DTO_ERR_PLUGIN", 3: "DTO_ERR_DEVICE",
                4: "DTO_ERR_UNSUPPORTED", 5: "DTO_ERR_NOT_CONVERGED"}

(IDX_STATE, IDX_ACTION, IDX_STATE_ACTION, IDX_STATE_ACTION_NEXT, IDX_DYNAMICS_CONSTRAINT,
 IDX_DYNAMICS_JACOBIAN, IDX_DYNAMICS_HESSIAN, IDX_STAGE_CONSTRAINT, IDX_STAGE_JACOBIAN,
 IDX_STAGE_HESSIAN, IDX_OBJECTIVE_HESSIAN) = range(11)


class ProblemSpec(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int),
        ("model_library", C.c_char_p),
        ("horizon", C.c_int),
        ("stage_kind", c_int32_p),
        ("variable_lower", c_double_p),
        ("variable_upper", c_double_p),
        ("parameters", c_double_p),
        ("num_parameters", C.c_int64),
        ("evaluate_hessian", C.c_int),
    ]


class Sizes(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "num_variables", "num_parameters", "num_constraint", "num_constraint_dynamics",
        "num_constraint_stage", "num_constraint_general", "num_jacobian", "num_jacobian_dynamics",
        "num_jacobian_stage", "num_jacobian_general", "nnz_hess_key", "nnz_hess_raw", "horizon",
        "num_state_max", "num_action_max")]


class Batch(C.Structure):
    _fields_ = [("B", C.c_int64), ("x", C.c_void_p), ("ldx", C.c_int64), ("params", C.c_void_p),
                ("ldp", C.c_int64), ("stream", C.c_void_p)]


class KktSystem(C.Structure):
    _fields_ = [("mu", C.c_void_p), ("ldmu", C.c_int64), ("sigma_x", C.c_void_p), ("ldsx", C.c_int64),
                ("sigma_c", C.c_void_p), ("ldsc", C.c_int64), ("delta_w", C.c_double), ("delta_c", C.c_double)]


class COptions(C.Structure):
    _fields_ = [("tol", C.c_double), ("s_max", C.c_double), ("max_iter", C.c_int), ("dual_inf_tol", C.c_double),
                ("constr_viol_tol", C.c_double), ("compl_inf_tol", C.c_double), ("mu_init", C.c_double),
                ("delta_c", C.c_double), ("delta_w_init", C.c_double), ("check_every", C.c_int), ("max_cpu_time", C.c_double),
                ("acceptable_tol", C.c_double), ("acceptable_iter", C.c_int), ("acceptable_dual_inf_tol", C.c_double),
                ("acceptable_constr_viol_tol", C.c_double), ("acceptable_compl_inf_tol", C.c_double),
                ("acceptable_obj_change_tol", C.c_double), ("diverging_iterates_tol", C.c_double), ("mu_target", C.c_double),
                ("line_search", C.c_int), ("penalty_switch_theta", C.c_double),
                ("hessian_approximation", C.c_int), ("kkt_refinement", C.c_int)]


DTO_LS_FILTER, DTO_LS_PENALTY_FILTER = 0, 1
DTO_HESSIAN_EXACT, DTO_HESSIAN_LBFGS, DTO_HESSIAN_SR1_BLOCKS = 0, 1, 2
DTO_STATUS_CPU_TIME = 6


# enum dto_scal (csrc/dto_kkt_kernels.hpp)
SCALARS = ["status", "iter", "mu", "penalty", "delta_w", "f", "theta1", "theta_inf", "dinf", "compl", "e0", "logbar",
           "alpha_pmax", "alpha_dmax", "dmerit", "alpha", "ls_fail", "nfact", "merit0", "delta_last",
           "theta_max", "theta_min", "filter_n", "ls_kind", "gamma", "need", "try_dw", "try_gam", "attempt", "qn_reset", "full_streak", "short_streak", "watchdog",
           "acc_count", "f_last", "xmax", "nneg", "ls_mode", "ascale", "qn_sigma", "qn_skip", "qn_gcorr"]


class DtoError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{STATUS_NAMES.get(code, code)}: {msg}")
        self.code = code


_lib = None


def lib() -> C.CDLL:
    """Load libdto_hip.so (building it first if the sources changed). Fails loudly if it cannot."""
    global _lib
    if _lib is not None:
        return _lib
    path = build_runtime()
    if not os.path.exists(path):
        raise RuntimeError(f"HIP runtime library missing: {LIB}")
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    L.dto_last_error.restype = C.c_char_p
    vp = C.c_void_p
    sigs = {
        "dto_problem_create": [C.POINTER(ProblemSpec), C.POINTER(vp)],
        "dto_problem_destroy": [vp],
        "dto_sizes": [vp, C.POINTER(Sizes)],
        "dto_features_available": [vp, C.POINTER(C.c_int)],
        "dto_jacobian_structure": [vp, c_int64_p, c_int64_p],
        "dto_hessian_structure": [vp, c_int64_p, c_int64_p],
        "dto_variable_bounds": [vp, c_double_p, c_double_p],
        "dto_constraint_bounds": [vp, c_double_p, c_double_p],
        "dto_stage_indices": [vp, C.c_int, C.c_int, c_int64_p, c_int64_p],
        "dto_eval_f": [vp, c_double_p, c_double_p],
        "dto_eval_grad_f": [vp, c_double_p, c_double_p],
        "dto_eval_g": [vp, c_double_p, c_double_p],
        "dto_eval_jac_g": [vp, c_double_p, c_double_p],
        "dto_eval_h": [vp, c_double_p, C.c_double, c_double_p, c_double_p],
        "dto_eval_f_batch": [vp, C.POINTER(Batch), vp],
        "dto_eval_grad_f_batch": [vp, C.POINTER(Batch), vp, C.c_int64],
        "dto_eval_g_batch": [vp, C.POINTER(Batch), vp, C.c_int64],
        "dto_eval_jac_g_batch": [vp, C.POINTER(Batch), vp, C.c_int64],
        "dto_eval_h_batch": [vp, C.POINTER(Batch), C.c_double, vp, C.c_int64, vp, C.c_int64],
        "dto_kkt_csr_structure": [vp, c_int64_p, c_int64_p, c_int64_p, c_int64_p],
        "dto_kkt_csr_values_batch": [vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, C.c_double, C.c_double, vp, C.c_int64, vp],
        "dto_options_default": [C.POINTER(COptions)],
        "dto_kkt_step_batch": [vp, C.POINTER(Batch), vp, C.c_int64, C.c_double, C.c_double, vp, C.c_int64, vp, C.c_int64,
                               C.POINTER(C.c_int)],
        "dto_solve_batch": [vp, C.POINTER(COptions), C.POINTER(Batch), vp, C.c_int64, vp, C.c_int64, c_int32_p, c_int32_p],
        "dto_solver_begin": [vp, C.POINTER(COptions), C.POINTER(Batch)],
        "dto_solver_begin_warm": [vp, C.POINTER(COptions), C.POINTER(Batch), C.c_double],
        "dto_solver_repack": [vp, C.POINTER(C.c_int), vp],
        "dto_solver_run": [vp, vp, C.c_int64, vp, C.c_int64, c_int32_p, c_int32_p, vp],
        "dto_kkt_assemble": [vp, C.POINTER(Batch), C.POINTER(KktSystem)],
        "dto_kkt_factor": [vp, c_int32_p, c_int32_p, vp],
        "dto_kkt_solve": [vp, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp],
        "dto_shard_range": [C.c_int64, C.c_int, C.c_int, c_int64_p, c_int64_p],
        "dto_solver_iterate": [vp, C.c_int, vp],
        "dto_solver_stats": [vp, c_int32_p, c_int32_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p],
        "dto_solver_end": [vp, vp, C.c_int64, vp, C.c_int64, vp],
        "dto_solver_scalar": [vp, C.c_int, c_double_p],
        "dto_solver_peek": [vp, C.c_int, vp, C.c_int64, vp],
        "dto_solver_hessian_mode": [vp, C.POINTER(C.c_int)],
        "dto_solver_trace": [vp, C.c_int],
        "dto_solver_trace_read": [vp, c_int32_p, c_int32_p, c_double_p, c_double_p, C.c_int64, c_int64_p],
        "dto_solver_launch_op": [vp, C.c_int, vp],
        "dto_solver_footprint": [vp, c_int64_p, c_int64_p, c_int64_p, C.POINTER(C.c_int)],
        "dto_solver_set_partitions": [vp, C.c_int],
        "dto_solver_partitions": [vp, C.POINTER(C.c_int)],
        "dto_solver_fused_update": [vp, C.POINTER(C.c_int)],
        "dto_solver_set_engine": [vp, C.c_int],
        "dto_solver_engine": [vp, C.POINTER(C.c_int)],
        "dto_solver_release": [vp],
        "dto_solver_shift": [vp, C.c_int, vp],
        "dto_solve": [vp, C.POINTER(COptions), c_double_p, c_double_p, c_double_p, c_int32_p, c_int32_p],
        "dto_device_alloc": [C.POINTER(vp), C.c_int64],
        "dto_device_free": [vp],
        "dto_copy_to_device": [vp, vp, C.c_int64],
        "dto_copy_to_host": [vp, vp, C.c_int64],
        "dto_device_synchronize": [],
        "dto_device_count": [C.POINTER(C.c_int)],
    }
    for name, argtypes in sigs.items():
        fn = getattr(L, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    _lib = L
    return L


def check(rc: int):
    if rc != DTO_OK:
        raise DtoError(rc, lib().dto_last_error().decode(errors="replace"))


def dptr(a):
    """double* of a contiguous float64 numpy array."""
    return a.ctypes.data_as(c_double_p)
