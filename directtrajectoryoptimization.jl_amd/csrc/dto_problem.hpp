// Problem handle behind the opaque `dto_problem*` of include/dto.h.
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "dto_layout.hpp"
#include "dto_model_plugin.h"

namespace dto {

int set_error(int code, const std::string& msg);
int hip_fail(hipError_t e, const char* what);

struct SolverState;  // dto_solver.cpp
struct ImState;      // dto_solver.cpp: instance-major engine

// dto_solver_trace: a HIP event pair around every kernel launch of the solver entry points, on the stream the kernel is launched
// on (the caller's, or the low-priority one of the early back substitutions) -- so that a caller can report what each kernel
// took INSIDE the region it timed instead of replaying the launches afterwards.  Events are pooled; nothing is synchronised
// until the trace is read.
struct LaunchTrace {
  struct Rec { int op, iteration, e0, e1; };
  bool on = false;
  int iteration = 0;                 // iterations of dto_solver_iterate since the trace was switched on
  std::vector<hipEvent_t> pool;      // events created so far (re-used by the next trace)
  size_t used = 0;
  std::vector<Rec> recs;
  static constexpr size_t MAX_RECS = 1u << 18;
  int take() {
    if (used == pool.size()) {
      hipEvent_t e = nullptr;
      if (hipEventCreate(&e) != hipSuccess) return -1;
      pool.push_back(e);
    }
    return (int)used++;
  }
  void clear() { used = 0; recs.clear(); iteration = 0; }
  ~LaunchTrace() { for (hipEvent_t e : pool) (void)hipEventDestroy(e); }
};

struct Problem {
  void* dl = nullptr;
  const dto_model_vtable* vt = nullptr;
  Layout L;
  // device tables
  bool dev_ready = false;
  int *d_kind = nullptr, *d_zoff = nullptr, *d_woff = nullptr, *d_cdoff = nullptr, *d_ccoff = nullptr;
  int *d_jdoff = nullptr, *d_jcoff = nullptr, *d_hoff = nullptr;
  int *d_hmap_cost = nullptr, *d_hmap_dyn_own = nullptr, *d_hmap_dyn_next = nullptr, *d_hmap_con = nullptr;
  double* d_params = nullptr;
  // single-instance staging for host-pointer callbacks
  double *d_x1 = nullptr, *d_mu1 = nullptr, *d_out1 = nullptr;
  double* d_scratch = nullptr;
  size_t scratch_len = 0;
  hipStream_t stream = nullptr;
  SolverState* solver = nullptr;
  ImState* im = nullptr;
  LaunchTrace* trace = nullptr;   // dto_solver_trace
  int hessian_mode_last = -1;     // dto_solver_hessian_mode: what the last solve / begun batch used
  int engine_req = 0;      // dto_solver_set_engine: 0 automatic, 1 SoA tiles, 2 instance-major
  bool im_active = false;  // which engine holds the batch that was begun last
  // factor storage and inertia flags of the wide-stage KKT kernels
  double* wide_fac = nullptr;
  size_t wide_fac_len = 0;
  int* wide_flags = nullptr;
  size_t wide_flags_len = 0;
  // device workspace of the bordered (multi-knot GeneralConstraint) step, kept between steps (eight hipMalloc /
  // hipFree per Newton step -- every hipFree is a device synchronisation)
  double* border_ws = nullptr;
  size_t border_ws_len = 0;
  // Jacobian pattern by columns (entries of a column in COO order) for the device-side products of the bordered step
  int *d_csc_ptr = nullptr, *d_csc_k = nullptr, *d_csc_row = nullptr;
  int* d_var_fixed = nullptr;   // [Nz] 1 where lo == hi

  // CSR pattern of the KKT matrix in the reference ordering (dto_kkt_csr_structure / dto_kkt_csr_values_batch), built on first use
  std::vector<int64_t> csr_rowptr, csr_col;   // 1-based
  int *d_csr_h = nullptr, *d_csr_j = nullptr;  // per CSR slot: 0-based index into the Hessian key / Jacobian values, -1 = none
  signed char* d_csr_diag = nullptr;           // per CSR slot: 1 = primal diagonal (+delta_w), 2 = dual diagonal (-delta_c), 0 = off-diagonal
  int build_kkt_csr();

  int ensure_device();
  int ensure_scratch(int64_t B);
  void fill_args(dto_eval_args& a, int64_t B, const double* z, int64_t ldz, const double* w, int64_t ldw) const;
  int launch(int op, const dto_eval_args& a, hipStream_t s);
  void free_solver();
  ~Problem();
};

}  // namespace dto
