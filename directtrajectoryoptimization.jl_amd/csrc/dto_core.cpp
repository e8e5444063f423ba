// libdto_hip.so -- runtime behind include/dto.h.
//
// Owns: plugin loading, the layout contract (dto_layout.hpp), device tables/workspaces, and the
// evaluator entry points that replace the MOI callbacks of reference src/moi.jl:1-125.
// There is no CPU evaluation path in this library: without a GPU every compute entry point
// returns DTO_ERR_DEVICE (structure queries, which the reference computes at construction time
// on the host as well, still work).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>

#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/dto.h"
#include "dto_layout.hpp"
#include "dto_model_plugin.h"
#include "dto_problem.hpp"

namespace {
thread_local std::string g_error;
}

namespace dto {
int set_error(int code, const std::string& msg) {
  g_error = msg;
  return code;
}
int hip_fail(hipError_t e, const char* what) {
  // HIP keeps the last error until it is read: left in place, the NEXT launch check of this library or of the caller's framework
  // (torch's C10 launch check) would report this failure again -- e.g. after a hipMalloc that a caller recovers from by
  // retrying with a smaller batch
  (void)hipGetLastError();
  return set_error(DTO_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace dto

using dto::hip_fail;
using dto::set_error;

#define HIP_TRY(expr)                                     \
  do {                                                    \
    hipError_t e_ = (expr);                               \
    if (e_ != hipSuccess) return hip_fail(e_, #expr);     \
  } while (0)

// ------------------------------------------------------------------------------------------------
// device tables
// ------------------------------------------------------------------------------------------------
namespace dto {

static int upload_ints(const std::vector<int>& v, int** dptr) {
  const size_t bytes = std::max<size_t>(1, v.size()) * sizeof(int);
  HIP_TRY(hipMalloc((void**)dptr, bytes));
  if (!v.empty()) HIP_TRY(hipMemcpy(*dptr, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
  return DTO_OK;
}

int Problem::ensure_device() {
  if (dev_ready) return DTO_OK;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) return set_error(DTO_ERR_DEVICE, "no HIP device available (the evaluator has no CPU path)");
  int rc;
  if ((rc = upload_ints(L.kind, &d_kind))) return rc;
  if ((rc = upload_ints(L.zoff, &d_zoff))) return rc;
  if ((rc = upload_ints(L.woff, &d_woff))) return rc;
  if ((rc = upload_ints(L.cdoff, &d_cdoff))) return rc;
  if ((rc = upload_ints(L.ccoff, &d_ccoff))) return rc;
  if ((rc = upload_ints(L.jdoff, &d_jdoff))) return rc;
  if ((rc = upload_ints(L.jcoff, &d_jcoff))) return rc;
  if ((rc = upload_ints(L.hoff, &d_hoff))) return rc;
  if ((rc = upload_ints(L.hmap_cost, &d_hmap_cost))) return rc;
  if ((rc = upload_ints(L.hmap_dyn_own, &d_hmap_dyn_own))) return rc;
  if ((rc = upload_ints(L.hmap_dyn_next, &d_hmap_dyn_next))) return rc;
  if ((rc = upload_ints(L.hmap_con, &d_hmap_con))) return rc;
  HIP_TRY(hipMalloc((void**)&d_params, std::max<size_t>(1, L.Nw) * sizeof(double)));
  if (L.Nw) HIP_TRY(hipMemcpy(d_params, L.params.data(), L.Nw * sizeof(double), hipMemcpyHostToDevice));
  // single-instance staging for the host-pointer callbacks
  const size_t out_len = (size_t)std::max<int64_t>({L.Nz, L.Nc, L.nnzJ, L.nnzH, (int64_t)L.T, 1});
  HIP_TRY(hipMalloc((void**)&d_x1, std::max<size_t>(1, L.Nz) * sizeof(double)));
  HIP_TRY(hipMalloc((void**)&d_mu1, std::max<size_t>(1, L.Nc) * sizeof(double)));
  HIP_TRY(hipMalloc((void**)&d_out1, out_len * sizeof(double)));
  HIP_TRY(hipStreamCreate(&stream));
  dev_ready = true;
  return DTO_OK;
}

int Problem::ensure_scratch(int64_t B) {
  const size_t need = (size_t)B * (size_t)L.T;
  if (need <= scratch_len) return DTO_OK;
  if (d_scratch) HIP_TRY(hipFree(d_scratch));
  d_scratch = nullptr;
  scratch_len = 0;
  HIP_TRY(hipMalloc((void**)&d_scratch, need * sizeof(double)));
  scratch_len = need;
  return DTO_OK;
}

void Problem::fill_args(dto_eval_args& a, int64_t B, const double* z, int64_t ldz, const double* w, int64_t ldw) const {
  std::memset(&a, 0, sizeof(a));
  a.T = L.T;
  a.B = B;
  a.kind = d_kind; a.zoff = d_zoff; a.woff = d_woff; a.cdoff = d_cdoff; a.ccoff = d_ccoff;
  a.jdoff = d_jdoff; a.jcoff = d_jcoff; a.hoff = d_hoff;
  a.hmap_cost = d_hmap_cost; a.hmap_dyn_own = d_hmap_dyn_own; a.hmap_dyn_next = d_hmap_dyn_next; a.hmap_con = d_hmap_con;
  a.hmap_stride = L.hmap_stride;
  a.general_row0 = (int)(L.Ndyn + L.Nstage);
  a.general_jac0 = (int)(L.nnzJd + L.nnzJs);
  a.z = z; a.ldz = ldz;
  if (w) { a.w = w; a.ldw = ldw; } else { a.w = d_params; a.ldw = 0; }
  a.sigma = 1.0;
}

int Problem::launch(int op, const dto_eval_args& a, hipStream_t s) {
  if (!vt->launch)
    return set_error(DTO_ERR_UNSUPPORTED,
                     "wide-stage model: the evaluator callbacks are not built for it yet, only the KKT step (DESIGN.md)");
  const int rc = vt->launch(op, &a, (void*)s);
  if (rc != 0) return hip_fail((hipError_t)rc, "kernel launch");
  return DTO_OK;
}

// ---- KKT matrix in CSR form -------------------------------------------------------------------------------------------
// K = [ H + delta_w I   J' ;  J   -delta_c I ] over [z; constraint rows] -- the matrix the reference's scratch builds with
// spzeros + index loops (examples/pendulum/pendulum.jl:138-198) and hands to QDLDL.  Pattern: Hessian key (row-major sorted,
// both triangles: src/data.jl:184) merged with the primal diagonal, J' to the right of it, J below, the dual diagonal.
int Problem::build_kkt_csr() {
  if (!csr_rowptr.empty()) return DTO_OK;
  if (!L.hessian) return set_error(DTO_ERR_UNSUPPORTED, "the KKT matrix needs the Hessian of the Lagrangian (evaluate_hessian = true)");
  const int64_t n = L.Nz, m = L.Nc, dim = n + m;
  struct Ent { int64_t col; int h, j; signed char d; };
  std::vector<std::vector<Ent>> rows((size_t)dim);
  for (int64_t k = 0; k < L.nnzH; ++k) rows[(size_t)(L.hess_rows[k] - 1)].push_back({L.hess_cols[k] - 1, (int)k, -1, 0});
  for (int64_t i = 0; i < n; ++i) rows[(size_t)i].push_back({i, -1, -1, 1});
  for (int64_t k = 0; k < L.nnzJ; ++k) {
    const int64_t r = L.jac_rows[k] - 1, c = L.jac_cols[k] - 1;
    rows[(size_t)c].push_back({n + r, -1, (int)k, 0});       // J' block
    rows[(size_t)(n + r)].push_back({c, -1, (int)k, 0});     // J block
  }
  for (int64_t r = 0; r < m; ++r) rows[(size_t)(n + r)].push_back({n + r, -1, -1, 2});
  std::vector<int> sh, sj;
  std::vector<signed char> sd;
  csr_rowptr.assign((size_t)dim + 1, 1);
  for (int64_t i = 0; i < dim; ++i) {
    auto& rw = rows[(size_t)i];
    std::stable_sort(rw.begin(), rw.end(), [](const Ent& a, const Ent& b) { return a.col < b.col; });
    for (size_t k = 0; k < rw.size(); ++k) {
      if (k > 0 && rw[k].col == rw[k - 1].col) {   // the diagonal entry meets a Hessian key entry: one slot, both sources
        if (rw[k].h >= 0) sh.back() = rw[k].h;
        if (rw[k].j >= 0) sj.back() = rw[k].j;
        if (rw[k].d) sd.back() = rw[k].d;
        continue;
      }
      csr_col.push_back(rw[k].col + 1);
      sh.push_back(rw[k].h); sj.push_back(rw[k].j); sd.push_back(rw[k].d);
    }
    csr_rowptr[(size_t)i + 1] = (int64_t)csr_col.size() + 1;
  }
  int rc = ensure_device();
  if (rc) { csr_rowptr.clear(); csr_col.clear(); return rc; }
  const size_t nnz = csr_col.size();
  HIP_TRY(hipMalloc((void**)&d_csr_h, std::max<size_t>(1, nnz) * sizeof(int)));
  HIP_TRY(hipMalloc((void**)&d_csr_j, std::max<size_t>(1, nnz) * sizeof(int)));
  HIP_TRY(hipMalloc((void**)&d_csr_diag, std::max<size_t>(1, nnz)));
  HIP_TRY(hipMemcpy(d_csr_h, sh.data(), nnz * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_csr_j, sj.data(), nnz * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_csr_diag, sd.data(), nnz, hipMemcpyHostToDevice));
  return DTO_OK;
}

// one thread per CSR slot and instance: a gather, so every instance's value row is written with full coalescing
static __global__ void k_kkt_csr_values(int64_t nnz, const int* src_h, const int* src_j, const signed char* diag, const double* H,
                                        int64_t ldh, const double* J, int64_t ldj, double delta_w, double delta_c, double* vals,
                                        int64_t ldv) {
  const int64_t b = blockIdx.y;
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nnz) return;
  double v = 0.0;
  const int h = src_h[s], j = src_j[s];
  if (h >= 0) v += H[b * ldh + h];
  if (j >= 0) v += J[b * ldj + j];
  const signed char d = diag[s];
  if (d == 1) v += delta_w;
  else if (d == 2) v -= delta_c;
  vals[b * ldv + s] = v;
}

Problem::~Problem() {
  if (d_csr_h) (void)hipFree(d_csr_h);
  if (d_csr_j) (void)hipFree(d_csr_j);
  if (d_csr_diag) (void)hipFree(d_csr_diag);
  for (int* p : {d_kind, d_zoff, d_woff, d_cdoff, d_ccoff, d_jdoff, d_jcoff, d_hoff, d_hmap_cost, d_hmap_dyn_own,
                 d_hmap_dyn_next, d_hmap_con, d_csc_ptr, d_csc_k, d_csc_row, d_var_fixed})
    if (p) (void)hipFree(p);
  for (double* p : {d_params, d_x1, d_mu1, d_out1, d_scratch, wide_fac, border_ws})
    if (p) (void)hipFree(p);
  if (wide_flags) (void)hipFree(wide_flags);
  free_solver();
  delete trace;
  if (stream) (void)hipStreamDestroy(stream);
  if (dl) dlclose(dl);
}

}  // namespace dto

using dto::Problem;

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* dto_last_error(void) { return g_error.c_str(); }

int dto_problem_create(const dto_problem_spec* spec, dto_problem** out) {
  if (!spec || !out) return set_error(DTO_ERR_INVALID, "null argument");
  *out = nullptr;
  if (spec->abi_version != DTO_ABI_VERSION) return set_error(DTO_ERR_INVALID, "dto_problem_spec.abi_version mismatch");
  if (!spec->model_library || !spec->stage_kind) return set_error(DTO_ERR_INVALID, "model_library and stage_kind are required");
  std::unique_ptr<Problem> p(new Problem());
  p->dl = dlopen(spec->model_library, RTLD_NOW | RTLD_LOCAL);
  if (!p->dl) return set_error(DTO_ERR_PLUGIN, std::string("dlopen failed: ") + dlerror());
  typedef const dto_model_vtable* (*get_fn)(void);
  get_fn get = (get_fn)dlsym(p->dl, "dto_model_get");
  if (!get) return set_error(DTO_ERR_PLUGIN, "plugin does not export dto_model_get");
  p->vt = get();
  if (!p->vt || p->vt->abi != DTO_PLUGIN_ABI) return set_error(DTO_ERR_PLUGIN, "plugin ABI version mismatch");
  if (!p->L.build(p->vt, spec->horizon, spec->stage_kind, spec->evaluate_hessian != 0, spec->variable_lower,
                  spec->variable_upper, spec->parameters, spec->parameters ? spec->num_parameters : -1))
    return set_error(DTO_ERR_INVALID, p->L.error);
  *out = reinterpret_cast<dto_problem*>(p.release());
  return DTO_OK;
}

int dto_problem_destroy(dto_problem* h) {
  delete reinterpret_cast<Problem*>(h);
  return DTO_OK;
}

int dto_sizes(const dto_problem* h, dto_sizes_t* o) {
  if (!h || !o) return set_error(DTO_ERR_INVALID, "null argument");
  const dto::Layout& L = reinterpret_cast<const Problem*>(h)->L;
  o->num_variables = L.Nz; o->num_parameters = L.Nw;
  o->num_constraint = L.Nc; o->num_constraint_dynamics = L.Ndyn; o->num_constraint_stage = L.Nstage;
  o->num_constraint_general = L.Ngen;
  o->num_jacobian = L.nnzJ; o->num_jacobian_dynamics = L.nnzJd; o->num_jacobian_stage = L.nnzJs;
  o->num_jacobian_general = L.nnzJg;
  o->nnz_hess_key = L.nnzH; o->nnz_hess_raw = L.nnzH_raw;
  o->horizon = L.T; o->num_state_max = L.max_nx; o->num_action_max = L.max_nu;
  return DTO_OK;
}

int dto_features_available(const dto_problem* h, int* bits) {
  if (!h || !bits) return set_error(DTO_ERR_INVALID, "null argument");
  const Problem* p = reinterpret_cast<const Problem*>(h);
  *bits = DTO_FEATURE_GRAD | DTO_FEATURE_JAC | (p->L.hessian ? DTO_FEATURE_HESS : 0);
  return DTO_OK;
}

int dto_jacobian_structure(const dto_problem* h, int64_t* rows, int64_t* cols) {
  if (!h || !rows || !cols) return set_error(DTO_ERR_INVALID, "null argument");
  const dto::Layout& L = reinterpret_cast<const Problem*>(h)->L;
  std::copy(L.jac_rows.begin(), L.jac_rows.end(), rows);
  std::copy(L.jac_cols.begin(), L.jac_cols.end(), cols);
  return DTO_OK;
}

int dto_hessian_structure(const dto_problem* h, int64_t* rows, int64_t* cols) {
  if (!h || !rows || !cols) return set_error(DTO_ERR_INVALID, "null argument");
  const dto::Layout& L = reinterpret_cast<const Problem*>(h)->L;
  std::copy(L.hess_rows.begin(), L.hess_rows.end(), rows);
  std::copy(L.hess_cols.begin(), L.hess_cols.end(), cols);
  return DTO_OK;
}

int dto_variable_bounds(const dto_problem* h, double* lo, double* hi) {
  if (!h || !lo || !hi) return set_error(DTO_ERR_INVALID, "null argument");
  const dto::Layout& L = reinterpret_cast<const Problem*>(h)->L;
  std::copy(L.var_lo.begin(), L.var_lo.end(), lo);
  std::copy(L.var_hi.begin(), L.var_hi.end(), hi);
  return DTO_OK;
}

int dto_constraint_bounds(const dto_problem* h, double* lo, double* hi) {
  if (!h || !lo || !hi) return set_error(DTO_ERR_INVALID, "null argument");
  const dto::Layout& L = reinterpret_cast<const Problem*>(h)->L;
  std::copy(L.con_lo.begin(), L.con_lo.end(), lo);
  std::copy(L.con_hi.begin(), L.con_hi.end(), hi);
  return DTO_OK;
}

int dto_stage_indices(const dto_problem* h, int which, int t1, int64_t* out, int64_t* n) {
  if (!h || !n) return set_error(DTO_ERR_INVALID, "null argument");
  const dto::Layout& L = reinterpret_cast<const Problem*>(h)->L;
  const int t = t1 - 1;
  if (t < 0 || t >= L.T) return set_error(DTO_ERR_INVALID, "stage out of range");
  std::vector<int64_t> v;
  auto range = [&](int64_t first0, int64_t count) { for (int64_t i = 0; i < count; ++i) v.push_back(first0 + i + 1); };
  const dto_kind& k = L.vt->kinds[L.kind[t]];
  switch (which) {
    case DTO_IDX_STATE: range(L.zoff[t], L.nx[t]); break;
    case DTO_IDX_ACTION: range(L.zoff[t] + L.nx[t], L.nu[t]); break;
    case DTO_IDX_STATE_ACTION: range(L.zoff[t], L.nx[t] + L.nu[t]); break;
    case DTO_IDX_STATE_ACTION_NEXT:
      if (t >= L.T - 1) return set_error(DTO_ERR_INVALID, "state_action_next_state is defined for t < T");
      range(L.zoff[t], L.nx[t] + L.nu[t] + L.nx[t + 1]);
      break;
    case DTO_IDX_DYNAMICS_CONSTRAINT: range(L.cdoff[t], L.cdoff[t + 1] - L.cdoff[t]); break;
    case DTO_IDX_DYNAMICS_JACOBIAN: range(L.jdoff[t], L.jdoff[t + 1] - L.jdoff[t]); break;
    case DTO_IDX_DYNAMICS_HESSIAN: v = L.dyn_h.empty() ? v : L.dyn_h[t]; break;
    case DTO_IDX_STAGE_CONSTRAINT: range(L.ccoff[t], L.ccoff[t + 1] - L.ccoff[t]); break;
    case DTO_IDX_STAGE_JACOBIAN: range(L.jcoff[t], L.jcoff[t + 1] - L.jcoff[t]); break;
    case DTO_IDX_STAGE_HESSIAN: v = L.con_h.empty() ? v : L.con_h[t]; break;
    case DTO_IDX_OBJECTIVE_HESSIAN: v = L.obj_h.empty() ? v : L.obj_h[t]; break;
    default: return set_error(DTO_ERR_INVALID, "unknown index kind");
  }
  (void)k;
  *n = (int64_t)v.size();
  if (out) std::copy(v.begin(), v.end(), out);
  return DTO_OK;
}

// ---- batched device-pointer evaluators ----------------------------------------------------------
static int check_batch(Problem* p, const dto_batch* b) {
  if (!p || !b || !b->x) return set_error(DTO_ERR_INVALID, "null argument");
  if (b->B <= 0) return set_error(DTO_ERR_INVALID, "batch size must be positive");
  if (b->ldx < p->L.Nz) return set_error(DTO_ERR_INVALID, "ldx < num_variables");
  if (b->params && b->ldp < p->L.Nw) return set_error(DTO_ERR_INVALID, "ldp < num_parameters");
  return p->ensure_device();
}

int dto_eval_f_batch(dto_problem* h, const dto_batch* b, double* f) {
  Problem* p = reinterpret_cast<Problem*>(h);
  int rc = check_batch(p, b);
  if (rc) return rc;
  if ((rc = p->ensure_scratch(b->B))) return rc;
  dto_eval_args a;
  p->fill_args(a, b->B, b->x, b->ldx, b->params, b->ldp);
  if (!f) return set_error(DTO_ERR_INVALID, "null output");
  a.scratch = p->d_scratch; a.out = f; a.ldout = 1;
  return p->launch(DTO_OP_OBJ, a, (hipStream_t)b->stream);
}

int dto_eval_grad_f_batch(dto_problem* h, const dto_batch* b, double* g, int64_t ldg) {
  Problem* p = reinterpret_cast<Problem*>(h);
  int rc = check_batch(p, b);
  if (rc) return rc;
  if (!g || ldg < p->L.Nz) return set_error(DTO_ERR_INVALID, "bad gradient buffer");
  dto_eval_args a;
  p->fill_args(a, b->B, b->x, b->ldx, b->params, b->ldp);
  a.out = g; a.ldout = ldg;
  return p->launch(DTO_OP_GRAD, a, (hipStream_t)b->stream);
}

int dto_eval_g_batch(dto_problem* h, const dto_batch* b, double* c, int64_t ldc) {
  Problem* p = reinterpret_cast<Problem*>(h);
  int rc = check_batch(p, b);
  if (rc) return rc;
  if (!c || ldc < p->L.Nc) return set_error(DTO_ERR_INVALID, "bad constraint buffer");
  dto_eval_args a;
  p->fill_args(a, b->B, b->x, b->ldx, b->params, b->ldp);
  a.out = c; a.ldout = ldc;
  if ((rc = p->launch(DTO_OP_CON, a, (hipStream_t)b->stream))) return rc;
  if (p->L.Ngen) return p->launch(DTO_OP_GENERAL_CON, a, (hipStream_t)b->stream);
  return DTO_OK;
}

int dto_kkt_csr_structure(dto_problem* h, int64_t* row_ptr, int64_t* col_ind, int64_t* dim, int64_t* nnz) {
  Problem* p = reinterpret_cast<Problem*>(h);
  if (!p) return set_error(DTO_ERR_INVALID, "null argument");
  int rc = p->build_kkt_csr();
  if (rc) return rc;
  if (dim) *dim = p->L.Nz + p->L.Nc;
  if (nnz) *nnz = (int64_t)p->csr_col.size();
  if (row_ptr) std::copy(p->csr_rowptr.begin(), p->csr_rowptr.end(), row_ptr);
  if (col_ind) std::copy(p->csr_col.begin(), p->csr_col.end(), col_ind);
  return DTO_OK;
}

int dto_kkt_csr_values_batch(dto_problem* h, int64_t B, const double* H, int64_t ldh, const double* J, int64_t ldj,
                             double delta_w, double delta_c, double* vals, int64_t ldv, void* stream) {
  Problem* p = reinterpret_cast<Problem*>(h);
  if (!p || !H || !J || !vals || B < 1) return set_error(DTO_ERR_INVALID, "null argument");
  int rc = p->build_kkt_csr();
  if (rc) return rc;
  const int64_t nnz = (int64_t)p->csr_col.size();
  if (ldh < p->L.nnzH || ldj < p->L.nnzJ || ldv < nnz) return set_error(DTO_ERR_INVALID, "leading dimension too small");
  if (B > 65535) return set_error(DTO_ERR_INVALID, "at most 65535 instances per call");
  dim3 grid((unsigned)((nnz + 255) / 256), (unsigned)B);
  hipLaunchKernelGGL(dto::k_kkt_csr_values, grid, dim3(256), 0, (hipStream_t)stream, nnz, (const int*)p->d_csr_h,
                     (const int*)p->d_csr_j, (const signed char*)p->d_csr_diag, H, ldh, J, ldj, delta_w, delta_c, vals, ldv);
  HIP_TRY(hipGetLastError());
  return DTO_OK;
}

int dto_eval_jac_g_batch(dto_problem* h, const dto_batch* b, double* J, int64_t ldj) {
  Problem* p = reinterpret_cast<Problem*>(h);
  int rc = check_batch(p, b);
  if (rc) return rc;
  if (!J || ldj < p->L.nnzJ) return set_error(DTO_ERR_INVALID, "bad Jacobian buffer");
  dto_eval_args a;
  p->fill_args(a, b->B, b->x, b->ldx, b->params, b->ldp);
  a.out = J; a.ldout = ldj;
  if ((rc = p->launch(DTO_OP_JAC, a, (hipStream_t)b->stream))) return rc;
  if (p->L.Ngen) return p->launch(DTO_OP_GENERAL_JAC, a, (hipStream_t)b->stream);
  return DTO_OK;
}

int dto_eval_h_batch(dto_problem* h, const dto_batch* b, double sigma, const double* mu, int64_t ldmu, double* H,
                     int64_t ldh) {
  Problem* p = reinterpret_cast<Problem*>(h);
  int rc = check_batch(p, b);
  if (rc) return rc;
  if (!p->L.hessian) return set_error(DTO_ERR_UNSUPPORTED, "problem was created with evaluate_hessian = false");
  if (!mu || ldmu < p->L.Nc || !H || ldh < p->L.nnzH) return set_error(DTO_ERR_INVALID, "bad Hessian buffers");
  dto_eval_args a;
  p->fill_args(a, b->B, b->x, b->ldx, b->params, b->ldp);
  a.mu = mu; a.ldmu = ldmu; a.sigma = sigma;
  a.out = H; a.ldout = ldh;
  return p->launch(DTO_OP_HESS, a, (hipStream_t)b->stream);
}

// ---- single-instance host-pointer callbacks (the MOI methods) ------------------------------------
static int host_call(Problem* p, const double* x, const double* mu, double sigma, int which, double* out, int64_t nout) {
  if (!p || !x || !out) return set_error(DTO_ERR_INVALID, "null argument");
  int rc = p->ensure_device();
  if (rc) return rc;
  const dto::Layout& L = p->L;
  HIP_TRY(hipMemcpyAsync(p->d_x1, x, L.Nz * sizeof(double), hipMemcpyHostToDevice, p->stream));
  if (mu) HIP_TRY(hipMemcpyAsync(p->d_mu1, mu, L.Nc * sizeof(double), hipMemcpyHostToDevice, p->stream));
  dto_batch b;
  b.B = 1; b.x = p->d_x1; b.ldx = L.Nz; b.params = nullptr; b.ldp = 0; b.stream = (void*)p->stream;
  dto_problem* h = reinterpret_cast<dto_problem*>(p);
  switch (which) {
    case DTO_OP_OBJ: rc = dto_eval_f_batch(h, &b, p->d_out1); break;
    case DTO_OP_GRAD: rc = dto_eval_grad_f_batch(h, &b, p->d_out1, L.Nz); break;
    case DTO_OP_CON: rc = dto_eval_g_batch(h, &b, p->d_out1, L.Nc); break;
    case DTO_OP_JAC: rc = dto_eval_jac_g_batch(h, &b, p->d_out1, L.nnzJ); break;
    case DTO_OP_HESS: rc = dto_eval_h_batch(h, &b, sigma, p->d_mu1, L.Nc, p->d_out1, L.nnzH); break;
    default: rc = set_error(DTO_ERR_INVALID, "bad op");
  }
  if (rc) return rc;
  if (nout) HIP_TRY(hipMemcpyAsync(out, p->d_out1, nout * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return DTO_OK;
}

int dto_eval_f(dto_problem* h, const double* x, double* f) {
  return host_call(reinterpret_cast<Problem*>(h), x, nullptr, 1.0, DTO_OP_OBJ, f, 1);
}
int dto_eval_grad_f(dto_problem* h, const double* x, double* g) {
  Problem* p = reinterpret_cast<Problem*>(h);
  return host_call(p, x, nullptr, 1.0, DTO_OP_GRAD, g, p ? p->L.Nz : 0);
}
int dto_eval_g(dto_problem* h, const double* x, double* c) {
  Problem* p = reinterpret_cast<Problem*>(h);
  return host_call(p, x, nullptr, 1.0, DTO_OP_CON, c, p ? p->L.Nc : 0);
}
int dto_eval_jac_g(dto_problem* h, const double* x, double* J) {
  Problem* p = reinterpret_cast<Problem*>(h);
  return host_call(p, x, nullptr, 1.0, DTO_OP_JAC, J, p ? p->L.nnzJ : 0);
}
int dto_eval_h(dto_problem* h, const double* x, double sigma, const double* mu, double* H) {
  Problem* p = reinterpret_cast<Problem*>(h);
  if (!mu) return set_error(DTO_ERR_INVALID, "null multipliers");
  return host_call(p, x, mu, sigma, DTO_OP_HESS, H, p ? p->L.nnzH : 0);
}

// ---- device helpers ------------------------------------------------------------------------------
int dto_device_alloc(void** ptr, int64_t bytes) {
  if (!ptr || bytes < 0) return set_error(DTO_ERR_INVALID, "bad argument");
  HIP_TRY(hipMalloc(ptr, (size_t)std::max<int64_t>(bytes, 1)));
  return DTO_OK;
}
int dto_device_free(void* ptr) {
  HIP_TRY(hipFree(ptr));
  return DTO_OK;
}
int dto_copy_to_device(void* dst, const void* src, int64_t bytes) {
  HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice));
  return DTO_OK;
}
int dto_copy_to_host(void* dst, const void* src, int64_t bytes) {
  HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
  return DTO_OK;
}
int dto_device_synchronize(void) {
  HIP_TRY(hipDeviceSynchronize());
  return DTO_OK;
}
int dto_device_count(int* n) {
  if (!n) return set_error(DTO_ERR_INVALID, "null argument");
  *n = 0;
  hipError_t e = hipGetDeviceCount(n);
  if (e != hipSuccess) *n = 0;
  return DTO_OK;
}

}  // extern "C"
