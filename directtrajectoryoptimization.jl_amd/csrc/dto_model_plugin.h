/*
 * Internal ABI between the runtime (libdto_hip.so) and a generated model plugin.
 *
 * A plugin is one shared object produced by plugin.py from the traced per-stage objects of a
 * problem (the role the `eval`'d Symbolics closures play in the reference, src/dynamics.jl:26-34).
 * It carries (a) the class tables: dims and local sparsity patterns exactly as the reference keeps
 * them on Dynamics/Cost/Constraint structs (src/dynamics.jl:1-16, src/costs.jl:1-11,
 * src/constraints.jl:1-17), 1-based, CSC order, and (b) launchers for the hand-written stage
 * kernels of csrc/dto_eval_kernels.hpp instantiated for this model's expression code.
 */
#ifndef DTO_MODEL_PLUGIN_H
#define DTO_MODEL_PLUGIN_H

#include <stdint.h>

#define DTO_PLUGIN_ABI 5

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dto_dyn_class {
  int num_next_state, num_state, num_action, num_parameter;
  int num_jacobian, num_hessian;
  const int* jac_rows; const int* jac_cols;   /* 1-based, columns over [x; u; y] */
  const int* hess_rows; const int* hess_cols; /* 1-based, full symmetric */
} dto_dyn_class;

typedef struct dto_cost_class {
  int num_state, num_action, num_parameter;
  int num_hessian;
  const int* hess_rows; const int* hess_cols; /* over [x; u] */
} dto_cost_class;

typedef struct dto_con_class {
  int num_state, num_action, num_parameter;
  int num_constraint, num_jacobian, num_hessian;
  const int* jac_rows; const int* jac_cols;
  const int* hess_rows; const int* hess_cols;
  int num_inequality; const int* indices_inequality; /* 1-based rows that are <= 0 */
} dto_con_class;

typedef struct dto_general_class {
  int num_variables, num_parameter;
  int num_constraint, num_jacobian, num_hessian;
  const int* jac_rows; const int* jac_cols;
  const int* hess_rows; const int* hess_cols;
  int num_inequality; const int* indices_inequality;
} dto_general_class;

/* one stage kind = which classes meet at a knot; -1 = none */
typedef struct dto_kind {
  int dyn;       /* dynamics class of d_t(x_t,u_t,x_{t+1}), -1 at t = T */
  int prev_dyn;  /* dynamics class of d_{t-1}, -1 at t = 1 */
  int cost;
  int con;       /* -1 = Constraint() */
} dto_kind;

/* operations a plugin can launch */
enum dto_op {
  DTO_OP_OBJ = 0,   /* per-stage costs -> scratch[B][T], then summed in stage order -> out[B] */
  DTO_OP_GRAD = 1,
  DTO_OP_CON = 2,
  DTO_OP_JAC = 3,
  DTO_OP_HESS = 4,
  DTO_OP_GENERAL_CON = 5,
  DTO_OP_GENERAL_JAC = 6,
  DTO_OP_COUNT
};

/* device-side argument block of the AoS (instance-major) evaluator kernels */
typedef struct dto_eval_args {
  int T;
  int64_t B;
  /* per-stage tables, device int32 */
  const int* kind;   /* [T]   */
  const int* zoff;   /* [T+1] offset of x_t in z; zoff[T] = num_variables */
  const int* woff;   /* [T+1] offset of w_t in the flattened parameters */
  const int* cdoff;  /* [T+1] dynamics rows of stage t start here (0-based); cdoff[T-1] = cdoff[T] = N_dyn */
  const int* ccoff;  /* [T+1] stage-constraint rows, absolute (already shifted by N_dyn) */
  const int* jdoff;  /* [T+1] dynamics Jacobian slots */
  const int* jcoff;  /* [T+1] stage Jacobian slots, absolute (shifted by nnz of dynamics) */
  const int* hoff;   /* [T+1] Hessian key slots owned by the rows of stage t */
  /* Hessian scatter maps, device int32 [n_kind][hmap_stride]: slot relative to hoff[t], -1 = not here */
  const int* hmap_cost;      /* cost(t)  local nnz -> rows of stage t */
  const int* hmap_dyn_own;   /* dyn(t)   local nnz with row in [x_t;u_t] -> rows of stage t */
  const int* hmap_dyn_next;  /* indexed by kind(t+1): dyn(t) local nnz with row in y -> rows of stage t+1 */
  const int* hmap_con;       /* con(t)   local nnz -> rows of stage t */
  int hmap_stride;
  int general_row0, general_jac0; /* offsets of the general block in c and J */
  /* data */
  const double* z;  int64_t ldz;
  const double* w;  int64_t ldw;   /* ldw = 0: shared */
  const double* mu; int64_t ldmu;
  double sigma;
  double* out;      int64_t ldout;
  double* scratch;  /* [B][T] per-stage costs for DTO_OP_OBJ */
} dto_eval_args;

struct dto_kkt_args;
struct dto_kkt_info;
struct dto_wide_args;
struct dto_wide_info;
struct dto_im_args;
struct dto_im_info;

typedef struct dto_model_vtable {
  int abi;            /* DTO_PLUGIN_ABI */
  const char* name;
  int n_dyn, n_cost, n_con, n_kind;
  const dto_dyn_class* dyn;
  const dto_cost_class* cost;
  const dto_con_class* con;
  const dto_kind* kinds;
  const dto_general_class* general; /* NULL = GeneralConstraint() */
  int evaluate_hessian;
  int max_key; /* LDS bound the Hessian kernel was compiled with: max key slots owned by one stage */
  /* launch `op` on `stream` (hipStream_t); returns 0 or a hipError_t value */
  int (*launch)(int op, const dto_eval_args* args, void* stream);
  /* KKT / solver kernels (dto_kkt_kernels.hpp), same convention */
  int (*launch_kkt)(int op, const struct dto_kkt_args* args, void* stream);
  int (*kkt_info)(struct dto_kkt_info* out);
  /* wide-stage (tile / MFMA) KKT kernels of dto_wide_kernels.hpp; NULL for register-path models, and then
   * `launch`/`launch_kkt` are NULL for wide models */
  int (*launch_wide)(int op, const struct dto_wide_args* args, void* stream);
  int (*wide_info)(struct dto_wide_info* out);
  /* instance-major engine of the solver path (dto_im_kernels.hpp); NULL where a plugin has none */
  int (*launch_im)(int op, const struct dto_im_args* args, void* stream);
  int (*im_info)(struct dto_im_info* out);
} dto_model_vtable;

/* the one symbol every plugin exports */
const dto_model_vtable* dto_model_get(void);

#ifdef __cplusplus
}
#endif
#endif
