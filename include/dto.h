/*
 * dto.h -- C ABI of the MI355X-native sparse NLP callback + KKT engine.
 *
 * Drop-in boundary for DirectTrajectoryOptimization.jl's collocation hot path.  The reference
 * has no FFI of its own: its boundary is the MathOptInterface evaluator protocol implemented on
 * `NLPData` (reference src/data.jl:106, src/moi.jl:1-125) and consumed by Ipopt.jl.  Every entry
 * point below names the reference method it replaces; a Julia `ccall` shim that implements the
 * MOI methods on top of these calls is shown in INTEGRATION.md.
 *
 * Conventions (same as the reference unless stated):
 *   - all numerics are double, all structure indices are int64 and 1-BASED (Julia convention),
 *   - variables      z = [x_1; u_1; ...; x_{T-1}; u_{T-1}; x_T]          (src/dynamics.jl:188-195)
 *   - constraints    [dynamics t=1..T-1; stage t=1..T; general]           (src/data.jl:64-75)
 *   - Jacobian       COO list, dynamics ++ stage ++ general, local CSC order (src/data.jl:170-175)
 *   - Hessian        sort(unique(raw)) key, row-major, BOTH triangles       (src/data.jl:184)
 *   - output buffers are caller-owned and fully overwritten (src/moi.jl:16,33,53,73),
 *   - every function returns an int status (DTO_OK = 0); nothing throws across the ABI,
 *   - a handle is not thread-safe (the reference evaluator is not re-entrant either).
 *
 * Pointers are HOST pointers for the dto_eval_* family (the MOI-callback replacement: one
 * instance, copied over PCIe) and DEVICE pointers for the *_batch / kkt / solve families
 * (B instances resident in HBM, instance-major: instance b starts at ptr + b*ld).
 */
#ifndef DTO_H
#define DTO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DTO_ABI_VERSION 4

enum dto_status {
  DTO_OK = 0,
  DTO_ERR_INVALID = 1,      /* bad argument / inconsistent spec */
  DTO_ERR_PLUGIN = 2,       /* model plugin missing or wrong ABI */
  DTO_ERR_DEVICE = 3,       /* HIP error (no GPU, OOM, launch failure) */
  DTO_ERR_UNSUPPORTED = 4,  /* feature outside the built path */
  DTO_ERR_NOT_CONVERGED = 5
};

/* feature bits, MOI.features_available (src/moi.jl:122) */
#define DTO_FEATURE_GRAD 1
#define DTO_FEATURE_JAC 2
#define DTO_FEATURE_HESS 4

typedef struct dto_problem dto_problem; /* opaque; plays the role of NLPData (src/data.jl:106-121) */

/*
 * Problem description: what `Solver(dynamics, objective, constraints, bounds; ...)` receives
 * (src/solver.jl:6-21) after the per-stage objects have been compiled into a model plugin.
 * `stage_kind[t]` selects, for stage t (0-based here), one entry of the plugin's kind table
 * = (dynamics class, previous dynamics class, cost class, constraint class).
 */
typedef struct dto_problem_spec {
  int abi_version;            /* DTO_ABI_VERSION */
  const char* model_library;  /* path of the generated plugin .so (gfx950 code object inside) */
  int horizon;                /* T */
  const int32_t* stage_kind;  /* [T] */
  const double* variable_lower; /* [num_variables] or NULL = -Inf  (src/data.jl:123-133) */
  const double* variable_upper; /* [num_variables] or NULL = +Inf */
  const double* parameters;     /* [num_parameters] flattened w_1..w_T (src/data.jl:218) or NULL */
  int64_t num_parameters;       /* length of `parameters`; must equal dto_sizes_t.num_parameters (sum over the stages of the
                                   largest num_parameter among the stage's dynamics, cost and constraint) when it is given */
  int evaluate_hessian;         /* Solver(...; evaluate_hessian) (src/solver.jl:7) */
} dto_problem_spec;

int dto_problem_create(const dto_problem_spec* spec, dto_problem** out);
int dto_problem_destroy(dto_problem* p);
const char* dto_last_error(void);

/* totals of NLPData (src/data.jl:155-167,187): nnz_hess_key = length(hessian_lagrangian_structure),
 * nnz_hess_raw = nlp.num_hessian_lagrangian (duplicate-counting). */
typedef struct dto_sizes_t {
  int64_t num_variables, num_parameters;
  int64_t num_constraint, num_constraint_dynamics, num_constraint_stage, num_constraint_general;
  int64_t num_jacobian, num_jacobian_dynamics, num_jacobian_stage, num_jacobian_general;
  int64_t nnz_hess_key, nnz_hess_raw;
  int64_t horizon, num_state_max, num_action_max;
} dto_sizes_t;
int dto_sizes(const dto_problem* p, dto_sizes_t* out);

/* MOI.features_available (src/moi.jl:122) */
int dto_features_available(const dto_problem* p, int* feature_bits);
/* MOI.jacobian_structure (src/moi.jl:124): rows/cols [num_jacobian], 1-based */
int dto_jacobian_structure(const dto_problem* p, int64_t* rows, int64_t* cols);
/* MOI.hessian_lagrangian_structure (src/moi.jl:125): rows/cols [nnz_hess_key], 1-based */
int dto_hessian_structure(const dto_problem* p, int64_t* rows, int64_t* cols);
/* The KKT matrix of the reference's scratch (examples/pendulum/pendulum.jl:138-198, built there with spzeros + index loops and
 * handed to QDLDL), in CSR form and in the reference ordering [z; dynamics rows; stage rows; general rows]:
 *     K = [ H + delta_w I   J' ;  J   -delta_c I ],   dimension num_variables + num_constraint,
 * full symmetric pattern (both triangles, like the Hessian key: src/data.jl:184), columns sorted inside every row, 1-based.
 * dto_kkt_csr_structure: row_ptr [dim + 1], col_ind [nnz]; either may be NULL to query *dim / *nnz only (HOST arrays).
 * dto_kkt_csr_values_batch: values [B][ldv] (DEVICE) from evaluated Hessian-of-the-Lagrangian values H [B][ldh] (order of
 * dto_hessian_structure: dto_eval_h_batch) and Jacobian values J [B][ldj] (dto_eval_jac_g_batch): one gather per CSR slot,
 * fully coalesced stores -- for a caller that wants K itself (an A/B against QDLDL / MUMPS, or its own factorisation). */
int dto_kkt_csr_structure(dto_problem* p, int64_t* row_ptr, int64_t* col_ind, int64_t* dim, int64_t* nnz);
int dto_kkt_csr_values_batch(dto_problem* p, int64_t B, const double* H, int64_t ldh, const double* J, int64_t ldj,
                             double delta_w, double delta_c, double* values, int64_t ldv, void* stream);
/* nlp.variable_bounds / nlp.constraint_bounds (src/data.jl:123-148) */
int dto_variable_bounds(const dto_problem* p, double* lower, double* upper);
int dto_constraint_bounds(const dto_problem* p, double* lower, double* upper);

/* TrajectoryOptimizationIndices (src/data.jl:44-104): 1-based index vector of stage t (1-based t).
 * Returns the length in *n; `out` may be NULL to query the length. */
enum dto_index_kind {
  DTO_IDX_STATE = 0,                 /* indices.states[t]               src/dynamics.jl:188-191 */
  DTO_IDX_ACTION = 1,                /* indices.actions[t]              src/dynamics.jl:193-195 */
  DTO_IDX_STATE_ACTION = 2,          /* indices.state_action[t]         src/dynamics.jl:197-200 */
  DTO_IDX_STATE_ACTION_NEXT = 3,     /* indices.state_action_next_state src/dynamics.jl:202-204 */
  DTO_IDX_DYNAMICS_CONSTRAINT = 4,   /* indices.dynamics_constraints[t] src/dynamics.jl:162-165 */
  DTO_IDX_DYNAMICS_JACOBIAN = 5,     /* indices.dynamics_jacobians[t]   src/dynamics.jl:167-170 */
  DTO_IDX_DYNAMICS_HESSIAN = 6,      /* indices.dynamics_hessians[t]    src/dynamics.jl:172-186 */
  DTO_IDX_STAGE_CONSTRAINT = 7,      /* indices.stage_constraints[t]    src/constraints.jl:141-153 */
  DTO_IDX_STAGE_JACOBIAN = 8,        /* indices.stage_jacobians[t]      src/constraints.jl:155-166 */
  DTO_IDX_STAGE_HESSIAN = 9,         /* indices.stage_hessians[t]       src/constraints.jl:168-183 */
  DTO_IDX_OBJECTIVE_HESSIAN = 10     /* indices.objective_hessians[t]   src/costs.jl:88-104 */
};
int dto_stage_indices(const dto_problem* p, int which, int t, int64_t* out, int64_t* n);

/* ---- the five MOI evaluator methods, one instance, HOST pointers --------------------------- */
/* MOI.eval_objective (src/moi.jl:1-13) */
int dto_eval_f(dto_problem* p, const double* x, double* f);
/* MOI.eval_objective_gradient (src/moi.jl:15-30): g[num_variables] */
int dto_eval_grad_f(dto_problem* p, const double* x, double* g);
/* MOI.eval_constraint (src/moi.jl:32-50): c[num_constraint] */
int dto_eval_g(dto_problem* p, const double* x, double* c);
/* MOI.eval_constraint_jacobian (src/moi.jl:52-70): J[num_jacobian], order of dto_jacobian_structure */
int dto_eval_jac_g(dto_problem* p, const double* x, double* J);
/* MOI.eval_hessian_lagrangian (src/moi.jl:72-120): H[nnz_hess_key], order of dto_hessian_structure */
int dto_eval_h(dto_problem* p, const double* x, double sigma, const double* mu, double* H);

/* ---- batched forms: B independent instances, DEVICE pointers, asynchronous on `stream` -------
 * x: [B][ldx] (ldx >= num_variables); outputs instance-major with the given leading dimension.
 * params: NULL = the spec's parameters shared by all instances, else [B][ldp] (also honoured by the solver entry points
 * dto_kkt_step_batch / dto_solve_batch / dto_solver_begin: one parameter set per instance, e.g. MPC rollouts).
 * These run the same kernels as above without the PCIe copies. `stream` is a hipStream_t. */
typedef struct dto_batch {
  int64_t B;
  const double* x;      int64_t ldx;
  const double* params; int64_t ldp;
  void* stream;
} dto_batch;
int dto_eval_f_batch(dto_problem* p, const dto_batch* b, double* f /* [B] */);
int dto_eval_grad_f_batch(dto_problem* p, const dto_batch* b, double* g, int64_t ldg);
int dto_eval_g_batch(dto_problem* p, const dto_batch* b, double* c, int64_t ldc);
int dto_eval_jac_g_batch(dto_problem* p, const dto_batch* b, double* J, int64_t ldj);
/* sigma: host scalar applied to every instance; mu: [B][ldmu] */
int dto_eval_h_batch(dto_problem* p, const dto_batch* b, double sigma, const double* mu, int64_t ldmu,
                     double* H, int64_t ldh);

/* ---- KKT step and solver: the work the reference delegates to Ipopt (src/solver.jl:45-47,
 *      src/data.jl:229-255) -------------------------------------------------------------------- */

/* Solver options: the subset of reference `Options` (src/options.jl:6-36) that defines convergence,
 * plus the interior-point constants Ipopt documents as defaults.  Print/file options of the
 * reference are out of scope. */
typedef struct dto_options {
  double tol;               /* 1e-6  src/options.jl:7  */
  double s_max;             /* 100   src/options.jl:8  */
  int max_iter;             /* 1000  src/options.jl:9  */
  double dual_inf_tol;      /* 1.0   src/options.jl:12 */
  double constr_viol_tol;   /* 1e-3  src/options.jl:13 */
  double compl_inf_tol;     /* 1e-3  src/options.jl:14 */
  double mu_init;           /* 0.1   (Ipopt default)   */
  double delta_c;           /* 1e-8  dual regularisation, examples/pendulum/pendulum.jl:195 uses 1e-5 */
  double delta_w_init;      /* 1e-4  first primal regularisation tried when the inertia is wrong */
  int check_every;          /* host polls the batch for completion every this many iterations */
  double max_cpu_time;      /* 300   src/options.jl:10: wall-clock limit of one dto_solve[_batch] call in seconds; instances
                               still running when it expires are returned as they are with DTO_STATUS_CPU_TIME (6) */
  /* Ipopt's "acceptable" termination (src/options.jl:15-20): an instance stops with status 4 after `acceptable_iter`
   * consecutive iterations whose scaled error is <= acceptable_tol, whose unscaled residuals are within the three
   * acceptable_*_tol values and whose objective changed by less than acceptable_obj_change_tol (relative). 0 = off. */
  double acceptable_tol;             /* 1e-6  src/options.jl:15 */
  int acceptable_iter;               /* 15    src/options.jl:16 */
  double acceptable_dual_inf_tol;    /* 1e10  src/options.jl:17 */
  double acceptable_constr_viol_tol; /* 1e-2  src/options.jl:18 */
  double acceptable_compl_inf_tol;   /* 1e-2  src/options.jl:19 */
  double acceptable_obj_change_tol;  /* 1e-5  src/options.jl:20 */
  double diverging_iterates_tol;     /* 1e8   src/options.jl:21: status 5 once max|z_i| exceeds it */
  double mu_target;                  /* 1e-4  src/options.jl:22: the barrier parameter is not driven below it and the
                                        complementarity in every termination test is measured against it (Ipopt's
                                        mu_target semantics); only matters for problems with bounds / inequality rows */
  /* ABI 3: globalisation.  DTO_LS_FILTER = Ipopt's filter line search from the first iteration (what the reference runs,
   * src/solver.jl:45-47).  DTO_LS_PENALTY_FILTER (default) = while the iterate is far from the constraint manifold
   * (max |c_i| > penalty_switch_theta) the step size is chosen on the l1 exact-penalty function, then the filter takes over:
   * same minimisers class, acrobot T = 1000 from the reference's straight-line guess in a median of ~60 instead of ~630
   * iterations.  Lane-per-instance solver path; the tile (64-state) and bordered paths run the filter. */
  int line_search;                   /* DTO_LS_PENALTY_FILTER */
  double penalty_switch_theta;       /* 1.0 */
  /* ABI 3: what stands in for the Hessian of the Lagrangian.  DTO_HESSIAN_EXACT: second derivatives of the traced expressions.
   * DTO_HESSIAN_LBFGS: Ipopt's hessian_approximation = limited-memory, i.e. what the reference runs when a problem is built
   * with evaluate_hessian = false (src/solver.jl:7, its default and its own acrobot / car examples): compact L-BFGS, history 6,
   * sigma = s'y / s's, updates skipped without curvature -- no second derivatives are evaluated.  Lane-per-instance solver
   * path.  Where the mode does not exist the library says what it runs instead (dto_solver_hessian_mode): a model plugin built
   * without Hessians (evaluate_hessian = 0 in its generator) keeps per-stage SR1 blocks; the tile (17 .. 64 states) and the
   * bordered (multi-knot GeneralConstraint) paths use the exact second derivatives of the traced expressions. */
  int hessian_approximation;         /* DTO_HESSIAN_EXACT */
  /* ABI 4: passes of iterative refinement per KKT step (0 = none).  One pass: the residual r = b - K v of the step just computed,
   * evaluated stage by stage from the code the sweeps factorise, one more factor + solve for K e = r, v := v + e.  Measured on the
   * acrobot T = 1000 bench state against an extended-precision solve of the oracle's K (profiles/r06/step_truth_*.json): the
   * sequential sweeps (batches that fill the GPU) are within 1e-9 of it without any pass; the time-partitioned sweeps (small
   * batches: chunks joined through a separator system) within 2.5e-8, 5e-6 at delta_w = 0, and within 1e-10 after one pass, which
   * costs them one more factor + solve per iteration.  Lane-per-instance path, exact-Hessian plugins; ignored elsewhere. */
  int kkt_refinement;                /* 0 */
} dto_options;
enum { DTO_LS_FILTER = 0, DTO_LS_PENALTY_FILTER = 1 };
enum { DTO_HESSIAN_EXACT = 0, DTO_HESSIAN_LBFGS = 1, DTO_HESSIAN_SR1_BLOCKS = 2 /* reported only: dto_solver_hessian_mode */ };
/* per-instance status reported by dto_solve[_batch] / dto_solver_run / dto_solver_stats */
enum { DTO_STATUS_RUNNING = 0, DTO_STATUS_CONVERGED = 1, DTO_STATUS_MAX_ITER = 2, DTO_STATUS_NONFINITE = 3,
       DTO_STATUS_ACCEPTABLE = 4, DTO_STATUS_DIVERGING = 5, DTO_STATUS_CPU_TIME = 6 /* cut off by max_cpu_time */ };
int dto_options_default(dto_options* o);

/* One regularised Newton-KKT step at given (x, mu), all constraints treated as equalities and bounds
 * ignored -- exactly the system of examples/pendulum/pendulum.jl:138-198:
 *     [ H + delta_w I   J' ] [dx ]     [ grad f + J' mu ]
 *     [ J         -delta_c I ] [dmu] = - [ c              ]
 * assembled stage-interleaved and solved by the block-tridiagonal LDL^T.  DEVICE pointers,
 * instance-major; dx: [B][lddx], dmu: [B][lddmu].  *inertia_ok (host, may be NULL) is set to 0 if any
 * instance had a pivot of the wrong sign (the matrix is not quasi-definite for this delta_w). */
int dto_kkt_step_batch(dto_problem* p, const dto_batch* b, const double* mu, int64_t ldmu, double delta_w,
                       double delta_c, double* dx, int64_t lddx, double* dmu, int64_t lddmu, int* inertia_ok);

/* ---- the linear solver alone: for a caller that keeps its own outer iteration (Ipopt's augmented system,
 *      cf. the sketch at examples/pendulum/pendulum.jl:138-198 with delta_w / delta_c on the diagonals):
 *     [ W(x, mu) + diag(sigma_x) + delta_w I            J(x)'               ] [ sol_x ]   [ rhs_x ]
 *     [ J(x)                              -diag(sigma_c) - delta_c I  ] [ sol_c ] = [ rhs_c ]
 *  W = hess f + sum_i mu_i hess c_i (exact Hessians), every constraint row treated as an equality, bounds ignored -- barrier
 *  terms enter through sigma_x / sigma_c.  DEVICE pointers, instance-major.  dto_kkt_assemble packs the point and the
 *  diagonals; dto_kkt_factor runs the block-tridiagonal LDL^T once and reports the inertia (HOST arrays [B], may be NULL:
 *  inertia_ok[i] = the matrix has exactly num_constraint negative and no tiny pivots; num_negative[i] = negative pivots);
 *  dto_kkt_solve solves for one right-hand side.  The factors are not stored between calls (the sweeps recompute each
 *  stage's LDL^T -- cheaper than the HBM traffic on this hardware), so every dto_kkt_solve costs a forward + backward sweep. */
typedef struct dto_kkt_system {
  const double* mu;      int64_t ldmu;   /* [B][ldmu] multipliers inside W */
  const double* sigma_x; int64_t ldsx;   /* [B][ldsx] >= 0, or NULL */
  const double* sigma_c; int64_t ldsc;   /* [B][ldsc] >= 0, or NULL */
  double delta_w, delta_c;
} dto_kkt_system;
int dto_kkt_assemble(dto_problem* p, const dto_batch* b, const dto_kkt_system* sys);
int dto_kkt_factor(dto_problem* p, int32_t* inertia_ok, int32_t* num_negative, void* stream);
int dto_kkt_solve(dto_problem* p, const double* rhs_x, int64_t ldrx, const double* rhs_c, int64_t ldrc, double* sol_x,
                  int64_t ldsx, double* sol_c, int64_t ldsc, void* stream);

/* Batched interior-point solve, one independent NLP per instance, same structure, different guesses.
 * x0: DEVICE [B][ldx] initial guesses (what initialize_states!/initialize_controls! set,
 * src/solver.jl:23-39); x_out/mu_out: DEVICE [B][ld*] final accepted iterates (get_trajectory,
 * src/solver.jl:41-43, returns the last *evaluated* point in the reference -- here it is the accepted one);
 * status/iterations: HOST [B] (0 running, 6 cut off by max_cpu_time, 1 converged, 2 max_iter, 3 failed: non-finite iterate,
 * 4 converged to the acceptable level, 5 diverging iterates).
 * Paths by model: lane-per-instance tiles (states <= 16; bounds, inequality rows, per-instance parameters); the tile (MFMA)
 * kernels for 64-state models (variables free, fixed or bounded; no stage constraints, one to four actions, shared or per-instance parameters); a
 * GeneralConstraint whose rows couple several knots is solved through a border (general rows equalities or inequalities,
 * dynamics / stage rows equalities, variables free or fixed):
 * one factorisation and n_g + 1 sweeps per step, the border algebra on the device since round 4 (DTO_BORDER_HOST=1: on the
 * host), the filter line-search loop around it driven from the host. */
int dto_solve_batch(dto_problem* p, const dto_options* opt, const dto_batch* b, double* x_out, int64_t ldxo,
                    double* mu_out, int64_t ldmuo, int32_t* status, int32_t* iterations);

/* The same solve split in three so a caller (bench.py) can time exactly K iterations. */
int dto_solver_begin(dto_problem* p, const dto_options* opt, const dto_batch* b);
/* Warm start for receding-horizon (MPC) re-solves -- what repeated initialize_states!/initialize_controls! + solve! calls
 * (src/solver.jl:23-47) amount to, without re-initialising the interior-point state: the multipliers, bound multipliers, slacks
 * and the barrier parameter of the previous solve of this handle stay on the device.  b->x (DEVICE, may be NULL = keep the
 * final iterate) replaces the primal iterate (e.g. the shifted trajectory), b->params the parameters (e.g. the newly
 * measured state); b->B must equal the previous batch size.  mu0 > 0 resets the barrier parameter, mu0 <= 0 keeps it. */
int dto_solver_begin_warm(dto_problem* p, const dto_options* opt, const dto_batch* b, double mu0);
/* Receding horizon: shift the device-resident iterate of the previous solve forward by `knots` knots -- x_t <- x_{t+k},
 * u_t <- u_{t+k}, dynamics multipliers and bound multipliers with them; the knots that enter at the end of the horizon hold the
 * final state and repeat the last action (the usual MPC warm start).  Stage-constraint multipliers stay with their knots.
 * Follow with dto_solver_begin_warm(x = NULL, params = the newly measured state) and dto_solver_run.  Needs the same state /
 * action dimensions at every knot. */
int dto_solver_shift(dto_problem* p, int knots, void* stream);
/* Move the instances that are still running to the leading tiles of the batch so that the following iterations only cover
 * tiles with work (a batch otherwise pays for every tile until its last lane has terminated); results keep coming back in
 * the caller's instance order.  dto_solver_run / dto_solve_batch do this by themselves; a caller that drives
 * dto_solver_iterate directly (bench.py) calls it between slices.  *num_running (may be NULL) = instances still iterating. */
int dto_solver_repack(dto_problem* p, int* num_running, void* stream);
/* iterate the begun batch to termination (the polling loop of dto_solve_batch) and hand the results over */
int dto_solver_run(dto_problem* p, double* x_out, int64_t ldxo, double* mu_out, int64_t ldmuo, int32_t* status,
                   int32_t* iterations, void* stream);
int dto_solver_iterate(dto_problem* p, int n_iterations, void* stream);
/* per-instance scalars, HOST [B] each, any may be NULL */
int dto_solver_stats(dto_problem* p, int32_t* status, int32_t* iterations, double* objective, double* constr_viol,
                     double* dual_inf, double* mu, double* delta_w, double* alpha);
/* diagnostic: launch ONE kernel of the iteration (enum dto_kkt_op in csrc/dto_kkt_kernels.hpp: 3 EVAL, 4 CONV,
 * 5 FACTOR_SOLVE, 6 LINESEARCH, 7 LS_REDUCE, 8 UPDATE, 9..12 the kernels behind FACTOR_SOLVE, 15 UPDATE_EVAL: UPDATE of one
 * iteration and EVAL of the next in one pass -- it writes into the second iterate buffers and swaps them: use it an even
 * number of times between calls that touch finished tiles) so a caller can time it with events on `stream`.
 * dto_solver_iterate itself runs UPDATE_EVAL wherever it can and, for batches of more than 1 024 tiles (65 536 instances), the
 * back substitutions of finished tiles on a second low-priority stream next to the forward launch (joined before it
 * returns to the caller's stream order); both bit-identical to the plain sequence (DTO_FUSE_UPDATE=0, DTO_OVERLAP_SWEEPS=0). */
int dto_solver_launch_op(dto_problem* p, int op, void* stream);
/* chunks of the time-partitioned block-tridiagonal factorisation: 0 = chosen from the batch size so that
 * tiles x chunks fills the GPU's SIMDs; 1 = plain sequential sweep.  Takes effect at the next begin/step. */
int dto_solver_set_partitions(dto_problem* p, int partitions);
int dto_solver_partitions(dto_problem* p, int* partitions);
/* 1 if dto_solver_iterate runs UPDATE of one iteration and EVAL of the next as one pass for the batch begun last (the second
 * iterate buffers could be allocated: an eighth of the device stays free after them, DTO_FUSE_RESERVE_GB overrides; and
 * DTO_FUSE_UPDATE is not 0), else 0: the two-kernel sequence, bit-identical */
int dto_solver_fused_update(dto_problem* p, int* fused);
/* Engine behind dto_solver_begin / dto_solve_batch: 0 = automatic, 1 = SoA tiles (64 instances share a wavefront for the whole
 * solve; time-partitioned sweeps for small batches), 2 = instance-major (csrc/dto_im_kernels.hpp: per-instance stage records,
 * work lists, one factorisation attempt per instance and pass; exact-Hessian models).  Automatic = SoA tiles (the faster one
 * on every measured workload) unless the environment says DTO_ENGINE=im.  Takes effect at the next
 * dto_solver_begin.  dto_solver_engine reports the engine of the batch begun last (1 or 2). */
int dto_solver_set_engine(dto_problem* p, int engine);
/* free the device state of the solver entry points (both engines); the next dto_solver_begin allocates again */
int dto_solver_release(dto_problem* p);
int dto_solver_engine(dto_problem* p, int* engine);
/* per-instance footprint of the solver state in doubles: stage records, factors (for roofline arithmetic) */
int dto_solver_footprint(dto_problem* p, int64_t* record_doubles, int64_t* factor_doubles, int64_t* num_slacks,
                         int* factor_rounds /* (k_kkt_fwd, k_kkt_sep) launch pairs per iteration */);
/* diagnostic: one per-instance scalar slot of the device state (enum dto_scal in csrc/dto_kkt_kernels.hpp), HOST [B] */
int dto_solver_scalar(dto_problem* p, int slot, double* out);
/* diagnostic: one vector of the device state, instance-major into DEVICE out[B][ld].  which: 0 z, 1 multipliers, 2 dz,
 * 3 d multipliers, 5 z_L, 6 z_U (bound multipliers, [num_variables]), 7 slacks, 8 slack multipliers, 9 d slacks (one per
 * inequality row, in stage order).  Used by the tests that re-derive the interior-point step in numpy. */
int dto_solver_peek(dto_problem* p, int which, double* out, int64_t ld, void* stream);
int dto_solver_end(dto_problem* p, double* x_out, int64_t ldxo, double* mu_out, int64_t ldmuo, void* stream);
/* What stood in for the Hessian of the Lagrangian in the solve / batch begun last on this handle: DTO_HESSIAN_EXACT,
 * DTO_HESSIAN_LBFGS or DTO_HESSIAN_SR1_BLOCKS -- a caller that asked for DTO_HESSIAN_LBFGS on a path without that mode reads
 * here what it got (-1: nothing solved yet). */
int dto_solver_hessian_mode(dto_problem* p, int* mode);
/* Diagnostic: time every kernel launch of the solver entry points (dto_solver_iterate, dto_solver_launch_op, dto_solver_run ...)
 * with a HIP event pair on the stream the kernel is launched on -- the caller's, or the library's low-priority stream for the
 * early back substitutions.  on = 1 clears the records and starts, on = 0 stops.  dto_solver_trace_read synchronises the device
 * and returns, launch by launch in issue order: the op (enum dto_kkt_op in csrc/dto_kkt_kernels.hpp; FACTOR_SOLVE = 5 covers the
 * kernels behind it when they are not launched one by one), the iteration index since the trace was switched on, the start
 * relative to the first traced launch and the duration in milliseconds.  HOST arrays of `capacity` entries (any may be NULL);
 * *count = launches recorded (may exceed capacity).  bench.py uses it to report per-kernel times of the iterations it timed. */
int dto_solver_trace(dto_problem* p, int on);
int dto_solver_trace_read(dto_problem* p, int32_t* op, int32_t* iteration, double* start_ms, double* duration_ms, int64_t capacity,
                          int64_t* count);

/* single instance, HOST pointers: solve!(solver) (src/solver.jl:45-47) */
int dto_solve(dto_problem* p, const dto_options* opt, const double* x0, double* x, double* mu, int32_t* status,
              int32_t* iterations);

/* device memory helpers so a host language without a HIP binding can stay on this ABI */
int dto_device_alloc(void** ptr, int64_t bytes);
int dto_device_free(void* ptr);
int dto_copy_to_device(void* dst, const void* src, int64_t bytes);
int dto_copy_to_host(void* dst, const void* src, int64_t bytes);
int dto_device_synchronize(void);
int dto_device_count(int* n);
/* instance sharding across ranks (one process per GPU): contiguous block [first, first + count) of `total` for `rank` */
int dto_shard_range(int64_t total, int rank, int world, int64_t* first, int64_t* count);

#ifdef __cplusplus
}
#endif
#endif /* DTO_H */
