// KKT assembly, block-tridiagonal LDL^T factor/solve and the interior-point outer iteration,
// hand-written for gfx950.  Everything here is work the reference leaves to Ipopt (external):
// KKT assembly + symmetric indefinite factorisation + step acceptance.  The only in-tree trace of
// the linear system is the scratch at reference examples/pendulum/pendulum.jl:138-198:
//     K = [ H + delta_w I   C' ;  C   -delta_c I ],   rhs = [ grad L ; c ]
// which is exactly the system solved here (stage-interleaved so that it is block tridiagonal,
// SURVEY.md Appendix F).
//
// Data layout ("SoA tiles"): B problem instances are grouped in tiles of 64; every per-instance
// vector v[N] is stored as v[tile][i][lane] so that lane = instance and each row i is one fully
// coalesced 512-byte line.  A wavefront therefore owns one time step t of 64 instances
// (k_stage_eval, k_linesearch) or the whole horizon chain of 64 instances (k_kkt): all lanes of a
// wave see the same stage kind, so the compile-time kind dispatch is wave-uniform and the generated
// expression code and the dense block algebra run entirely in VGPRs with literal indices.
//
// Stage blocks (kind K: NP = nx+nu primal, Q stage-constraint rows, NY = n_{t+1} dynamics rows):
//     v_t = (p_t, nu_t, lam_t),  S_t = [ W+Sigma+dw I + [P_t]_xx   G'        F'      ]
//                                      [ G                     -Dc        0       ]
//                                      [ F                      0       -dc I     ]
//     coupling to x_{t+1}:  O_t = [ V ; 0 ; E ]   (only the x columns of the next block are touched)
// forward sweep:  S_t = L D L',  X = L^-1 O_t,  w = L^-1 y_t,
//                 P_{t+1} = YY_t - X' D^-1 X,  y_{t+1}[x] -= X' D^-1 w
// backward sweep: v_t = L^-T D^-1 (w - X x_{t+1}).
// Inertia: the negative pivots of all stage factorisations are counted (Sylvester's law: the count is
// the number of negative eigenvalues of K).  K is accepted when the count equals the number of
// constraint rows and no pivot is tiny; otherwise delta_w is raised (Ipopt's schedule) and the sweep
// is repeated inside the kernel.  This is Ipopt's inertia criterion (reduced Hessian positive
// definite), not the stricter quasi-definiteness of W + delta_w I.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>

#include "dto_eval_kernels.hpp"
#include "dto_model_plugin.h"

// ---- ops --------------------------------------------------------------------------------------
// 0 compiles the limited-memory mode's three touches of the sweep kernels out (A/B measurements of their cost: the mode then
// fails its tests, never ship a build with it)
#ifndef DTO_QN_HOT
#define DTO_QN_HOT 1
#endif

// the linear-solver entry points (dto_kkt_step_batch, dto_kkt_assemble / factor / solve) share the sweep kernels with the solver:
// `newton_only` switches bounds / slack handling off and the caller's Sigma_x / Sigma_c on.  -DDTO_NO_NEWTON_ONLY=1 folds those
// branches away inside the stage algebra (measurement: what the shared instantiation costs the solver's sweeps)
#ifdef DTO_NO_NEWTON_ONLY
#define DTO_NEWTON(o) false
#else
#define DTO_NEWTON(o) ((o).newton_only != 0)
#endif

enum dto_kkt_op {
  DTO_KKT_PACK = 0,        // instance-major z (and lam) -> SoA tiles
  DTO_KKT_UNPACK = 1,      // SoA tiles -> instance-major
  DTO_KKT_INIT = 2,        // bound push, slack and multiplier initialisation, scalar state
  DTO_KKT_EVAL = 3,        // per-stage derivative blocks + residual partials
  DTO_KKT_CONV = 4,        // reduce partials, convergence test, barrier update
  DTO_KKT_FACTOR_SOLVE = 5,  // all rounds of (chunk forward sweeps, separator system) + back substitution + post
  DTO_KKT_LINESEARCH = 6,  // merit partials of the trial step sizes
  DTO_KKT_LS_REDUCE = 7,   // pick the step size
  DTO_KKT_UPDATE = 8,      // take the step
  DTO_KKT_FWD = 9, DTO_KKT_SEP = 10, DTO_KKT_BWD = 11, DTO_KKT_POST = 12,  // the four kernels behind FACTOR_SOLVE
  DTO_KKT_RHS = 13,        // linear-solver entry points: caller's right-hand side -> stage records
  DTO_KKT_REARM = 14,      // linear-solver entry points: request one factorisation with the fixed delta_w
  DTO_KKT_UPDATE_EVAL = 15,  // UPDATE of one iteration and EVAL of the next in one pass (z, lam -> z_next, lam_next)
  DTO_KKT_BWD_EARLY = 16,    // back substitution of the tiles whose forward sweep has published its tag (second stream)
  DTO_KKT_BWD_REST = 17,     // ... and of the tiles DTO_KKT_BWD_EARLY left
  DTO_KKT_BWD_GATE = 18,     // one wavefront that returns when every forward block of the launch has started
  // limited-memory BFGS mode (dto_options.hessian_approximation = DTO_HESSIAN_LBFGS; round 5): see "limited-memory BFGS" below
  DTO_KKT_QN_BEGIN = 19,     // after EVAL: secant pair of the last step, history, sigma, the small matrices S'S and S'Y
  DTO_KKT_QN_RHS = 20,       // stage records r_p := r_p0 - (a column of U | U q | nothing), one more factorisation requested
  DTO_KKT_QN_COL = 21,       // Z_col := dz - v0 (col = -1: v0 := dz)
  DTO_KKT_QN_SMALL = 22,     // U'Z, U'v0, C = M - U'Z, q = C^-1 U'v0, correction of the directional derivative
  DTO_KKT_QN_SAVE = 23,      // after LS_REDUCE: grad_x L(x_k, lam_{k+1}) and alpha dz for the next secant pair
  DTO_KKT_QN_COLS_RHS = 24,  // column state (the columns of U as instances of their own): r_p -= column (slot mod QN_M2) of the main state's U
  // iterative refinement of the step (dto_options.kkt_refinement; round 6): see "iterative refinement" below
  DTO_KKT_REFINE = 25,       // stage records := residual b - K v of the step just computed (v saved), one more factorisation requested at the accepted (delta_w, gamma)
  DTO_KKT_REFINE_APPLY = 26, // step := saved step + correction; slack steps, step-length limits and the merit derivative recomputed from it
  DTO_KKT_OP_COUNT
};

// per-instance scalar slots (SoA rows of `scal`)
enum dto_scal {
  SC_STATUS = 0,  // 0 running, 1 converged, 2 max_iter, 3 failed (non-finite), 4 acceptable level, 5 diverging iterates
  SC_ITER, SC_MU, SC_PENALTY, SC_DELTA_W, SC_F, SC_THETA1, SC_THETA_INF, SC_DINF, SC_COMPL, SC_E0,
  SC_LOGBAR, SC_ALPHA_PMAX, SC_ALPHA_DMAX, SC_DMERIT, SC_ALPHA, SC_LS_FAIL, SC_NFACT, SC_MERIT0, SC_DELTA_LAST,
  SC_THETA_MAX, SC_THETA_MIN, SC_FILTER_N, SC_LS_KIND, SC_GAMMA, SC_NEED, SC_TRY_DW, SC_TRY_GAM, SC_ATTEMPT, SC_QN_RESET, SC_FULL_STREAK, SC_SHORT_STREAK, SC_WATCHDOG,
  SC_ACC_COUNT, SC_F_LAST, SC_XMAX, SC_NNEG,
  SC_LS_MODE,   // line-search phase: 1 = l1-penalty (far from the constraint manifold), 2 = filter (ls_reduce_body)
  SC_ASCALE,    // penalty phase: scale of the trial step sizes (shrinks 256 x when all eight trials fail, recovers 4 x per full step)
  SC_QN_SIGMA, SC_QN_SKIP, SC_QN_GCORR,   // limited-memory BFGS: sigma of B_0 = sigma I, consecutive skipped updates, q'(U'dz)
  SC_COUNT
};

constexpr int DTO_NPART = 10;   // residual partials written by k_stage_eval
constexpr int DTO_SB = 8;       // consecutive stages one wavefront of the stage-parallel kernels walks (partials summed in registers):
                                // the value for batches that fill the GPU; dto_kkt_args.sb carries the one in use (1 for a batch of one)
constexpr int DTO_LS_TRIALS = 8;
constexpr double DTO_LS_NULL_STEP = 100.0;  // k_ls_reduce: no step at all when even the most feasible trial multiplies the violation by more
constexpr int DTO_FILTER_CAP = 24;  // filter entries kept per instance (ring)

struct dto_kkt_info {
  int supported;
  int n_kind;
  int rec_size[16];   // doubles per stage record, by kind
  int fac_size[16];   // doubles per stage factor record, by kind
  int fac_size_seq[16];  // the same without the spike coupling: all a batch that only ever runs the sequential sweep stores
  int n_ineq[16];     // inequality rows (slacks) by kind
  int npart, nscal, ls_trials, filter_cap;
  int chunk_sum_size, sep_fac_size, nx;  // per (tile, chunk) doubles of the partitioned factorisation
  int quasi_newton;                      // 1: the stage records hold persistent quasi-Newton blocks
  int has_general;                       // 1: the model has GeneralConstraint rows -- the kernels cover the stage part of K, the
                                         //    host adds the border (dto_solver.cpp: bordered_step)
};

struct dto_solver_opts {
  double tol, s_max, dual_inf_tol, constr_viol_tol, compl_inf_tol;
  int max_iter;
  double acceptable_tol, acceptable_dual_inf_tol, acceptable_constr_viol_tol, acceptable_compl_inf_tol, acceptable_obj_change_tol;
  int acceptable_iter;
  double diverging_iterates_tol, mu_target;
  double mu_init, kappa_eps, kappa_mu, theta_mu, tau_min, bound_push, bound_frac;
  double delta_c, delta_w_init, delta_w_min, delta_w_max, kappa_w_minus, kappa_w_plus, kappa_w_plus_first;
  double delta_w_exact_cap;  // largest delta_w tried with the exact Hessian before the Gauss-Newton fallback
  double eta_armijo, rho_penalty, piv_tol;
  int max_refactor;
  int watchdog_trigger, watchdog_trials;  // Ipopt: watchdog_shortened_iter_trigger (10), watchdog_trial_iter_max (3); 0 = off
  int ls_penalty;       // 1: l1-penalty line search while theta_inf > ls_switch, then the filter (dto_options.line_search)
  int pen_gn;           // 1: Gauss-Newton Hessian model during the penalty phase (conv_body)
  int qn_lbfgs;         // 1: limited-memory BFGS instead of the exact Hessian of the Lagrangian (dto_options.hessian_approximation)
  double cost_hess_scale;  // 1, or 0 in limited-memory mode: factor on the objective Hessian inside the stage blocks
  double ls_switch;     // dto_options.penalty_switch_theta
  int newton_only;      // 1: ignore bounds/inequality structure, fixed delta_w (dto_kkt_step_batch, dto_kkt_factor/solve)
  double fixed_delta_w;
  int warm;             // 1: dto_solver_begin_warm -- keep multipliers, bound multipliers, slacks (and mu unless mu_warm > 0)
  double mu_warm;
};

struct dto_stage_run {
  int kind, t0, t1;                     // stages [t0, t1) are of this kind
  int z0, cd0, cc0, io0, w0;            // offsets of stage t0 (rows of z, dynamics rows, stage rows, slacks, parameters)
  int zs, cds, ccs, ios, ws;            // ... and their strides per stage
  int bounded;                          // 1: some variable of the run is fixed or has a finite bound
  long long rec0, fac0, recs, facs;     // record / carry offsets of stage t0 and strides (byte 56 on)
  int pad_, pad2_;
};
static_assert(sizeof(dto_stage_run) == 96, "load_run reads the fields by position");

struct dto_kkt_args {
  int T;
  int64_t B;
  int G;  // tiles of 64 instances
  int64_t Nz, Nc, Ni;
  int64_t n_mult, n_bnd;  // multiplier / bound-multiplier counts for Ipopt's error scaling
  const int* kind; const int* zoff; const int* woff; const int* cdoff; const int* ccoff;
  const int* ioff;    // [T+1] slack offsets
  const int64_t* recoff;  // [T+1] record offsets (doubles per lane)
  const int64_t* facoff;  // [T+1]
  int64_t rec_total, fac_total;
  // the horizon as runs of consecutive stages of one kind (the sequential sweeps walk runs: inside a run every offset advances
  // by a constant stride, so the hot loop has no table look-ups and no kind dispatch)
  const struct dto_stage_run* runs; int n_runs;
  const double* lo; const double* hi;  // [Nz] shared variable bounds
  const double* params;                // shared parameters
  const double* wtile;                 // per-instance parameters as SoA tiles [G][Nw][64], or NULL (then `params` is used)
  int64_t Nw;
  // SoA state, doubles
  double* z; double* lam; double* zl; double* zu; double* s; double* zs;
  double* z_next; double* lam_next;   // DTO_KKT_UPDATE_EVAL: where the updated iterate goes (the host swaps the pairs)
  // sequential sweeps overlapped through a second stream: per tile, the tag (iteration counter of the host) of the last
  // finished forward sweep / back substitution; NULL = plain launches
  int* tile_fwd_tag; int* tile_bwd_tag; int sweep_tag;
  int* fwd_started;   // forward blocks of this launch that have started (zeroed by the host before the launch)
  double* dz; double* dlam; double* ds;
  double* rec; double* fac; double* part; double* lspart; double* scal;
  double* filt;  // [G][2*DTO_FILTER_CAP][64] filter entries (theta, phi)
  int P;               // chunks of the time-partitioned factorisation
  const int* cstart;   // [P+1] first stage of every chunk
  double* csum; double* sfac; double* xsep; double* cacc;  // chunk summaries, separator factors/solutions, step partials
  long long* prof;  // debug: cycle stamps of workgroup 0 (tools/kkt_profile.py), NULL otherwise
  double* cpart;       // [G][P][16][64] chunk-level partial reductions (k_part_reduce)
  // linear-solver entry points (dto_kkt_assemble / factor / solve): optional extra diagonals as SoA tiles, and the caller's
  // right-hand side (instance-major) that DTO_KKT_RHS writes into the stage records
  const double* sigx; const double* sigc;
  const double* rhs_x; int64_t ld_rhs_x; const double* rhs_c; int64_t ld_rhs_c;
  // slot -> instance map after dto_solver_repack moved the running instances to the front (NULL: identity)
  const int* inst_of_slot;
  int fwd_rounds;  // sequential sweep: inertia-correction rounds per launch (0 = all)
  // time-partitioned form: per tile four arrival counters (chunk sums -> convergence test, chunk sweeps -> separator system, back
  // substitutions -> step partials, merit partials -> step size): the wavefront that arrives LAST does the tile's joining step in
  // the same launch (tile_last_arrival); NULL: one launch per step as in rounds 2-4 (DTO_FUSE_JOIN=0, bit-identical)
  int* csync;
  int sb;          // consecutive stages per wavefront of the stage-parallel kernels (k_stage_eval, k_linesearch, k_update_eval): DTO_SB, less for small batches
  int sep_cr;      // separator system by cyclic reduction, lanes = separators (kkt_sep_cr): 1 = inside the tile's wavefront (batches of
                   // at most DTO_SEP_CR_MAX_INST instances), 2 = one wavefront per instance (k_kkt_sep_cr: larger batches, many chunks)
  double* qn;      // limited-memory BFGS: per tile (4 QN_M + 4) Nz + QN_SMALL rows (S, Y, Z, r_p0, grad L, s, v0, small matrices), or NULL
  const double* qn_main; // DTO_KKT_QN_COLS_RHS (launched on the column state): the main state's `qn`
  int qn_mode, qn_col;   // DTO_KKT_QN_RHS: 0 column qn_col of U, 1 U q, 2 restore; DTO_KKT_QN_COL: column (-1: save v0)
  // instance-major mirrors for pack/unpack
  const double* aos_in; double* aos_out; int64_t ld_aos; int aos_which;  // 0: z, 1: lam, 2: dz, 3: dlam
  dto_solver_opts opt;
  // iterative refinement (DTO_KKT_REFINE / _APPLY): [G][T][MAX_NX][64] what stage t - 1 contributes to the x_t rows of K v, and the
  // step being refined (laid out like dz / dlam).  Behind `opt`: no offset of the fields the sweeps read moves.
  double* refq; double* refvz; double* refvl;
};

namespace dto {

// index helpers -----------------------------------------------------------------------------------
__host__ __device__ constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }  // i >= j

__device__ __forceinline__ double* soa(double* base, int64_t tile, int64_t n, int64_t i) {
  return base + ((tile * n + i) << 6) + threadIdx.x;
}
__device__ __forceinline__ const double* soa(const double* base, int64_t tile, int64_t n, int64_t i) {
  return base + ((tile * n + i) << 6) + threadIdx.x;
}

// Read-only tables that every lane indexes with the same (wave-uniform) index -- stage kinds, offsets, the shared variable
// bounds and parameters -- are read through the CONSTANT address space: the compiler then uses the scalar memory path
// (s_load into SGPRs, scalar compares and branches) instead of a vector load per lane.  Measured reason (profiles/r03, SQ
// counters + ISA of k_kkt_bwd_seq): as plain global loads the look-ups compiled to per-lane global_load_dword whose result
// fed the ADDRESS of the data loads, and the bound tests to a chain of basic blocks each waiting for its own small load --
// six to ten dependent memory round trips per stage; the sweeps sat in s_waitcnt for 58 % of their cycles.
template <class T>
__device__ __forceinline__ T uload(const T* p, int64_t i) {
  return ((const __attribute__((address_space(4))) T*)p)[i];
}

// Pair-interleaved rows of a stage block (stage records, carry records): logical rows 2k, 2k + 1 share one 1 KiB line, 16 bytes
// per lane.  `base` points at the block's first row + 2 * lane; the block has an even number of rows (fill_info rounds up).
__device__ __forceinline__ int64_t pair_at(int e) { return ((int64_t)(e >> 1) << 7) + (e & 1); }

// One SoA array of ONE tile as a buffer resource (rows of 64 doubles).  Every access is
//     buffer_load/store_dwordx2  v, v_lane, s[rsrc], s_row  offen
// -- resource and row offset in SGPRs (scalar ALU), ONE VGPR (8 x lane) shared by every array.  Measured reason
// (profiles/r03: ISA of the sweeps + SQ/TCP counters): with pointer arithmetic the compiler kept one 64-bit VGPR pointer per
// array and per offset class alive across the stage loop (loop strength reduction), the sweeps needed more than the 256
// VGPRs two wavefronts per SIMD allow, and the spills landed INSIDE the stage loop: 16-44 scratch reloads per stage, each
// followed by s_waitcnt vmcnt(0), i.e. a full drain of the in-order vector-memory counter (average 270 cycles) -- 35 drains
// per stage were the 58 % of the wave cycles the sweeps spent waiting.  An array that does not exist (base = NULL: zero
// bytes) reads as 0 and drops writes; the row offset itself travels in the scalar offset, which the hardware's range check
// does not include -- callers pass valid rows only.
typedef int dto_v2i __attribute__((ext_vector_type(2)));
typedef int dto_v4i __attribute__((ext_vector_type(4)));
struct TileBuf {
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ void bind(const double* base, int64_t tile, int64_t rows) {
    const int64_t bytes = base ? rows * 512 : 0;   // the host refuses state whose tile arrays reach 2 GiB (ensure_state)
    r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base + ((tile * rows) << 6)), 0, (int)bytes, 0x00020000);
  }
  __device__ __forceinline__ double ld(int row) const {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, threadIdx.x * 8u, row << 9, 0));
  }
  __device__ __forceinline__ void st(int row, double v) const {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(dto_v2i, v), r, threadIdx.x * 8u, row << 9, 0);
  }
  // PAIR-INTERLEAVED rows (the carry records, round 4): inside a block of rows that starts at `row0` the logical rows 2k and
  // 2k + 1 share one 1 KiB line, 16 bytes per lane, so that a wavefront moves two rows with ONE buffer_load/store_dwordx4.
  // The sweeps at one wavefront per SIMD have nobody to hide their vector-memory issue behind: 14 carry rows per stage are 7
  // instructions instead of 14 going forward and coming back.  A lane that must not store (its attempt is lost, or it did not
  // ask for a factorisation) gets an offset beyond the resource -- the hardware drops the write -- so the stores are
  // straight-line code the scheduler can spread over the arithmetic instead of a branch around a burst.
  __device__ __forceinline__ void ld2(int row0, int k, double& v0, double& v1) const {
    const dto_v4i q = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16u, (row0 << 9) + (k << 10), 0);
    v0 = __builtin_bit_cast(double, dto_v2i{q[0], q[1]});
    v1 = __builtin_bit_cast(double, dto_v2i{q[2], q[3]});
  }
  __device__ __forceinline__ void st2(int row0, int k, double v0, double v1, bool on) const {
    const dto_v2i a0 = __builtin_bit_cast(dto_v2i, v0), a1 = __builtin_bit_cast(dto_v2i, v1);
    __builtin_amdgcn_raw_buffer_store_b128(dto_v4i{a0[0], a0[1], a1[0], a1[1]}, r, on ? threadIdx.x * 16u : 0x80000000u,
                                           (row0 << 9) + (k << 10), 0);
  }
  __device__ __forceinline__ double ld1p(int row0, int i) const {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, threadIdx.x * 16u + (i & 1) * 8u, (row0 << 9) + ((i >> 1) << 10), 0));
  }
  __device__ __forceinline__ void st1p(int row0, int i, double v, bool on) const {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(dto_v2i, v), r, on ? threadIdx.x * 16u + (i & 1) * 8u : 0x80000000u,
                                          (row0 << 9) + ((i >> 1) << 10), 0);
  }
};
struct SoaBufs {
  TileBuf z, lam, zl, zu, s, zs, dz, dlam, ds, rec, fac, sigx, sigc, wt;
  __device__ __forceinline__ SoaBufs(const dto_kkt_args& a, int64_t g);
};

template <class M, int K>
struct KindDims {
  using KD = typename M::template Kind<K>;
  using C = typename M::template Cost<KD::COST>;
  static constexpr int NX = C::NX, NU = C::NU, NP = NX + NU;
  static constexpr int Q = []() { if constexpr (KD::CON >= 0) return M::template Con<KD::CON>::NC; else return 0; }();
  static constexpr int QI = []() { if constexpr (KD::CON >= 0) return M::template Con<KD::CON>::NI; else return 0; }();
  static constexpr int NY = []() { if constexpr (KD::DYN >= 0) return M::template Dyn<KD::DYN>::NY; else return 0; }();
  static constexpr int BD = NP + Q + NY;
  // Stage record.  Exact-Hessian models (FUSED): only the RESIDUALS of the stage (r_p = grad_p L, d, c) -- the
  // derivative values themselves are NOT stored: the sweeps re-evaluate the generated Jacobian / Hessian code from
  // (z, lambda) in registers, which costs a few hundred flops per stage against 55 rows (28 KB per wavefront and stage)
  // of HBM traffic, their LDS staging and the registers that held them (round 1 stored them: the sweeps were bound by
  // exactly that handling, not by the factorisation).  Quasi-Newton models keep their per-stage SR1 block and the previous
  // Jacobian values here (they cannot be recomputed).
  static constexpr bool QN = (M::EVALUATE_HESSIAN == 0);
  static constexpr bool FUSED = !QN;
  static constexpr int N_CH = FUSED ? 0 : C::NHL;   // objective Hessian, lower triangle
  static constexpr int N_DJ = []() { if constexpr (KD::DYN >= 0 && !FUSED) return M::template Dyn<KD::DYN>::NJ; else return 0; }();
  static constexpr int N_DH = []() { if constexpr (KD::DYN >= 0 && !FUSED) return M::template Dyn<KD::DYN>::NHL; else return 0; }();
  static constexpr int N_KJ = []() { if constexpr (KD::CON >= 0 && !FUSED) return M::template Con<KD::CON>::NJ; else return 0; }();
  static constexpr int N_KH = []() { if constexpr (KD::CON >= 0 && !FUSED) return M::template Con<KD::CON>::NHL; else return 0; }();
  // quasi-Newton mode (problem built with evaluate_hessian=false, the reference default under which Ipopt uses a
  // limited-memory BFGS Hessian): a partitioned SR1 approximation B_t of the ELEMENT Hessian
  // d^2/d(p_t,x_{t+1})^2 [ l_t + lam_t'd_t + nu_t'c_t ] is kept per stage (block structure preserved), together
  // with the previous objective gradient for the secant pair.
  static constexpr int NE = NP + NY;
  static constexpr int N_B = QN ? NE * (NE + 1) / 2 : 0;
  static constexpr int N_GC = QN ? NP : 0;
  static constexpr int R_CH = 0;
  static constexpr int R_B = R_CH + N_CH;
  static constexpr int R_GC = R_B + N_B;
  static constexpr int R_DJ = R_GC + N_GC;
  static constexpr int R_DH = R_DJ + N_DJ;
  static constexpr int R_KJ = R_DH + N_DH;
  static constexpr int R_KH = R_KJ + N_KJ;
  static constexpr int R_RP = R_KH + N_KH;
  static constexpr int R_D = R_RP + NP;
  static constexpr int R_C = R_D + NY;
  static constexpr int REC = R_C + Q;
  // "factor" record: only what the forward recursion hands from stage to stage (the carry-in of this stage).
  // The stage's L, D, X, w, Z are NOT stored: the backward sweep rebuilds them from the stage record and
  // this carry -- a 9x9..13x13 LDL^T is ~0.4 us of arithmetic, the 162 doubles it replaces are 83 KB of HBM
  // writes + reads per wave and stage (measured: the stored-factor version was write-bandwidth bound).
  static constexpr int F_P = 0;                          // Schur complement on x_t from stage t-1 (packed lower)
  static constexpr int F_PY = F_P + NX * (NX + 1) / 2;   // rhs carry
  static constexpr int F_CX = F_PY + NX;                 // coupling of x_t to the chunk's left separator (NX x NX)
  static constexpr int FAC = F_CX + NX * NX;
  __host__ __device__ static constexpr bool ineq(int j) {
    if constexpr (KD::CON >= 0) return M::template Con<KD::CON>::ineq(j); else return false;
  }
  __host__ __device__ static constexpr int slack(int j) {
    if constexpr (KD::CON >= 0) return M::template Con<KD::CON>::slack(j); else return -1;
  }
};

// chunk summary layout (doubles per (tile, chunk)), n = M::MAX_NX
template <class M>
struct ChunkSum {
  static constexpr int N = M::MAX_NX, NT = N * (N + 1) / 2;
  static constexpr int P = 0, PY = P + NT, RLL = PY + N, RL = RLL + NT, CX = RL + N, OK = CX + N * N, NNEG = OK + 1;
  static constexpr int SIZE = NNEG + 1;
};
// separator factor layout
template <class M>
struct SepFac {
  static constexpr int N = M::MAX_NX;
  static constexpr int L = 0, DI = L + N * (N - 1) / 2, W = DI + N, MM = W + N, SIZE = MM + N * N;
};

template <class M, int K = 0>
void fill_info(dto_kkt_info* o) {
  if constexpr (K < M::N_KIND) {
    using D = KindDims<M, K>;
    o->rec_size[K] = (D::REC + 1) & ~1;        // even: pair-interleaved rows
    o->fac_size[K] = (D::FAC + 1) & ~1;        // even: the carry rows are stored pair-interleaved (TileBuf::st2)
    o->fac_size_seq[K] = (D::F_CX + 1) & ~1;
    o->n_ineq[K] = D::QI;
    fill_info<M, K + 1>(o);
  }
}

template <class M>
int kkt_info(dto_kkt_info* out) {
  out->supported = (M::N_KIND <= 16) ? 1 : 0;
  out->has_general = M::HAS_GENERAL ? 1 : 0;
  out->n_kind = M::N_KIND;
  for (int i = 0; i < 16; ++i) out->rec_size[i] = out->fac_size[i] = out->fac_size_seq[i] = out->n_ineq[i] = 0;
  fill_info<M>(out);
  out->npart = DTO_NPART;
  out->nscal = SC_COUNT;
  out->ls_trials = DTO_LS_TRIALS;
  out->filter_cap = DTO_FILTER_CAP;
  out->chunk_sum_size = ChunkSum<M>::SIZE;
  out->sep_fac_size = SepFac<M>::SIZE;
  out->nx = M::MAX_NX;
  out->quasi_newton = (M::EVALUATE_HESSIAN == 0) ? 1 : 0;
  return 0;
}

__device__ __forceinline__ SoaBufs::SoaBufs(const dto_kkt_args& a, int64_t g) {
  z.bind(a.z, g, a.Nz); lam.bind(a.lam, g, a.Nc); zl.bind(a.zl, g, a.Nz); zu.bind(a.zu, g, a.Nz);
  s.bind(a.s, g, a.Ni); zs.bind(a.zs, g, a.Ni); dz.bind(a.dz, g, a.Nz); dlam.bind(a.dlam, g, a.Nc); ds.bind(a.ds, g, a.Ni);
  rec.bind(a.rec, g, a.rec_total); fac.bind(a.fac, g, a.fac_total);
  sigx.bind(a.sigx, g, a.Nz); sigc.bind(a.sigc, g, a.Nc); wt.bind(a.wtile, g, a.Nw);
}

// parameters w_t of stage t: shared by all instances (the reference's one `parameters` vector, src/solver.jl:10), or one
// set per instance (dto_batch.params: MPC rollouts that differ in initial state / target)
template <int N>
__device__ __forceinline__ void load_params(arr<N>& w, const dto_kkt_args& a, int64_t g, int t) {
  if (a.wtile) {
    const double* src = a.wtile + ((g * a.Nw + uload(a.woff, t)) << 6) + threadIdx.x;
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = src[(int64_t)i << 6];
  } else {
    const int w0 = uload(a.woff, t);
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = uload(a.params, w0 + i);
  }
}

// Everything a stage needs to treat its variable bounds, requested in ONE batch of independent loads before any of it
// is used.  Written the obvious way (load lo/hi, branch, then load the multipliers inside the branch) every variable
// costs two dependent memory round trips, and with one wavefront per SIMD those waits were most of the sweep time
// (SQ_WAIT_ANY 65 % of the wave cycles).  The bound multipliers are only touched when the problem has finite bounds.
template <int NP>
struct StageBounds {
  double lo[NP > 0 ? NP : 1], hi[NP > 0 ? NP : 1], p[NP > 0 ? NP : 1], zl[NP > 0 ? NP : 1], zu[NP > 0 ? NP : 1];
};
template <int NP>
__device__ __forceinline__ void load_stage_bounds(const dto_kkt_args& a, int64_t g, int z0, StageBounds<NP>& b) {
  const bool duals = a.zl != nullptr;   // allocated iff some variable has a finite, non-fixing bound
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    b.lo[i] = uload(a.lo, z0 + i);
    b.hi[i] = uload(a.hi, z0 + i);
    b.p[i] = duals ? *soa(a.z, g, a.Nz, z0 + i) : 0.0;
    b.zl[i] = duals ? *soa(a.zl, g, a.Nz, z0 + i) : 0.0;
    b.zu[i] = duals ? *soa(a.zu, g, a.Nz, z0 + i) : 0.0;
  }
}

// How the sweeps reach the data of one stage.  SoaIO: the SoA tiles of this file (lane = column of the tile).  The
// instance-major engine (dto_im_kernels.hpp) implements the same interface over stage records staged in LDS, so that the
// block algebra below (stage_factor / stage_forward / stage_backward) exists once.
template <class M, int K>
struct SoaIO {
  using D = KindDims<M, K>;
  const dto_kkt_args& a;
  const int64_t g;
  const int t, z0;
  const double* recp;
  double* facp;
  __device__ __forceinline__ SoaIO(const dto_kkt_args& a_, int64_t g_, int t_)
      : a(a_), g(g_), t(t_), z0(uload(a_.zoff, t_)), recp(a_.rec + ((g_ * a_.rec_total + uload(a_.recoff, t_)) << 6) + 2 * threadIdx.x),
        facp(a_.fac + ((g_ * a_.fac_total + uload(a_.facoff, t_)) << 6) + 2 * threadIdx.x) {}
  __device__ __forceinline__ double rec(int e) const { return recp[pair_at(e)]; }   // pair-interleaved like the carries
  __device__ __forceinline__ double p(int i) const { return *soa(a.z, g, a.Nz, z0 + i); }
  __device__ __forceinline__ double y(int i) const { return *soa(a.z, g, a.Nz, uload(a.zoff, t + 1) + i); }
  __device__ __forceinline__ double lam(int k) const { return *soa(a.lam, g, a.Nc, uload(a.cdoff, t) + k); }
  __device__ __forceinline__ double nu(int j) const { return *soa(a.lam, g, a.Nc, uload(a.ccoff, t) + j); }
  template <int N>
  __device__ __forceinline__ void params(arr<N>& w) const { load_params(w, a, g, t); }
  __device__ __forceinline__ void bounds(StageBounds<D::NP>& b) const { load_stage_bounds<D::NP>(a, g, z0, b); }
  __device__ __forceinline__ bool has_sigx() const { return a.sigx != nullptr; }
  __device__ __forceinline__ bool has_sigc() const { return a.sigc != nullptr; }
  __device__ __forceinline__ double sigx(int i) const { return *soa(a.sigx, g, a.Nz, z0 + i); }
  __device__ __forceinline__ double sigc_con(int j) const { return *soa(a.sigc, g, a.Nc, uload(a.ccoff, t) + j); }
  __device__ __forceinline__ double sigc_dyn(int k) const { return *soa(a.sigc, g, a.Nc, uload(a.cdoff, t) + k); }
  __device__ __forceinline__ double slack(int j) const { return *soa(a.s, g, a.Ni, uload(a.ioff, t) + D::slack(j)); }
  __device__ __forceinline__ double slack_mult(int j) const { return *soa(a.zs, g, a.Ni, uload(a.ioff, t) + D::slack(j)); }
  // carry-in record of the stage (written by the forward sweep, read by the backward sweep)
  // carry rows: pair-interleaved inside the stage's block (see TileBuf::st2)
  static constexpr bool PAIR_CARRY = true;
  __device__ __forceinline__ void put_carry(int i, double v, bool on = true) const { if (on) facp[((int64_t)(i >> 1) << 7) + (i & 1)] = v; }
  __device__ __forceinline__ double carry(int i) const { return facp[((int64_t)(i >> 1) << 7) + (i & 1)]; }
  __device__ __forceinline__ void put_carry2(int i, double v0, double v1, bool on) const {
    if (on) *reinterpret_cast<double2*>(facp + ((int64_t)(i >> 1) << 7)) = double2{v0, v1};
  }
  __device__ __forceinline__ void carry2(int i, double& v0, double& v1) const {
    const double2 q = *reinterpret_cast<const double2*>(facp + ((int64_t)(i >> 1) << 7));
    v0 = q.x; v1 = q.y;
  }
  // the step
  __device__ __forceinline__ void put_dp(int i, double v) const { *soa(a.dz, g, a.Nz, z0 + i) = v; }
  __device__ __forceinline__ void put_dnu(int j, double v) const { *soa(a.dlam, g, a.Nc, uload(a.ccoff, t) + j) = v; }
  __device__ __forceinline__ void put_dlam(int k, double v) const { *soa(a.dlam, g, a.Nc, uload(a.cdoff, t) + k) = v; }
  __device__ __forceinline__ void put_ds(int j, double v) const { *soa(a.ds, g, a.Ni, uload(a.ioff, t) + D::slack(j)) = v; }
  __device__ __forceinline__ long long* prof() const {
#if DTO_KKT_PROFILE == 2   // (tools/micro/fused_sweeps_check.py: the same stamps behind a wave-UNIFORM condition)
    return a.prof;
#elif DTO_KKT_PROFILE
    return (a.prof && blockIdx.x == 1 && threadIdx.x == 0) ? a.prof : nullptr;
#else
    return nullptr;   // the cycle stamps are compiled in only with -DDTO_KKT_PROFILE=1 (DTO_PLUGIN_CXXFLAGS, tools/kkt_profile.py):
                      // even a never-taken stamp is a branch that cuts the stage into basic blocks the scheduler cannot cross
#endif
  }
};

// SoaIO for a stage inside a run (dto_stage_run): every offset is arithmetic -- offset of the run's first stage + stride x
// position -- so nothing has to be looked up before the stage's rows can be requested.  Measured background (profiles/r03):
// with look-ups, a stage of the sweeps began with a chain of dependent memory round trips (kind -> offsets -> rows) that no
// other instruction of the wavefront could overlap; a wavefront sat in s_waitcnt for 58 % of its cycles.  BOUNDED = false:
// the run has no fixed variable and no finite bound, the bound tests of the block algebra fold away at compile time.
template <class M, int K, bool BOUNDED>
struct SoaRunIO {
  using D = KindDims<M, K>;
  const dto_kkt_args& a;
  const SoaBufs& b;
  const int t, z0, cd0, cc0, io0, w0, rec0, fac0;
  __device__ __forceinline__ SoaRunIO(const dto_kkt_args& a_, const SoaBufs& b_, const dto_stage_run& r, int t_)
      : a(a_), b(b_), t(t_), z0(r.z0 + (t_ - r.t0) * r.zs), cd0(r.cd0 + (t_ - r.t0) * r.cds), cc0(r.cc0 + (t_ - r.t0) * r.ccs),
        io0(r.io0 + (t_ - r.t0) * r.ios), w0(r.w0 + (t_ - r.t0) * r.ws), rec0((int)r.rec0 + (t_ - r.t0) * (int)r.recs),
        fac0((int)r.fac0 + (t_ - r.t0) * (int)r.facs) {}
  __device__ __forceinline__ double rec(int e) const { return b.rec.ld1p(rec0, e); }
  __device__ __forceinline__ double p(int i) const { return b.z.ld(z0 + i); }
  __device__ __forceinline__ double y(int i) const { return b.z.ld(z0 + D::NP + i); }   // x_{t+1} follows [x_t; u_t]
  __device__ __forceinline__ double lam(int k) const { return b.lam.ld(cd0 + k); }
  __device__ __forceinline__ double nu(int j) const { return b.lam.ld(cc0 + j); }
  template <int N>
  __device__ __forceinline__ void params(arr<N>& w) const {
    if (a.wtile) {
#pragma unroll
      for (int i = 0; i < N; ++i) w[i] = b.wt.ld(w0 + i);
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) w[i] = uload(a.params, w0 + i);
    }
  }
  __device__ __forceinline__ void bounds(StageBounds<D::NP>& sb) const {
    if constexpr (BOUNDED) {
      const bool duals = a.zl != nullptr;   // allocated iff some variable has a finite, non-fixing bound
#pragma unroll
      for (int i = 0; i < D::NP; ++i) {
        sb.lo[i] = uload(a.lo, z0 + i);
        sb.hi[i] = uload(a.hi, z0 + i);
        sb.p[i] = duals ? b.z.ld(z0 + i) : 0.0;
        sb.zl[i] = b.zl.ld(z0 + i);   // zero rows when the arrays do not exist: reads 0
        sb.zu[i] = b.zu.ld(z0 + i);
      }
    } else {
#pragma unroll
      for (int i = 0; i < D::NP; ++i) {
        sb.lo[i] = -__builtin_huge_val();
        sb.hi[i] = __builtin_huge_val();
        sb.p[i] = sb.zl[i] = sb.zu[i] = 0.0;
      }
    }
  }
  __device__ __forceinline__ bool has_sigx() const { return a.sigx != nullptr; }
  __device__ __forceinline__ bool has_sigc() const { return a.sigc != nullptr; }
  __device__ __forceinline__ double sigx(int i) const { return b.sigx.ld(z0 + i); }
  __device__ __forceinline__ double sigc_con(int j) const { return b.sigc.ld(cc0 + j); }
  __device__ __forceinline__ double sigc_dyn(int k) const { return b.sigc.ld(cd0 + k); }
  __device__ __forceinline__ double slack(int j) const { return b.s.ld(io0 + D::slack(j)); }
  __device__ __forceinline__ double slack_mult(int j) const { return b.zs.ld(io0 + D::slack(j)); }
  static constexpr bool PAIR_CARRY = true;
  __device__ __forceinline__ void put_carry(int i, double v, bool on = true) const { b.fac.st1p(fac0, i, v, on); }
  __device__ __forceinline__ double carry(int i) const { return b.fac.ld1p(fac0, i); }
  __device__ __forceinline__ void put_carry2(int i, double v0, double v1, bool on) const { b.fac.st2(fac0, i >> 1, v0, v1, on); }
  __device__ __forceinline__ void carry2(int i, double& v0, double& v1) const { b.fac.ld2(fac0, i >> 1, v0, v1); }
  __device__ __forceinline__ void put_dp(int i, double v) const { b.dz.st(z0 + i, v); }
  __device__ __forceinline__ void put_dnu(int j, double v) const { b.dlam.st(cc0 + j, v); }
  __device__ __forceinline__ void put_dlam(int k, double v) const { b.dlam.st(cd0 + k, v); }
  __device__ __forceinline__ void put_ds(int j, double v) const { b.ds.st(io0 + D::slack(j), v); }
  __device__ __forceinline__ long long* prof() const {
#if DTO_KKT_PROFILE == 2   // (tools/micro/fused_sweeps_check.py: the same stamps behind a wave-UNIFORM condition)
    return a.prof;
#elif DTO_KKT_PROFILE
    return (a.prof && blockIdx.x == 1 && threadIdx.x == 0) ? a.prof : nullptr;
#else
    return nullptr;   // the cycle stamps are compiled in only with -DDTO_KKT_PROFILE=1 (DTO_PLUGIN_CXXFLAGS, tools/kkt_profile.py):
                      // even a never-taken stamp is a branch that cuts the stage into basic blocks the scheduler cannot cross
#endif
  }
};

// Everything the block algebra reads of one stage, in registers.  The sequential sweeps request the rows of stage t+1
// (t-1 going backwards) BEFORE they start the arithmetic of stage t: with one wavefront per SIMD (512 registers, nothing
// spilled) the ~4000 cycles of arithmetic of a stage cover the memory latency of the next one, which nothing else would --
// the sweeps are one dependent chain per instance.
#ifndef DTO_SEQ_FWD_OCC
#define DTO_SEQ_FWD_OCC 1
#endif
#ifndef DTO_SEQ_BWD_OCC
#define DTO_SEQ_BWD_OCC 1
#endif
#ifndef DTO_SEQ_PREFETCH_FWD
#define DTO_SEQ_PREFETCH_FWD 1
#endif
#ifndef DTO_SEQ_PREFETCH_BWD
#define DTO_SEQ_PREFETCH_BWD 1
#endif
#ifndef DTO_SEQ_PREFETCH_MAX
#define DTO_SEQ_PREFETCH_MAX 56   // doubles per StageIn up to which the next stage is requested ahead
#endif
#ifndef DTO_KKT_PROFILE
#define DTO_KKT_PROFILE 0
#endif
#ifndef DTO_CHUNK_PREFETCH
#define DTO_CHUNK_PREFETCH 1   // the chunk sweeps of the time-partitioned form request the next stage's rows ahead too (round 5)
#endif
// SPK: the carry record of a chunk behind the first also holds the spike coupling (D::FAC instead of D::F_CX rows)
template <class M, int K, bool BOUNDED, bool BWD, bool SPK = false>
struct StageIn {
  using D = KindDims<M, K>;
  static constexpr int NP = D::NP, NY = D::NY, Q = D::Q, NCAR = BWD ? (SPK ? D::FAC : D::F_CX) : 0;
  static constexpr bool HAS_P = D::FUSED || BOUNDED;
  double rec[D::REC > 0 ? D::REC : 1], p[HAS_P && NP > 0 ? NP : 1], y[D::FUSED && NY > 0 ? NY : 1], lam[NY > 0 ? NY : 1],
      nu[Q > 0 ? Q : 1], zl[BOUNDED && NP > 0 ? NP : 1], zu[BOUNDED && NP > 0 ? NP : 1], s[Q > 0 ? Q : 1], zs[Q > 0 ? Q : 1],
      car[NCAR > 0 ? NCAR : 1];
  __device__ __forceinline__ void load(const SoaBufs& b, const dto_stage_run& r, int t) {
    const int n = t - r.t0;
    const int z0 = r.z0 + n * r.zs, cd0 = r.cd0 + n * r.cds, cc0 = r.cc0 + n * r.ccs, io0 = r.io0 + n * r.ios;
    const int rec0 = (int)r.rec0 + n * (int)r.recs, fac0 = (int)r.fac0 + n * (int)r.facs;
#pragma unroll
    for (int i = 0; i + 1 < NCAR; i += 2) b.fac.ld2(fac0, i >> 1, car[i], car[i + 1]);
    if constexpr (NCAR % 2 == 1) car[NCAR - 1] = b.fac.ld1p(fac0, NCAR - 1);
#pragma unroll
    for (int i = 0; i + 1 < D::REC; i += 2) b.rec.ld2(rec0, i >> 1, rec[i], rec[i + 1]);
    if constexpr (D::REC % 2 == 1) rec[D::REC - 1] = b.rec.ld1p(rec0, D::REC - 1);
    if constexpr (HAS_P) {
#pragma unroll
      for (int i = 0; i < NP; ++i) p[i] = b.z.ld(z0 + i);
    }
    if constexpr (D::FUSED) {
#pragma unroll
      for (int i = 0; i < NY; ++i) y[i] = b.z.ld(z0 + NP + i);
    }
#pragma unroll
    for (int i = 0; i < NY; ++i) lam[i] = b.lam.ld(cd0 + i);
#pragma unroll
    for (int j = 0; j < Q; ++j) {
      nu[j] = b.lam.ld(cc0 + j);
      if (D::slack(j) >= 0) {
        s[j] = b.s.ld(io0 + D::slack(j));
        zs[j] = b.zs.ld(io0 + D::slack(j));
      }
    }
    if constexpr (BOUNDED) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        zl[i] = b.zl.ld(z0 + i);   // arrays that do not exist read as 0
        zu[i] = b.zu.ld(z0 + i);
      }
    }
  }
};

// SoaRunIO whose reads come out of a StageIn (writes, the linear-solver extras and the parameters stay direct)
template <class M, int K, bool BOUNDED, bool BWD, bool SPK = false>
struct SoaPreIO : SoaRunIO<M, K, BOUNDED> {
  using Base = SoaRunIO<M, K, BOUNDED>;
  using D = KindDims<M, K>;
  using In = StageIn<M, K, BOUNDED, BWD, SPK>;
  const In& in;
  __device__ __forceinline__ SoaPreIO(const dto_kkt_args& a_, const SoaBufs& b_, const dto_stage_run& r, int t_, const In& in_)
      : Base(a_, b_, r, t_), in(in_) {}
  __device__ __forceinline__ double rec(int e) const { return in.rec[e]; }
  __device__ __forceinline__ double p(int i) const { return in.p[i]; }
  __device__ __forceinline__ double y(int i) const { return in.y[i]; }
  __device__ __forceinline__ double lam(int k) const { return in.lam[k]; }
  __device__ __forceinline__ double nu(int j) const { return in.nu[j]; }
  __device__ __forceinline__ double slack(int j) const { return in.s[j]; }
  __device__ __forceinline__ double slack_mult(int j) const { return in.zs[j]; }
  __device__ __forceinline__ double carry(int i) const { return in.car[i]; }
  __device__ __forceinline__ void carry2(int i, double& v0, double& v1) const { v0 = in.car[i]; v1 = in.car[i + 1]; }
  __device__ __forceinline__ void bounds(StageBounds<D::NP>& sb) const {
    if constexpr (BOUNDED) {
#pragma unroll
      for (int i = 0; i < D::NP; ++i) {
        sb.lo[i] = uload(this->a.lo, this->z0 + i);
        sb.hi[i] = uload(this->a.hi, this->z0 + i);
        sb.p[i] = in.p[i];
        sb.zl[i] = in.zl[i];
        sb.zu[i] = in.zu[i];
      }
    } else {
      Base::bounds(sb);
    }
  }
};

// one entry of the run table through scalar loads
__device__ __forceinline__ dto_stage_run load_run(const dto_stage_run* runs, int r) {
  const int* q = reinterpret_cast<const int*>(runs + r);
  const long long* ql = reinterpret_cast<const long long*>(runs + r);
  dto_stage_run o;
  o.kind = uload(q, 0); o.t0 = uload(q, 1); o.t1 = uload(q, 2);
  o.z0 = uload(q, 3); o.cd0 = uload(q, 4); o.cc0 = uload(q, 5); o.io0 = uload(q, 6); o.w0 = uload(q, 7);
  o.zs = uload(q, 8); o.cds = uload(q, 9); o.ccs = uload(q, 10); o.ios = uload(q, 11); o.ws = uload(q, 12);
  o.bounded = uload(q, 13);
  o.rec0 = uload(ql, 7); o.fac0 = uload(ql, 8); o.recs = uload(ql, 9); o.facs = uload(ql, 10);
  o.pad_ = 0;
  return o;
}

// wave-uniform kind dispatch (all lanes of a tile are at the same stage)
template <class M, int K = 0, class F>
__device__ __forceinline__ void dispatch_uniform(int kind, F&& f) {
  if constexpr (K < M::N_KIND) {
    if (kind == K) f(std::integral_constant<int, K>{});
    else dispatch_uniform<M, K + 1>(kind, static_cast<F&&>(f));
  }
}

__device__ __forceinline__ bool finite_lo(double v) { return v > -1e300; }
__device__ __forceinline__ bool finite_hi(double v) { return v < 1e300; }

// ------------------------------------------------------------------------------------------------
// pack / unpack between the C-ABI's instance-major buffers and SoA tiles
// ------------------------------------------------------------------------------------------------
// SC_STATUS of a lane that holds no instance (the tail of the last tile): any non-zero status means "not running" to every kernel
constexpr double DTO_ST_NO_INSTANCE = 9.0;
__device__ __forceinline__ bool lane_is_instance(const dto_kkt_args& a, int64_t g) {
  const int64_t slot = g * 64 + threadIdx.x;
  return (a.inst_of_slot ? (int64_t)a.inst_of_slot[slot] : slot) < a.B;
}
static __global__ __launch_bounds__(256) void k_pack(dto_kkt_args a, int64_t n, double* dst, unsigned nrb) {
  // grid: G * nrb, nrb = ceil(n/4) row blocks (tile index in grid.x: grid.y is limited to 65535); block 256 = 4 rows x 64 lanes
  const int64_t tile = blockIdx.x / nrb;
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)(blockIdx.x % nrb) * 4 + (threadIdx.x >> 6);
  const int64_t slot = tile * 64 + lane;
  const int64_t inst = a.inst_of_slot ? a.inst_of_slot[slot] : slot;
  if (i >= n) return;
  double v = 0.0;
  if (inst < a.B) v = a.aos_in[inst * a.ld_aos + i];
  dst[((tile * n + i) << 6) + lane] = v;
}

static __global__ __launch_bounds__(256) void k_unpack(dto_kkt_args a, int64_t n, const double* src, unsigned nrb) {
  const int64_t tile = blockIdx.x / nrb;
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)(blockIdx.x % nrb) * 4 + (threadIdx.x >> 6);
  const int64_t slot = tile * 64 + lane;
  const int64_t inst = a.inst_of_slot ? a.inst_of_slot[slot] : slot;
  if (i >= n || inst >= a.B) return;
  a.aos_out[inst * a.ld_aos + i] = src[((tile * n + i) << 6) + lane];
}

// ------------------------------------------------------------------------------------------------
// initialisation (Ipopt-style): push the guess into the bounds, slacks from the inequality values,
// multipliers on the central path of mu_init
// ------------------------------------------------------------------------------------------------
template <class M>
__global__ __launch_bounds__(WAVE) void k_init(dto_kkt_args a) {
  const int64_t g = blockIdx.x / a.T;
  const int t = blockIdx.x % a.T;
  const dto_solver_opts& o = a.opt;
  // warm start (dto_solver_begin_warm): the multipliers, bound multipliers and slacks of the previous solve stay; the barrier
  // parameter too unless the caller resets it.  Only entries that are unusable (non-positive) are re-initialised.
  const double mu_prev = *soa(a.scal, g, SC_COUNT, SC_MU);
  const double mu0 = !o.warm ? o.mu_init : (o.mu_warm > 0.0 ? o.mu_warm : (mu_prev > 0.0 ? mu_prev : o.mu_init));
  dispatch_uniform<M>(uload(a.kind, t), [&](auto kc) {
    constexpr int K = decltype(kc)::value;
    using D = KindDims<M, K>;
    using KD = typename D::KD;
    const int z0 = uload(a.zoff, t);
    arr<D::NP> p;
#pragma unroll
    for (int i = 0; i < D::NP; ++i) {
      double v = *soa(a.z, g, a.Nz, z0 + i);
      const double lo = uload(a.lo, z0 + i), hi = uload(a.hi, z0 + i);
      double zl = 0.0, zu = 0.0;
      if (!o.newton_only) {
        if (lo == hi) {
          v = lo;
        } else {
          const bool fl = finite_lo(lo), fh = finite_hi(hi);
          if (fl && fh) {
            const double pl = fmin(o.bound_push * fmax(1.0, fabs(lo)), o.bound_frac * (hi - lo));
            const double pu = fmin(o.bound_push * fmax(1.0, fabs(hi)), o.bound_frac * (hi - lo));
            v = fmin(fmax(v, lo + pl), hi - pu);
          } else if (fl) {
            v = fmax(v, lo + o.bound_push * fmax(1.0, fabs(lo)));
          } else if (fh) {
            v = fmin(v, hi - o.bound_push * fmax(1.0, fabs(hi)));
          }
          if (fl) zl = mu0 / (v - lo);
          if (fh) zu = mu0 / (hi - v);
          if (o.warm && a.zl) {
            const double pl = *soa(a.zl, g, a.Nz, z0 + i), pu = *soa(a.zu, g, a.Nz, z0 + i);
            if (fl && pl > 0.0) zl = pl;
            if (fh && pu > 0.0) zu = pu;
          }
        }
      }
      p[i] = v;
      *soa(a.z, g, a.Nz, z0 + i) = v;
      if (a.zl) {
        *soa(a.zl, g, a.Nz, z0 + i) = zl;
        *soa(a.zu, g, a.Nz, z0 + i) = zu;
      }
    }
    // multipliers: lam = 0; inequality rows: slack from c(z), nu = zs = mu0 / s
    // (newton_only keeps the caller's multipliers: dto_kkt_step evaluates at a given (z, lam); a warm start keeps them too)
    if constexpr (KD::DYN >= 0) {
      if (!o.newton_only && !o.warm) {
#pragma unroll
        for (int i = 0; i < D::NY; ++i) *soa(a.lam, g, a.Nc, uload(a.cdoff, t) + i) = 0.0;
      }
    }
    if constexpr (KD::CON >= 0) {
      if (!o.newton_only) {
      using C = typename M::template Con<KD::CON>;
      arr<C::NW> w; arr<C::NC> c;
      load_params(w, a, g, t);
      C::eval(p.data(), p.data() + C::NX, w.data(), c.data());
#pragma unroll
      for (int j = 0; j < C::NC; ++j) {
        double nu = o.warm ? *soa(a.lam, g, a.Nc, uload(a.ccoff, t) + j) : 0.0;
        if (D::ineq(j)) {
          double sv = fmax(-c[j], o.bound_push * fmax(1.0, fabs(c[j])));
          double zv = mu0 / sv;
          if (o.warm) {
            const double ps = *soa(a.s, g, a.Ni, uload(a.ioff, t) + D::slack(j)), pz = *soa(a.zs, g, a.Ni, uload(a.ioff, t) + D::slack(j));
            if (ps > 0.0 && pz > 0.0) { sv = ps; zv = pz; } else nu = zv;
          } else {
            nu = zv;
          }
          *soa(a.s, g, a.Ni, uload(a.ioff, t) + D::slack(j)) = sv;
          *soa(a.zs, g, a.Ni, uload(a.ioff, t) + D::slack(j)) = zv;
        }
        *soa(a.lam, g, a.Nc, uload(a.ccoff, t) + j) = nu;
      }
      }
    }
  });
  if (t == 0) {
    // the lanes of the last tile behind the batch are no instances: "finished" from the start (until round 5 they ran a solve
    // from the all-zero guess -- harmless, but a tile's inertia-correction rounds then waited for them too)
    *soa(a.scal, g, SC_COUNT, SC_STATUS) = lane_is_instance(a, g) ? 0.0 : DTO_ST_NO_INSTANCE;
    *soa(a.scal, g, SC_COUNT, SC_ITER) = 0.0;
    *soa(a.scal, g, SC_COUNT, SC_MU) = o.newton_only ? 0.0 : mu0;
    *soa(a.scal, g, SC_COUNT, SC_PENALTY) = 0.0;   // nu of the l1-penalty phase (ls_reduce_body)
    *soa(a.scal, g, SC_COUNT, SC_DELTA_W) = 0.0;
    *soa(a.scal, g, SC_COUNT, SC_DELTA_LAST) = 0.0;
    *soa(a.scal, g, SC_COUNT, SC_LS_FAIL) = 0.0;
    *soa(a.scal, g, SC_COUNT, SC_NFACT) = 0.0;
    *soa(a.scal, g, SC_COUNT, SC_ALPHA) = 0.0;
    *soa(a.scal, g, SC_COUNT, SC_THETA_MAX) = -1.0;  // set from theta_0 at the first convergence check
    *soa(a.scal, g, SC_COUNT, SC_THETA_MIN) = -1.0;
    *soa(a.scal, g, SC_COUNT, SC_FILTER_N) = 0.0;
    *soa(a.scal, g, SC_COUNT, SC_LS_KIND) = 0.0;
    *soa(a.scal, g, SC_COUNT, SC_QN_RESET) = 1.0;  // quasi-Newton blocks start from the objective Hessian
    *soa(a.scal, g, SC_COUNT, SC_FULL_STREAK) = 0.0;
    *soa(a.scal, g, SC_COUNT, SC_SHORT_STREAK) = 0.0;
    *soa(a.scal, g, SC_COUNT, SC_WATCHDOG) = 0.0;
    *soa(a.scal, g, SC_COUNT, SC_ACC_COUNT) = 0.0;
    *soa(a.scal, g, SC_COUNT, SC_F_LAST) = 1e300;
    *soa(a.scal, g, SC_COUNT, SC_XMAX) = 0.0;
    *soa(a.scal, g, SC_COUNT, SC_NNEG) = 0.0;
    *soa(a.scal, g, SC_COUNT, SC_LS_MODE) = (o.ls_penalty && !o.newton_only) ? 1.0 : 2.0;
    *soa(a.scal, g, SC_COUNT, SC_ASCALE) = 1.0;
    *soa(a.scal, g, SC_COUNT, SC_QN_SIGMA) = 1.0;   // Ipopt: limited_memory_init_val = 1
    *soa(a.scal, g, SC_COUNT, SC_QN_SKIP) = 0.0;
    *soa(a.scal, g, SC_COUNT, SC_QN_GCORR) = 0.0;
  }
}

// ------------------------------------------------------------------------------------------------
// linear-solver entry points (dto_kkt_assemble / dto_kkt_factor / dto_kkt_solve): the caller's right-hand side
// K v = (rhs_x, rhs_c) goes into the stage records (the sweeps solve K v = -(r_p, c, d)), and one factorisation with the
// fixed delta_w is requested
// ------------------------------------------------------------------------------------------------
template <class M>
__global__ __launch_bounds__(WAVE) void k_rhs_record(dto_kkt_args a) {
  const int64_t g = blockIdx.x / a.T;
  const int t = blockIdx.x % a.T;
  const int64_t slot = g * 64 + threadIdx.x;
  const int64_t inst = a.inst_of_slot ? a.inst_of_slot[slot] : slot;
  const bool live = inst < a.B;
  dispatch_uniform<M>(uload(a.kind, t), [&](auto kc) {
    constexpr int K = decltype(kc)::value;
    using D = KindDims<M, K>;
    double* rec = a.rec + ((g * a.rec_total + uload(a.recoff, t)) << 6) + 2 * threadIdx.x;   // pair-interleaved rows: pair_at()
    const double* rx = a.rhs_x + inst * a.ld_rhs_x;
    const double* rc = a.rhs_c + inst * a.ld_rhs_c;
#pragma unroll
    for (int i = 0; i < D::NP; ++i) rec[pair_at(D::R_RP + i)] = live ? -rx[uload(a.zoff, t) + i] : 0.0;
#pragma unroll
    for (int k = 0; k < D::NY; ++k) rec[pair_at(D::R_D + k)] = live ? -rc[uload(a.cdoff, t) + k] : 0.0;
#pragma unroll
    for (int j = 0; j < D::Q; ++j) rec[pair_at(D::R_C + j)] = live ? -rc[uload(a.ccoff, t) + j] : 0.0;
  });
}

static __global__ __launch_bounds__(WAVE) void k_rearm(dto_kkt_args a) {
  double* sc = a.scal + (((int64_t)blockIdx.x * SC_COUNT) << 6) + threadIdx.x;
  sc[SC_STATUS << 6] = lane_is_instance(a, blockIdx.x) ? 0.0 : DTO_ST_NO_INSTANCE;
  sc[SC_NEED << 6] = 1.0;
  sc[SC_ATTEMPT << 6] = 0.0;
  sc[SC_TRY_DW << 6] = a.opt.fixed_delta_w;
  sc[SC_TRY_GAM << 6] = 1.0;
}

// ------------------------------------------------------------------------------------------------
// per-stage derivative blocks.  grid = G*T waves; wave = (tile, stage); lane = instance.
// ------------------------------------------------------------------------------------------------
// UPD (k_update_eval, DTO_KKT_UPDATE_EVAL): the step of the iteration that just ended is taken on the way (k_update's
// arithmetic, expression for expression): the rows of a stage are read as z + alpha dz, lam + alpha dlam, written to z_next /
// lam_next (other buffers: a block also reads the rows of its neighbour stages, which another wavefront may be writing) and
// used from registers.  One pass over 36 rows per stage instead of 27 (k_update) + 22 (k_stage_eval).  Bound multipliers and
// slacks have no reader outside their own stage and are updated in place.
// Two instantiations of one body: every value the model's generated bodies consume passes through an empty asm in BOTH, so
// that the compiler sees the same expression graph over opaque registers and contracts it the same way (as loads in one
// and multiply-adds in the other it paired the products of a cost body  0.1 u^2 + 0.1 (x3^2 + x4^2)  differently: the
// objective of the fused pass differed from the two-kernel sequence in the last bit).  A run-time switch inside one kernel
// was tried instead: 256 + 162 registers, one wavefront per SIMD, both modes 1.5-2x slower.
template <class M, bool UPD>
__device__ __forceinline__ void stage_eval_body(const dto_kkt_args& a) {
  // a wavefront walks DTO_SB consecutive stages: the residual partials are summed in registers in stage order (one row set
  // per block instead of one per stage goes to memory), and E_t' lambda_t is handed to the next stage instead of
  // re-evaluating the previous stage's Jacobian there
  const int nblk = (a.T + a.sb - 1) / a.sb;
  const int64_t g = blockIdx.x / nblk;
  const int blk = blockIdx.x % nblk;
  const dto_solver_opts& o = a.opt;
  const int t_begin = blk * a.sb;
  double al = 0.0, ad = 0.0, mu_u = 0.0;
  // A finished instance takes no step.  In the fused pass its iterate still has to reach the other buffer: a tile whose lanes
  // have ALL finished copies the rows and leaves; in a tile that mixes finished and running lanes the finished ones go through
  // the pass like the others with the step masked out (z' = z, lam' = lam by a select -- their dz may be anything --, multipliers
  // and slacks untouched, the record recomputed from the unchanged iterate: the values it already holds).  As a branch of their
  // own (rounds 3 - 5) the few finished lanes of a tile made the wavefront run both paths one after the other: k_update_eval went
  // from 30 to 52 ms between iterations 15 and 30 of the bench's batch (tools/update_eval_growth.py).
  // (plugins with per-stage quasi-Newton blocks in their records keep the branch: a recomputed record would update those blocks)
  const bool fin = *soa(a.scal, g, SC_COUNT, SC_STATUS) != 0.0;
  constexpr bool MASKED = UPD && M::EVALUATE_HESSIAN != 0;
  if (MASKED ? __all(fin) : fin) {
    if constexpr (UPD) {
      const int te = (blk + 1) * a.sb < a.T ? (blk + 1) * a.sb : a.T;
      const int zi0 = uload(a.zoff, t_begin), zi1 = uload(a.zoff, te), di0 = uload(a.cdoff, t_begin), di1 = uload(a.cdoff, te),
                ci0 = uload(a.ccoff, t_begin), ci1 = uload(a.ccoff, te);
      // (eight loads, then eight stores: the compiler cannot tell the two buffers apart and keeps a load behind the store before it)
      auto copy_rows = [&](double* dst, const double* src, int64_t n, int i0, int i1) {
        for (int i = i0; i < i1; i += 8) {
          double tmp[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) tmp[k] = (i + k < i1) ? *soa(src, g, n, i + k) : 0.0;
#pragma unroll
          for (int k = 0; k < 8; ++k) if (i + k < i1) *soa(dst, g, n, i + k) = tmp[k];
        }
      };
      copy_rows(a.z_next, a.z, a.Nz, zi0, zi1);
      copy_rows(a.lam_next, a.lam, a.Nc, di0, di1);
      copy_rows(a.lam_next, a.lam, a.Nc, ci0, ci1);
    }
    return;
  }
  if constexpr (UPD) {
    al = *soa(a.scal, g, SC_COUNT, SC_ALPHA);
    ad = *soa(a.scal, g, SC_COUNT, SC_ALPHA_DMAX);
    mu_u = *soa(a.scal, g, SC_COUNT, SC_MU);
    if (blk == 0 && !fin) *soa(a.scal, g, SC_COUNT, SC_ITER) += 1.0;
  }
  // the iterate as this pass sees it
  // (the updated value passes through an empty asm: the compiler then cannot fuse its multiply-add into the arithmetic that
  // consumes it -- AMDGPU contracts aggressively, fma(x, y, fma(u, v, z)) for fma(x, y, u v) + z -- and the evaluation sees
  // exactly the double k_update would have stored: the objective differed in the last bit without it)
  auto zrow = [&](int i) -> double {
    double v;
    if constexpr (UPD) {
      const double v0 = *soa(a.z, g, a.Nz, i);
      v = v0 + al * *soa(a.dz, g, a.Nz, i);
      v = fin ? v0 : v;
    } else {
      v = *soa(a.z, g, a.Nz, i);
    }
    asm("" : "+v"(v));
    return v;
  };
  auto lrow = [&](int i) -> double {
    double v;
    if constexpr (UPD) {
      const double v0 = *soa(a.lam, g, a.Nc, i);
      v = v0 + al * *soa(a.dlam, g, a.Nc, i);
      v = fin ? v0 : v;
    } else {
      v = *soa(a.lam, g, a.Nc, i);
    }
    asm("" : "+v"(v));
    return v;
  };
  double A_f = 0.0, A_th1 = 0.0, A_thinf = 0.0, A_dinf = 0.0, A_szmax = 0.0, A_isz = 0.0, A_sumlam = 0.0, A_sumz = 0.0,
         A_logbar = 0.0, A_xmax = 0.0;
  double ecarry[M::MAX_NX], enext[M::MAX_NX];   // E_{t-1}' lambda_{t-1} from the previous stage / E_t' lambda_t for the next
  bool have_carry = false;
  const int t_end = (blk + 1) * a.sb < a.T ? (blk + 1) * a.sb : a.T;
  for (int t = t_begin; t < t_end; ++t) {
  dispatch_uniform<M>(uload(a.kind, t), [&](auto kc) {
    constexpr int K = decltype(kc)::value;
    using D = KindDims<M, K>;
    using KD = typename D::KD;
    using CO = typename M::template Cost<KD::COST>;
    const int z0 = uload(a.zoff, t);
    double* rec = a.rec + ((g * a.rec_total + uload(a.recoff, t)) << 6) + 2 * threadIdx.x;   // pair-interleaved rows: pair_at()
    // exact-Hessian models: the record is just the residuals (r_p, d, c) -- collected in registers and stored two rows at a
    // time (one 16-byte store per lane and pair) after the last of them; quasi-Newton records are written entry by entry
    double recv[D::FUSED && D::REC > 0 ? D::REC + 1 : 1];
    auto put = [&](int e, double v) { if constexpr (D::FUSED) recv[e] = v; else rec[pair_at(e)] = v; };

    arr<D::NP> p;
    if constexpr (UPD) {
      constexpr double KSIG = 1e10;
      StageBounds<D::NP> sbu;
      load_stage_bounds<D::NP>(a, g, z0, sbu);
#pragma unroll
      for (int i = 0; i < D::NP; ++i) {
        const double pold = *soa(a.z, g, a.Nz, z0 + i);
        const double dp = *soa(a.dz, g, a.Nz, z0 + i);
        double pn = pold + al * dp;
        pn = fin ? pold : pn;
        asm("" : "+v"(pn));
        if (!o.newton_only && !fin) {
          const double lo = sbu.lo[i], hi = sbu.hi[i];
          if (lo != hi) {
            if (finite_lo(lo)) {
              const double zl = sbu.zl[i];
              const double gap = pold - lo;
              const double dzl = mu_u / gap - zl - (zl / gap) * dp;
              double zn = zl + ad * dzl;
              const double gn = pn - lo;
              zn = fmin(fmax(zn, mu_u / (KSIG * gn)), KSIG * mu_u / gn);
              *soa(a.zl, g, a.Nz, z0 + i) = zn;
            }
            if (finite_hi(hi)) {
              const double zu = sbu.zu[i];
              const double gap = hi - pold;
              const double dzu = mu_u / gap - zu + (zu / gap) * dp;
              double zn = zu + ad * dzu;
              const double gn = hi - pn;
              zn = fmin(fmax(zn, mu_u / (KSIG * gn)), KSIG * mu_u / gn);
              *soa(a.zu, g, a.Nz, z0 + i) = zn;
            }
          }
        }
        p[i] = pn;
      }
    } else {
#pragma unroll
      for (int i = 0; i < D::NP; ++i) p[i] = zrow(z0 + i);
    }
    // every row of the stage is requested here, before any is used and before anything is stored (the compiler does not move
    // a load above a store it cannot tell apart): x_{t+1}, lambda_t, nu_t
    arr<D::NY> y, lam;
    arr<D::Q> nu;
#pragma unroll
    for (int i = 0; i < D::NY; ++i) {
      y[i] = zrow(uload(a.zoff, t + 1) + i);
      lam[i] = lrow(uload(a.cdoff, t) + i);
    }
#pragma unroll
    for (int j = 0; j < D::Q; ++j) nu[j] = lrow(uload(a.ccoff, t) + j);
    arr<CO::NW> wc;
    load_params(wc, a, g, t);

    // quasi-Newton mode: last iteration's Jacobian nonzeros (still in the record) for the secant pair
    double qn_old_dj[D::QN && D::N_DJ > 0 ? D::N_DJ : 1], qn_old_kj[D::QN && D::N_KJ > 0 ? D::N_KJ : 1];
    if constexpr (D::QN) {
#pragma unroll
      for (int i = 0; i < D::N_DJ; ++i) qn_old_dj[i] = rec[pair_at(D::R_DJ + i)];
#pragma unroll
      for (int i = 0; i < D::N_KJ; ++i) qn_old_kj[i] = rec[pair_at(D::R_KJ + i)];
    }
    arr<D::NP> rp;
    double cost_val;
    {
      double o1[1];
      CO::eval(p.data(), p.data() + CO::NX, wc.data(), o1);
      cost_val = o1[0];
      asm("" : "+v"(cost_val));   // summed below as a value: not contracted into the sum differently per instantiation
      CO::grad(p.data(), p.data() + CO::NX, wc.data(), rp.data());
      if constexpr (CO::SNH > 0 && !D::FUSED) {
        arr<CO::SNH> hv;
        arr<CO::NHL> hl;
        CO::shess(p.data(), p.data() + CO::NX, wc.data(), hv.data());
        CO::pack_hess_lower(hv.data(), hl.data());
#pragma unroll
        for (int i = 0; i < CO::NHL; ++i) put(D::R_CH + i, hl[i]);
      }
    }
    double th1 = 0.0, thinf = 0.0, sumlam = 0.0;

    if constexpr (KD::DYN >= 0) {
      using DY = typename M::template Dyn<KD::DYN>;
      arr<DY::NY> d;
      arr<DY::NW> w;
      load_params(w, a, g, t);
      arr<DY::NJ> jv;
      DY::eval_jac(p.data(), p.data() + DY::NX, y.data(), w.data(), d.data(), jv.data());
      DY::jtlam(jv.data(), lam.data(), rp.data());
#pragma unroll
      for (int i = 0; i < M::MAX_NX; ++i) enext[i] = 0.0;
      DY::etlam(jv.data(), lam.data(), enext);
      if constexpr (!D::FUSED) {
#pragma unroll
        for (int i = 0; i < DY::NJ; ++i) put(D::R_DJ + i, jv[i]);
      }
      if constexpr (DY::NH > 0 && !D::FUSED) {
        arr<DY::NH> hv;
        arr<DY::NHL> hl;
        DY::hess(p.data(), p.data() + DY::NX, y.data(), w.data(), lam.data(), hv.data());
        DY::pack_hess_lower(hv.data(), hl.data());
#pragma unroll
        for (int i = 0; i < DY::NHL; ++i) put(D::R_DH + i, hl[i]);
      }
#pragma unroll
      for (int i = 0; i < DY::NY; ++i) {
        put(D::R_D + i, d[i]);
        th1 += fabs(d[i]);
        thinf = fmax(thinf, fabs(d[i]));
        sumlam += fabs(lam[i]);
      }
    }
    // complementarity products s*z of this stage: largest and 1/smallest (k_conv measures max|s z - m| against any m:
    // the current barrier parameter, mu_target, and the candidates of the fast monotone decrease)
    double dinf = 0.0, szmax = 0.0, iszmax = 0.0, sumz = 0.0, logbar = 0.0, xmax = 0.0;
    if constexpr (KD::CON >= 0) {
      using C = typename M::template Con<KD::CON>;
      arr<C::NW> w; arr<C::NC> c; arr<C::NJ> jv;
      load_params(w, a, g, t);
#pragma unroll
      for (int j = 0; j < C::NC; ++j) {
        if constexpr (UPD) {
          if (!o.newton_only && D::ineq(j) && !fin) {
            constexpr double KSIG = 1e10;
            const int si = uload(a.ioff, t) + D::slack(j);
            const double sv = *soa(a.s, g, a.Ni, si);
            const double zv = *soa(a.zs, g, a.Ni, si);
            const double dsv = *soa(a.ds, g, a.Ni, si);
            const double dzs = mu_u / sv - zv - (zv / sv) * dsv;
            double sn = sv + al * dsv;
            asm("" : "+v"(sn));
            double zn = zv + ad * dzs;
            zn = fmin(fmax(zn, mu_u / (KSIG * sn)), KSIG * mu_u / sn);
            *soa(a.s, g, a.Ni, si) = sn;
            *soa(a.zs, g, a.Ni, si) = zn;
          }
        }
      }
      C::eval(p.data(), p.data() + C::NX, w.data(), c.data());
      C::jac(p.data(), p.data() + C::NX, w.data(), jv.data());
      C::jtlam(jv.data(), nu.data(), rp.data());
      if constexpr (!D::FUSED) {
#pragma unroll
        for (int i = 0; i < C::NJ; ++i) put(D::R_KJ + i, jv[i]);
      }
      if constexpr (C::NH > 0 && !D::FUSED) {
        arr<C::NH> hv;
        arr<C::NHL> hl;
        C::hess(p.data(), p.data() + C::NX, w.data(), nu.data(), hv.data());
        C::pack_hess_lower(hv.data(), hl.data());
#pragma unroll
        for (int i = 0; i < C::NHL; ++i) put(D::R_KH + i, hl[i]);
      }
#pragma unroll
      for (int j = 0; j < C::NC; ++j) {
        double r = c[j];
        if (!o.newton_only && D::ineq(j)) {
          const double sv = *soa(a.s, g, a.Ni, uload(a.ioff, t) + D::slack(j));
          const double zv = *soa(a.zs, g, a.Ni, uload(a.ioff, t) + D::slack(j));
          r = c[j] + sv;
          dinf = fmax(dinf, fabs(nu[j] - zv));
          szmax = fmax(szmax, sv * zv);
          iszmax = fmax(iszmax, 1.0 / (sv * zv));
          sumz += fabs(zv);
          logbar += log(sv);
        }
        put(D::R_C + j, r);
        th1 += fabs(r);
        thinf = fmax(thinf, fabs(r));
        sumlam += fabs(nu[j]);
      }
    }
    // E_{t-1}' lam_{t-1}: handed over by the previous stage of this block; the first stage of a block re-evaluates the
    // previous stage's Jacobian (cheaper than a carry pass through memory)
    if constexpr (KD::PREV >= 0) {
      using DP = typename M::template Dyn<KD::PREV>;
      if (have_carry) {
#pragma unroll
        for (int i = 0; i < DP::NY; ++i) rp[i] += ecarry[i];
      } else {
      arr<DP::NX + DP::NU> pp; arr<DP::NY> lamp; arr<DP::NW> w; arr<DP::NJ> jv;
      load_params(w, a, g, t - 1);
#pragma unroll
      for (int i = 0; i < DP::NX + DP::NU; ++i) pp[i] = zrow(uload(a.zoff, t - 1) + i);
#pragma unroll
      for (int i = 0; i < DP::NY; ++i) lamp[i] = lrow(uload(a.cdoff, t - 1) + i);
      DP::jac(pp.data(), pp.data() + DP::NX, p.data(), w.data(), jv.data());
      DP::etlam(jv.data(), lamp.data(), rp.data());
      }
    }
    if constexpr (D::QN) {
      // ---- partitioned SR1 on the element Hessian of this stage
      constexpr int NE = D::NE, NBQ = NE * (NE + 1) / 2;
      const bool reset = (*soa(a.scal, g, SC_COUNT, SC_QN_RESET) != 0.0);
      const double alpha = *soa(a.scal, g, SC_COUNT, SC_ALPHA);
      double B[NBQ > 0 ? NBQ : 1];
      // new and old element gradients, both with the CURRENT multipliers (Ipopt's L-BFGS secant pair)
      double gn[NE > 0 ? NE : 1], go[NE > 0 ? NE : 1], sv[NE > 0 ? NE : 1];
#pragma unroll
      for (int i = 0; i < NE; ++i) gn[i] = go[i] = sv[i] = 0.0;
      {
        arr<D::NP> gc;
        CO::grad(p.data(), p.data() + CO::NX, wc.data(), gc.data());
#pragma unroll
        for (int i = 0; i < D::NP; ++i) {
          gn[i] = gc[i];
          go[i] = rec[pair_at(D::R_GC + i)];
          put(D::R_GC + i, gc[i]);
          sv[i] = alpha * *soa(a.dz, g, a.Nz, z0 + i);
        }
      }
      if constexpr (KD::DYN >= 0) {
        using DY = typename M::template Dyn<KD::DYN>;
        arr<DY::NW> w; arr<DY::NJ> jn, jo;
        load_params(w, a, g, t);
#pragma unroll
        for (int i = 0; i < DY::NY; ++i) {
          sv[D::NP + i] = alpha * *soa(a.dz, g, a.Nz, uload(a.zoff, t + 1) + i);
        }
        DY::jac(p.data(), p.data() + DY::NX, y.data(), w.data(), jn.data());
#pragma unroll
        for (int i = 0; i < DY::NJ; ++i) jo[i] = qn_old_dj[i];
        DY::jtlam(jn.data(), lam.data(), gn);
        DY::etlam(jn.data(), lam.data(), gn + D::NP);
        DY::jtlam(jo.data(), lam.data(), go);
        DY::etlam(jo.data(), lam.data(), go + D::NP);
      }
      if constexpr (KD::CON >= 0) {
        using C = typename M::template Con<KD::CON>;
        arr<C::NW> w; arr<C::NJ> jn, jo;
        load_params(w, a, g, t);
        C::jac(p.data(), p.data() + C::NX, w.data(), jn.data());
#pragma unroll
        for (int i = 0; i < C::NJ; ++i) jo[i] = qn_old_kj[i];
        C::jtlam(jn.data(), nu.data(), gn);
        C::jtlam(jo.data(), nu.data(), go);
      }
      if (reset) {
        // start from the (exact) objective Hessian plus a small multiple of the identity
#pragma unroll
        for (int i = 0; i < NBQ; ++i) B[i] = 0.0;
        if constexpr (CO::SNH > 0) {
          arr<CO::SNH> hv; arr<CO::NHL> hl;
          CO::shess(p.data(), p.data() + CO::NX, wc.data(), hv.data());
          CO::pack_hess_lower(hv.data(), hl.data());
          CO::scatter_hess_lower(hl.data(), B);
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) B[tri(i, i)] += 1e-8;
      } else {
#pragma unroll
        for (int i = 0; i < NBQ; ++i) B[i] = rec[pair_at(D::R_B + i)];
        // symmetric rank-one update: element Hessians of lam'd are indefinite, which SR1 can represent (BFGS cannot);
        // the inertia ladder of the factorization takes care of the resulting indefinite reduced Hessians
        double v[NE > 0 ? NE : 1];
        double vs = 0.0, vv = 0.0, ss = 0.0;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
          double acc = 0.0;
#pragma unroll
          for (int j = 0; j < NE; ++j) acc += B[i >= j ? tri(i, j) : tri(j, i)] * sv[j];
          v[i] = (gn[i] - go[i]) - acc;
          vs += v[i] * sv[i];
          vv += v[i] * v[i];
          ss += sv[i] * sv[i];
        }
        if (ss > 1e-24 && fabs(vs) > 1e-8 * sqrt(vv * ss)) {
          const double ivs = 1.0 / vs;
#pragma unroll
          for (int i = 0; i < NE; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) B[tri(i, j)] += v[i] * v[j] * ivs;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NBQ; ++i) put(D::R_B + i, B[i]);
    }
    StageBounds<D::NP> sb;
    load_stage_bounds<D::NP>(a, g, z0, sb);
#pragma unroll
    for (int i = 0; i < D::NP; ++i) put(D::R_RP + i, rp[i]);
    if constexpr (D::FUSED) {
#pragma unroll
      for (int i = 0; i + 1 < D::REC; i += 2) *reinterpret_cast<double2*>(rec + ((int64_t)(i >> 1) << 7)) = double2{recv[i], recv[i + 1]};
      if constexpr (D::REC % 2 == 1) rec[pair_at(D::REC - 1)] = recv[D::REC - 1];
    }
#pragma unroll
    for (int i = 0; i < D::NP; ++i) {
      xmax = fmax(xmax, fabs(p[i]));
      const double lo = sb.lo[i], hi = sb.hi[i];
      if (o.newton_only) {
        dinf = fmax(dinf, fabs(rp[i]));
      } else if (lo != hi) {
        const double zl = sb.zl[i], zu = sb.zu[i];
        dinf = fmax(dinf, fabs(rp[i] - zl + zu));
        if (finite_lo(lo)) {
          szmax = fmax(szmax, (p[i] - lo) * zl);
          iszmax = fmax(iszmax, 1.0 / ((p[i] - lo) * zl));
          sumz += fabs(zl);
          logbar += log(p[i] - lo);
        }
        if (finite_hi(hi)) {
          szmax = fmax(szmax, (hi - p[i]) * zu);
          iszmax = fmax(iszmax, 1.0 / ((hi - p[i]) * zu));
          sumz += fabs(zu);
          logbar += log(hi - p[i]);
        }
      }
    }
    if constexpr (UPD) {
#pragma unroll
      for (int i = 0; i < D::NP; ++i) *soa(a.z_next, g, a.Nz, z0 + i) = p[i];
#pragma unroll
      for (int i = 0; i < D::NY; ++i) *soa(a.lam_next, g, a.Nc, uload(a.cdoff, t) + i) = lam[i];
#pragma unroll
      for (int j = 0; j < D::Q; ++j) *soa(a.lam_next, g, a.Nc, uload(a.ccoff, t) + j) = nu[j];
    }
    A_f += cost_val;
    A_th1 += th1;
    A_thinf = fmax(A_thinf, thinf);
    A_dinf = fmax(A_dinf, dinf);
    A_szmax = fmax(A_szmax, szmax);
    A_isz = fmax(A_isz, iszmax);
    A_sumlam += sumlam;
    A_sumz += sumz;
    A_logbar += logbar;
    A_xmax = fmax(A_xmax, xmax);
    have_carry = (KD::DYN >= 0);
#pragma unroll
    for (int i = 0; i < M::MAX_NX; ++i) ecarry[i] = enext[i];
  });
  }
  double* part = a.part + (((g * nblk + blk) * DTO_NPART) << 6) + threadIdx.x;
  part[0 << 6] = A_f;
  part[1 << 6] = A_th1;
  part[2 << 6] = A_thinf;
  part[3 << 6] = A_dinf;
  part[4 << 6] = A_szmax;
  part[5 << 6] = A_isz;
  part[6 << 6] = A_sumlam;
  part[7 << 6] = A_sumz;
  part[8 << 6] = A_logbar;
  part[9 << 6] = A_xmax;
}

template <class M>
__global__ __launch_bounds__(WAVE) void k_stage_eval(dto_kkt_args a) { stage_eval_body<M, false>(a); }
#ifndef DTO_UPD_OCC
#define DTO_UPD_OCC 0   // measurement: minimum wavefronts per SIMD asked of the compiler for k_update_eval (0: its own choice, 2)
#endif
#if DTO_UPD_OCC > 0
template <class M>
__global__ __launch_bounds__(WAVE, DTO_UPD_OCC) void k_update_eval(dto_kkt_args a) { stage_eval_body<M, true>(a); }
#else
template <class M>
__global__ __launch_bounds__(WAVE) void k_update_eval(dto_kkt_args a) { stage_eval_body<M, true>(a); }
#endif

// ------------------------------------------------------------------------------------------------
// first level of the deterministic reductions: every (tile, chunk) wave folds the per-stage partials of its
// chunk in stage order (sum, or max for the slots in `maxmask`); k_conv / k_ls_reduce then fold P chunk rows
// instead of T stage rows.  grid = G*P waves.
// ------------------------------------------------------------------------------------------------
// The per-tile join of a step that many wavefronts of one launch contribute to (P chunk wavefronts per tile): every wavefront
// arrives at the tile's counter after its own work (release), and the one that finds it at n - 1 is the last -- it resets the
// counter, acquires, and runs the join (separator system, convergence test, ...) in the same launch instead of a launch of its
// own.  No wavefront ever waits for another: no residency requirement, works for any grid.  A batch of one is bound by its
// chain of dependent launches (27 per iteration before, ~4 us each when there is little to do: profiles/r05/).
// (Measured: with an agent-scope release fence in every arriving wavefront the in-launch joins lost beyond 16 chunks --
// profiles/r05/join_in_launch_ab.txt; with the join data itself at agent scope, below, a batch of one gains at every chunk count
// -- join_in_launch_sc1_ab.txt: acrobot T=101 0.104 -> 0.089 ms per iteration, T=1000 0.323 -> 0.300, pendulum T=50 0.095 -> 0.045
// -- while batches of 16 - 384 tiles lose 0 - 5 %: the host hands out the counters up to two tiles, dto_solver.cpp: fill_kkt_args.)
// What a join reads of the other wavefronts' work -- chunk summaries, chunk sums, step partials: a few rows each -- is written and
// read at agent scope (global_store / global_load ... sc1: through to memory, past the per-XCD L2), so that the arrival needs no
// L2 write-back (`buffer_wbl2`: what a release fence at agent scope costs EVERY arriving wavefront -- with it the in-launch joins
// were a loss beyond 16 chunks, profiles/r05/join_in_launch_ab.txt) and the last wavefront no L2 invalidation: the arriving
// wavefront waits for its own stores (s_waitcnt vmcnt(0)), then counts itself in; everything else it wrote (carry records, steps)
// is for later launches and becomes visible at the end of this one as always.
__device__ __forceinline__ void st_join(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_join(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool tile_last_arrival(int* ctr, int n) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int old = 0;
  if (threadIdx.x == 0) old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = __builtin_amdgcn_readfirstlane(old);
  if (old + 1 != n) return false;
  if (threadIdx.x == 0) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" ::: "memory");
  return true;
}

template <int NV>
__device__ __forceinline__ void part_reduce_body(const dto_kkt_args& a, const double* in, unsigned maxmask) {
  const int64_t g = blockIdx.x / a.P;
  const int p = blockIdx.x % a.P;
  if (*soa(a.scal, g, SC_COUNT, SC_STATUS) != 0.0) return;
  double acc[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) acc[k] = 0.0;
  // a block of DTO_SB stages belongs to the chunk that holds its first stage (every block is counted exactly once)
  const int nblk = (a.T + a.sb - 1) / a.sb;
  const int b0 = (uload(a.cstart, p) + a.sb - 1) / a.sb, b1 = (uload(a.cstart, p + 1) + a.sb - 1) / a.sb;
  // (partially unrolled: the rows of several blocks in flight at once; the sums are still taken in block order)
#pragma unroll 4
  for (int t = b0; t < b1; ++t) {
    const double* row = in + (((g * nblk + t) * NV) << 6) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const double v = row[(int64_t)k << 6];
      acc[k] = ((maxmask >> k) & 1u) ? fmax(acc[k], v) : acc[k] + v;
    }
  }
  double* out = a.cpart + (((g * a.P + p) * 16) << 6) + threadIdx.x;
#pragma unroll
  for (int k = 0; k < NV; ++k) st_join(&out[(int64_t)k << 6], acc[k]);
}
template <int NV>
static __global__ __launch_bounds__(WAVE) void k_part_reduce(dto_kkt_args a, const double* in, unsigned maxmask) {
  part_reduce_body<NV>(a, in, maxmask);
}

// ------------------------------------------------------------------------------------------------
// reduce the stage partials in stage order (deterministic), test convergence, update mu.
// grid = G waves.
// ------------------------------------------------------------------------------------------------
// Convergence test, barrier update and the factorisation request of one instance, given the residual norms of its current
// iterate (summed over the stages by the caller).  SH as in retry_update_t.
struct ConvSums {
  double f, th1, thinf, dinf, szmax, iszmax, slam, sz, lb, xmax;
};
template <int SH>
__device__ __forceinline__ void conv_body(const dto_solver_opts& o, double* sc, const ConvSums& cs, int64_t n_mult, int64_t n_bnd) {
  const double f = cs.f, th1 = cs.th1, thinf = cs.thinf, dinf = cs.dinf, szmax = cs.szmax, iszmax = cs.iszmax, slam = cs.slam,
               sz = cs.sz, lb = cs.lb, xmax = cs.xmax;
  double mu = sc[SC_MU << SH];
  const double sd = fmax(o.s_max, (slam + sz) / (double)(n_mult + n_bnd > 0 ? n_mult + n_bnd : 1)) / o.s_max;
  const double scn = fmax(o.s_max, sz / (double)(n_bnd > 0 ? n_bnd : 1)) / o.s_max;
  // max |s_i z_i - m| over all bound / slack pairs, for any m
  const double szmin = iszmax > 0.0 ? 1.0 / iszmax : 1e300;
  auto compl_at = [&](double m) { return n_bnd > 0 ? fmax(szmax - m, m - szmin) : 0.0; };
  // Ipopt's mu_target: the termination tests measure complementarity against the target barrier parameter
  const double c0 = compl_at(o.mu_target);
  const double e0 = fmax(fmax(dinf / sd, thinf), c0 / scn);
  sc[SC_F << SH] = f;
  sc[SC_THETA1 << SH] = th1;
  sc[SC_THETA_INF << SH] = thinf;
  sc[SC_DINF << SH] = dinf;
  sc[SC_COMPL << SH] = c0;
  sc[SC_E0 << SH] = e0;
  sc[SC_LOGBAR << SH] = lb;
  sc[SC_XMAX << SH] = xmax;
  const double iter = sc[SC_ITER << SH];
  const bool nonfinite = !(f == f) || !(th1 == th1) || !(dinf == dinf) || fabs(f) > 1e300 || th1 > 1e300;
  // Ipopt's acceptable level (OptimalityErrorConvergenceCheck::CurrentIsAcceptable), counted over consecutive iterations
  const double f_last = sc[SC_F_LAST << SH];
  const bool acceptable = o.acceptable_iter > 0 && e0 <= o.acceptable_tol && dinf <= o.acceptable_dual_inf_tol &&
                          thinf <= o.acceptable_constr_viol_tol && c0 <= o.acceptable_compl_inf_tol &&
                          fabs(f - f_last) / fmax(1.0, fabs(f)) <= o.acceptable_obj_change_tol;
  const double acc_count = acceptable ? sc[SC_ACC_COUNT << SH] + 1.0 : 0.0;
  sc[SC_ACC_COUNT << SH] = acc_count;
  sc[SC_F_LAST << SH] = f;
  if (nonfinite) {
    sc[SC_STATUS << SH] = 3.0;
  } else if (e0 <= o.tol && dinf <= o.dual_inf_tol && thinf <= o.constr_viol_tol && c0 <= o.compl_inf_tol) {
    sc[SC_STATUS << SH] = 1.0;
  } else if (o.acceptable_iter > 0 && acc_count >= (double)o.acceptable_iter) {
    sc[SC_STATUS << SH] = 4.0;
  } else if (!o.newton_only && xmax > o.diverging_iterates_tol) {
    sc[SC_STATUS << SH] = 5.0;
  } else if (iter >= (double)o.max_iter) {
    sc[SC_STATUS << SH] = 2.0;
  } else if (n_bnd > 0) {
    // monotone barrier update (Ipopt MonotoneMuUpdate, mu_allow_fast_monotone_decrease = yes): while the barrier problem is
    // solved to kappa_eps * mu the parameter drops, but never below max(mu_target, min(tol, compl_inf_tol) / (kappa_eps + 1))
    const double mu_floor = fmax(o.mu_target, fmin(o.tol, o.compl_inf_tol) / (o.kappa_eps + 1.0));
    bool changed = false;
    for (int k = 0; k < 8; ++k) {
      const double emu = fmax(fmax(dinf / sd, thinf), compl_at(mu) / scn);
      if (!(emu <= o.kappa_eps * mu) || mu <= mu_floor) break;
      mu = fmax(mu_floor, fmin(o.kappa_mu * mu, pow(mu, o.theta_mu)));
      changed = true;
    }
    if (changed) {
      sc[SC_MU << SH] = mu;
      sc[SC_FILTER_N << SH] = 0.0;  // new barrier problem: the filter is reset (Ipopt, step A-3)
    }
  }
  if (sc[SC_THETA_MAX << SH] < 0.0) {
    sc[SC_THETA_MAX << SH] = 1e4 * fmax(1.0, th1);
    sc[SC_THETA_MIN << SH] = 1e-4 * fmax(1.0, th1);
  }
  // merit value of the current iterate with the (possibly updated) barrier parameter
  sc[SC_MERIT0 << SH] = f - mu * lb;
  // end of the penalty phase of the line search (ls_reduce_body): near the constraint manifold the filter takes over for good
  const bool penalty_phase = !o.newton_only && sc[SC_LS_MODE << SH] == 1.0 && thinf > o.ls_switch;
  if (!penalty_phase && sc[SC_LS_MODE << SH] == 1.0) {
    sc[SC_LS_MODE << SH] = 2.0;
    sc[SC_FILTER_N << SH] = 0.0;
  }
  // factorisation request of this iteration (consumed by k_kkt_fwd / k_kkt_sep)
  sc[SC_NEED << SH] = (o.newton_only || sc[SC_STATUS << SH] == 0.0) ? 1.0 : 0.0;
  sc[SC_ATTEMPT << SH] = 0.0;
  sc[SC_TRY_GAM << SH] = 1.0;  // exact Hessian of the Lagrangian first
  if (o.newton_only) {
    sc[SC_TRY_DW << SH] = o.fixed_delta_w;
  } else {
    const double dlast = sc[SC_DELTA_LAST << SH];
    // Ipopt's Algorithm IC: always try the unmodified matrix first (unless the last line search failed)
    // after a failed line search start from a larger regularisation -- but never beyond the exact-Hessian cap: without
    // the cap an instance whose trials keep being rejected by the filter multiplies delta_w by 10 every iteration
    // (1e163 was observed), its steps vanish and it can never leave that state
    // (limited-memory mode: delta_last is never recorded there (gam = 0), so the escalation starts from the delta_w the rejected
    //  direction was computed with -- with delta_w_init again and again a null step repeated itself for the rest of the
    //  iterations: same point, same direction; 24 of 4 096 acrobot T = 101 instances, round 6)
    if (sc[SC_LS_FAIL << SH] != 0.0)
      sc[SC_TRY_DW << SH] = fmin(o.delta_w_exact_cap, fmax(10.0 * fmax(dlast, o.qn_lbfgs ? sc[SC_DELTA_W << SH] : 0.0), o.delta_w_init));
    // Ipopt's Algorithm IC probes delta_w = 0 in every iteration.  While the last iteration needed a regularisation well
    // above the floor that probe almost always fails and costs a whole factorisation, so it is skipped and the ladder is
    // entered at kappa_w^- delta_last directly; 0 is probed again once delta_w has decayed to the floor or after two
    // consecutive full steps (fast local convergence needs the unmodified matrix).  C port, acrobot: T=101 -18 %
    // factorisations, -7 % iterations; T=301 -31 % / -10 %; pendulum unchanged; every instance still converges.
    // (round 6: the decaying delta_w is floored at delta_w^min = 1e-20 as in Ipopt's IpPDPerturbationHandler, not at delta_w_init:
    //  with a floor of 1e-4 the acrobot T = 1000 instances that enter a valley whose reduced Hessian has an eigenvalue of 2e-7 are
    //  frozen along it -- 227 of 256 seeds converged on the C port, 256 of 256 with Ipopt's floor)
    else if (dlast > 1.1 * o.delta_w_min && sc[SC_FULL_STREAK << SH] < 2.0)
      sc[SC_TRY_DW << SH] = fmax(o.delta_w_min, o.kappa_w_minus * dlast);
    else sc[SC_TRY_DW << SH] = 0.0;
    // penalty phase: Gauss-Newton model (constraint curvature dropped), delta_w >= delta_w_init.  Far from the manifold the exact
    // Hessian is so indefinite that the ladder ends at delta_w ~ 10 .. 100 anyway -- its curvature is swamped while every probe
    // costs a sweep of the whole tile; the Gauss-Newton matrix has the right inertia by construction (one factorisation) and, under
    // the penalty line search, needs fewer iterations (C port, acrobot T=1000 x 128: median 47 instead of 57, 87 % instead of 78 %
    // converged; T=101: 38 / 46; the configs that start near the manifold never enter the phase)
    if (penalty_phase && o.pen_gn) {
      sc[SC_TRY_GAM << SH] = 0.0;
      sc[SC_TRY_DW << SH] = fmax(sc[SC_TRY_DW << SH], o.delta_w_init);
    }
    if (o.qn_lbfgs) sc[SC_TRY_GAM << SH] = 0.0;   // no second derivatives at all: K0 = [sigma I + Sigma + delta_w I, J'; J, -D]
  }
}

__device__ __forceinline__ void conv_tile(const dto_kkt_args& a, const int64_t g, int64_t n_mult, int64_t n_bnd) {
  double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  if (sc[SC_STATUS << 6] != 0.0) return;
  double f = 0, th1 = 0, thinf = 0, dinf = 0, szmax = 0, iszmax = 0, slam = 0, sz = 0, lb = 0, xmax = 0;
  // (partially unrolled: the loads of several chunks in flight at once -- with one running lane this loop is a chain of
  //  memory latencies, 23 us for 64 chunks; the sums are still taken in chunk order)
#pragma unroll 8
  for (int c = 0; c < a.P; ++c) {
    const double* part = a.cpart + (((g * a.P + c) * 16) << 6) + threadIdx.x;
    f += ld_join(&part[0 << 6]);
    th1 += ld_join(&part[1 << 6]);
    thinf = fmax(thinf, ld_join(&part[2 << 6]));
    dinf = fmax(dinf, ld_join(&part[3 << 6]));
    szmax = fmax(szmax, ld_join(&part[4 << 6]));
    iszmax = fmax(iszmax, ld_join(&part[5 << 6]));
    slam += ld_join(&part[6 << 6]);
    sz += ld_join(&part[7 << 6]);
    lb += ld_join(&part[8 << 6]);
    xmax = fmax(xmax, ld_join(&part[9 << 6]));
  }
  conv_body<6>(a.opt, sc, ConvSums{f, th1, thinf, dinf, szmax, iszmax, slam, sz, lb, xmax}, n_mult, n_bnd);
}
static __global__ __launch_bounds__(WAVE) void k_conv(dto_kkt_args a, int64_t n_mult, int64_t n_bnd) { conv_tile(a, blockIdx.x, n_mult, n_bnd); }
// chunk sums and, in the last wavefront of a tile to finish them, the convergence test (grid = G * P)
static __global__ __launch_bounds__(WAVE) void k_part_reduce_conv(dto_kkt_args a, const double* in, unsigned maxmask, int64_t n_mult, int64_t n_bnd) {
  part_reduce_body<DTO_NPART>(a, in, maxmask);
  const int64_t g = blockIdx.x / a.P;
  if (tile_last_arrival(a.csync + g * 4 + 0, a.P)) conv_tile(a, g, n_mult, n_bnd);
}

// ------------------------------------------------------------------------------------------------
// dense block helpers (all indices are literals after unrolling)
// ------------------------------------------------------------------------------------------------
// 1 / x for a pivot: v_rcp_f64 and two Newton steps (5 instructions, a dependent chain of 5) instead of the IEEE division
// sequence (14 instructions, a chain of 10) -- the reciprocals of the pivots are the longest dependent chains of a stage and,
// at one wavefront per SIMD, exposed.  Within 1 ulp of the correctly rounded quotient (tools/micro/rcp_accuracy.hip).
#ifndef DTO_PIVOT_RCP
#define DTO_PIVOT_RCP 1
#endif
__device__ __forceinline__ double pivot_recip(double x) {
#if DTO_PIVOT_RCP
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}

template <int BD>
__device__ __forceinline__ void ldl_inplace(double* S, double* dinv, double piv_tol, bool& ok, int& nneg) {
  // right-looking LDL^T on the packed lower triangle, static order, no pivoting; L overwrites the strict
  // lower part.  Negative pivots are counted (Sylvester: their total over the whole block-tridiagonal
  // factorisation is the number of negative eigenvalues of K); a pivot that is tiny relative to its
  // column marks the factorisation as unusable so that the caller raises delta_w.
#pragma unroll
  for (int j = 0; j < BD; ++j) {
    double dj = S[tri(j, j)];
    double cmax = 0.0;
#pragma unroll
    for (int i = j + 1; i < BD; ++i) cmax = fmax(cmax, fabs(S[tri(i, j)]));
    if (!(fabs(dj) > piv_tol * fmax(1.0, cmax))) {
      ok = false;
      dj = (dj < 0.0 ? -1.0 : 1.0) * fmax(fabs(dj), piv_tol);
    }
    if (dj < 0.0) ++nneg;
    const double inv = pivot_recip(dj);
    dinv[j] = inv;
#pragma unroll
    for (int i = j + 1; i < BD; ++i) {
      const double lij = S[tri(i, j)] * inv;
#pragma unroll
      for (int k = j + 1; k <= i; ++k) S[tri(i, k)] -= lij * S[tri(k, j)];
    }
#pragma unroll
    for (int i = j + 1; i < BD; ++i) S[tri(i, j)] *= inv;
  }
}

// ------------------------------------------------------------------------------------------------
// block-tridiagonal LDL^T, parallel in time.
//
// The horizon is cut into P chunks [a_p, a_{p+1}).  The state x_{a_p} at the head of every chunk
// p >= 1 is a *separator*: with the separators fixed the chunks decouple, so
//   k_kkt_fwd  (G*P waves): every chunk eliminates its interior by the forward sweep below while
//              carrying the coupling of its interior to the left separator ("spike" Z = L^-1 C) and
//              accumulates its Schur-complement contributions R_LL, R_LR, r_L (left separator) and
//              P, py (right separator);
//   k_kkt_sep  (G waves):   the reduced block-tridiagonal system over the P-1 separators (n x n
//              blocks) is factorised and solved, the inertia of the whole KKT matrix is summed
//              (Sylvester) and the per-instance retry state machine (delta_w ladder / Gauss-Newton
//              fallback) advances; the two kernels are launched a fixed number of rounds and exit at
//              once when no instance of the tile needs another factorisation;
//   k_kkt_bwd  (G*P waves): back substitution inside every chunk given x_L and x_R, step-length
//              bounds and merit derivative partials;
//   k_kkt_post (G waves):   deterministic reduction of the chunk partials.
// P = 1 is the plain sequential sweep.  Work grows by ~1.6x (the spike) while the number of
// wavefronts grows P-fold, which is what fills the 1024 SIMDs at moderate batch sizes.
// ------------------------------------------------------------------------------------------------
template <class M>
struct Carry {
  double P[M::MAX_NX * (M::MAX_NX + 1) / 2];  // Schur complement on x_{t+1} (packed lower)
  double py[M::MAX_NX];                        // rhs carry
};

template <class M>
struct Spike {
  double Cx[M::MAX_NX * M::MAX_NX];              // K[x_t rows, x_L cols] of the stage about to be eliminated
  double RLL[M::MAX_NX * (M::MAX_NX + 1) / 2];   // separator diagonal block contribution
  double rL[M::MAX_NX];                          // separator rhs contribution
};

// Build the stage block S_t (with the carry-in P_t, py and, in chunks p >= 1, the spike coupling Cx), factorise
// it and run the forward substitutions.  On return S holds L (strict lower), dinv = 1/D, X = L^-1 O, y = w,
// Z = L^-1 C.  Used by BOTH sweeps: the backward sweep recomputes instead of reading stored factors.
// BWD (backward sweep): the step of the neighbouring blocks is already known (xn = dx_{t+1}, xL = the chunk's left
// separator), so the couplings are folded into the right-hand side BEFORE the factorisation,
//     v_t = L^-T D^-1 L^-1 (y - O xn - C xL),
// and the panels X = L^-1 O, Z = L^-1 C are never formed: the backward sweep needs neither their registers (72 doubles for
// the acrobot) nor their substitutions.
// ASSEMBLE_ONLY (k_kkt_refine, round 6): stop before the factorisation -- S, X = O_t, YYl, y then hold the stage's rows of K and
// of the right-hand side exactly as the sweeps factorise them (with an empty carry: of K itself).  A template constant so that
// the sweeps' instantiations are, token for token, what they were: their code generation is sensitive to the shape of this function.
template <class M, int K, bool SPK, bool BWD = false, bool ASSEMBLE_ONLY = false, class IO>
__device__ __forceinline__ void stage_factor(const dto_solver_opts& o, const IO& io, double mu, double dw, double gam,
                                             bool first, const Carry<M>& cy, Spike<M>& sp, double* S, double* y,
                                             double* X, double* YYl, double* Z, double* cx_direct, double* dinv,
                                             bool& ok, int& nneg, double* keep, const double* xn = nullptr,
                                             const double* xL = nullptr) {
  using D = KindDims<M, K>;
  constexpr int NP = D::NP, Q = D::Q, NY = D::NY, BD = D::BD, NX = D::NX;
  using KD = typename D::KD;
  using CO = typename M::template Cost<KD::COST>;
  // the stage record (residuals; plus the quasi-Newton blocks of models without exact Hessians): every row is requested
  // before any is used, together with the iterate the derivative code below needs -- one memory latency per stage
  double rr[D::REC > 0 ? D::REC : 1];
#pragma unroll
  for (int i = 0; i < D::REC; ++i) rr[i] = io.rec(i);
  auto R = [&](int e) { return rr[e]; };
  arr<NP> pv;
  arr<NY> yv, lamv;
  arr<Q> nuv;
  if constexpr (D::FUSED) {
#pragma unroll
    for (int i = 0; i < NP; ++i) pv[i] = io.p(i);
    if constexpr (KD::DYN >= 0) {
#pragma unroll
      for (int i = 0; i < NY; ++i) {
        yv[i] = io.y(i);
        lamv[i] = io.lam(i);
      }
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) nuv[j] = io.nu(j);
  }
  long long* const prof_ = io.prof();
  long long tq_ = prof_ ? clock64() : 0;
  if (prof_) prof_[13] = tq_;
#define DTO_KKT_TICK(slot) do { if (prof_) { const long long n_ = clock64(); prof_[slot] += n_ - tq_; tq_ = n_; } } while (0)

#pragma unroll
  for (int i = 0; i < BD * (BD + 1) / 2; ++i) S[i] = 0.0;
#pragma unroll
  for (int i = 0; i < BD * (NY > 0 ? NY : 1); ++i) X[i] = 0.0;
#pragma unroll
  for (int i = 0; i < (NY > 0 ? NY * (NY + 1) / 2 : 1); ++i) YYl[i] = 0.0;
  // --- scatter the structural nonzeros into the dense blocks (literal indices: registers only)
  if constexpr (CO::NHL > 0) {
    double hl[CO::NHL];
    if constexpr (D::FUSED) {
      arr<CO::NW> wc;
      io.params(wc);
      arr<CO::SNH> hv;
      CO::shess(pv.data(), pv.data() + CO::NX, wc.data(), hv.data());
      CO::pack_hess_lower(hv.data(), hl);
    } else {
#pragma unroll
      for (int i = 0; i < CO::NHL; ++i) hl[i] = R(D::R_CH + i);
    }
    // (limited-memory mode: B_0 = sigma I replaces the objective Hessian too -- scale 0; otherwise 1: x * 1.0 is exact.
    //  Not for the element-BFGS plugins, which never run in that mode: with the multiply there, AMD clang 22 stops with
    //  "Illegal instruction detected ... V_CMP_NE_U32_e32 0, $src_private_base" on the SR1 acrobot plugin.)
    if constexpr (!D::QN && DTO_QN_HOT) {
#pragma unroll
      for (int i = 0; i < CO::NHL; ++i) hl[i] *= o.cost_hess_scale;
    }
    CO::scatter_hess_lower(hl, S);  // the pp block of S is packed exactly like W
  }
  if constexpr (D::QN) {
    // element BFGS block B over (p_t, x_{t+1}): pp part -> S, (x_{t+1}, p) part -> coupling rows, x_{t+1} part -> YY.
    // gam = 0 (inertia fallback) keeps only the objective Hessian scattered above.
    if (gam != 0.0) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) S[tri(i, j)] = R(D::R_B + tri(i, j));
      }
#pragma unroll
      for (int c = 0; c < NY; ++c) {
#pragma unroll
        for (int i = 0; i < NP; ++i) X[i * NY + c] = R(D::R_B + tri(NP + c, i));
#pragma unroll
        for (int e = 0; e <= c; ++e) YYl[tri(c, e)] = R(D::R_B + tri(NP + c, NP + e));
      }
    }
  }
  if constexpr (KD::DYN >= 0) {
    using DY = typename M::template Dyn<KD::DYN>;
    double jv[DY::NJ], F[NY * NP];
    arr<DY::NW> wd;
    arr<D::FUSED && (DY::NHL > 0) ? DY::NH : 0> hvd;  // Jacobian and Hessian values come out of one generated body
    if constexpr (D::FUSED) {
      io.params(wd);
      if constexpr (DY::NHL > 0) DY::jac_hess(pv.data(), pv.data() + DY::NX, yv.data(), wd.data(), lamv.data(), jv, hvd.data());
      else DY::jac(pv.data(), pv.data() + DY::NX, yv.data(), wd.data(), jv);
    } else {
#pragma unroll
      for (int i = 0; i < DY::NJ; ++i) jv[i] = R(D::R_DJ + i);
    }
#pragma unroll
    for (int i = 0; i < NY * NP; ++i) F[i] = 0.0;
    DY::scatter_jac(jv, F, X + (NP + Q) * NY);  // E rows of the coupling block
#pragma unroll
    for (int k = 0; k < NY; ++k) {
#pragma unroll
      for (int i = 0; i < NP; ++i) S[tri(NP + Q + k, i)] = F[k * NP + i];
    }
    if constexpr (DY::NHL > 0) {
      double hl[DY::NHL];
      if constexpr (D::FUSED) {
        DY::pack_hess_lower(hvd.data(), hl);
      } else {
#pragma unroll
        for (int i = 0; i < DY::NHL; ++i) hl[i] = R(D::R_DH + i);
      }
      DY::scatter_hess_lower(hl, gam, S, X, YYl);  // W_D -> pp block, V -> p rows of the coupling block
    }
  }
  if constexpr (KD::CON >= 0) {
    using CN = typename M::template Con<KD::CON>;
    double jv[CN::NJ > 0 ? CN::NJ : 1], G[Q * NP > 0 ? Q * NP : 1];
    arr<CN::NW> wk;
    if constexpr (D::FUSED) {
      io.params(wk);
      CN::jac(pv.data(), pv.data() + CN::NX, wk.data(), jv);
    } else {
#pragma unroll
      for (int i = 0; i < CN::NJ; ++i) jv[i] = R(D::R_KJ + i);
    }
#pragma unroll
    for (int i = 0; i < Q * NP; ++i) G[i] = 0.0;
    CN::scatter_jac(jv, G);
#pragma unroll
    for (int j = 0; j < Q; ++j) {
#pragma unroll
      for (int i = 0; i < NP; ++i) S[tri(NP + j, i)] = G[j * NP + i];
    }
    if constexpr (CN::NHL > 0) {
      double hl[CN::NHL];
      if constexpr (D::FUSED) {
        arr<CN::NH> hv;
        CN::hess(pv.data(), pv.data() + CN::NX, wk.data(), nuv.data(), hv.data());
        CN::pack_hess_lower(hv.data(), hl);
      } else {
#pragma unroll
        for (int i = 0; i < CN::NHL; ++i) hl[i] = R(D::R_KH + i);
      }
      CN::scatter_hess_lower(hl, gam, S);
    }
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) S[tri(i, j)] += cy.P[tri(i, j)];
  }
  bool fixed[NP > 0 ? NP : 1];
  StageBounds<NP> sb;
  if (!DTO_NEWTON(o)) io.bounds(sb);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    double rp = R(D::R_RP + i);
    double sig = dw;
    fixed[i] = false;
    if (DTO_NEWTON(o) && io.has_sigx()) sig += io.sigx(i);
    if (!DTO_NEWTON(o)) {
      const double lo = sb.lo[i], hi = sb.hi[i];
      if (lo == hi) {
        fixed[i] = true;
      } else {
        const double p = sb.p[i];
        if (finite_lo(lo)) {
          const double zl = sb.zl[i];
          sig += zl / (p - lo);
          rp -= mu / (p - lo);
        }
        if (finite_hi(hi)) {
          const double zu = sb.zu[i];
          sig += zu / (hi - p);
          rp += mu / (hi - p);
        }
      }
    }
    S[tri(i, i)] += sig;
    y[i] = -rp - (i < NX ? cy.py[i] : 0.0);
  }
  // --- stage-constraint rows
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    double dc = o.delta_c;
    double r = R(D::R_C + j);
    if (DTO_NEWTON(o) && io.has_sigc()) dc += io.sigc_con(j);
    if (!DTO_NEWTON(o) && D::ineq(j)) {
      const double sv = io.slack(j);
      const double zv = io.slack_mult(j);
      const double nu = io.nu(j);
      dc += sv / zv;
      r -= (sv / zv) * (nu - mu / sv);
    }
    S[tri(NP + j, NP + j)] = -dc;
    y[NP + j] = -r;
  }
  // --- dynamics rows
#pragma unroll
  for (int k = 0; k < NY; ++k) {
    double dc = o.delta_c;
    if (DTO_NEWTON(o) && io.has_sigc()) dc += io.sigc_dyn(k);
    S[tri(NP + Q + k, NP + Q + k)] = -dc;
    y[NP + Q + k] = -R(D::R_D + k);
  }
  // --- fixed variables: identity rows/columns
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    if (fixed[i]) {
#pragma unroll
      for (int r2 = 0; r2 < BD; ++r2) {
        if (r2 > i) S[tri(r2, i)] = 0.0;
        if (r2 < i) S[tri(i, r2)] = 0.0;
      }
      S[tri(i, i)] = 1.0;
      y[i] = 0.0;
#pragma unroll
      for (int c = 0; c < NY; ++c) X[i * NY + c] = 0.0;
    }
  }
  // --- spike: coupling of this stage to the chunk's left separator x_L
  if constexpr (SPK) {
#pragma unroll
    for (int i = 0; i < (NY > 0 ? NY : 1) * NX; ++i) cx_direct[i] = 0.0;
    if (first) {
      // the head stage's own x IS the separator: export its diagonal block / rhs, turn its
      // couplings into the spike, and pin it inside the chunk
#pragma unroll
      for (int i = 0; i < NX; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) sp.RLL[tri(i, j)] = S[tri(i, j)];
        sp.rL[i] = y[i];
      }
#pragma unroll
      for (int i = 0; i < BD; ++i) {
#pragma unroll
        for (int c = 0; c < NX; ++c) Z[i * NX + c] = (i < NX) ? 0.0 : S[tri(i, c)];
      }
#pragma unroll
      for (int aa = 0; aa < NY; ++aa) {
#pragma unroll
        for (int c = 0; c < NX; ++c) cx_direct[aa * NX + c] = X[c * NY + aa];  // V_x': x_L <-> x_{t+1}
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) {
#pragma unroll
        for (int r2 = 0; r2 < BD; ++r2) {
          if (r2 > i) S[tri(r2, i)] = 0.0;
          if (r2 < i) S[tri(i, r2)] = 0.0;
        }
        S[tri(i, i)] = 1.0;
        y[i] = 0.0;
#pragma unroll
        for (int c = 0; c < NY; ++c) X[i * NY + c] = 0.0;
      }
    } else {
#pragma unroll
      for (int i = 0; i < BD; ++i) {
#pragma unroll
        for (int c = 0; c < NX; ++c) Z[i * NX + c] = (i < NX && !fixed[i < NP ? i : 0]) ? sp.Cx[i * NX + c] : 0.0;
      }
    }
  }
  // values of the record the backward sweep still needs after the factorisation: residuals r_p, c, d
  if (keep) {
#pragma unroll
    for (int i = 0; i < NP; ++i) keep[i] = R(D::R_RP + i);
#pragma unroll
    for (int j = 0; j < Q; ++j) keep[NP + j] = R(D::R_C + j);
#pragma unroll
    for (int k = 0; k < NY; ++k) keep[NP + Q + k] = R(D::R_D + k);
  }
  DTO_KKT_TICK(1);
  if constexpr (ASSEMBLE_ONLY) return;
  if constexpr (BWD) {
#pragma unroll
    for (int i = 0; i < BD; ++i) {
      double r = y[i];
#pragma unroll
      for (int c = 0; c < NY; ++c) r -= X[i * NY + c] * xn[c];
      if constexpr (SPK) {
#pragma unroll
        for (int c = 0; c < NX; ++c) r -= Z[i * NX + c] * xL[c];
      }
      y[i] = r;
    }
  }
  // --- factor
  ldl_inplace<BD>(S, dinv, o.piv_tol, ok, nneg);
  DTO_KKT_TICK(4);
  // --- X = L^-1 O, w = L^-1 y, Z = L^-1 C
#pragma unroll
  for (int i = 1; i < BD; ++i) {
#pragma unroll
    for (int k = 0; k < i; ++k) {
      const double l = S[tri(i, k)];
      if constexpr (!BWD) {
#pragma unroll
        for (int c = 0; c < NY; ++c) X[i * NY + c] -= l * X[k * NY + c];
      }
      y[i] -= l * y[k];
      if constexpr (SPK && !BWD) {
#pragma unroll
        for (int c = 0; c < NX; ++c) Z[i * NX + c] -= l * Z[k * NX + c];
      }
    }
  }
  DTO_KKT_TICK(5);
  if (prof_) prof_[14] = tq_;
}

template <class M, int K, bool SPK, class IO>
__device__ __forceinline__ void stage_forward(const dto_solver_opts& o, const IO& io, double mu, double dw,
                                              double gam, bool first, bool need, Carry<M>& cy, Spike<M>& sp,
                                              bool& ok, int& nneg, bool keep_lost = false) {
  using D = KindDims<M, K>;
  constexpr int NP = D::NP, Q = D::Q, NY = D::NY, BD = D::BD, NX = D::NX;
  double S[BD * (BD + 1) / 2];
  double y[BD];
  double X[BD * (NY > 0 ? NY : 1)];
  double YYl[NY > 0 ? NY * (NY + 1) / 2 : 1];
  double Z[SPK ? BD * NX : 1];
  double cx_direct[SPK ? (NY > 0 ? NY : 1) * NX : 1];
  double dinv[BD];
  // cycle stamps (tools/kkt_profile.py; blockIdx.x == 1, thread 0): 0 = between two stages (loop, prefetch issue, copy of the
  // prefetched rows), 2 = carry stores, 1 / 4 / 5 = stage_factor (loads + derivative code + scatter / LDL / substitutions),
  // 6 = Schur complement onto x_{t+1}, 7 = stages stamped
  long long* const pf_ = io.prof();
  const long long te_ = pf_ ? clock64() : 0;
  if (pf_ && pf_[15]) pf_[0] += te_ - pf_[15];
  // --- carry-in of this stage is all the backward sweep needs besides the stage record (not of a lost attempt)
  // `keep_lost`: the last attempt of the ladder (or the single attempt of the linear-solver entry points) is swept through and
  // stored even with the wrong inertia -- it is the factorisation the backward sweep will use
  {
    const bool on = need && (SPK || ok || keep_lost);
    constexpr int NT_ = NX * (NX + 1) / 2, NCY = SPK ? D::FAC : D::F_CX;
    static_assert(D::F_P == 0 && D::F_PY == NT_ && D::F_CX == NT_ + NX, "carry record: P, py, Cx back to back");
    double cv[NCY > 0 ? NCY : 1];
#pragma unroll
    for (int i = 0; i < NT_; ++i) cv[i] = cy.P[i];
#pragma unroll
    for (int i = 0; i < NX; ++i) cv[NT_ + i] = cy.py[i];
    if constexpr (SPK) {
#pragma unroll
      for (int i = 0; i < NX * NX; ++i) cv[NT_ + NX + i] = sp.Cx[i];
    }
    auto elem = [&](int i) { return cv[i]; };
    if constexpr (IO::PAIR_CARRY) {
#pragma unroll
      for (int i = 0; i + 1 < NCY; i += 2) io.put_carry2(i, elem(i), elem(i + 1), on);
      if constexpr (NCY % 2 == 1) io.put_carry(NCY - 1, elem(NCY - 1), on);
    } else {
      if (on) {
#pragma unroll
        for (int i = 0; i < NCY; ++i) io.put_carry(i, elem(i));
      }
    }
  }
  const int nneg_in = nneg;
  stage_factor<M, K, SPK>(o, io, mu, dw, gam, first, cy, sp, S, y, X, YYl, Z, cx_direct, dinv, ok, nneg, nullptr);
  // static pivot order: the primal pivots of the block come first and must be positive, the Q + NY constraint pivots
  // negative; any other count means the inertia of the whole matrix is off (the sequential sweep gives up early on it)
  if constexpr (!SPK) {
    if (nneg - nneg_in != Q + NY) ok = false;
  }
  // --- carry to the next stage: P = YY - X' D^-1 X, py = X' D^-1 w   (D^-1 X and D^-1 Z are formed once: the sweep is
  //     bound by dependent f64 arithmetic, not by memory)
  double XD[BD * (NY > 0 ? NY : 1)];
#pragma unroll
  for (int i = 0; i < BD; ++i) {
#pragma unroll
    for (int c = 0; c < NY; ++c) XD[i * NY + c] = X[i * NY + c] * dinv[i];
  }
#pragma unroll
  for (int c = 0; c < NY; ++c) {
#pragma unroll
    for (int e = 0; e <= c; ++e) {
      double acc = YYl[tri(c, e)];
#pragma unroll
      for (int i = 0; i < BD; ++i) acc -= XD[i * NY + c] * X[i * NY + e];
      cy.P[tri(c, e)] = acc;
    }
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < BD; ++i) acc += XD[i * NY + c] * y[i];
    cy.py[c] = acc;
  }
  if (pf_) {
    const long long now_ = clock64();
    pf_[6] += now_ - pf_[14];
    pf_[2] += pf_[13] - te_;
    pf_[15] = now_;
    pf_[7] += 1;
  }
  if constexpr (SPK) {
    // separator contributions and the coupling handed to the next stage's x rows
    double ZD[BD * NX];
#pragma unroll
    for (int i = 0; i < BD; ++i) {
#pragma unroll
      for (int c = 0; c < NX; ++c) ZD[i * NX + c] = Z[i * NX + c] * dinv[i];
    }
#pragma unroll
    for (int c = 0; c < NX; ++c) {
#pragma unroll
      for (int e = 0; e <= c; ++e) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < BD; ++i) acc += ZD[i * NX + c] * Z[i * NX + e];
        sp.RLL[tri(c, e)] -= acc;
      }
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < BD; ++i) acc += ZD[i * NX + c] * y[i];
      sp.rL[c] -= acc;
    }
#pragma unroll
    for (int aa = 0; aa < NY; ++aa) {
#pragma unroll
      for (int c = 0; c < NX; ++c) {
        double acc = cx_direct[aa * NX + c];
#pragma unroll
        for (int i = 0; i < BD; ++i) acc -= XD[i * NY + aa] * Z[i * NX + c];
        sp.Cx[aa * NX + c] = acc;
      }
    }
  }
}

// Stage kinds whose dense blocks are much larger than the rest (e.g. the end stages that carry the pin constraints: 13 x 13
// instead of 9 x 9 for the acrobot) would dictate the register allocation of the whole sweep although they run once per
// horizon: they are called out of line on copies of the loop-carried state, so that the common stages keep theirs in
// registers and the two-wavefront-per-SIMD form of the sweep does not spill in its hot loop.
template <class M, int K>
constexpr bool heavy_kind() {
  using D = KindDims<M, K>;
  return D::BD * (D::BD + 1) / 2 + D::BD * D::NY > 100;
}
template <class M, int K, bool SPK>
__device__ __attribute__((noinline)) void stage_forward_cold(const dto_kkt_args& a, int64_t g, int t, double mu, double dw, double gam,
                                                            bool first, bool need, Carry<M>* cy, Spike<M>* sp, int* okneg) {
  bool ok = okneg[0] != 0;
  int nneg = okneg[1];
  const bool keep_lost = okneg[2] != 0;
  stage_forward<M, K, SPK>(a.opt, SoaIO<M, K>(a, g, t), mu, dw, gam, first, need, *cy, *sp, ok, nneg, keep_lost);
  okneg[0] = ok ? 1 : 0;
  okneg[1] = nneg;
}

// Outcome of one factorisation attempt of a lane: accept it, or put the next (delta_w, gamma) of the inertia-correction
// ladder into SC_TRY_* and leave SC_NEED set.  Shared by k_kkt_sep (time-partitioned sweeps, one launch per round) and the
// sequential sweep, which loops over its rounds inside one launch.
// SH: log2 of the stride between the scalar slots of one instance (6: column of an SoA tile; 0: the instance-major engine's
// contiguous block)
template <int SH>
__device__ __forceinline__ void retry_update_t(const dto_solver_opts& o, int Nc, double* sc, bool ok, int nneg) {
  // inertia of the whole KKT matrix must be (n_primal, n_dual, 0): exactly Nc negative pivots
  if (nneg != Nc) ok = false;
  sc[SC_NFACT << SH] += 1.0;
  sc[SC_NNEG << SH] = (double)nneg;
  double dw = sc[SC_TRY_DW << SH], gam = sc[SC_TRY_GAM << SH];
  const double dlast = sc[SC_DELTA_LAST << SH];
  const int attempt = (int)sc[SC_ATTEMPT << SH];
  const bool done = ok || o.newton_only || attempt >= o.max_refactor;
  if (done) {
    sc[SC_NEED << SH] = 0.0;
    sc[SC_DELTA_W << SH] = dw;
    sc[SC_GAMMA << SH] = gam;
    if (dw > 0.0 && gam != 0.0) sc[SC_DELTA_LAST << SH] = dw;  // last nonzero regularisation of the exact Hessian
    if (dw == 0.0) sc[SC_DELTA_LAST << SH] = 0.0;
    sc[SC_LS_FAIL << SH] = ok ? 0.0 : 1.0;  // not ok: regularisation cap reached, force growth next time
    sc[SC_QN_RESET << SH] = (gam == 0.0) ? 1.0 : 0.0;  // quasi-Newton: restart the element blocks after a fallback
    return;
  }
  if (gam != 0.0) {
    // Ipopt's Algorithm IC on the exact Hessian, but only up to a moderate delta_w: beyond it the
    // constraint curvature lam'd'' + nu'c'' (proportional to the multipliers, which a large
    // delta_w I only inflates further) is dropped instead -- Gauss-Newton convexification.
    const bool skip_ladder = (sc[SC_GAMMA << SH] == 0.0) && (((int)sc[SC_ITER << SH]) % 4 != 0);
    if (dw == 0.0 && !skip_ladder) dw = (dlast == 0.0) ? o.delta_w_init : fmax(o.delta_w_min, o.kappa_w_minus * dlast);
    else if (!skip_ladder) dw *= (dlast == 0.0) ? o.kappa_w_plus_first : o.kappa_w_plus;
    if (skip_ladder || dw > o.delta_w_exact_cap) {
      gam = 0.0;
      dw = o.delta_w_init;
    }
  } else {
    dw = (dw == 0.0) ? o.delta_w_init : dw * o.kappa_w_plus;   // (dw = 0 with gam = 0: the limited-memory mode's first probe)
    if (dw > o.delta_w_max) dw = o.delta_w_max;
  }
  sc[SC_TRY_DW << SH] = dw;
  sc[SC_TRY_GAM << SH] = gam;
  sc[SC_ATTEMPT << SH] = (double)(attempt + 1);
}

__device__ __forceinline__ void retry_update(const dto_kkt_args& a, double* sc, bool ok, int nneg) {
  retry_update_t<6>(a.opt, (int)a.Nc, sc, ok, nneg);
}

// CHUNKED = false: the plain sequential sweep (P = 1) without any spike code -- a separate instantiation because the
// spike blocks are what pushes the register count of the chunked form beyond one wavefront per SIMD
template <class M, bool CHUNKED>
__device__ __forceinline__ void kkt_fwd_body(const dto_kkt_args& a) {
  const int64_t g = blockIdx.x / a.P;
  const int p = blockIdx.x % a.P;
  double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  // The sequential form owns its tile for the whole inertia-correction loop: sweep, judge the inertia, pick the next
  // (delta_w, gamma) and sweep again until every lane of the tile has a factorisation -- no launch per round.  The
  // time-partitioned form does one round per launch (k_kkt_sep joins its chunks in between).
  for (int round = 0; CHUNKED || a.fwd_rounds <= 0 || round < a.fwd_rounds; ++round) {
    const bool need = sc[SC_STATUS << 6] == 0.0 && sc[SC_NEED << 6] != 0.0;
    if (!__any(need)) return;
    const double mu = sc[SC_MU << 6];
    // (limited-memory mode: the diagonal sigma I of B_0 rides on the primal regularisation)
    const double dw = sc[SC_TRY_DW << 6] + (DTO_QN_HOT && a.opt.qn_lbfgs ? sc[SC_QN_SIGMA << 6] : 0.0), gam = sc[SC_TRY_GAM << 6];
    const int t0 = uload(a.cstart, p), t1 = uload(a.cstart, p + 1);
    Carry<M> cy;
    Spike<M> sp;
#pragma unroll
    for (int i = 0; i < M::MAX_NX * (M::MAX_NX + 1) / 2; ++i) cy.P[i] = sp.RLL[i] = 0.0;
#pragma unroll
    for (int i = 0; i < M::MAX_NX; ++i) cy.py[i] = sp.rL[i] = 0.0;
#pragma unroll
    for (int i = 0; i < M::MAX_NX * M::MAX_NX; ++i) sp.Cx[i] = 0.0;
    bool ok = true;
    int nneg = 0;
    // the attempt after which retry_update stops whatever the inertia: its factorisation is the one that gets used
    const bool keep_lost = DTO_NEWTON(a.opt) || (int)sc[SC_ATTEMPT << 6] >= a.opt.max_refactor;
    if constexpr (!CHUNKED) {
      // walk the horizon run by run: kind dispatch and table look-ups once per run, arithmetic offsets inside
      const SoaBufs bufs(a, g);
      bool lost = false;
      for (int r = 0; r < a.n_runs && !lost; ++r) {
        const dto_stage_run run = load_run(a.runs, r);
        dispatch_uniform<M>(run.kind, [&](auto kc) {
          constexpr int K = decltype(kc)::value;
          if constexpr (heavy_kind<M, K>()) {
            for (int t = run.t0; t < run.t1; ++t) {
              Carry<M> cyc = cy;
              Spike<M> spc = sp;
              int okneg[3] = {ok ? 1 : 0, nneg, keep_lost ? 1 : 0};
              // the callee gets its own copy of the argument block: handing out the address of the kernel's would move every
              // pointer of the hot loop to the stack as well (reloads, generic instead of global addressing)
              const dto_kkt_args acold = a;
              stage_forward_cold<M, K, false>(acold, g, t, mu, dw, gam, false, need, &cyc, &spc, okneg);
              cy = cyc;
              ok = okneg[0] != 0;
              nneg = okneg[1];
              if (!__any(need && (ok || keep_lost))) { lost = true; break; }
            }
          } else {
            auto sweep = [&](auto bounded) {
              using IO = SoaPreIO<M, K, decltype(bounded)::value, false>;
              // (stages with long records -- quasi-Newton blocks, stored derivatives -- are not double-buffered: twice their
              // rows do not fit the register file next to the block algebra)
              constexpr bool PRE = DTO_SEQ_PREFETCH_FWD && sizeof(typename IO::In) <= DTO_SEQ_PREFETCH_MAX * sizeof(double);
              typename IO::In nxt, cur;
              if (PRE) nxt.load(bufs, run, run.t0);
              for (int t = run.t0; t < run.t1; ++t) {
                if (PRE) {
                  cur = nxt;
                  nxt.load(bufs, run, min(t + 1, run.t1 - 1));   // in flight while stage t is worked on
                } else {
                  cur.load(bufs, run, t);
                }
                stage_forward<M, K, false>(a.opt, IO(a, bufs, run, t, cur), mu, dw, gam, false, need, cy, sp, ok, nneg, keep_lost);
                // the attempt of a lane is lost with the first stage whose pivots have the wrong signs (stage_forward clears
                // ok): once that has happened to every lane that asked for a factorisation the rest of the sweep is pointless
                if (!__any(need && (ok || keep_lost))) { lost = true; break; }
              }
            };
            if (run.bounded) sweep(std::integral_constant<bool, true>{}); else sweep(std::integral_constant<bool, false>{});
          }
        });
      }
    } else {
#if DTO_CHUNK_PREFETCH
    // the chunk's stages run by run, as the sequential form walks the horizon: offsets by arithmetic, the rows of the next stage
    // requested before the arithmetic of this one (a chunk sweep is one dependent chain per lane like the sequential one; until
    // round 5 every stage began with its own round trip to memory: 5.5 us per stage for 3.5 of arithmetic)
    const SoaBufs bufs(a, g);
    for (int r = 0; r < a.n_runs; ++r) {
      const dto_stage_run run = load_run(a.runs, r);
      const int ra = run.t0 > t0 ? run.t0 : t0, rb = run.t1 < t1 ? run.t1 : t1;
      if (ra >= rb) continue;
      dispatch_uniform<M>(run.kind, [&](auto kc) {
        constexpr int K = decltype(kc)::value;
        auto sweep = [&](auto bounded, auto spk) {
          constexpr bool SPKv = decltype(spk)::value;
          // a kind without a previous dynamics is stage 0 and can only be in chunk 0: no spike instantiation
          if constexpr (!SPKv || M::template Kind<K>::PREV >= 0) {
            using IO = SoaPreIO<M, K, decltype(bounded)::value, false, SPKv>;
            constexpr bool PRE = sizeof(typename IO::In) <= DTO_SEQ_PREFETCH_MAX * sizeof(double);
            typename IO::In nxt, cur;
            if (PRE) nxt.load(bufs, run, ra);
            for (int t = ra; t < rb; ++t) {
              if (PRE) {
                cur = nxt;
                nxt.load(bufs, run, t + 1 < rb ? t + 1 : rb - 1);
              } else {
                cur.load(bufs, run, t);
              }
              stage_forward<M, K, SPKv>(a.opt, IO(a, bufs, run, t, cur), mu, dw, gam, SPKv && t == t0, need, cy, sp, ok, nneg,
                                        SPKv ? false : keep_lost);
            }
          }
        };
        using T_ = std::integral_constant<bool, true>;
        using F_ = std::integral_constant<bool, false>;
        if (p == 0) { if (run.bounded) sweep(T_{}, F_{}); else sweep(F_{}, F_{}); }
        else { if (run.bounded) sweep(T_{}, T_{}); else sweep(F_{}, T_{}); }
      });
    }
#else
    for (int t = t0; t < t1; ++t) {
      if (p == 0) {
        dispatch_uniform<M>(uload(a.kind, t), [&](auto kc) {
          stage_forward<M, decltype(kc)::value, false>(a.opt, SoaIO<M, decltype(kc)::value>(a, g, t), mu, dw, gam, false, need, cy, sp,
                                                       ok, nneg, keep_lost);
        });
      } else {
        dispatch_uniform<M>(uload(a.kind, t), [&](auto kc) {
          // a kind without a previous dynamics is stage 0 and can only be in chunk 0: no spike instantiation
          if constexpr (M::template Kind<decltype(kc)::value>::PREV >= 0)
            stage_forward<M, decltype(kc)::value, true>(a.opt, SoaIO<M, decltype(kc)::value>(a, g, t), mu, dw, gam, t == t0, need,
                                                        cy, sp, ok, nneg);
        });
      }
    }
#endif
    }
    if constexpr (!CHUNKED) {
      if (need) retry_update(a, sc, ok, nneg);
      if (a.opt.newton_only) return;  // one factorisation with the caller's delta_w
    } else {
      if (!need) return;
      using CS = ChunkSum<M>;
      double* cs = a.csum + (((g * a.P + p) * CS::SIZE) << 6) + threadIdx.x;
#pragma unroll
      for (int i = 0; i < CS::NT; ++i) {
        st_join(&cs[(int64_t)(CS::P + i) << 6], cy.P[i]);
        st_join(&cs[(int64_t)(CS::RLL + i) << 6], sp.RLL[i]);
      }
#pragma unroll
      for (int i = 0; i < CS::N; ++i) {
        st_join(&cs[(int64_t)(CS::PY + i) << 6], cy.py[i]);
        st_join(&cs[(int64_t)(CS::RL + i) << 6], sp.rL[i]);
      }
#pragma unroll
      for (int i = 0; i < CS::N * CS::N; ++i) st_join(&cs[(int64_t)(CS::CX + i) << 6], sp.Cx[i]);
      st_join(&cs[(int64_t)CS::OK << 6], ok ? 1.0 : 0.0);
      st_join(&cs[(int64_t)CS::NNEG << 6], (double)nneg);
      return;
    }
  }
}

template <class M>
__global__ __launch_bounds__(WAVE) void k_kkt_fwd(dto_kkt_args a) { kkt_fwd_body<M, true>(a); }
// the plain sequential sweep: ONE wavefront per SIMD (512 registers, nothing spilled inside the stage loops; rounds 1-2 ran it
// at two with spills -- DESIGN.md section 4.2 for what that cost), the next stage's rows in flight during a stage
template <class M>
__global__ __launch_bounds__(WAVE, DTO_SEQ_FWD_OCC) void k_kkt_fwd_seq(dto_kkt_args a) {
  if (a.fwd_started && threadIdx.x == 0) atomicAdd(a.fwd_started, 1);
  kkt_fwd_body<M, false>(a);
  if (a.tile_fwd_tag) {
    // this tile's factorisation is complete: tell the early back substitutions (k_kkt_bwd_early, other stream)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (threadIdx.x == 0) __hip_atomic_store(a.tile_fwd_tag + blockIdx.x, a.sweep_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ------------------------------------------------------------------------------------------------
// The separator system of ONE instance by block cyclic reduction, lanes = separators (a batch of one / a few instances: the
// lane-per-instance elimination below walks the P - 1 separators one after the other with one active lane -- 4 us each, 250 us
// of a 830 us iteration at T = 1000, profiles/r05/single_instance_kernel_trace_*.txt).  The reduced matrix is symmetric block
// tridiagonal (diagonal blocks A_j, sub-diagonal blocks Lo_j = K[s_j, s_{j-1}]); at level h = 1, 2, 4, ... the nodes with
// j mod 2h = h - 1 are eliminated -- A_e = L D L', F = A_e^-1 Lo_e, G = A_e^-1 Lo_{e+h}', y = A_e^-1 r_e -- and their
// neighbours j mod 2h = 2h - 1 take the Schur complements; log2 levels down, log2 levels of back substitution up, neighbour
// blocks through LDS.  It is a block LDL' under a symmetric permutation: the count of negative pivots is the inertia all the
// same (Sylvester), and for the positive definite matrix this must be for the step to be accepted every pivot is positive.
// ------------------------------------------------------------------------------------------------
#ifndef DTO_SEP_CR_MAX_INST
#define DTO_SEP_CR_MAX_INST 4
#endif
__device__ __forceinline__ void sep_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
template <class M>
__device__ __forceinline__ void kkt_sep_cr(const dto_kkt_args& a, const int64_t g, const int li, bool& ok_out, int& nneg_out) {
  using CS = ChunkSum<M>;
  constexpr int N = CS::N, NT = CS::NT, NN = N * N;
  __shared__ double sh_lo[64 * NN], sh_f[64 * NN], sh_g[64 * NN], sh_y[64 * N], sh_x[64 * N];
  const dto_solver_opts& o = a.opt;
  const int j = threadIdx.x;   // separator j = x at the head of chunk j + 1
  const int ns = a.P - 1;
  const bool act = j < ns;
  auto csp = [&](int p) { return a.csum + (((g * a.P + p) * CS::SIZE) << 6) + li; };
  bool ok = true;
  int nneg = 0;
  if (j < a.P) {               // the chunks' own verdicts: lane j reads chunk j
    const double* cs = csp(j);
    if (ld_join(&cs[(int64_t)CS::OK << 6]) == 0.0) ok = false;
    nneg = (int)ld_join(&cs[(int64_t)CS::NNEG << 6]);
  }
  double A[NT], Lo[NN], r[N], F[NN], G[NN], y[N], dinv[N];
#pragma unroll
  for (int i = 0; i < NT; ++i) A[i] = 0.0;
#pragma unroll
  for (int i = 0; i < NN; ++i) Lo[i] = F[i] = G[i] = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) { A[tri(i, i)] = 1.0; r[i] = y[i] = 0.0; dinv[i] = 1.0; }
  if (act) {
    const int p = j + 1;
    const double* cl = csp(p - 1);
    const double* cr = csp(p);
#pragma unroll
    for (int i = 0; i < NT; ++i) A[i] = ld_join(&cl[(int64_t)(CS::P + i) << 6]) + ld_join(&cr[(int64_t)(CS::RLL + i) << 6]);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = ld_join(&cr[(int64_t)(CS::RL + i) << 6]) - ld_join(&cl[(int64_t)(CS::PY + i) << 6]);
    const int z0 = a.zoff[a.cstart[p]];
    bool fx[N];
#pragma unroll
    for (int i = 0; i < N; ++i) fx[i] = !o.newton_only && (a.lo[z0 + i] == a.hi[z0 + i]);
    if (p > 1) {
      const int zp = a.zoff[a.cstart[p - 1]];
#pragma unroll
      for (int c = 0; c < N; ++c) {
        const bool fc = !o.newton_only && (a.lo[zp + c] == a.hi[zp + c]);
#pragma unroll
        for (int aa = 0; aa < N; ++aa) Lo[aa * N + c] = (fc || fx[aa]) ? 0.0 : ld_join(&cl[(int64_t)(CS::CX + aa * N + c) << 6]);
      }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (fx[i]) {
#pragma unroll
        for (int r2 = 0; r2 < N; ++r2) {
          if (r2 > i) A[tri(r2, i)] = 0.0;
          if (r2 < i) A[tri(i, r2)] = 0.0;
        }
        A[tri(i, i)] = 1.0;
        r[i] = 0.0;
      }
    }
  }
  // ---- reduction
  for (int h = 1; h <= ns; h <<= 1) {
    const int m = j & (2 * h - 1);
    const bool elim = act && m == h - 1, surv = act && m == 2 * h - 1;
    if (elim || surv) {
#pragma unroll
      for (int i = 0; i < NN; ++i) sh_lo[j * NN + i] = Lo[i];
    }
    sep_lds_fence();
    if (elim) {
      const bool right = j + h < ns;
#pragma unroll
      for (int aa = 0; aa < N; ++aa) {
#pragma unroll
        for (int c = 0; c < N; ++c) {
          F[aa * N + c] = Lo[aa * N + c];
          G[aa * N + c] = right ? sh_lo[(j + h) * NN + c * N + aa] : 0.0;   // K[s_e, s_{e+h}] = Lo_{e+h}'
        }
        y[aa] = r[aa];
      }
      ldl_inplace<N>(A, dinv, o.piv_tol, ok, nneg);
#pragma unroll
      for (int i = 1; i < N; ++i) {
#pragma unroll
        for (int k = 0; k < i; ++k) {
          const double l = A[tri(i, k)];
#pragma unroll
          for (int c = 0; c < N; ++c) {
            F[i * N + c] -= l * F[k * N + c];
            G[i * N + c] -= l * G[k * N + c];
          }
          y[i] -= l * y[k];
        }
      }
#pragma unroll
      for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int c = 0; c < N; ++c) {
          F[i * N + c] *= dinv[i];
          G[i * N + c] *= dinv[i];
        }
        y[i] *= dinv[i];
      }
#pragma unroll
      for (int i = N - 1; i >= 1; --i) {
#pragma unroll
        for (int k = 0; k < i; ++k) {
          const double l = A[tri(i, k)];
#pragma unroll
          for (int c = 0; c < N; ++c) {
            F[k * N + c] -= l * F[i * N + c];
            G[k * N + c] -= l * G[i * N + c];
          }
          y[k] -= l * y[i];
        }
      }
#pragma unroll
      for (int i = 0; i < NN; ++i) {
        sh_f[j * NN + i] = F[i];
        sh_g[j * NN + i] = G[i];
      }
#pragma unroll
      for (int i = 0; i < N; ++i) sh_y[j * N + i] = y[i];
    }
    sep_lds_fence();
    if (surv) {
      const int e = j - h;   // eliminated neighbour on the left: always there
      double nl[NN];
#pragma unroll
      for (int aa = 0; aa < N; ++aa) {
#pragma unroll
        for (int bb = 0; bb < N; ++bb) {
          double accg = 0.0, accf = 0.0;
#pragma unroll
          for (int c = 0; c < N; ++c) {
            accg += Lo[aa * N + c] * sh_g[e * NN + c * N + bb];
            accf += Lo[aa * N + c] * sh_f[e * NN + c * N + bb];
          }
          if (bb <= aa) A[tri(aa, bb)] -= accg;
          nl[aa * N + bb] = -accf;
        }
        double accy = 0.0;
#pragma unroll
        for (int c = 0; c < N; ++c) accy += Lo[aa * N + c] * sh_y[e * N + c];
        r[aa] -= accy;
      }
      if (j + h < ns) {      // ... and on the right
        const int e2 = j + h;
#pragma unroll
        for (int aa = 0; aa < N; ++aa) {
#pragma unroll
          for (int bb = 0; bb <= aa; ++bb) {
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < N; ++c) acc += sh_lo[e2 * NN + c * N + aa] * sh_f[e2 * NN + c * N + bb];
            A[tri(aa, bb)] -= acc;
          }
          double accy = 0.0;
#pragma unroll
          for (int c = 0; c < N; ++c) accy += sh_lo[e2 * NN + c * N + aa] * sh_y[e2 * N + c];
          r[aa] -= accy;
        }
      }
#pragma unroll
      for (int i = 0; i < NN; ++i) Lo[i] = nl[i];
    }
    sep_lds_fence();
  }
  // ---- back substitution, top level first
  int htop = 1;
  while (2 * htop <= ns) htop <<= 1;
  for (int h = htop; h >= 1; h >>= 1) {
    const bool elim = act && (j & (2 * h - 1)) == h - 1;
    if (elim) {
      const bool left = j - h >= 0, right = j + h < ns;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        double v = y[i];
#pragma unroll
        for (int c = 0; c < N; ++c) {
          if (left) v -= F[i * N + c] * sh_x[(j - h) * N + c];
          if (right) v -= G[i * N + c] * sh_x[(j + h) * N + c];
        }
        r[i] = v;
      }
#pragma unroll
      for (int i = 0; i < N; ++i) sh_x[j * N + i] = r[i];
      double* xs = a.xsep + (((g * a.P + (j + 1)) * N) << 6) + li;
#pragma unroll
      for (int i = 0; i < N; ++i) xs[(int64_t)i << 6] = r[i];
    }
    sep_lds_fence();
  }
  // ---- verdict of the whole factorisation: every chunk and every separator block
  ok_out = __ballot(!ok) == 0ull;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) nneg += __shfl_xor(nneg, d);
  nneg_out = nneg;
}

// reduced system over the separators + inertia + retry state machine.  grid = G waves.
template <class M>
__device__ __forceinline__ void kkt_sep_body(const dto_kkt_args& a, const int64_t g) {
  const dto_solver_opts& o = a.opt;
  double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  const bool need = sc[SC_STATUS << 6] == 0.0 && sc[SC_NEED << 6] != 0.0;
  if (!__any(need)) return;
  using CS = ChunkSum<M>;
  using SF = SepFac<M>;
  constexpr int N = CS::N, NT = CS::NT;
  if constexpr (N <= 6) {     // (registers and LDS of kkt_sep_cr grow with N^2)
    const unsigned long long nm = __ballot(need);
    if (a.sep_cr == 1 && __popcll(nm) <= DTO_SEP_CR_MAX_INST) {
      for (unsigned long long rest = nm; rest; rest &= rest - 1) {
        const int li = __ffsll((long long)rest) - 1;
        bool okc;
        int nnegc;
        kkt_sep_cr<M>(a, g, li, okc, nnegc);
        if ((int)threadIdx.x == li) retry_update(a, sc, okc, nnegc);
      }
      return;
    }
  }
  auto csp = [&](int p) { return a.csum + (((g * a.P + p) * CS::SIZE) << 6) + threadIdx.x; };
  auto sfp = [&](int p) { return a.sfac + (((g * a.P + p) * SF::SIZE) << 6) + threadIdx.x; };
  bool ok = true;
  int nneg = 0;
  for (int p = 0; p < a.P; ++p) {
    const double* cs = csp(p);
    if (ld_join(&cs[(int64_t)CS::OK << 6]) == 0.0) ok = false;
    nneg += (int)ld_join(&cs[(int64_t)CS::NNEG << 6]);
  }
  // forward elimination over separators p = 1 .. P-1
  double Lp[N * (N + 1) / 2], dip[N], wp[N];  // factor of the previous separator block
  for (int p = 1; p < a.P; ++p) {
    const double* cl = csp(p - 1);
    const double* cr = csp(p);
    double Dg[NT], r[N], Bt[N * N];  // Bt[c][aa] = K[s_p(aa), s_{p-1}(c)] = CX of chunk p-1, transposed access below
#pragma unroll
    for (int i = 0; i < NT; ++i) Dg[i] = ld_join(&cl[(int64_t)(CS::P + i) << 6]) + ld_join(&cr[(int64_t)(CS::RLL + i) << 6]);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = ld_join(&cr[(int64_t)(CS::RL + i) << 6]) - ld_join(&cl[(int64_t)(CS::PY + i) << 6]);
    const int z0 = uload(a.zoff, uload(a.cstart, p));
    bool fx[N];
#pragma unroll
    for (int i = 0; i < N; ++i) fx[i] = !o.newton_only && (uload(a.lo, z0 + i) == uload(a.hi, z0 + i));
    if (p > 1) {
      // coupling with the previous separator: rows = this separator (x_R of chunk p-1), cols = previous (x_L)
      const double* cx = csp(p - 1) + ((int64_t)CS::CX << 6);
      double Mm[N * N];  // Mm = L_{p-1}^-1 B', B'[c][aa] = cx[aa][c]
      const int zp = uload(a.zoff, uload(a.cstart, p - 1));
#pragma unroll
      for (int c = 0; c < N; ++c) {
        const bool fc = !o.newton_only && (uload(a.lo, zp + c) == uload(a.hi, zp + c));
#pragma unroll
        for (int aa = 0; aa < N; ++aa) Mm[c * N + aa] = (fc || fx[aa]) ? 0.0 : ld_join(&cx[(int64_t)(aa * N + c) << 6]);
      }
#pragma unroll
      for (int i = 1; i < N; ++i) {
#pragma unroll
        for (int k = 0; k < i; ++k) {
          const double l = Lp[tri(i, k)];
#pragma unroll
          for (int aa = 0; aa < N; ++aa) Mm[i * N + aa] -= l * Mm[k * N + aa];
        }
      }
#pragma unroll
      for (int aa = 0; aa < N; ++aa) {
#pragma unroll
        for (int bb = 0; bb <= aa; ++bb) {
          double acc = 0.0;
#pragma unroll
          for (int i = 0; i < N; ++i) acc += Mm[i * N + aa] * Mm[i * N + bb] * dip[i];
          Dg[tri(aa, bb)] -= acc;
        }
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) acc += Mm[i * N + aa] * dip[i] * wp[i];
        r[aa] -= acc;
      }
      if (need) {
        double* sf = sfp(p - 1);
#pragma unroll
        for (int i = 0; i < N * N; ++i) sf[(int64_t)(SF::MM + i) << 6] = Mm[i];
      }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (fx[i]) {
#pragma unroll
        for (int r2 = 0; r2 < N; ++r2) {
          if (r2 > i) Dg[tri(r2, i)] = 0.0;
          if (r2 < i) Dg[tri(i, r2)] = 0.0;
        }
        Dg[tri(i, i)] = 1.0;
        r[i] = 0.0;
      }
    }
    ldl_inplace<N>(Dg, dip, o.piv_tol, ok, nneg);
#pragma unroll
    for (int i = 1; i < N; ++i) {
#pragma unroll
      for (int k = 0; k < i; ++k) r[i] -= Dg[tri(i, k)] * r[k];
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) Lp[i] = Dg[i];
#pragma unroll
    for (int i = 0; i < N; ++i) wp[i] = r[i];
    if (need) {
      double* sf = sfp(p);
#pragma unroll
      for (int i = 1; i < N; ++i) {
#pragma unroll
        for (int k = 0; k < i; ++k) sf[(int64_t)(SF::L + i * (i - 1) / 2 + k) << 6] = Dg[tri(i, k)];
      }
#pragma unroll
      for (int i = 0; i < N; ++i) {
        sf[(int64_t)(SF::DI + i) << 6] = dip[i];
        sf[(int64_t)(SF::W + i) << 6] = r[i];
      }
    }
  }
  // backward substitution over the separators
  double sn[N];
#pragma unroll
  for (int i = 0; i < N; ++i) sn[i] = 0.0;
  for (int p = a.P - 1; p >= 1; --p) {
    const double* sf = sfp(p);
    double v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double rr = sf[(int64_t)(SF::W + i) << 6];
      if (p < a.P - 1) {
#pragma unroll
        for (int aa = 0; aa < N; ++aa) rr -= sf[(int64_t)(SF::MM + i * N + aa) << 6] * sn[aa];
      }
      v[i] = rr * sf[(int64_t)(SF::DI + i) << 6];
    }
#pragma unroll
    for (int i = N - 1; i >= 1; --i) {
#pragma unroll
      for (int k = 0; k < i; ++k) v[k] -= sf[(int64_t)(SF::L + i * (i - 1) / 2 + k) << 6] * v[i];
    }
    if (need) {
      double* xs = a.xsep + (((g * a.P + p) * N) << 6) + threadIdx.x;
#pragma unroll
      for (int i = 0; i < N; ++i) xs[(int64_t)i << 6] = v[i];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) sn[i] = v[i];
  }
  if (need) retry_update(a, sc, ok, nneg);
}
template <class M>
__global__ __launch_bounds__(WAVE) void k_kkt_sep(dto_kkt_args a) { kkt_sep_body<M>(a, blockIdx.x); }
// The cyclic reduction for batches of MORE than DTO_SEP_CR_MAX_INST instances with many chunks (dto_kkt_args.sep_cr == 2: at least
// 16 chunks, decided by the batch): one wavefront per (tile, lane) instead of one per tile -- 64 instances x 63 separators take
// ~20 us side by side where the lane-per-instance elimination walks the separators one after the other (4 us each).  grid = G * 64.
template <class M>
__global__ __launch_bounds__(WAVE) void k_kkt_sep_cr(dto_kkt_args a) {
  using CS = ChunkSum<M>;
  if constexpr (CS::N <= 6) {
    const int64_t g = blockIdx.x >> 6;
    const int li = (int)(blockIdx.x & 63);
    double* scl = a.scal + ((g * SC_COUNT) << 6) + li;   // the instance's column of the scalar block
    if (!(scl[SC_STATUS << 6] == 0.0 && scl[SC_NEED << 6] != 0.0)) return;
    bool okc;
    int nnegc;
    kkt_sep_cr<M>(a, g, li, okc, nnegc);
    if (threadIdx.x == 0) retry_update(a, scl, okc, nnegc);
  }
}

struct StepAcc {
  double apmax, admax, gphid, rlam;
};

template <class M, int K, bool SPK, class IO>
__device__ __forceinline__ void stage_backward(const dto_solver_opts& o, const IO& io, double mu, double tau,
                                               double dw, double gam, bool first, const double* xL, double* xn,
                                               StepAcc& acc) {
  using D = KindDims<M, K>;
  constexpr int NP = D::NP, Q = D::Q, NY = D::NY, BD = D::BD, NX = D::NX;
  // residuals of the record, copied to registers by stage_factor before the LDS buffer is handed to the next stage
  double keep[BD];
  auto R = [&](int e) { return e >= D::R_C ? keep[NP + (e - D::R_C)] : (e >= D::R_D ? keep[NP + Q + (e - D::R_D)] : keep[e - D::R_RP]); };
  // --- rebuild this stage's factorisation from its record and the stored carry-in
  Carry<M> cy;
  Spike<M> sp;
  {
    constexpr int NT_ = NX * (NX + 1) / 2, NCY = SPK ? D::FAC : D::F_CX;
    double cv[NCY > 0 ? NCY + 1 : 1];
    if constexpr (IO::PAIR_CARRY) {
#pragma unroll
      for (int i = 0; i + 1 < NCY; i += 2) io.carry2(i, cv[i], cv[i + 1]);
      if constexpr (NCY % 2 == 1) cv[NCY - 1] = io.carry(NCY - 1);
    } else {
#pragma unroll
      for (int i = 0; i < NCY; ++i) cv[i] = io.carry(i);
    }
#pragma unroll
    for (int i = 0; i < NT_; ++i) cy.P[i] = cv[i];
#pragma unroll
    for (int i = 0; i < NX; ++i) cy.py[i] = cv[NT_ + i];
    if constexpr (SPK) {
#pragma unroll
      for (int i = 0; i < NX * NX; ++i) sp.Cx[i] = cv[NT_ + NX + i];
    }
  }
  double S[BD * (BD + 1) / 2];
  double w[BD];
  double X[BD * (NY > 0 ? NY : 1)];
  double YYl[NY > 0 ? NY * (NY + 1) / 2 : 1];
  double Z[SPK ? BD * NX : 1];
  double cx_direct[SPK ? (NY > 0 ? NY : 1) * NX : 1];
  double dinv[BD];
  bool ok_unused = true;
  int nneg_unused = 0;
  stage_factor<M, K, SPK, true>(o, io, mu, dw, gam, first, cy, sp, S, w, X, YYl, Z, cx_direct, dinv, ok_unused, nneg_unused, keep,
                                xn, xL);
  double v[BD];
#pragma unroll
  for (int i = 0; i < BD; ++i) v[i] = w[i] * dinv[i];
#pragma unroll
  for (int i = BD - 1; i >= 1; --i) {
#pragma unroll
    for (int k = 0; k < i; ++k) v[k] -= S[tri(i, k)] * v[i];
  }
  if constexpr (SPK) {
    if (first) {
#pragma unroll
      for (int i = 0; i < NX; ++i) v[i] = xL[i];  // the head stage's x is the separator itself
    }
  }
  // primal step, fraction to the boundary, barrier directional derivative
  StageBounds<NP> sb;
  if (!DTO_NEWTON(o)) io.bounds(sb);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const double dp = v[i];
    io.put_dp(i, dp);
    acc.gphid += R(D::R_RP + i) * dp;
    if (!DTO_NEWTON(o)) {
      const double lo = sb.lo[i], hi = sb.hi[i];
      if (lo != hi) {
        const double p = sb.p[i];
        if (finite_lo(lo)) {
          const double zl = sb.zl[i];
          const double gap = p - lo;
          const double dzl = mu / gap - zl - (zl / gap) * dp;
          if (dp < 0.0) acc.apmax = fmin(acc.apmax, -tau * gap / dp);
          if (dzl < 0.0) acc.admax = fmin(acc.admax, -tau * zl / dzl);
          acc.gphid -= mu / gap * dp;
        }
        if (finite_hi(hi)) {
          const double zu = sb.zu[i];
          const double gap = hi - p;
          const double dzu = mu / gap - zu + (zu / gap) * dp;
          if (dp > 0.0) acc.apmax = fmin(acc.apmax, tau * gap / dp);
          if (dzu < 0.0) acc.admax = fmin(acc.admax, -tau * zu / dzu);
          acc.gphid += mu / gap * dp;
        }
      }
    }
  }
  // constraint multipliers; grad f' dp = rp' dp + sum lam_j (r_j - dc dlam_j [+ ds_j])
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    const double dnu = v[NP + j];
    const double nu = io.nu(j);
    const double r = R(D::R_C + j);
    io.put_dnu(j, dnu);
    double dsv = 0.0;
    if (!DTO_NEWTON(o) && D::ineq(j)) {
      const double sv = io.slack(j);
      const double zv = io.slack_mult(j);
      dsv = -(sv / zv) * (nu - mu / sv + dnu);
      const double dzs = mu / sv - zv - (zv / sv) * dsv;
      io.put_ds(j, dsv);
      if (dsv < 0.0) acc.apmax = fmin(acc.apmax, -tau * sv / dsv);
      if (dzs < 0.0) acc.admax = fmin(acc.admax, -tau * zv / dzs);
      acc.gphid -= mu / sv * dsv;
    }
    acc.gphid += nu * (r - o.delta_c * dnu + dsv);
    acc.rlam += r * (nu + dnu);
  }
#pragma unroll
  for (int k = 0; k < NY; ++k) {
    const double dl = v[NP + Q + k];
    const double lam = io.lam(k);
    const double r = R(D::R_D + k);
    io.put_dlam(k, dl);
    acc.gphid += lam * (r - o.delta_c * dl);
    acc.rlam += r * (lam + dl);
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) xn[i] = v[i];
}

// out-of-line form for the heavy stage kinds (see stage_forward_cold)
template <class M, int K>
__device__ __attribute__((noinline)) void stage_backward_cold(const dto_kkt_args& a, int64_t g, int t, double mu, double tau, double dw,
                                                             double gam, const double* xL, double* xn, StepAcc* acc) {
  StepAcc ac = *acc;
  double xl[M::MAX_NX], xv[M::MAX_NX];
#pragma unroll
  for (int i = 0; i < M::MAX_NX; ++i) {
    xl[i] = xL[i];
    xv[i] = xn[i];
  }
  stage_backward<M, K, false>(a.opt, SoaIO<M, K>(a, g, t), mu, tau, dw, gam, false, xl, xv, ac);
#pragma unroll
  for (int i = 0; i < M::MAX_NX; ++i) xn[i] = xv[i];
  *acc = ac;
}

template <class M, bool CHUNKED>
__device__ __forceinline__ void kkt_bwd_body(const dto_kkt_args& a) {
  const int64_t g = blockIdx.x / a.P;
  const int p = blockIdx.x % a.P;
  const dto_solver_opts& o = a.opt;
  const double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  const bool running = sc[SC_STATUS << 6] == 0.0;
  if (__all(!running)) return;
  constexpr int N = M::MAX_NX;
  const double mu = sc[SC_MU << 6];
  const double tau = fmax(o.tau_min, 1.0 - mu);
  const double dw = sc[SC_DELTA_W << 6] + (DTO_QN_HOT && a.opt.qn_lbfgs ? sc[SC_QN_SIGMA << 6] : 0.0), gam = sc[SC_GAMMA << 6];  // the accepted factorisation
  const int t0 = uload(a.cstart, p), t1 = uload(a.cstart, p + 1);
  double xL[N], xn[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    xL[i] = (p > 0) ? a.xsep[(((g * a.P + p) * N + i) << 6) + threadIdx.x] : 0.0;
    xn[i] = (p < a.P - 1) ? a.xsep[(((g * a.P + p + 1) * N + i) << 6) + threadIdx.x] : 0.0;
  }
  StepAcc acc{1.0, 1.0, 0.0, 0.0};
  if constexpr (!CHUNKED) {
    const SoaBufs bufs(a, g);
    for (int r = a.n_runs - 1; r >= 0; --r) {
      const dto_stage_run run = load_run(a.runs, r);
      dispatch_uniform<M>(run.kind, [&](auto kc) {
        constexpr int K = decltype(kc)::value;
        if constexpr (heavy_kind<M, K>()) {
          for (int t = run.t1 - 1; t >= run.t0; --t) {
            const dto_kkt_args acold = a;  // own copy: see kkt_fwd_body
            double xnc[N];
#pragma unroll
            for (int i = 0; i < N; ++i) xnc[i] = xn[i];
            StepAcc accc = acc;
            stage_backward_cold<M, K>(acold, g, t, mu, tau, dw, gam, xL, xnc, &accc);
#pragma unroll
            for (int i = 0; i < N; ++i) xn[i] = xnc[i];
            acc = accc;
          }
        } else {
          auto sweep = [&](auto bounded) {
            using IO = SoaPreIO<M, K, decltype(bounded)::value, true>;
            constexpr bool PRE = DTO_SEQ_PREFETCH_BWD && sizeof(typename IO::In) <= DTO_SEQ_PREFETCH_MAX * sizeof(double);
            typename IO::In nxt, cur;
            if (PRE) nxt.load(bufs, run, run.t1 - 1);
            for (int t = run.t1 - 1; t >= run.t0; --t) {
              if (PRE) {
                cur = nxt;
                nxt.load(bufs, run, max(t - 1, run.t0));   // in flight while stage t is worked on
              } else {
                cur.load(bufs, run, t);
              }
              stage_backward<M, K, false>(a.opt, IO(a, bufs, run, t, cur), mu, tau, dw, gam, false, xL, xn, acc);
            }
          };
          if (run.bounded) sweep(std::integral_constant<bool, true>{}); else sweep(std::integral_constant<bool, false>{});
        }
      });
    }
  } else {
#if DTO_CHUNK_PREFETCH
    const SoaBufs bufs(a, g);
    for (int r = a.n_runs - 1; r >= 0; --r) {
      const dto_stage_run run = load_run(a.runs, r);
      const int ra = run.t0 > t0 ? run.t0 : t0, rb = run.t1 < t1 ? run.t1 : t1;
      if (ra >= rb) continue;
      dispatch_uniform<M>(run.kind, [&](auto kc) {
        constexpr int K = decltype(kc)::value;
        auto sweep = [&](auto bounded, auto spk) {
          constexpr bool SPKv = decltype(spk)::value;
          if constexpr (!SPKv || M::template Kind<K>::PREV >= 0) {
            using IO = SoaPreIO<M, K, decltype(bounded)::value, true, SPKv>;
            constexpr bool PRE = sizeof(typename IO::In) <= DTO_SEQ_PREFETCH_MAX * sizeof(double);
            typename IO::In nxt, cur;
            if (PRE) nxt.load(bufs, run, rb - 1);
            for (int t = rb - 1; t >= ra; --t) {
              if (PRE) {
                cur = nxt;
                nxt.load(bufs, run, t - 1 > ra ? t - 1 : ra);
              } else {
                cur.load(bufs, run, t);
              }
              stage_backward<M, K, SPKv>(a.opt, IO(a, bufs, run, t, cur), mu, tau, dw, gam, SPKv && t == t0, xL, xn, acc);
            }
          }
        };
        using T_ = std::integral_constant<bool, true>;
        using F_ = std::integral_constant<bool, false>;
        if (p == 0) { if (run.bounded) sweep(T_{}, F_{}); else sweep(F_{}, F_{}); }
        else { if (run.bounded) sweep(T_{}, T_{}); else sweep(F_{}, T_{}); }
      });
    }
#else
    for (int t = t1 - 1; t >= t0; --t) {
      if (p == 0) {
        dispatch_uniform<M>(uload(a.kind, t), [&](auto kc) {
          stage_backward<M, decltype(kc)::value, false>(a.opt, SoaIO<M, decltype(kc)::value>(a, g, t), mu, tau, dw, gam, false, xL, xn,
                                                        acc);
        });
      } else {
        dispatch_uniform<M>(uload(a.kind, t), [&](auto kc) {
          if constexpr (M::template Kind<decltype(kc)::value>::PREV >= 0)
            stage_backward<M, decltype(kc)::value, true>(a.opt, SoaIO<M, decltype(kc)::value>(a, g, t), mu, tau, dw, gam, t == t0, xL,
                                                         xn, acc);
        });
      }
    }
#endif
  }
  double* ca = a.cacc + (((g * a.P + p) * 4) << 6) + threadIdx.x;
  st_join(&ca[0 << 6], acc.apmax);
  st_join(&ca[1 << 6], acc.admax);
  st_join(&ca[2 << 6], acc.gphid);
  st_join(&ca[3 << 6], acc.rlam);
}

template <class M>
__global__ __launch_bounds__(WAVE) void k_kkt_bwd(dto_kkt_args a) { kkt_bwd_body<M, true>(a); }
template <class M>
__global__ __launch_bounds__(WAVE, DTO_SEQ_BWD_OCC) void k_kkt_bwd_seq(dto_kkt_args a) { kkt_bwd_body<M, false>(a); }
#ifndef DTO_FUSE_SWEEPS
#define DTO_FUSE_SWEEPS 0
#endif
#if DTO_FUSE_SWEEPS
// EXPERIMENT (tools/micro/fused_sweeps_check.py, -DDTO_FUSE_SWEEPS=1): both sweeps of a tile in one wavefront -- the form that gave
// deterministic wrong results in round 3 when both bodies were inline and the forward prefetch was on.  Not used by the product.
template <class M>
__global__ __launch_bounds__(WAVE, 1) void k_kkt_fwdbwd_seq(dto_kkt_args a) {
  kkt_fwd_body<M, false>(a);
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  kkt_bwd_body<M, false>(a);
}
#endif
// A forward launch ends with its wavefront slots draining: the tiles that start late and climb a long regularisation ladder
// (up to six rounds, 20 ms) run alone at the end -- 105-110 ms for 80 ms of work at 8 192 tiles on 1 024 slots
// (profiles/r03/sq_counters_soa_sweeps_B524288_final.txt).  k_kkt_bwd_early runs on a second, low-priority stream next to the
// forward launch: its blocks get the slots the forward launch no longer fills and do the back substitution of every tile
// whose forward sweep has published this iteration's tag (release / acquire at agent scope).  A block never waits: a tile that
// is not ready is left to k_kkt_bwd_rest, which the host launches after both.
// The early kernel must not take wavefront slots while the forward launch still has blocks to start (stream priorities only
// weight the arbitration: without the gate the forward launch got 9 ms longer).  One wavefront in front of it on the second
// stream: returns when every forward block has started -- from then on free slots are slots the forward launch cannot use.
// Bounded (~0.3 s), one slot of 1 024.
static __global__ __launch_bounds__(WAVE) void k_kkt_bwd_gate(dto_kkt_args a) {
  for (int spin = 0; spin < 100000; ++spin) {
    if (__hip_atomic_load(a.fwd_started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= a.G) return;
    __builtin_amdgcn_s_sleep(127);
  }
}
template <class M>
__global__ __launch_bounds__(WAVE, DTO_SEQ_BWD_OCC) void k_kkt_bwd_early(dto_kkt_args a) {
  if (a.tile_bwd_tag[blockIdx.x] == a.sweep_tag) return;   // done by an earlier pass of this kernel
  if (__hip_atomic_load(a.tile_fwd_tag + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.sweep_tag) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  kkt_bwd_body<M, false>(a);
  if (threadIdx.x == 0) a.tile_bwd_tag[blockIdx.x] = a.sweep_tag;
}
template <class M>
__global__ __launch_bounds__(WAVE, DTO_SEQ_BWD_OCC) void k_kkt_bwd_rest(dto_kkt_args a) {
  if (a.tile_bwd_tag[blockIdx.x] == a.sweep_tag) return;
  kkt_bwd_body<M, false>(a);
}

__device__ __forceinline__ void kkt_post_body(const dto_kkt_args& a, int64_t g) {
  double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  if (sc[SC_STATUS << 6] != 0.0) return;
  double apmax = 1.0, admax = 1.0, gphid = 0.0;
  // (partially unrolled: the loads of several chunks in flight at once -- with one running lane this loop is a chain of
  //  memory latencies, 23 us for 64 chunks; the sums are still taken in chunk order)
#pragma unroll 8
  for (int p = 0; p < a.P; ++p) {
    const double* ca = a.cacc + (((g * a.P + p) * 4) << 6) + threadIdx.x;
    apmax = fmin(apmax, ld_join(&ca[0 << 6]));
    admax = fmin(admax, ld_join(&ca[1 << 6]));
    gphid += ld_join(&ca[2 << 6]);
  }
  // directional derivative of the barrier objective along the step (filter line search, switching condition)
  sc[SC_DMERIT << 6] = gphid;
  // penalty phase: the eight trial steps are alpha_pmax * ascale * 2^-k (ls_reduce_body adapts ascale)
  if (sc[SC_LS_MODE << 6] == 1.0) apmax *= sc[SC_ASCALE << 6];
  sc[SC_ALPHA_PMAX << 6] = apmax;
  sc[SC_ALPHA_DMAX << 6] = admax;
}
static __global__ __launch_bounds__(WAVE) void k_kkt_post(dto_kkt_args a) { kkt_post_body(a, blockIdx.x); }

// Time-partitioned form with the joins inside the launches (tile_last_arrival): one round = ONE launch (the chunk sweeps, and in
// the last wavefront of a tile to finish, the separator system with the inertia verdict), the back substitutions and the
// reduction of their step partials another.
template <class M>
__global__ __launch_bounds__(WAVE) void k_kkt_fwd_sep(dto_kkt_args a) {
  const int64_t g = blockIdx.x / a.P;
  const double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  if (!__any(sc[SC_STATUS << 6] == 0.0 && sc[SC_NEED << 6] != 0.0)) return;   // the same for every chunk wavefront of the tile
  kkt_fwd_body<M, true>(a);
  if (tile_last_arrival(a.csync + g * 4 + 1, a.P)) kkt_sep_body<M>(a, g);
}
template <class M>
__global__ __launch_bounds__(WAVE) void k_kkt_bwd_post(dto_kkt_args a) {
  const int64_t g = blockIdx.x / a.P;
  const double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  if (__all(sc[SC_STATUS << 6] != 0.0)) return;                                // ... of the tile
  kkt_bwd_body<M, true>(a);
  if (tile_last_arrival(a.csync + g * 4 + 2, a.P)) kkt_post_body(a, g);
}


template <class M>
__global__ __launch_bounds__(WAVE) void k_linesearch(dto_kkt_args a) {
  // a wavefront walks DTO_SB consecutive stages and sums the merit partials of the eight trial step sizes in registers
  const int nblk = (a.T + a.sb - 1) / a.sb;
  const int64_t g = blockIdx.x / nblk;
  const int blk = blockIdx.x % nblk;
  const dto_solver_opts& o = a.opt;
  const double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  if (sc[SC_STATUS << 6] != 0.0) return;
  const double mu = sc[SC_MU << 6];
  const double amax = sc[SC_ALPHA_PMAX << 6];
  // the trial loop stays rolled (one copy of the model code, few registers: three wavefronts per SIMD); its sixteen
  // running sums are indexed by the trial number, so they live in LDS (8 KiB per wavefront, lane-private columns)
  __shared__ double s_acc[2 * DTO_LS_TRIALS * WAVE];
  double* acc = s_acc + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 2 * DTO_LS_TRIALS; ++k) acc[k * WAVE] = 0.0;
  const int t_end = (blk + 1) * a.sb < a.T ? (blk + 1) * a.sb : a.T;
  for (int t = blk * a.sb; t < t_end; ++t) {
  dispatch_uniform<M>(uload(a.kind, t), [&](auto kc) {
    constexpr int K = decltype(kc)::value;
    using D = KindDims<M, K>;
    using KD = typename D::KD;
    using CO = typename M::template Cost<KD::COST>;
    const int z0 = uload(a.zoff, t);
    arr<D::NP> p, dp;
    arr<D::NY> y, dy;
#pragma unroll
    for (int i = 0; i < D::NP; ++i) {
      p[i] = *soa(a.z, g, a.Nz, z0 + i);
      dp[i] = *soa(a.dz, g, a.Nz, z0 + i);
    }
#pragma unroll
    for (int i = 0; i < D::NY; ++i) {
      y[i] = *soa(a.z, g, a.Nz, uload(a.zoff, t + 1) + i);
      dy[i] = *soa(a.dz, g, a.Nz, uload(a.zoff, t + 1) + i);
    }
    arr<CO::NW> w;
    load_params(w, a, g, t);
    double blo[D::NP > 0 ? D::NP : 1], bhi[D::NP > 0 ? D::NP : 1];  // bounds: loaded once, not once per trial
#pragma unroll
    for (int i = 0; i < D::NP; ++i) { blo[i] = uload(a.lo, z0 + i); bhi[i] = uload(a.hi, z0 + i); }
    double sl[D::QI > 0 ? D::QI : 1], dsl[D::QI > 0 ? D::QI : 1];
    if (!o.newton_only) {
#pragma unroll
      for (int j = 0; j < D::QI; ++j) {
        sl[j] = *soa(a.s, g, a.Ni, uload(a.ioff, t) + j);
        dsl[j] = *soa(a.ds, g, a.Ni, uload(a.ioff, t) + j);
      }
    }
    // Trial points x + alpha_k dx, alpha_k = alpha_max 2^-k, visited from the SHORTEST step up.  Where every sin / cos
    // argument of the dynamics is affine in (x, u, y) (Dyn::NTRIG > 0) the argument at trial k is a0 + 2^(7-k) da: sin / cos
    // come from ONE sincos of a0 and one of da per argument, the angle-addition identity and the double-angle recurrence,
    // instead of a library sincos per argument and trial (the trigonometry was most of this kernel's arithmetic).  The
    // recurrence doubles the absolute error seven times: ~1e-14 at alpha_max, far below what the filter tests resolve.
    constexpr int NTRIG = []() { if constexpr (KD::DYN >= 0) return M::template Dyn<KD::DYN>::NTRIG; else return 0; }();
    double tS0[NTRIG > 0 ? NTRIG : 1], tC0[NTRIG > 0 ? NTRIG : 1], ts[NTRIG > 0 ? NTRIG : 1], tc[NTRIG > 0 ? NTRIG : 1];
    const double amin = amax * (1.0 / (double)(1 << (DTO_LS_TRIALS - 1)));
    if constexpr (NTRIG > 0) {
      using DY = typename M::template Dyn<KD::DYN>;
      arr<D::NP> pm;
      arr<DY::NY> ym;
#pragma unroll
      for (int i = 0; i < D::NP; ++i) pm[i] = p[i] + amin * dp[i];
#pragma unroll
      for (int i = 0; i < DY::NY; ++i) ym[i] = y[i] + amin * dy[i];
      double a0[NTRIG], a1[NTRIG];
      DY::trig_args(p.data(), p.data() + DY::NX, y.data(), w.data(), a0);
      DY::trig_args(pm.data(), pm.data() + DY::NX, ym.data(), w.data(), a1);
#pragma unroll
      for (int j = 0; j < NTRIG; ++j) {
        DTO_SINCOS(a0[j], &tS0[j], &tC0[j]);
        DTO_SINCOS(a1[j] - a0[j], &ts[j], &tc[j]);
      }
    }
    double alpha = amin;
#pragma unroll 1
    for (int kk = 0; kk < DTO_LS_TRIALS; ++kk) {
      const int k = DTO_LS_TRIALS - 1 - kk;
      arr<D::NP> pk;
      double phi, th = 0.0;
#pragma unroll
      for (int i = 0; i < D::NP; ++i) pk[i] = p[i] + alpha * dp[i];
      {
        double o1[1];
        CO::eval(pk.data(), pk.data() + CO::NX, w.data(), o1);
        phi = o1[0];
      }
      if (!o.newton_only) {
#pragma unroll
        for (int i = 0; i < D::NP; ++i) {
          const double lo = blo[i], hi = bhi[i];
          if (lo != hi) {
            if (finite_lo(lo)) phi -= mu * log(pk[i] - lo);
            if (finite_hi(hi)) phi -= mu * log(hi - pk[i]);
          }
        }
      }
      if constexpr (KD::DYN >= 0) {
        using DY = typename M::template Dyn<KD::DYN>;
        arr<DY::NY> yk, d;
#pragma unroll
        for (int i = 0; i < DY::NY; ++i) yk[i] = y[i] + alpha * dy[i];
        if constexpr (NTRIG > 0) {
          double sn[NTRIG], cs[NTRIG];
#pragma unroll
          for (int j = 0; j < NTRIG; ++j) {
            sn[j] = tS0[j] * tc[j] + tC0[j] * ts[j];
            cs[j] = tC0[j] * tc[j] - tS0[j] * ts[j];
          }
          DY::eval_trig(pk.data(), pk.data() + DY::NX, yk.data(), w.data(), sn, cs, d.data());
#pragma unroll
          for (int j = 0; j < NTRIG; ++j) {   // angle of the next (twice as long) step
            const double s2 = 2.0 * ts[j] * tc[j];
            tc[j] = 1.0 - 2.0 * ts[j] * ts[j];
            ts[j] = s2;
          }
        } else {
          DY::eval(pk.data(), pk.data() + DY::NX, yk.data(), w.data(), d.data());
        }
#pragma unroll
        for (int i = 0; i < DY::NY; ++i) th += fabs(d[i]);
      }
      if constexpr (KD::CON >= 0) {
        using C = typename M::template Con<KD::CON>;
        arr<C::NC> c;
        C::eval(pk.data(), pk.data() + C::NX, w.data(), c.data());
#pragma unroll
        for (int j = 0; j < C::NC; ++j) {
          double r = c[j];
          if (!o.newton_only && D::ineq(j)) {
            const double sk = sl[D::slack(j)] + alpha * dsl[D::slack(j)];
            r += sk;
            phi -= mu * log(sk);
          }
          th += fabs(r);
        }
      }
      acc[(2 * k) * WAVE] += phi;
      acc[(2 * k + 1) * WAVE] += th;
      alpha *= 2.0;
    }
  });
  }
  double* out = a.lspart + (((g * nblk + blk) * (2 * DTO_LS_TRIALS)) << 6) + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 2 * DTO_LS_TRIALS; ++k) out[(int64_t)k << 6] = acc[k * WAVE];
}

// Filter line search of Ipopt (Waechter & Biegler 2006, Algorithm A, steps A-5..A-8) over the precomputed trial step sizes
// alpha_k = alpha_max 2^-k (phi / th: barrier objective and l1 violation of the trials, summed over the stages by the
// caller); no restoration phase: if every trial is rejected the most feasible trial is taken and more regularisation is
// requested.  fl: the instance's filter entries, same stride as the scalars.  SH as in retry_update_t.
template <int SH>
__device__ __forceinline__ void ls_reduce_body(const dto_solver_opts& o, double* sc, double* fl, const double* phi, const double* th) {
  constexpr double G_TH = 1e-5, G_PHI = 1e-8, S_TH = 1.1, S_PHI = 2.3, ETA = 1e-8, DELTA = 1.0;
  const double th0 = sc[SC_THETA1 << SH];
  const double phi0 = sc[SC_MERIT0 << SH];
  const double dphi = sc[SC_DMERIT << SH];
  const double thmax = sc[SC_THETA_MAX << SH], thmin = sc[SC_THETA_MIN << SH];
  // ---- two-phase globalisation (round 5; DESIGN.md section 5, profiles/r05/third_party_cfg3_T1000.json).  Far from the
  //      constraint manifold (theta_inf > ls_switch) the filter takes any step that lowers the violation, whatever it does to the
  //      objective: from the reference's straight-line guesses (examples/acrobot/acrobot.jl:126-127) that is a jump to 20 x the
  //      guess's objective followed by hundreds of iterations back down along the manifold (acrobot T=1000: median 630 iterations
  //      to f ~ 1000; scipy's trust-constr on the oracle's callbacks: 94-269 iterations to f = 273-319).  There the step size is
  //      chosen on the l1 exact-penalty function phi + nu theta_1 (Armijo, nu >= dphi / ((1 - rho) theta_1) + 1; Nocedal & Wright
  //      18.3, Ipopt's line_search_method=penalty), with a persistent scale on the eight trial steps instead of more trials
  //      (all eight rejected: no step, scale / 256; a full step: scale x 4).  Once theta_inf <= ls_switch (conv_body) the filter takes
  //      over for good.  While the phase lasts the Hessian model is Gauss-Newton (conv_body).  Same decisions in
  //      oracle/cpu_port/solver_port.c: convergence, factor_solve, line_search.
  if (sc[SC_LS_MODE << SH] == 1.0) {   // (the phase ends in conv_body, before the factorisation of the iteration)
    {
      constexpr double RHO = 0.1, ETA_P = 1e-4;
      double nu = sc[SC_PENALTY << SH];
      if (th0 > 0.0) {
        const double need = dphi / ((1.0 - RHO) * th0);
        if (nu < need) nu = need + 1.0;
      }
      sc[SC_PENALTY << SH] = nu;
      const double m0 = phi0 + nu * th0, D = dphi - nu * th0, amax = sc[SC_ALPHA_PMAX << SH];
      double alpha = amax, chosen = -1.0;
#pragma unroll 1
      for (int k = 0; k < DTO_LS_TRIALS; ++k) {
        const double mk = phi[k] + nu * th[k];
        if (mk == mk && mk <= m0 + ETA_P * alpha * D + 1e-13 * fabs(m0)) { chosen = alpha; break; }
        alpha *= 0.5;
      }
      const double asc = sc[SC_ASCALE << SH];
      if (chosen < 0.0) {
        chosen = 0.0;
        sc[SC_ALPHA_DMAX << SH] = 0.0;
        sc[SC_ASCALE << SH] = fmax(asc / 256.0, 1e-12);
      } else if (chosen >= amax) {
        sc[SC_ASCALE << SH] = fmin(1.0, asc * 4.0);
      }
      sc[SC_LS_FAIL << SH] = 0.0;
      sc[SC_LS_KIND << SH] = chosen > 0.0 ? 5.0 : -1.0;
      sc[SC_ALPHA << SH] = chosen;
      sc[SC_FULL_STREAK << SH] = (chosen >= amax) ? sc[SC_FULL_STREAK << SH] + 1.0 : 0.0;
      return;
    }
  }
  const int nf_total = (int)sc[SC_FILTER_N << SH];
  const int nf = nf_total < DTO_FILTER_CAP ? nf_total : DTO_FILTER_CAP;
  double alpha = sc[SC_ALPHA_PMAX << SH];
  double chosen = -1.0;
  bool ftype = false;
  int best = 0;
  // Watchdog (Ipopt: watchdog_shortened_iter_trigger = 10, watchdog_trial_iter_max = 3; Chamberlain et al. 1982): after
  // `trigger` consecutive iterations whose step was cut by the filter, the next `trials` iterations take the
  // fraction-to-the-boundary step without consulting the filter (only theta <= theta_max), then the filter decides again
  // from wherever that led.  Ipopt keeps the watchdog iterate and returns to it if the trial iterations end outside the
  // filter; here there is no rollback (no second copy of the iterate per instance).  Effect: DESIGN.md section 5.
  const int wd_left = (int)sc[SC_WATCHDOG << SH];
  const bool watchdog = wd_left > 0;
#pragma unroll 1
  for (int k = 0; k < DTO_LS_TRIALS; ++k) {
    const double tk = th[k], pk = phi[k];
    if (th[k] < th[best] || !(th[best] == th[best])) best = k;
    bool ok = (tk == tk) && (pk == pk) && tk <= thmax;
    const bool sw = dphi < 0.0 && alpha * pow(-dphi, S_PHI) > DELTA * pow(th0, S_TH);
    bool armijo = false;
    if (ok) {
      if (sw && th0 <= thmin) {
        armijo = pk <= phi0 + ETA * alpha * dphi + 1e-13 * fabs(phi0);
        ok = armijo;
      } else {
        ok = (tk <= (1.0 - G_TH) * th0) || (pk <= phi0 - G_PHI * th0);
      }
    }
    // watchdog steps skip the filter, but -- there being no rollback -- they may not let the violation explode:
    // theta <= 3 max(theta_0, 1).  C port: with the bound, runs no longer blow up to f ~ 1e8 and the
    // trigger can be as early as two shortened steps; acrobot T=1000 x 128: 82 % converge within 1000 iterations, 100 of the
    // 105 in a sensible minimiser (f < 2000); without it 50 %, half of them in junk minima.
    if (watchdog) ok = (tk == tk) && (pk == pk) && tk <= thmax && tk <= 3.0 * fmax(th0, 1.0);
    if (ok && !watchdog) {
      for (int i = 0; i < nf; ++i) {
        const double tf = fl[(int64_t)(2 * i) << SH], pf = fl[(int64_t)(2 * i + 1) << SH];
        if (!(tk <= (1.0 - G_TH) * tf || pk <= pf - G_PHI * tf)) { ok = false; break; }
      }
    }
    if (ok) {
      chosen = alpha;
      ftype = sw && (pk <= phi0 + ETA * alpha * dphi + 1e-13 * fabs(phi0));
      break;
    }
    alpha *= 0.5;
  }
  bool augment = false;
  if (chosen < 0.0) {
    // no acceptable trial: take the most feasible one if it improves feasibility, else the shortest step
    double ab = sc[SC_ALPHA_PMAX << SH];
    for (int k = 0; k < best; ++k) ab *= 0.5;
    if (th[best] == th[best] && th[best] < th0) chosen = ab;
    else chosen = alpha * 2.0;  // alpha_max 2^-(TRIALS-1)
    // every trial is catastrophic (even the most feasible one multiplies the violation by > 100; Ipopt would enter its
    // restoration phase): such a direction is not worth any step -- stay, primal and dual, and let SC_LS_FAIL regularise the
    // next system more.  acrobot T = 101, 1024 seeds: 1022 -> 1024 converge, the other workloads unchanged.
    if (!(th[best] <= DTO_LS_NULL_STEP * fmax(th0, 1.0))) {
      chosen = 0.0;
      sc[SC_ALPHA_DMAX << SH] = 0.0;
    }
    sc[SC_LS_FAIL << SH] = 1.0;
    augment = true;
  } else {
    sc[SC_LS_FAIL << SH] = 0.0;
    augment = !ftype && !watchdog;
  }
  if (watchdog) {
    sc[SC_WATCHDOG << SH] = (double)(wd_left - 1);
    sc[SC_SHORT_STREAK << SH] = 0.0;
  } else if (o.watchdog_trigger > 0) {
    const double streak = (chosen < sc[SC_ALPHA_PMAX << SH]) ? sc[SC_SHORT_STREAK << SH] + 1.0 : 0.0;
    if (streak >= (double)o.watchdog_trigger) {
      sc[SC_WATCHDOG << SH] = (double)o.watchdog_trials;
      sc[SC_SHORT_STREAK << SH] = 0.0;
    } else {
      sc[SC_SHORT_STREAK << SH] = streak;
    }
  }
  if (augment) {
    const int slot = nf_total % DTO_FILTER_CAP;
    fl[(int64_t)(2 * slot) << SH] = (1.0 - G_TH) * th0;
    fl[(int64_t)(2 * slot + 1) << SH] = phi0 - G_PHI * th0;
    sc[SC_FILTER_N << SH] = (double)(nf_total + 1);
  }
  sc[SC_LS_KIND << SH] = chosen < 0.0 ? -1.0 : (watchdog ? 3.0 : (ftype ? 1.0 : 2.0));
  sc[SC_ALPHA << SH] = chosen;
  // consecutive full (fraction-to-the-boundary) steps: consulted by k_conv when it picks the first delta_w to try
  sc[SC_FULL_STREAK << SH] = (chosen >= sc[SC_ALPHA_PMAX << SH]) ? sc[SC_FULL_STREAK << SH] + 1.0 : 0.0;
}

__device__ __forceinline__ void ls_reduce_tile(const dto_kkt_args& a, const int64_t g) {
  double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  if (sc[SC_STATUS << 6] != 0.0) return;
  double phi[DTO_LS_TRIALS], th[DTO_LS_TRIALS];
#pragma unroll
  for (int k = 0; k < DTO_LS_TRIALS; ++k) phi[k] = th[k] = 0.0;
  // (partially unrolled: the loads of several chunks in flight at once -- with one running lane this loop is a chain of
  //  memory latencies, 23 us for 64 chunks; the sums are still taken in chunk order)
#pragma unroll 8
  for (int c = 0; c < a.P; ++c) {
    const double* in = a.cpart + (((g * a.P + c) * 16) << 6) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < DTO_LS_TRIALS; ++k) {
      phi[k] += ld_join(&in[(int64_t)(2 * k) << 6]);
      th[k] += ld_join(&in[(int64_t)(2 * k + 1) << 6]);
    }
  }
  ls_reduce_body<6>(a.opt, sc, a.filt + ((g * (2 * DTO_FILTER_CAP)) << 6) + threadIdx.x, phi, th);
}
static __global__ __launch_bounds__(WAVE) void k_ls_reduce(dto_kkt_args a) { ls_reduce_tile(a, blockIdx.x); }
// chunk sums of the merit partials and, in the last wavefront of a tile to finish them, the choice of the step size (grid = G * P)
static __global__ __launch_bounds__(WAVE) void k_part_reduce_ls(dto_kkt_args a, const double* in) {
  part_reduce_body<2 * DTO_LS_TRIALS>(a, in, 0u);
  const int64_t g = blockIdx.x / a.P;
  if (tile_last_arrival(a.csync + g * 4 + 3, a.P)) ls_reduce_tile(a, g);
}

// ------------------------------------------------------------------------------------------------
// take the step.  grid = G*T waves.
// ------------------------------------------------------------------------------------------------
template <class M>
__global__ __launch_bounds__(WAVE) void k_update(dto_kkt_args a) {
  const int64_t g = blockIdx.x / a.T;
  const int t = blockIdx.x % a.T;
  const dto_solver_opts& o = a.opt;
  double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  const bool running = sc[SC_STATUS << 6] == 0.0;
  if (!running) return;
  const double mu = sc[SC_MU << 6];
  const double al = sc[SC_ALPHA << 6];
  const double ad = sc[SC_ALPHA_DMAX << 6];
  constexpr double KSIG = 1e10;
  dispatch_uniform<M>(uload(a.kind, t), [&](auto kc) {
    constexpr int K = decltype(kc)::value;
    using D = KindDims<M, K>;
    if (!running) return;
    const int z0 = uload(a.zoff, t);
    StageBounds<D::NP> sb;
    load_stage_bounds<D::NP>(a, g, z0, sb);
    double dpv[D::NP > 0 ? D::NP : 1], pv[D::NP > 0 ? D::NP : 1];
#pragma unroll
    for (int i = 0; i < D::NP; ++i) {
      pv[i] = *soa(a.z, g, a.Nz, z0 + i);
      dpv[i] = *soa(a.dz, g, a.Nz, z0 + i);
    }
#pragma unroll
    for (int i = 0; i < D::NP; ++i) {
      const double p = pv[i];
      const double dp = dpv[i];
      const double pn = p + al * dp;
      if (!o.newton_only) {
        const double lo = sb.lo[i], hi = sb.hi[i];
        if (lo != hi) {
          if (finite_lo(lo)) {
            const double zl = sb.zl[i];
            const double gap = p - lo;
            const double dzl = mu / gap - zl - (zl / gap) * dp;
            double zn = zl + ad * dzl;
            const double gn = pn - lo;
            zn = fmin(fmax(zn, mu / (KSIG * gn)), KSIG * mu / gn);
            *soa(a.zl, g, a.Nz, z0 + i) = zn;
          }
          if (finite_hi(hi)) {
            const double zu = sb.zu[i];
            const double gap = hi - p;
            const double dzu = mu / gap - zu + (zu / gap) * dp;
            double zn = zu + ad * dzu;
            const double gn = hi - pn;
            zn = fmin(fmax(zn, mu / (KSIG * gn)), KSIG * mu / gn);
            *soa(a.zu, g, a.Nz, z0 + i) = zn;
          }
        }
      }
      *soa(a.z, g, a.Nz, z0 + i) = pn;
    }
#pragma unroll
    for (int j = 0; j < D::Q; ++j) {
      const double nu = *soa(a.lam, g, a.Nc, uload(a.ccoff, t) + j);
      const double dnu = *soa(a.dlam, g, a.Nc, uload(a.ccoff, t) + j);
      if (!o.newton_only && D::ineq(j)) {
        const double sv = *soa(a.s, g, a.Ni, uload(a.ioff, t) + D::slack(j));
        const double zv = *soa(a.zs, g, a.Ni, uload(a.ioff, t) + D::slack(j));
        const double dsv = *soa(a.ds, g, a.Ni, uload(a.ioff, t) + D::slack(j));
        const double dzs = mu / sv - zv - (zv / sv) * dsv;
        const double sn = sv + al * dsv;
        double zn = zv + ad * dzs;
        zn = fmin(fmax(zn, mu / (KSIG * sn)), KSIG * mu / sn);
        *soa(a.s, g, a.Ni, uload(a.ioff, t) + D::slack(j)) = sn;
        *soa(a.zs, g, a.Ni, uload(a.ioff, t) + D::slack(j)) = zn;
      }
      *soa(a.lam, g, a.Nc, uload(a.ccoff, t) + j) = nu + al * dnu;
    }
#pragma unroll
    for (int k = 0; k < D::NY; ++k) {
      const double lam = *soa(a.lam, g, a.Nc, uload(a.cdoff, t) + k);
      *soa(a.lam, g, a.Nc, uload(a.cdoff, t) + k) = lam + al * *soa(a.dlam, g, a.Nc, uload(a.cdoff, t) + k);
    }
  });
  if (t == 0 && running) sc[SC_ITER << 6] += 1.0;
}

// ------------------------------------------------------------------------------------------------
// limited-memory BFGS (round 5).  The reference's default -- Solver(...; evaluate_hessian=false),
// src/solver.jl:7, and its own acrobot / car examples -- leaves Ipopt on hessian_approximation = limited-memory: the Hessian of
// the Lagrangian is the compact L-BFGS matrix of Byrd, Nocedal & Schnabel (1994)
//     B = sigma I - W M^-1 W',   W = [sigma S, Y],   M = [[sigma S'S, L], [L', -D]]     (S, Y: the last QN_M = 6 secant pairs;
//                                                                                        L, D: strictly lower part / diagonal of S'Y)
// with y_k = grad_x L(x_{k+1}, lam_{k+1}) - grad_x L(x_k, lam_{k+1}), sigma = s'y / s's of the newest pair
// (limited_memory_initialization = scalar1), the update skipped when s'y <= sqrt(eps) |s| |y|, the history dropped after two
// skips in a row.  sigma I is block diagonal, so K0 = [sigma I + Sigma + delta_w I, J'; J, -D_c] is what the sweeps factorise
// with the constraint curvature off (gam = 0), the objective Hessian scaled away (cost_hess_scale = 0) and sigma riding on
// delta_w; the 2 QN_M columns of U = [W; 0] are a dense border:
//     K v = b,  K = K0 - U M^-1 U':    v = v0 + Z q,   v0 = K0^-1 b,   Z = K0^-1 U,   C = M - U'Z,   q = C^-1 U'v0.
// Every solve with K0 is a plain forward + backward sweep of the SOLVER kernels at the same point with a modified stage record:
// K0 v_a = -(r_p0 - u + barrier terms; c; d) gives v_a - v0 = K0^-1 (u; 0; 0) by linearity (no kernel learns about right-hand
// sides), and the corrected step itself is one more such solve with u = U q -- its fraction-to-the-boundary limits and slack
// steps come out of the back substitution as always; only grad phi' d needs q'(U'dz) added, and U'dz = U'v0 + (U'Z) q is known.
// The kernels below are one wavefront per tile, lane = instance, plain loops over the rows: everything per instance (dot
// products, the 12 x 12 algebra) is lane-local.  Mirrors oracle/cpu_port/solver_port.c: qn_update / qn_factor_solve / qn_save.
// ------------------------------------------------------------------------------------------------
constexpr int QN_M = 6, QN_M2 = 2 * QN_M;
constexpr int QN_SS = 0, QN_SY = QN_SS + QN_M * QN_M, QN_UZ = QN_SY + QN_M * QN_M, QN_UV = QN_UZ + QN_M2 * QN_M2, QN_Q = QN_UV + QN_M2,
              QN_SMALL = QN_Q + QN_M2;
// the row loops of this mode run on QN_NB wavefronts per tile (a batch of one has ONE running lane: a loop over the N_z rows
// in one wavefront is a chain of N_z memory latencies -- 0.55 ms for k_qn_small at T = 101, profiles/r05/)
constexpr int QN_NB = 16, QN_GRAM = QN_M2 * QN_M2 + QN_M2;
struct QnRows {
  int64_t Nz;
  __host__ __device__ int64_t S(int j) const { return (int64_t)j * Nz; }
  __host__ __device__ int64_t Y(int j) const { return (int64_t)(QN_M + j) * Nz; }
  __host__ __device__ int64_t Z(int j) const { return (int64_t)(2 * QN_M + j) * Nz; }
  __host__ __device__ int64_t rp0() const { return (int64_t)(4 * QN_M) * Nz; }
  __host__ __device__ int64_t gl() const { return rp0() + Nz; }
  __host__ __device__ int64_t qs() const { return gl() + Nz; }
  __host__ __device__ int64_t v0() const { return qs() + Nz; }
  __host__ __device__ int64_t small() const { return v0() + Nz; }
  __host__ __device__ int64_t part() const { return small() + QN_SMALL; }     // QN_NB partial sums of U'Z and U'v0 (k_qn_gram)
  __host__ __device__ int64_t total() const { return part() + (int64_t)QN_NB * QN_GRAM; }
};
// rows [r0, r1) of block `blk` of QN_NB
__device__ __forceinline__ void qn_row_block(int64_t n, int blk, int64_t& r0, int64_t& r1) {
  r0 = (n * blk) / QN_NB;
  r1 = (n * (blk + 1)) / QN_NB;
}
__device__ __forceinline__ double* qn_tile(const dto_kkt_args& a, int64_t g) {
  return a.qn + ((g * QnRows{a.Nz}.total()) << 6) + threadIdx.x;
}
__device__ __forceinline__ double qn_u(const double* q, const QnRows& R, int col, int64_t row, double sigma) {
  return col < QN_M ? sigma * q[(R.S(col) + row) << 6] : q[(R.Y(col - QN_M) + row) << 6];
}

// QN_BEGIN in three launches, each on QN_NB wavefronts per tile (k_qn_fin: one) -- partial sums go through the `part` rows:
//   k_qn_pair: the Lagrangian gradient of the new point (stage records) -> r_p0; y = r_p0 - grad_x L(x_k, lam_{k+1}) into the gl rows,
//              s into the qs rows; partial s'y, s's, y'y of the block's stages
//   k_qn_hist: curvature test (every block sums the partials and decides alike: Ipopt skips the update; two skips in a row restart
//              the approximation), history rows of the block: oldest pair out, newest in; partial S'S and S'Y of its rows
//   k_qn_fin:  S'S, S'Y; sigma and the skip count
constexpr int QN_P_SY = 0, QN_P_SS = 1, QN_P_YY = 2, QN_P_G = 3;   // offsets inside a block's QN_GRAM partial rows
struct QnDecision { bool have, accept, reset; double sy, ss, skipped; };
__device__ __forceinline__ QnDecision qn_decide(const double* sc, const double* q, const QnRows& R) {
  QnDecision d;
  d.have = sc[SC_ITER << 6] > 0.0 && sc[SC_ALPHA << 6] > 0.0;
  double sy = 0.0, ss = 0.0, yy = 0.0;
#pragma unroll
  for (int b = 0; b < QN_NB; ++b) {
    sy += q[(R.part() + (int64_t)b * QN_GRAM + QN_P_SY) << 6];
    ss += q[(R.part() + (int64_t)b * QN_GRAM + QN_P_SS) << 6];
    yy += q[(R.part() + (int64_t)b * QN_GRAM + QN_P_YY) << 6];
  }
  d.sy = sy; d.ss = ss;
  d.accept = d.have && sy > 1.4901161193847656e-08 * sqrt(ss) * sqrt(yy);
  d.skipped = d.accept ? 0.0 : sc[SC_QN_SKIP << 6] + 1.0;
  d.reset = d.have && d.skipped >= 2.0;
  return d;
}
template <class M>
__global__ __launch_bounds__(WAVE) void k_qn_pair(dto_kkt_args a) {   // grid = G * QN_NB (stages in QN_NB blocks)
  const int64_t g = blockIdx.x / QN_NB;
  const int blk = (int)(blockIdx.x % QN_NB);
  const double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  if (sc[SC_STATUS << 6] != 0.0) return;
  const QnRows R{a.Nz};
  double* q = qn_tile(a, g);
  const bool have = sc[SC_ITER << 6] > 0.0 && sc[SC_ALPHA << 6] > 0.0;
  double sy = 0.0, ss = 0.0, yy = 0.0;
  const int t0 = (int)(((int64_t)a.T * blk) / QN_NB), t1 = (int)(((int64_t)a.T * (blk + 1)) / QN_NB);
  for (int t = t0; t < t1; ++t) {
    dispatch_uniform<M>(uload(a.kind, t), [&](auto kc) {
      constexpr int K = decltype(kc)::value;
      using D = KindDims<M, K>;
      const SoaIO<M, K> io(a, g, t);
#pragma unroll
      for (int i = 0; i < D::NP; ++i) {
        const int64_t row = io.z0 + i;
        const double rp = io.rec(D::R_RP + i);
        const bool fx = uload(a.lo, row) == uload(a.hi, row);
        const double y = (fx || !have) ? 0.0 : rp - q[(R.gl() + row) << 6];
        const double sv = (fx || !have) ? 0.0 : q[(R.qs() + row) << 6];
        q[(R.rp0() + row) << 6] = rp;
        q[(R.gl() + row) << 6] = y;
        q[(R.qs() + row) << 6] = sv;
        sy += sv * y; ss += sv * sv; yy += y * y;
      }
    });
  }
  double* out = q + ((R.part() + (int64_t)blk * QN_GRAM) << 6);
  out[(int64_t)QN_P_SY << 6] = sy;
  out[(int64_t)QN_P_SS << 6] = ss;
  out[(int64_t)QN_P_YY << 6] = yy;
}
static __global__ __launch_bounds__(WAVE) void k_qn_hist(dto_kkt_args a) {   // grid = G * QN_NB (rows in QN_NB blocks)
  const int64_t g = blockIdx.x / QN_NB;
  const int blk = (int)(blockIdx.x % QN_NB);
  const double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  if (sc[SC_STATUS << 6] != 0.0) return;
  const QnRows R{a.Nz};
  double* q = qn_tile(a, g);
  const QnDecision d = qn_decide(sc, q, R);
  const bool accept = d.accept, reset = d.reset;
  int64_t r0, r1;
  qn_row_block(a.Nz, blk, r0, r1);
  double SS[QN_M * QN_M], SY[QN_M * QN_M];
#pragma unroll
  for (int i = 0; i < QN_M * QN_M; ++i) SS[i] = SY[i] = 0.0;
#pragma unroll 2
  for (int64_t row = r0; row < r1; ++row) {
    double sj[QN_M], yj[QN_M];
#pragma unroll
    for (int j = 0; j < QN_M; ++j) { sj[j] = q[(R.S(j) + row) << 6]; yj[j] = q[(R.Y(j) + row) << 6]; }
    const double sn = q[(R.qs() + row) << 6], yn = q[(R.gl() + row) << 6];
#pragma unroll
    for (int j = 0; j < QN_M; ++j) {
      const double s_new = j + 1 < QN_M ? sj[j + 1 < QN_M ? j + 1 : j] : sn, y_new = j + 1 < QN_M ? yj[j + 1 < QN_M ? j + 1 : j] : yn;
      const double so = reset ? 0.0 : (accept ? s_new : sj[j]), yo = reset ? 0.0 : (accept ? y_new : yj[j]);
      q[(R.S(j) + row) << 6] = so;
      q[(R.Y(j) + row) << 6] = yo;
      sj[j] = so; yj[j] = yo;      // (only after every old value of the row has been read: s_new / y_new above use the loads)
    }
#pragma unroll
    for (int aa = 0; aa < QN_M; ++aa) {
#pragma unroll
      for (int bb = 0; bb < QN_M; ++bb) {
        if (bb <= aa) SS[aa * QN_M + bb] += sj[aa] * sj[bb];
        SY[aa * QN_M + bb] += sj[aa] * yj[bb];
      }
    }
  }
  double* out = q + ((R.part() + (int64_t)blk * QN_GRAM + QN_P_G) << 6);
#pragma unroll
  for (int i = 0; i < QN_M * QN_M; ++i) {
    out[(int64_t)i << 6] = SS[i];
    out[(int64_t)(QN_M * QN_M + i) << 6] = SY[i];
  }
}
static __global__ __launch_bounds__(WAVE) void k_qn_fin(dto_kkt_args a) {   // grid = G
  const int64_t g = blockIdx.x;
  double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  if (sc[SC_STATUS << 6] != 0.0) return;
  const QnRows R{a.Nz};
  double* q = qn_tile(a, g);
  const QnDecision d = qn_decide(sc, q, R);
  if (d.have) sc[SC_QN_SKIP << 6] = d.reset ? 0.0 : d.skipped;
  if (d.accept) sc[SC_QN_SIGMA << 6] = fmin(1e8, fmax(1e-8, d.sy / d.ss));
  if (d.reset) sc[SC_QN_SIGMA << 6] = 1.0;
#pragma unroll 4
  for (int i = 0; i < 2 * QN_M * QN_M; ++i) {
    double acc = 0.0;
#pragma unroll
    for (int b = 0; b < QN_NB; ++b) acc += q[(R.part() + (int64_t)b * QN_GRAM + QN_P_G + i) << 6];
    const int e = i % (QN_M * QN_M), aa = e / QN_M, bb = e % QN_M;
    if (i < QN_M * QN_M) {
      if (bb <= aa) {      // S'S: the lower triangle was accumulated, both halves are stored
        q[(R.small() + QN_SS + aa * QN_M + bb) << 6] = acc;
        q[(R.small() + QN_SS + bb * QN_M + aa) << 6] = acc;
      }
    } else {
      q[(R.small() + QN_SY + aa * QN_M + bb) << 6] = acc;
    }
  }
}

// stage records: r_p := r_p0 - u  (mode 0: u = column qn_col of U; 1: u = U q; 2: u = 0 and grad phi' d gets its correction);
// modes 0 / 1 request ONE more factorisation with the (delta_w, gam = 0) the iteration accepted
template <class M>
__global__ __launch_bounds__(WAVE) void k_qn_rhs(dto_kkt_args a) {   // grid = G * T waves (tile, stage)
  const int64_t g = blockIdx.x / a.T;
  const int t = blockIdx.x % a.T;
  double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  if (sc[SC_STATUS << 6] != 0.0) return;
  const QnRows R{a.Nz};
  double* q = qn_tile(a, g);
  const double sigma = sc[SC_QN_SIGMA << 6];
  double qc[QN_M2];
#pragma unroll
  for (int j = 0; j < QN_M2; ++j) qc[j] = a.qn_mode == 1 ? q[(R.small() + QN_Q + j) << 6] : 0.0;
  {
    dispatch_uniform<M>(uload(a.kind, t), [&](auto kc) {
      constexpr int K = decltype(kc)::value;
      using D = KindDims<M, K>;
      const SoaIO<M, K> io(a, g, t);
      double* recw = const_cast<double*>(io.recp);
#pragma unroll
      for (int i = 0; i < D::NP; ++i) {
        const int64_t row = io.z0 + i;
        double v = q[(R.rp0() + row) << 6];
        if (a.qn_mode == 0) v -= qn_u(q, R, a.qn_col, row, sigma);
        if (a.qn_mode == 1) {
#pragma unroll
          for (int j = 0; j < QN_M2; ++j) v -= qc[j] * qn_u(q, R, j, row, sigma);
        }
        recw[pair_at(D::R_RP + i)] = v;
      }
    });
  }
  if (t != 0) return;
  if (a.qn_mode == 2) {
    sc[SC_DMERIT << 6] += sc[SC_QN_GCORR << 6];
  } else {
    sc[SC_NEED << 6] = 1.0;
    sc[SC_ATTEMPT << 6] = (double)a.opt.max_refactor;   // this attempt is the one that gets used, whatever its inertia
    sc[SC_TRY_DW << 6] = sc[SC_DELTA_W << 6];
    sc[SC_TRY_GAM << 6] = 0.0;
  }
}

// The QN_M2 solves K0 z_c = u_c of an iteration side by side (small batches): a second solver state whose slot s * QN_M2 + c is a
// copy of the main state's slot s (dto_solver.cpp: k_qn_cols_copy) gets column c of that slot's U subtracted from its r_p here --
// one FACTOR_SOLVE of the column state then does what QN_M2 x (QN_RHS, FACTOR_SOLVE, QN_COL) did one after the other, and
// k_qn_cols_gather takes Z_c = dz_c - v0 back.  grid = G' * T waves (column tile, stage).
template <class M>
__global__ __launch_bounds__(WAVE) void k_qn_cols_rhs(dto_kkt_args a) {
  const int64_t g = blockIdx.x / a.T;
  const int t = blockIdx.x % a.T;
  double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  if (sc[SC_STATUS << 6] != 0.0) return;
  const int64_t slot = g * 64 + threadIdx.x, ms = slot / QN_M2;     // main slot
  const int col = (int)(slot % QN_M2);
  const QnRows R{a.Nz};
  const double* q = a.qn_main + (((ms >> 6) * R.total()) << 6) + (ms & 63);
  const double sigma = sc[SC_QN_SIGMA << 6];
  dispatch_uniform<M>(uload(a.kind, t), [&](auto kc) {
    constexpr int K = decltype(kc)::value;
    using D = KindDims<M, K>;
    const SoaIO<M, K> io(a, g, t);
    double* recw = const_cast<double*>(io.recp);
#pragma unroll
    for (int i = 0; i < D::NP; ++i) recw[pair_at(D::R_RP + i)] -= qn_u(q, R, col, io.z0 + i, sigma);
  });
  if (t == 0) {
    sc[SC_NEED << 6] = 1.0;
    sc[SC_ATTEMPT << 6] = (double)a.opt.max_refactor;   // this attempt is the one that gets used, whatever its inertia
    sc[SC_TRY_DW << 6] = sc[SC_DELTA_W << 6];
    sc[SC_TRY_GAM << 6] = 0.0;
  }
}

// Z_col := dz - v0 (qn_col >= 0), v0 := dz (qn_col = -1)
static __global__ __launch_bounds__(WAVE) void k_qn_col(dto_kkt_args a) {   // grid = G * QN_NB
  const int64_t g = blockIdx.x / QN_NB;
  const double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  if (sc[SC_STATUS << 6] != 0.0) return;
  const QnRows R{a.Nz};
  double* q = qn_tile(a, g);
  int64_t r0, r1;
  qn_row_block(a.Nz, (int)(blockIdx.x % QN_NB), r0, r1);
#pragma unroll 4
  for (int64_t row = r0; row < r1; ++row) {
    const double d = *soa(a.dz, g, a.Nz, row);
    if (a.qn_col < 0) q[(R.v0() + row) << 6] = d;
    else q[(R.Z(a.qn_col) + row) << 6] = d - q[(R.v0() + row) << 6];
  }
}

// partial sums of U'Z (QN_M2 x QN_M2) and U'v0 over the rows of one block; grid = G * QN_NB
static __global__ __launch_bounds__(WAVE) void k_qn_gram(dto_kkt_args a) {
  const int64_t g = blockIdx.x / QN_NB;
  const int blk = (int)(blockIdx.x % QN_NB);
  const double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  if (sc[SC_STATUS << 6] != 0.0) return;
  const QnRows R{a.Nz};
  double* q = qn_tile(a, g);
  const double sigma = sc[SC_QN_SIGMA << 6];
  int64_t r0, r1;
  qn_row_block(a.Nz, blk, r0, r1);
  for (int half = 0; half < 2; ++half) {     // two passes of six rows of U (72 accumulators each)
    double acc[QN_M * QN_M2], av[QN_M];
#pragma unroll
    for (int i = 0; i < QN_M * QN_M2; ++i) acc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < QN_M; ++i) av[i] = 0.0;
#pragma unroll 2
    for (int64_t row = r0; row < r1; ++row) {
      double u[QN_M], z[QN_M2];
#pragma unroll
      for (int j = 0; j < QN_M; ++j) u[j] = half == 0 ? sigma * q[(R.S(j) + row) << 6] : q[(R.Y(j) + row) << 6];
#pragma unroll
      for (int j = 0; j < QN_M2; ++j) z[j] = q[(R.Z(j) + row) << 6];
      const double v0 = q[(R.v0() + row) << 6];
#pragma unroll
      for (int aa = 0; aa < QN_M; ++aa) {
        av[aa] += u[aa] * v0;
#pragma unroll
        for (int bb = 0; bb < QN_M2; ++bb) acc[aa * QN_M2 + bb] += u[aa] * z[bb];
      }
    }
    double* out = q + ((R.part() + (int64_t)blk * QN_GRAM) << 6);
#pragma unroll
    for (int aa = 0; aa < QN_M; ++aa) {
      out[(int64_t)(QN_M2 * QN_M2 + half * QN_M + aa) << 6] = av[aa];
#pragma unroll
      for (int bb = 0; bb < QN_M2; ++bb) out[(int64_t)((half * QN_M + aa) * QN_M2 + bb) << 6] = acc[aa * QN_M2 + bb];
    }
  }
}

// U'Z and U'v0 from k_qn_gram's partial sums, then per lane: M, C = M - U'Z, q = C^-1 U'v0 (Gaussian elimination with partial pivoting in the
// lane's LDS column), q'(U'dz) for the directional derivative.  Empty history slots (zero columns) are decoupled.
static __global__ __launch_bounds__(WAVE) void k_qn_small(dto_kkt_args a) {
  const int64_t g = blockIdx.x;
  double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  if (sc[SC_STATUS << 6] != 0.0) return;
  const QnRows R{a.Nz};
  double* q = qn_tile(a, g);
  const double sigma = sc[SC_QN_SIGMA << 6];
  __shared__ double lds[(QN_M2 * QN_M2 + 2 * QN_M2) * WAVE];
  double* Cm = lds + threadIdx.x;                        // Cm[(a * QN_M2 + b) * WAVE]
  double* tv = lds + QN_M2 * QN_M2 * WAVE + threadIdx.x;  // U'v0, then q
  double* uzq = tv + QN_M2 * WAVE;                        // (U'Z) q
  // U'Z and U'v0: the QN_NB partial sums of k_qn_gram, in block order
#pragma unroll 4
  for (int i = 0; i < QN_M2 * QN_M2; ++i) {
    double acc = 0.0;
#pragma unroll
    for (int b = 0; b < QN_NB; ++b) acc += q[(R.part() + (int64_t)b * QN_GRAM + i) << 6];
    Cm[i * WAVE] = acc;
  }
  for (int j = 0; j < QN_M2; ++j) {
    double acc = 0.0;
#pragma unroll
    for (int b = 0; b < QN_NB; ++b) acc += q[(R.part() + (int64_t)b * QN_GRAM + QN_M2 * QN_M2 + j) << 6];
    tv[j * WAVE] = acc;
  }
  // keep U'Z (for the correction of the directional derivative), form C = M - sym(U'Z)
  double gcorr_t[QN_M2];
#pragma unroll
  for (int j = 0; j < QN_M2; ++j) gcorr_t[j] = tv[j * WAVE];
  for (int aa = 0; aa < QN_M2; ++aa)
    for (int bb = 0; bb < QN_M2; ++bb) q[(R.small() + QN_UZ + aa * QN_M2 + bb) << 6] = Cm[(aa * QN_M2 + bb) * WAVE];
  for (int aa = 0; aa < QN_M2; ++aa)
    for (int bb = 0; bb <= aa; ++bb) {
      const double uz = 0.5 * (Cm[(aa * QN_M2 + bb) * WAVE] + Cm[(bb * QN_M2 + aa) * WAVE]);
      double m;
      if (aa < QN_M) m = sigma * q[(R.small() + QN_SS + aa * QN_M + bb) << 6];                                    // sigma S'S
      else if (bb < QN_M) m = (aa - QN_M) < bb ? q[(R.small() + QN_SY + bb * QN_M + (aa - QN_M)) << 6] : 0.0;     // L' (row a of Y, column b of S: s_b'y_a for b > a)
      else m = aa == bb ? -q[(R.small() + QN_SY + (aa - QN_M) * QN_M + (aa - QN_M)) << 6] : 0.0;                 // -D
      Cm[(aa * QN_M2 + bb) * WAVE] = Cm[(bb * QN_M2 + aa) * WAVE] = m - uz;
    }
  // empty slots: s_j = 0 (history not full yet, or just restarted): rows / columns j and QN_M + j are decoupled
  for (int j = 0; j < QN_M; ++j) {
    if (q[(R.small() + QN_SS + j * QN_M + j) << 6] == 0.0) {
      for (int k = 0; k < QN_M2; ++k) {
        Cm[(j * QN_M2 + k) * WAVE] = Cm[(k * QN_M2 + j) * WAVE] = 0.0;
        Cm[((QN_M + j) * QN_M2 + k) * WAVE] = Cm[(k * QN_M2 + QN_M + j) * WAVE] = 0.0;
      }
      Cm[(j * QN_M2 + j) * WAVE] = 1.0; Cm[((QN_M + j) * QN_M2 + QN_M + j) * WAVE] = 1.0;
      tv[j * WAVE] = 0.0; tv[(QN_M + j) * WAVE] = 0.0;
    }
  }
  // q = C^-1 t
  bool singular = false;
  for (int k = 0; k < QN_M2; ++k) {
    int piv = k;
    for (int r = k + 1; r < QN_M2; ++r) if (fabs(Cm[(r * QN_M2 + k) * WAVE]) > fabs(Cm[(piv * QN_M2 + k) * WAVE])) piv = r;
    if (piv != k) {
      for (int c = 0; c < QN_M2; ++c) { const double t_ = Cm[(k * QN_M2 + c) * WAVE]; Cm[(k * QN_M2 + c) * WAVE] = Cm[(piv * QN_M2 + c) * WAVE]; Cm[(piv * QN_M2 + c) * WAVE] = t_; }
      const double t_ = tv[k * WAVE]; tv[k * WAVE] = tv[piv * WAVE]; tv[piv * WAVE] = t_;
    }
    const double d = Cm[(k * QN_M2 + k) * WAVE];
    if (!(fabs(d) > 1e-300)) { singular = true; continue; }
    for (int r = k + 1; r < QN_M2; ++r) {
      const double f = Cm[(r * QN_M2 + k) * WAVE] / d;
      if (f == 0.0) continue;
      for (int c = k; c < QN_M2; ++c) Cm[(r * QN_M2 + c) * WAVE] -= f * Cm[(k * QN_M2 + c) * WAVE];
      tv[r * WAVE] -= f * tv[k * WAVE];
    }
  }
  for (int k = QN_M2 - 1; k >= 0; --k) {
    double acc = tv[k * WAVE];
    for (int c = k + 1; c < QN_M2; ++c) acc -= Cm[(k * QN_M2 + c) * WAVE] * tv[c * WAVE];
    const double d = Cm[(k * QN_M2 + k) * WAVE];
    tv[k * WAVE] = (fabs(d) > 1e-300) ? acc / d : 0.0;
  }
  // q'(U'dz) = q'(U'v0 + (U'Z) q)
  double gcorr = 0.0;
  for (int aa = 0; aa < QN_M2; ++aa) {
    double r = gcorr_t[0];
#pragma unroll
    for (int j = 0; j < QN_M2; ++j) r = (j == aa) ? gcorr_t[j] : r;
    for (int bb = 0; bb < QN_M2; ++bb) r += q[(R.small() + QN_UZ + aa * QN_M2 + bb) << 6] * (singular ? 0.0 : tv[bb * WAVE]);
    gcorr += (singular ? 0.0 : tv[aa * WAVE]) * r;
  }
  for (int j = 0; j < QN_M2; ++j) q[(R.small() + QN_Q + j) << 6] = singular ? 0.0 : tv[j * WAVE];
  sc[SC_QN_GCORR << 6] = gcorr;
  (void)uzq;
}

// after the line search: grad_x L(x_k, lam_{k+1}) = r_p0 + alpha J(x_k)' dlam, with J'dlam from the first block row of the system
// just solved, (B + Sigma + delta_w I) dz + J'dlam = -(r_p0 + barrier terms) and B dz = sigma dz - U q; and s = alpha dz
static __global__ __launch_bounds__(WAVE) void k_qn_save(dto_kkt_args a) {   // grid = G * QN_NB
  const int64_t g = blockIdx.x / QN_NB;
  const double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  if (sc[SC_STATUS << 6] != 0.0) return;
  const QnRows R{a.Nz};
  double* q = qn_tile(a, g);
  const double sigma = sc[SC_QN_SIGMA << 6], al = sc[SC_ALPHA << 6], dw = sc[SC_DELTA_W << 6], mu = sc[SC_MU << 6];
  double qc[QN_M2];
#pragma unroll
  for (int j = 0; j < QN_M2; ++j) qc[j] = q[(R.small() + QN_Q + j) << 6];
  int64_t r0, r1;
  qn_row_block(a.Nz, (int)(blockIdx.x % QN_NB), r0, r1);
#pragma unroll 2
  for (int64_t row = r0; row < r1; ++row) {
    const double lo = uload(a.lo, row), hi = uload(a.hi, row);
    const double rp0 = q[(R.rp0() + row) << 6], dzv = *soa(a.dz, g, a.Nz, row);
    double uq = 0.0;
#pragma unroll
    for (int j = 0; j < QN_M2; ++j) uq += qc[j] * qn_u(q, R, j, row, sigma);
    double sig = sigma + dw, bt = 0.0;
    if (lo != hi && a.zl) {
      const double p = *soa(a.z, g, a.Nz, row);
      if (finite_lo(lo)) { sig += *soa(a.zl, g, a.Nz, row) / (p - lo); bt -= mu / (p - lo); }
      if (finite_hi(hi)) { sig += *soa(a.zu, g, a.Nz, row) / (hi - p); bt += mu / (hi - p); }
    }
    const bool fx = lo == hi;
    q[(R.gl() + row) << 6] = fx ? 0.0 : rp0 + al * (-(rp0 + bt) - sig * dzv + uq);
    q[(R.qs() + row) << 6] = fx ? 0.0 : al * dzv;
  }
}

// ------------------------------------------------------------------------------------------------
// iterative refinement of the KKT step (dto_options.kkt_refinement = number of passes; round 6).
//
// Measured background (tools/step_truth.py, profiles/r06/step_truth_*: the step of the acrobot T = 1000 bench state against the
// solution of the oracle's K in extended precision): the sequential sweeps are within 7e-10 of it, the time-partitioned form
// within 1e-8 .. 2.5e-8 and, at delta_w = 0, 5e-6 -- the systems themselves are well conditioned (float64 sparse LU: 1e-14; half
// an ulp of noise in the data moves the solution by 1e-15).  The loss is the partition's own: a chunk whose head state is the
// separator eliminates lambda_a against u_a alone, three of its four pivots are -delta_c = -1e-8, and the separator's diagonal
// block R_LL and right-hand side r_L collect +1e8 and -1e8 terms that cancel to O(1) -- eight digits.
//
// One pass: r = b - K v from the SAME assembly code the sweeps factorise (stage_factor<..., ASSEMBLE_ONLY> with an empty carry
// gives the stage's rows of K and b), a solve K e = r with the sweeps at the accepted (delta_w, gamma) -- the stage records hold
// r for it, the host keeps a copy of the real ones and puts it back --, and v := v + e with everything the back substitution
// derives from the step recomputed from the sum.  (First built as ONE solve with b + r: the cancellation hits the right-hand
// side path as well as the matrix, i.e. its error is relative to |b + r| = |b| again: 9.9e-9 -> 1.2e-8 on the test state.  The
// correction solve's error is relative to |r| = 1e-8 |b|.)
//   k_kkt_refine       (tile, stage): v_t -> refv;  own rows  rec := -(y - S v - O x_{t+1});  q_{t+1} := O' v + YY x_{t+1} -> refq
//   k_kkt_refine_join  (tile, stage): x rows    rec += q_t;  stage 0: the tile's lanes ask for one more factorisation
//   k_kkt_refine_apply (tile, chunk): dz, dlam := refv + e; ds, step-length limits, merit derivative as stage_backward forms them
// ------------------------------------------------------------------------------------------------
template <class M>
__global__ __launch_bounds__(WAVE) void k_kkt_refine(dto_kkt_args a) {
  const int64_t g = blockIdx.x / a.T;
  const int t = blockIdx.x % a.T;
  const double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  if (sc[SC_STATUS << 6] != 0.0) return;
  const double mu = sc[SC_MU << 6], dw = sc[SC_DELTA_W << 6], gam = sc[SC_GAMMA << 6];   // the accepted factorisation
  dispatch_uniform<M>(uload(a.kind, t), [&](auto kc) {
    constexpr int K = decltype(kc)::value;
    using D = KindDims<M, K>;
    constexpr int NP = D::NP, Q = D::Q, NY = D::NY, BD = D::BD;
    const SoaIO<M, K> io(a, g, t);
    Carry<M> cy;
    Spike<M> sp;
#pragma unroll
    for (int i = 0; i < M::MAX_NX * (M::MAX_NX + 1) / 2; ++i) cy.P[i] = 0.0;
#pragma unroll
    for (int i = 0; i < M::MAX_NX; ++i) cy.py[i] = 0.0;
    double S[BD * (BD + 1) / 2], y[BD], X[BD * (NY > 0 ? NY : 1)], YYl[NY > 0 ? NY * (NY + 1) / 2 : 1], Z[1], cxd[1], dinv[BD];
    bool ok = true;
    int nneg = 0;
    stage_factor<M, K, false, false, true>(a.opt, io, mu, dw, gam, false, cy, sp, S, y, X, YYl, Z, cxd, dinv, ok, nneg, nullptr);
    double v[BD], xn[NY > 0 ? NY : 1];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      v[i] = *soa(a.dz, g, a.Nz, io.z0 + i);
      *soa(a.refvz, g, a.Nz, io.z0 + i) = v[i];
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) {
      v[NP + j] = *soa(a.dlam, g, a.Nc, uload(a.ccoff, t) + j);
      *soa(a.refvl, g, a.Nc, uload(a.ccoff, t) + j) = v[NP + j];
    }
#pragma unroll
    for (int k = 0; k < NY; ++k) {
      v[NP + Q + k] = *soa(a.dlam, g, a.Nc, uload(a.cdoff, t) + k);
      *soa(a.refvl, g, a.Nc, uload(a.cdoff, t) + k) = v[NP + Q + k];
      xn[k] = *soa(a.dz, g, a.Nz, uload(a.zoff, t + 1) + k);
    }
    double* recw = const_cast<double*>(io.recp);
    StageBounds<NP> sb;
    if (!DTO_NEWTON(a.opt)) io.bounds(sb);
#pragma unroll
    for (int i = 0; i < BD; ++i) {
      double r = y[i];
#pragma unroll
      for (int k = 0; k < BD; ++k) r -= S[i >= k ? tri(i, k) : tri(k, i)] * v[k];
#pragma unroll
      for (int c = 0; c < NY; ++c) r -= X[i * NY + c] * xn[c];
      // The correction solve assembles its right-hand side from the record like any other: y = -(entry + what the assembly adds
      // for barrier terms and eliminated slacks).  The entry that makes y = r is therefore -r plus those terms (the same
      // expressions, in the order the assembly subtracts them).  A fixed variable's row is the identity with y = 0 whatever the
      // record says, and its step is 0.
      if (i < NP) {
        double e = -r;
        bool fx = false;
        if (!DTO_NEWTON(a.opt)) {
          const double lo = sb.lo[i], hi = sb.hi[i];
          fx = lo == hi;
          if (!fx) {
            if (finite_hi(hi)) e -= mu / (hi - sb.p[i]);
            if (finite_lo(lo)) e += mu / (sb.p[i] - lo);
          }
        }
        if (!fx) recw[pair_at(D::R_RP + i)] = e;
      } else if (i < NP + Q) {
        const int j = i - NP;
        double e = -r;
        if (!DTO_NEWTON(a.opt) && D::ineq(j)) {
          const double sv = io.slack(j), zv = io.slack_mult(j);
          e += (sv / zv) * (io.nu(j) - mu / sv);
        }
        recw[pair_at(D::R_C + j)] = e;
      } else {
        recw[pair_at(D::R_D + (i - NP - Q))] = -r;
      }
    }
    if constexpr (NY > 0) {
      double* q = a.refq + (((g * a.T + (t + 1)) * M::MAX_NX) << 6) + threadIdx.x;
#pragma unroll
      for (int c = 0; c < NY; ++c) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < BD; ++i) acc += X[i * NY + c] * v[i];
#pragma unroll
        for (int e = 0; e < NY; ++e) acc += YYl[c >= e ? tri(c, e) : tri(e, c)] * xn[e];
        q[(int64_t)c << 6] = acc;
      }
    }
  });
}

template <class M>
__global__ __launch_bounds__(WAVE) void k_kkt_refine_join(dto_kkt_args a) {
  const int64_t g = blockIdx.x / a.T;
  const int t = blockIdx.x % a.T;
  double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  if (sc[SC_STATUS << 6] != 0.0) return;
  dispatch_uniform<M>(uload(a.kind, t), [&](auto kc) {
    constexpr int K = decltype(kc)::value;
    using D = KindDims<M, K>;
    using KD = typename D::KD;
    if constexpr (KD::PREV >= 0) {
      const SoaIO<M, K> io(a, g, t);
      double* recw = const_cast<double*>(io.recp);
      const double* q = a.refq + (((g * a.T + t) * M::MAX_NX) << 6) + threadIdx.x;
#pragma unroll
      for (int i = 0; i < D::NX; ++i) {
        const bool fx = !DTO_NEWTON(a.opt) && uload(a.lo, io.z0 + i) == uload(a.hi, io.z0 + i);
        // r_i -= q_i  <=>  entry += q_i
        if (!fx) recw[pair_at(D::R_RP + i)] += q[(int64_t)i << 6];
      }
    }
  });
  if (t != 0) return;
  sc[SC_NEED << 6] = 1.0;
  sc[SC_ATTEMPT << 6] = (double)a.opt.max_refactor;   // this attempt is the one that gets used, whatever its inertia
  sc[SC_TRY_DW << 6] = sc[SC_DELTA_W << 6];
  sc[SC_TRY_GAM << 6] = sc[SC_GAMMA << 6];
}

// v := saved step + correction, and what the back substitution derives from a step (the tail of stage_backward, expression for
// expression and in its order -- stages of a chunk from the last to the first, so that a zero correction reproduces its sums bit
// for bit): slack steps, fraction-to-the-boundary limits, directional derivative of the barrier objective.  The stage records are
// the REAL ones again when this runs (the host has put its copy back).  grid = G * P; k_kkt_post folds the chunk partials.
template <class M>
__global__ __launch_bounds__(WAVE) void k_kkt_refine_apply(dto_kkt_args a) {
  const int64_t g = blockIdx.x / a.P;
  const int p = blockIdx.x % a.P;
  const dto_solver_opts& o = a.opt;
  const double* sc = a.scal + ((g * SC_COUNT) << 6) + threadIdx.x;
  if (sc[SC_STATUS << 6] != 0.0) return;
  const double mu = sc[SC_MU << 6];
  const double tau = fmax(o.tau_min, 1.0 - mu);
  const int t0 = uload(a.cstart, p), t1 = uload(a.cstart, p + 1);
  StepAcc acc{1.0, 1.0, 0.0, 0.0};
  for (int t = t1 - 1; t >= t0; --t) {
    dispatch_uniform<M>(uload(a.kind, t), [&](auto kc) {
      constexpr int K = decltype(kc)::value;
      using D = KindDims<M, K>;
      constexpr int NP = D::NP, Q = D::Q, NY = D::NY;
      const SoaIO<M, K> io(a, g, t);
      StageBounds<NP> sb;
      if (!DTO_NEWTON(o)) io.bounds(sb);
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const double dp = *soa(a.refvz, g, a.Nz, io.z0 + i) + *soa(a.dz, g, a.Nz, io.z0 + i);
        io.put_dp(i, dp);
        acc.gphid += io.rec(D::R_RP + i) * dp;
        if (!DTO_NEWTON(o)) {
          const double lo = sb.lo[i], hi = sb.hi[i];
          if (lo != hi) {
            const double pv = sb.p[i];
            if (finite_lo(lo)) {
              const double zl = sb.zl[i];
              const double gap = pv - lo;
              const double dzl = mu / gap - zl - (zl / gap) * dp;
              if (dp < 0.0) acc.apmax = fmin(acc.apmax, -tau * gap / dp);
              if (dzl < 0.0) acc.admax = fmin(acc.admax, -tau * zl / dzl);
              acc.gphid -= mu / gap * dp;
            }
            if (finite_hi(hi)) {
              const double zu = sb.zu[i];
              const double gap = hi - pv;
              const double dzu = mu / gap - zu + (zu / gap) * dp;
              if (dp > 0.0) acc.apmax = fmin(acc.apmax, tau * gap / dp);
              if (dzu < 0.0) acc.admax = fmin(acc.admax, -tau * zu / dzu);
              acc.gphid += mu / gap * dp;
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < Q; ++j) {
        const int row = uload(a.ccoff, t) + j;
        const double dnu = *soa(a.refvl, g, a.Nc, row) + *soa(a.dlam, g, a.Nc, row);
        const double nu = io.nu(j);
        const double r = io.rec(D::R_C + j);
        io.put_dnu(j, dnu);
        double dsv = 0.0;
        if (!DTO_NEWTON(o) && D::ineq(j)) {
          const double sv = io.slack(j);
          const double zv = io.slack_mult(j);
          dsv = -(sv / zv) * (nu - mu / sv + dnu);
          const double dzs = mu / sv - zv - (zv / sv) * dsv;
          io.put_ds(j, dsv);
          if (dsv < 0.0) acc.apmax = fmin(acc.apmax, -tau * sv / dsv);
          if (dzs < 0.0) acc.admax = fmin(acc.admax, -tau * zv / dzs);
          acc.gphid -= mu / sv * dsv;
        }
        acc.gphid += nu * (r - o.delta_c * dnu + dsv);
        acc.rlam += r * (nu + dnu);
      }
#pragma unroll
      for (int k = 0; k < NY; ++k) {
        const int row = uload(a.cdoff, t) + k;
        const double dl = *soa(a.refvl, g, a.Nc, row) + *soa(a.dlam, g, a.Nc, row);
        const double lam = io.lam(k);
        const double r = io.rec(D::R_D + k);
        io.put_dlam(k, dl);
        acc.gphid += lam * (r - o.delta_c * dl);
        acc.rlam += r * (lam + dl);
      }
    });
  }
  double* ca = a.cacc + (((g * a.P + p) * 4) << 6) + threadIdx.x;
  st_join(&ca[0 << 6], acc.apmax);
  st_join(&ca[1 << 6], acc.admax);
  st_join(&ca[2 << 6], acc.gphid);
  st_join(&ca[3 << 6], acc.rlam);
}

// ------------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------------
template <class M>
int launch_kkt(int op, const dto_kkt_args* args, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  const dto_kkt_args& a = *args;
  const unsigned gt = (unsigned)((int64_t)a.G * a.T);
  const unsigned gb = (unsigned)((int64_t)a.G * ((a.T + a.sb - 1) / a.sb));  // blocks of DTO_SB stages
  {
    switch (op) {
      case DTO_KKT_PACK:
      case DTO_KKT_UNPACK: {
        int64_t n = 0;
        double* buf = nullptr;
        switch (a.aos_which) {
          case 0: n = a.Nz; buf = a.z; break;
          case 1: n = a.Nc; buf = a.lam; break;
          case 2: n = a.Nz; buf = a.dz; break;
          case 3: n = a.Nc; buf = a.dlam; break;
          case 4: n = a.Nw; buf = const_cast<double*>(a.wtile); break;
          case 5: n = a.Nz; buf = a.zl; break;
          case 6: n = a.Nz; buf = a.zu; break;
          case 7: n = a.Ni; buf = a.s; break;
          case 8: n = a.Ni; buf = a.zs; break;
          case 9: n = a.Ni; buf = a.ds; break;
          case 10: n = a.Nz; buf = const_cast<double*>(a.sigx); break;
          case 11: n = a.Nc; buf = const_cast<double*>(a.sigc); break;
          default: return -1;
        }
        if (n == 0 || !buf) break;   // vectors the problem does not have (no slacks, no bound multipliers)
        const unsigned nrb = (unsigned)((n + 3) / 4);
        if ((uint64_t)nrb * (uint64_t)a.G > 0x7fffffffull) return (int)hipErrorInvalidValue;
        const dim3 grid(nrb * (unsigned)a.G);
        if (op == DTO_KKT_PACK) hipLaunchKernelGGL(k_pack, grid, dim3(256), 0, st, a, n, buf, nrb);
        else hipLaunchKernelGGL(k_unpack, grid, dim3(256), 0, st, a, n, (const double*)buf, nrb);
        break;
      }
      case DTO_KKT_INIT: hipLaunchKernelGGL(k_init<M>, dim3(gt), dim3(WAVE), 0, st, a); break;
      case DTO_KKT_EVAL: hipLaunchKernelGGL(k_stage_eval<M>, dim3(gb), dim3(WAVE), 0, st, a); break;
      case DTO_KKT_CONV:
        // slots 2..5 and 9 (theta_inf, dual inf, max s*z, max 1/(s*z), max |x|) are maxima, the others sums
        if (a.P > 1 && a.csync) {
          hipLaunchKernelGGL(k_part_reduce_conv, dim3((unsigned)((int64_t)a.G * a.P)), dim3(WAVE), 0, st, a, (const double*)a.part,
                             0x23Cu, a.n_mult, a.n_bnd);
          break;
        }
        hipLaunchKernelGGL(k_part_reduce<DTO_NPART>, dim3((unsigned)((int64_t)a.G * a.P)), dim3(WAVE), 0, st, a,
                           (const double*)a.part, 0x23Cu);
        hipLaunchKernelGGL(k_conv, dim3((unsigned)a.G), dim3(WAVE), 0, st, a, a.n_mult, a.n_bnd);
        break;
      case DTO_KKT_FACTOR_SOLVE: {
        const unsigned gp = (unsigned)((int64_t)a.G * a.P);
        const int rounds = a.opt.newton_only ? 1 : a.opt.max_refactor + 1;
        // separator system: in the last chunk wavefront of a tile (small batches), one wavefront per instance (cyclic reduction:
        // many chunks), or one wavefront per tile (lane-per-instance elimination)
        const bool sep_wide = a.P > 1 && a.sep_cr == 2 && ChunkSum<M>::N <= 6;
        if (a.P > 1 && a.csync && !sep_wide) {
          for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_kkt_fwd_sep<M>, dim3(gp), dim3(WAVE), 0, st, a);
          hipLaunchKernelGGL(k_kkt_bwd_post<M>, dim3(gp), dim3(WAVE), 0, st, a);
          break;
        }
        if (a.P > 1) {
          for (int r = 0; r < rounds; ++r) {
            hipLaunchKernelGGL(k_kkt_fwd<M>, dim3(gp), dim3(WAVE), 0, st, a);
            if (sep_wide) hipLaunchKernelGGL(k_kkt_sep_cr<M>, dim3((unsigned)a.G * 64u), dim3(WAVE), 0, st, a);
            else hipLaunchKernelGGL(k_kkt_sep<M>, dim3((unsigned)a.G), dim3(WAVE), 0, st, a);
          }
          if (a.csync) {
            hipLaunchKernelGGL(k_kkt_bwd_post<M>, dim3(gp), dim3(WAVE), 0, st, a);
            break;
          }
          hipLaunchKernelGGL(k_kkt_bwd<M>, dim3(gp), dim3(WAVE), 0, st, a);
        } else {
          // the sequential sweep loops over its rounds inside the launch, a.fwd_rounds at most per launch (0: all of them)
          const int per = a.fwd_rounds > 0 ? a.fwd_rounds : rounds;
#if DTO_FUSE_SWEEPS
          if (per == rounds) { hipLaunchKernelGGL(k_kkt_fwdbwd_seq<M>, dim3(gp), dim3(WAVE), 0, st, a); } else
#endif
          {
          for (int r = 0; r < rounds; r += per) hipLaunchKernelGGL(k_kkt_fwd_seq<M>, dim3(gp), dim3(WAVE), 0, st, a);
          hipLaunchKernelGGL(k_kkt_bwd_seq<M>, dim3(gp), dim3(WAVE), 0, st, a);
          }
        }
        hipLaunchKernelGGL(k_kkt_post, dim3((unsigned)a.G), dim3(WAVE), 0, st, a);
        break;
      }
      case DTO_KKT_FWD:
        if (a.P > 1) hipLaunchKernelGGL(k_kkt_fwd<M>, dim3((unsigned)((int64_t)a.G * a.P)), dim3(WAVE), 0, st, a);
        else hipLaunchKernelGGL(k_kkt_fwd_seq<M>, dim3((unsigned)((int64_t)a.G * a.P)), dim3(WAVE), 0, st, a);
        break;
      case DTO_KKT_SEP: hipLaunchKernelGGL(k_kkt_sep<M>, dim3((unsigned)a.G), dim3(WAVE), 0, st, a); break;
      case DTO_KKT_BWD:
        if (a.P > 1) hipLaunchKernelGGL(k_kkt_bwd<M>, dim3((unsigned)((int64_t)a.G * a.P)), dim3(WAVE), 0, st, a);
        else hipLaunchKernelGGL(k_kkt_bwd_seq<M>, dim3((unsigned)((int64_t)a.G * a.P)), dim3(WAVE), 0, st, a);
        break;
      case DTO_KKT_BWD_GATE:
        if (!a.fwd_started) return (int)hipErrorInvalidValue;
        hipLaunchKernelGGL(k_kkt_bwd_gate, dim3(1), dim3(WAVE), 0, st, a);
        break;
      case DTO_KKT_BWD_EARLY:
      case DTO_KKT_BWD_REST:
        if (a.P != 1 || !a.tile_fwd_tag || !a.tile_bwd_tag) return (int)hipErrorInvalidValue;
        if (op == DTO_KKT_BWD_EARLY) hipLaunchKernelGGL(k_kkt_bwd_early<M>, dim3((unsigned)a.G), dim3(WAVE), 0, st, a);
        else hipLaunchKernelGGL(k_kkt_bwd_rest<M>, dim3((unsigned)a.G), dim3(WAVE), 0, st, a);
        break;
      case DTO_KKT_POST: hipLaunchKernelGGL(k_kkt_post, dim3((unsigned)a.G), dim3(WAVE), 0, st, a); break;
      case DTO_KKT_LINESEARCH: hipLaunchKernelGGL(k_linesearch<M>, dim3(gb), dim3(WAVE), 0, st, a); break;
      case DTO_KKT_LS_REDUCE:
        if (a.P > 1 && a.csync) {
          hipLaunchKernelGGL(k_part_reduce_ls, dim3((unsigned)((int64_t)a.G * a.P)), dim3(WAVE), 0, st, a, (const double*)a.lspart);
          break;
        }
        hipLaunchKernelGGL(k_part_reduce<2 * DTO_LS_TRIALS>, dim3((unsigned)((int64_t)a.G * a.P)), dim3(WAVE), 0, st, a,
                           (const double*)a.lspart, 0u);
        hipLaunchKernelGGL(k_ls_reduce, dim3((unsigned)a.G), dim3(WAVE), 0, st, a);
        break;
      case DTO_KKT_UPDATE: hipLaunchKernelGGL(k_update<M>, dim3(gt), dim3(WAVE), 0, st, a); break;
      case DTO_KKT_UPDATE_EVAL:
        if (!a.z_next || !a.lam_next) return (int)hipErrorInvalidValue;
        hipLaunchKernelGGL(k_update_eval<M>, dim3(gb), dim3(WAVE), 0, st, a);
        break;
      case DTO_KKT_RHS: hipLaunchKernelGGL(k_rhs_record<M>, dim3(gt), dim3(WAVE), 0, st, a); break;
      case DTO_KKT_REARM: hipLaunchKernelGGL(k_rearm, dim3((unsigned)a.G), dim3(WAVE), 0, st, a); break;
      case DTO_KKT_QN_BEGIN:
        hipLaunchKernelGGL(k_qn_pair<M>, dim3((unsigned)a.G * QN_NB), dim3(WAVE), 0, st, a);
        hipLaunchKernelGGL(k_qn_hist, dim3((unsigned)a.G * QN_NB), dim3(WAVE), 0, st, a);
        hipLaunchKernelGGL(k_qn_fin, dim3((unsigned)a.G), dim3(WAVE), 0, st, a);
        break;
      case DTO_KKT_QN_RHS: hipLaunchKernelGGL(k_qn_rhs<M>, dim3(gt), dim3(WAVE), 0, st, a); break;
      case DTO_KKT_QN_COL: hipLaunchKernelGGL(k_qn_col, dim3((unsigned)a.G * QN_NB), dim3(WAVE), 0, st, a); break;
      case DTO_KKT_QN_SMALL:
        hipLaunchKernelGGL(k_qn_gram, dim3((unsigned)a.G * QN_NB), dim3(WAVE), 0, st, a);
        hipLaunchKernelGGL(k_qn_small, dim3((unsigned)a.G), dim3(WAVE), 0, st, a);
        break;
      case DTO_KKT_QN_SAVE: hipLaunchKernelGGL(k_qn_save, dim3((unsigned)a.G * QN_NB), dim3(WAVE), 0, st, a); break;
      case DTO_KKT_QN_COLS_RHS:
        if (!a.qn_main) return (int)hipErrorInvalidValue;
        hipLaunchKernelGGL(k_qn_cols_rhs<M>, dim3(gt), dim3(WAVE), 0, st, a);
        break;
      case DTO_KKT_REFINE:
        if (!a.refq || !a.refvz || !a.refvl) return (int)hipErrorInvalidValue;
        hipLaunchKernelGGL(k_kkt_refine<M>, dim3(gt), dim3(WAVE), 0, st, a);
        hipLaunchKernelGGL(k_kkt_refine_join<M>, dim3(gt), dim3(WAVE), 0, st, a);
        break;
      case DTO_KKT_REFINE_APPLY:
        if (!a.refvz || !a.refvl) return (int)hipErrorInvalidValue;
        hipLaunchKernelGGL(k_kkt_refine_apply<M>, dim3((unsigned)((int64_t)a.G * a.P)), dim3(WAVE), 0, st, a);
        hipLaunchKernelGGL(k_kkt_post, dim3((unsigned)a.G), dim3(WAVE), 0, st, a);
        break;
      default: return -1;
    }
    return (int)hipGetLastError();
  }
}

}  // namespace dto
