// Wide-stage KKT kernels: block-tridiagonal LDL^T with DENSE per-stage blocks on the f64 matrix cores.
//
// The lane-per-instance kernels of dto_kkt_kernels.hpp keep a whole stage block in registers, which stops
// working when state + action dimension reaches MFMA tile sizes (BASELINE.json configs[4]: acrobot embedded in
// n = 64 states, blocks of n + m + n = 129).  Here ONE WORKGROUP owns one
// problem instance and walks the horizon; the stage blocks live in LDS as 16 x 16 tiles and every O(n^3)
// operation is a v_mfma_f64_16x16x4_f64 tile product.
//
// System solved (the reference's in-tree KKT sketch, examples/pendulum/pendulum.jl:138-198):
//     [ H + dw I   J' ] [dz ]     [ grad f + J' mu ]
//     [ J      -dc I  ] [dmu] = - [ c              ]
// Stage t couples (x_t, u_t, lam_t) with y = x_{t+1}.  Elimination order inside a stage: u (scalar pivots),
// x (A = L_A D_A L_A'), lam (-(D + F~ D_A^-1 F~') = -L_M D_M L_M'); the Schur complement
//     P' = YY - V~' D_A^-1 V~ + E~' D_M^-1 E~          (F~ = F L_A^-T, V~ = L_A^-1 V, E~ = L_M^-1 (E - F~ D_A^-1 V~))
// is carried to stage t+1.  Factors go to HBM for the backward sweep (same kernel, same workgroup).
// Inertia = number of negative pivots (Sylvester), must equal the number of constraints.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "dto_math.hpp"
#include "dto_model_plugin.h"

enum dto_wide_op { DTO_WIDE_STEP = 0, DTO_WIDE_MERIT = 1 };

struct dto_wide_info {
  int supported;
  int n, nu;
  int64_t fac_stage;  // doubles of factor storage per stage and instance
  int lds_bytes;
};

struct dto_wide_args {
  int T;
  int64_t B;
  const int* kind;   // [T]
  const int* zoff;   // [T+1]
  const int* woff;   // [T+1]
  const int* cdoff;  // [T+1]
  const double* params; int64_t ldw;   // stage parameters w_t at params[b * ldw + woff[t]] (ldw = 0: one set shared by all instances)
  const double* z; int64_t ldz;
  const double* mu; int64_t ldmu;
  double delta_w, delta_c, piv_tol;
  double* dz; int64_t lddz;
  double* dmu; int64_t lddmu;
  double* fac;   // [B][T][fac_stage]
  int* flags;    // [B]: 1 = inertia (n, m, 0) and no tiny pivot
  int64_t Nc;
  long long* prof;  // optional [32] cycle counters of workgroup 0 (tools/wide_profile.py), else NULL
  // ---- solver use (dto_solve_batch on wide models); all NULL / 0 for the plain dto_kkt_step_batch
  const double* fixed_lo; const double* fixed_hi;  // [Nz] variable bounds: components with lo == hi get identity rows
  const double* dw_inst;   // [B] per-instance delta_w (overrides delta_w)
  const double* gam_inst;  // [B] 1: exact Hessian of the Lagrangian, 0: Gauss-Newton (constraint curvature lam' d'' dropped)
  const int* active;       // [B] 0 = skip this instance
  double* stats;           // [B][DTO_WIDE_NSTAT]: f, theta_1, theta_inf, dual infeasibility, grad f' dz, sum |lam|
  double* merit;           // [B][2 * DTO_WIDE_TRIALS]: (phi, theta_1) at z + alpha_pmax 2^-k dz, k = 0..TRIALS-1 (DTO_WIDE_MERIT)
  // ---- finite variable bounds (round 4; NULL: variables are free or fixed): primal-dual barrier terms of the components
  //      with lo < hi, one of them finite -- Sigma = z_L / (x - lo) + z_U / (hi - x) on the diagonal, mu / (x - lo) - mu / (hi - x)
  //      in the right-hand side (the bound multipliers are eliminated, as in the lane-per-instance path: dto_kkt_kernels.hpp)
  const double* zl; const double* zu;   // [B][Nz] bound multipliers
  const double* mu_inst;                // [B] barrier parameter
  double tau_min;                       // fraction-to-the-boundary parameter floor (0.99)
};
// stats: 0 f (barrier terms excluded), 1 theta_1, 2 theta_inf, 3 dual infeasibility, 4 grad phi' dz, 5 sum |lam|, 6/7 scratch,
// 8 alpha_pmax, 9 alpha_dmax, 10 max s z, 11 max 1 / (s z), 12 sum z, 13 sum log s  (8..13 only with bounds)
enum { DTO_WIDE_F = 0, DTO_WIDE_TH1, DTO_WIDE_THINF, DTO_WIDE_DINF, DTO_WIDE_GPHID, DTO_WIDE_SUMLAM, DTO_WIDE_APMAX = 8, DTO_WIDE_ADMAX,
       DTO_WIDE_SZMAX, DTO_WIDE_ISZMAX, DTO_WIDE_SUMZ, DTO_WIDE_LOGBAR, DTO_WIDE_NSTAT = 16 };
#define DTO_WIDE_TRIALS 8

namespace dto {
namespace wide {

typedef double d4 __attribute__((ext_vector_type(4)));

#ifndef DTO_WIDE_LDL_RANK1
#define DTO_WIDE_LDL_RANK1 1   // 0: the blocked LDL^T with one-wavefront diagonal tiles (rounds 1-3), kept for A/B runs
#endif
#ifndef DTO_WIDE_PROFILE
#define DTO_WIDE_PROFILE 0     // 1: cycle stamps of workgroup 0 compiled in (tools/wide_profile.py builds its plugin with it)
#endif
// ---- debug shapes of the forward sweep (round 5; tests/test_wide_flagsets_gpu.py, tools/wide_debug.py)
#ifndef DTO_WIDE_DBG_SYNC
#define DTO_WIDE_DBG_SYNC 0    // 1: every lds_barrier() also drains the vector-memory counter (s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier)
#endif
#ifndef DTO_WIDE_DBG_POISON
#define DTO_WIDE_DBG_POISON 0  // 1: the whole LDS image of k_wide_step starts as signalling NaNs: a read of something never written shows
#endif
#ifndef DTO_WIDE_DBG_JITTER
#define DTO_WIDE_DBG_JITTER 0  // 1: every wavefront sleeps a pseudo-random time after each barrier: a hand-off that is not ordered fails
#endif
#ifndef DTO_WIDE_SPLIT_BWD
#define DTO_WIDE_SPLIT_BWD 1   // 1: the backward sweep is its own kernel (k_wide_bwd) that prefetches the next stage's factor record
#endif                         //    into registers while it works on the current one; 0: the tail of k_wide_step (rounds 1-4)
#ifndef DTO_WIDE_PACK_L
#define DTO_WIDE_PACK_L DTO_WIDE_SPLIT_BWD   // 1: the two triangular factors of a stage go to the record as their ten lower 16 x 16
#endif                                        //    tiles (2 560 instead of 4 160 doubles each); needs the split backward sweep
#ifndef DTO_WIDE_LDL_INLINE
#define DTO_WIDE_LDL_INLINE __attribute__((noinline))
#endif
constexpr int WG = 256;  // 4 wavefronts
constexpr int TB = 16;   // MFMA tile edge

template <int N, int NU = 1>
struct Dims {
  static constexpr int LD = N + 1;             // row stride of the LDS matrices (odd: column walks are conflict free)
  static constexpr int MAT = N * LD;           // doubles per matrix
  static constexpr int NT = N / TB;            // tiles per edge
  static constexpr int LI_LD = TB + 1;
  static constexpr int LI = NT * TB * LI_LD;   // inverses of the unit-lower diagonal tiles
  // factor record of one stage in HBM
  // triangular factors: lower tiles only (DTO_WIDE_PACK_L), tile-major [NT (NT + 1) / 2][TB][TB]
  static constexpr int LTILES = NT * (NT + 1) / 2, PKL = DTO_WIDE_PACK_L ? LTILES * TB * TB : MAT;
  static constexpr int F_LA = 0, F_FT = PKL, F_VT = PKL + MAT, F_LM = PKL + 2 * MAT, F_ET = 2 * PKL + 2 * MAT, F_VEC = 2 * PKL + 3 * MAT;
  // vectors: D_A^-1, D_M^-1, bx~, bd^, the NU action rows (A_xu, F_u, V_u after the elimination inside the action block), cost
  // gradient [N + 8], scalars of the action block (V_SC: 1 / pivot [NU], reduced right-hand side [NU], unit-lower factor [NU][NU])
  static constexpr int V_DA = 0, V_DM = N, V_BX = 2 * N, V_BD = 3 * N, V_AU = 4 * N, V_FU = V_AU + NU * N, V_VU = V_FU + NU * N,
                       V_GC = V_VU + NU * N, V_SC = V_GC + N + 8;
  static constexpr int FAC = F_VEC + V_SC + ((NU * (NU + 2) + 7) & ~7);
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ---- 16x16 tile <-> accumulator (C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 j)
__device__ __forceinline__ d4 tile_load(const double* Mx, int ld, int m0, int n0) {
  const int r = lane_id() & 15, q = lane_id() >> 4;
  d4 c;
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] = Mx[(m0 + q + 4 * j) * ld + n0 + r];
  return c;
}
__device__ __forceinline__ void tile_store(double* Mx, int ld, int m0, int n0, d4 c) {
  const int r = lane_id() & 15, q = lane_id() >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) Mx[(m0 + q + 4 * j) * ld + n0 + r] = c[j];
}

// C(m,n) += sum_k A[m0+m][k] * s[k] * B[n0+n][k]        (A, B row-major; "NT")
__device__ __forceinline__ d4 mm_nt(d4 c, const double* A, int lda, int m0, const double* Bm, int ldb, int n0, int k0,
                                    int k1, const double* s, double sgn) {
  const int r = lane_id() & 15, q = lane_id() >> 4;
  for (int k = k0; k < k1; k += 4) {
    double a = A[(m0 + r) * lda + k + q];
    if (s) a *= s[k + q];
    const double b = Bm[(n0 + r) * ldb + k + q];
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(sgn * a, b, c, 0, 0, 0);
  }
  return c;
}
// C(m,n) += sum_k A[m0+m][k] * s[k] * B[k][n0+n]          ("NN")
__device__ __forceinline__ d4 mm_nn(d4 c, const double* A, int lda, int m0, const double* Bm, int ldb, int n0, int k0,
                                    int k1, const double* s, double sgn, int ka0 = -1) {
  const int r = lane_id() & 15, q = lane_id() >> 4;
  // ka0: column offset of A for k = k0 (A may be a small tile whose columns start at 0)
  const int ka = (ka0 < 0) ? k0 : ka0;
  for (int k = k0; k < k1; k += 4) {
    double a = A[(m0 + r) * lda + (k - k0 + ka) + q];
    if (s) a *= s[k + q];
    const double b = Bm[(k + q) * ldb + n0 + r];
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(sgn * a, b, c, 0, 0, 0);
  }
  return c;
}
// C(m,n) += sum_k A[k][m0+m] * s[k] * B[k][n0+n]          ("TN")
__device__ __forceinline__ d4 mm_tn(d4 c, const double* A, int lda, int m0, const double* Bm, int ldb, int n0, int k0,
                                    int k1, const double* s, double sgn) {
  const int r = lane_id() & 15, q = lane_id() >> 4;
  for (int k = k0; k < k1; k += 4) {
    double a = A[(k + q) * lda + m0 + r];
    if (s) a *= s[k + q];
    const double b = Bm[(k + q) * ldb + n0 + r];
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(sgn * a, b, c, 0, 0, 0);
  }
  return c;
}

// ---- one tile-row (four 16x16 tiles) of a 64-deep product, k outermost: per k-step one A fragment feeds four
//      independent accumulators, so the matrix pipe always has an MFMA to issue while the next fragments load.
//      MODE 0: C += A s B'   (A[(m0+m)][k], B[(n)][k]);  1: C += A s B  (A[(m0+m)][k], B[k][n]);  2: C += A' s B  (A[k][m0+m], B[k][n])
template <int MODE, int N>
__device__ __forceinline__ void mm_row4(d4 (&c)[4], const double* A, int m0, const double* Bm, const double* s, double sgn) {
  constexpr int LD = Dims<N>::LD;
  const int r = lane_id() & 15, q = lane_id() >> 4;
  // operands of step k + 4 are requested before the four MFMAs of step k are issued (left to the compiler, the B operands of
  // MODE 1 / 2 were read one at a time, each behind its own lgkmcnt(0))
  const double* Ap = (MODE == 2) ? A + q * LD + m0 + r : A + (m0 + r) * LD + q;
  const double* Bp = (MODE == 0) ? Bm + r * LD + q : Bm + q * LD + r;
  constexpr int AS = (MODE == 2) ? LD : 1, BK = (MODE == 0) ? 1 : LD, BJ = (MODE == 0) ? TB * LD : TB;
  double an = Ap[0], sn = s[q], bn[4];
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) bn[jb] = Bp[jb * BJ];
#pragma unroll 4
  for (int k = 0; k < N; k += 4) {
    const double a = an * (sgn * sn);
    double b[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) b[jb] = bn[jb];
    if (k + 4 < N) {
      an = Ap[(k + 4) * AS]; sn = s[k + 4 + q];
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) bn[jb] = Bp[(k + 4) * BK + jb * BJ];
    }
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) c[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[jb], c[jb], 0, 0, 0);
  }
}

__device__ __forceinline__ double readlane_d(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

// ---- unblocked LDL^T of the 16x16 diagonal tile at (o, o) by one wavefront: lane i (< 16) holds row i of the
//      symmetric tile in registers; values travel between lanes by v_readlane (lane indices are literals after
//      unrolling), each broadcast value is consumed by the next instruction.  Step j, critical path: reciprocal of the
//      lane's own diagonal -> scaled pivot row s_c = a_jc / d_j (which is also L[c][j]) -> broadcast -> one FMA.
//      The inverse X of the unit-lower factor (lane c = column c) is built row by row inside the same loop: row j+1
//      of L is final after step j, and its work fills the latency gaps of the factorisation chain.
//      d/dinv get the pivots; cnt[0] += negative pivots, cnt[1] |= tiny pivot seen.
__device__ __forceinline__ void diag_tile(double* Mx, int ld, int o, double* d, double* dinv, double* LIk, int li_ld,
                                          double piv_tol, int* cnt, long long* prof = nullptr) {
  const int l = lane_id();
  const int i = l & 15;
  long long tq_ = prof ? clock64() : 0;
  double a[TB];
#pragma unroll
  for (int c = 0; c < TB; ++c) a[c] = (c <= i) ? Mx[(o + i) * ld + o + c] : Mx[(o + c) * ld + o + i];
  const double diag0 = fabs(Mx[(o + i) * ld + o + i]);
  if (prof && threadIdx.x == 0) { const long long t_ = clock64(); prof[24] += t_ - tq_; tq_ = t_; }
  int nneg = 0, tiny = 0;
  double dj_mine = 0.0, idj_mine = 0.0;
  double X[TB];
  X[0] = (l == 0) ? 1.0 : 0.0;
#pragma unroll
  for (int j = 0; j < TB; ++j) {
    const double pv = a[j];
    double rc = __builtin_amdgcn_rcp(pv);
    rc = fma(fma(-pv, rc, 1.0), rc, rc);
    rc = fma(fma(-pv, rc, 1.0), rc, rc);
    const double aij = a[j];
#pragma unroll
    for (int c = j + 1; c < TB; ++c) {
      const double sc = readlane_d(a[c] * rc, j);
      if (i > j) a[c] -= aij * sc;
    }
    const double dj = readlane_d(pv, j);
    const double d0 = readlane_d(diag0, j);
    const double rcj = readlane_d(rc, j);
    if (!(fabs(dj) > piv_tol * fmax(1.0, d0))) tiny = 1;
    if (dj < 0.0) ++nneg;
    if (l == j) { dj_mine = pv; idj_mine = rc; }
    if (i > j) a[j] = aij * rcj;
    // row j+1 of the inverse: X[j+1][c] = -sum_{k<=j} L[j+1][k] X[k][c]
    if (j + 1 < TB) {
      double sacc = 0.0;
#pragma unroll
      for (int k = 0; k <= j; ++k) sacc += readlane_d(a[k], j + 1) * X[k];
      X[j + 1] = (l == j + 1) ? 1.0 : (l < j + 1 ? -sacc : 0.0);
    }
  }
  if (prof && threadIdx.x == 0) { const long long t_ = clock64(); prof[25] += t_ - tq_; tq_ = t_; }
  if (l < TB) {
    d[o + l] = dj_mine;
    dinv[o + l] = idj_mine;
#pragma unroll
    for (int k = 0; k < TB - 1; ++k)
      if (k < l) Mx[(o + l) * ld + o + k] = a[k];
#pragma unroll
    for (int r = 0; r < TB; ++r) LIk[r * li_ld + l] = X[r];
  }
  if (l == 0) { cnt[0] += nneg; cnt[1] |= tiny; }
  if (prof && threadIdx.x == 0) { const long long t_ = clock64(); prof[26] += t_ - tq_; tq_ = t_; }
}

// barrier for phases that exchange data through LDS only: __syncthreads() also waits for every outstanding global STORE of the
// wavefront (the factor records streaming to HBM) -- a full memory round trip at each of the barriers that follow a copy
// INVARIANT: only LDS data is handed from thread to thread across this barrier.  No global write of k_wide_step is read
// by another thread of the same launch (the factor records are consumed by k_wide_bwd, the next launch); a hand-off through
// global memory needs __syncthreads() (+ a fence), as the in-kernel backward sweep (DTO_WIDE_SPLIT_BWD = 0) does.
#if DTO_WIDE_DBG_JITTER
__device__ __forceinline__ void dbg_jitter() {
  // xorshift of (wavefront, clock): 0..63 x 64 cycles of s_sleep, different in every wavefront at every barrier
  unsigned x = (unsigned)__builtin_readcyclecounter() * 2654435761u + (threadIdx.x >> 6) * 40503u;
  x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
  const unsigned n = __builtin_amdgcn_readfirstlane(x) & 63u;
  for (unsigned i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
}
#endif
__device__ __forceinline__ void lds_barrier() {
#if DTO_WIDE_DBG_SYNC
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
#if DTO_WIDE_DBG_JITTER
  dbg_jitter();
#endif
}

// Hand-off through LDS INSIDE one wavefront: lanes store their own entries, then wave-uniform code (the generated model bodies,
// the row products) loads ALL of them.  The hardware runs a wavefront's LDS instructions in order, but to the compiler a lane's
// load of what ANOTHER lane stored is not ordered after that store: it may hoist the load above a store it sees as conditional
// (k_wide_merit: `if (l < NU) pk[N + l] = ...` followed by the model code reading pk[N] -- in -fno-strict-aliasing builds the
// load went first and theta of every trial point was evaluated with the action of the previous one; the solves then ran into
// the iteration limit.  Second root cause of round 5).  A wavefront-scope fence orders the two for the
// compiler and emits no instruction.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---- blocked right-looking LDL^T of the N x N matrix in LDS (lower tiles), all WG threads.
//      On exit: strict lower part = L, d/dinv = pivots, LI = inverses of the unit-lower diagonal tiles.
template <int N>
__device__ __forceinline__ void ldl_blocked(double* Mx, double* d, double* dinv, double* LI, double piv_tol, int* cnt, long long* prof = nullptr) {
  using D = Dims<N>;
  constexpr int LD = D::LD, NT = D::NT;
  const int w = wave_id();
  for (int kb = 0; kb < NT; ++kb) {
    const int o = kb * TB;
    double* LIk = LI + kb * TB * D::LI_LD;
    long long t0_ = 0;
    if (prof && threadIdx.x == 0) t0_ = clock64();
    if (w == 0) diag_tile(Mx, LD, o, d, dinv, LIk, D::LI_LD, piv_tol, cnt, prof);
    if (prof && threadIdx.x == 0) { const long long t1_ = clock64(); prof[20] += t1_ - t0_; t0_ = t1_; }
    __syncthreads();
    if (prof && threadIdx.x == 0) { const long long t1_ = clock64(); prof[21] += t1_ - t0_; t0_ = t1_; }
    // panel: L(ib,kb) = A(ib,kb) * Linv' * D^-1
    for (int ib = kb + 1 + w; ib < NT; ib += 4) {
      d4 c = {0.0, 0.0, 0.0, 0.0};
      c = mm_nt(c, Mx + o, LD, ib * TB, LIk, D::LI_LD, 0, 0, TB, nullptr, 1.0);
      const double sc = dinv[o + (lane_id() & 15)];
#pragma unroll
      for (int j = 0; j < 4; ++j) c[j] *= sc;
      tile_store(Mx, LD, ib * TB, o, c);
    }
    __syncthreads();
    if (prof && threadIdx.x == 0) { const long long t1_ = clock64(); prof[22] += t1_ - t0_; t0_ = t1_; }
    // trailing update of the lower tiles: A(ib,jb) -= L(ib,kb) D_kb L(jb,kb)'
    int idx = 0;
    for (int ib = kb + 1; ib < NT; ++ib) {
      for (int jb = kb + 1; jb <= ib; ++jb, ++idx) {
        if ((idx & 3) != w) continue;
        d4 c = tile_load(Mx, LD, ib * TB, jb * TB);
        c = mm_nt(c, Mx + o, LD, ib * TB, Mx + o, LD, jb * TB, 0, TB, d + o, -1.0);
        tile_store(Mx, LD, ib * TB, jb * TB, c);
      }
    }
    __syncthreads();
    if (prof && threadIdx.x == 0) { const long long t1_ = clock64(); prof[23] += t1_ - t0_; t0_ = t1_; }
  }
}

// ---- right-looking LDL^T of the N x N matrix in LDS with the matrix held in REGISTERS (round 4).
//      The blocked form above spends its time in the eight 16 x 16 diagonal tiles of a stage, each factorised by ONE wavefront
//      through v_readlane broadcasts while the other three wait at a barrier (profiles/r04/wide_phase_cycles_*: 15.6 k cycles per
//      tile, 125 k of the 363 k cycles of a stage).  Here all 256 threads hold a 4 x 4 comb of the symmetric matrix
//      (rows ti + 16 a, columns tj + 16 b: 16 doubles per thread) for the whole factorisation; step k costs one exchange of
//      column k through a 64-double LDS buffer (ping-pong: one barrier per step), nine LDS reads and at most sixteen
//      multiply-adds per thread -- the chain is 64 short steps instead of 8 long tiles plus 24 barriers.  L (strict lower part),
//      the pivots and the inverses of the unit-lower diagonal tiles (for the MFMA triangular solves that follow) are written back
//      at the end; negative / tiny pivots are counted like in ldl_blocked (same test, same order of pivots).
// (out of line, LDS pointers by address space: the caller's kernel is already at 256 + 182 registers, inlined three times the
// comb pushed it into scratch)
typedef __attribute__((address_space(3))) double lds_double;
typedef __attribute__((address_space(3))) int lds_int;
template <int N>
__device__ DTO_WIDE_LDL_INLINE void ldl_rank1(lds_double* Mx, lds_double* d, lds_double* dinv, lds_double* LI, lds_double* colb,
                                                    lds_double* dg0, double piv_tol, lds_int* cnt, long long* prof = nullptr) {
  long long tq_ = (DTO_WIDE_PROFILE && prof) ? clock64() : 0;
#if DTO_WIDE_PROFILE
#define DTO_LDL_TICK(slot) do { if (prof && threadIdx.x == 0) { const long long t_ = clock64(); prof[slot] += t_ - tq_; tq_ = t_; } } while (0)
#else
#define DTO_LDL_TICK(slot) do { (void)tq_; } while (0)
#endif
  using D = Dims<N>;
  constexpr int LD = D::LD;
  static_assert(N == 64 && WG == 256, "thread comb: 16 x 16 threads, 4 x 4 elements each");
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  double r[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int b = 0; b < 4; ++b) r[a][b] = Mx[(ti + 16 * a) * LD + tj + 16 * b];
  }
  if (tid < N) dg0[tid] = fabs(Mx[tid * LD + tid]);
  DTO_LDL_TICK(20);
  int nneg = 0, tiny = 0;
  // four columns per exchange: the owners publish the raw columns k0 .. k0+3 (updated by all earlier blocks), every thread
  // factorises the 4 x 4 pivot block itself (redundantly: 10 values, four short reciprocal chains), forward-substitutes its four
  // rows and its four columns against it (y_p = L_ip d_p) and applies the rank-4 update to its comb: 16 barriers per
  // factorisation instead of 64.
  // (the comb is indexed with the block column kb: it must be a compile-time constant or the comb lands in scratch memory)
  auto block_col = [&](auto kbc) {
    constexpr int kb = decltype(kbc)::value;
#pragma unroll 1
    for (int kq = 0; kq < 4; ++kq) {
      const int kk = kq * 4, k0 = kb * 16 + kk;
      lds_double* cb = colb + (kq & 1) * 4 * N;
      if ((tj >> 2) == kq) {
        const int q = tj & 3;
#pragma unroll
        for (int a = kb; a < 4; ++a) cb[q * N + ti + 16 * a] = r[a][kb];
      }
      lds_barrier();
      // the thread's own rows / columns of the four published columns: requested before the (long, dependent) pivot-block
      // chain below so that their LDS latency hides behind it
      // (rows and columns above block column kb are final: nothing of them is read, substituted or updated any more)
      double cv[4][4], cu[4][4];
#pragma unroll
      for (int a = kb; a < 4; ++a) {
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) { cv[a][p2] = cb[p2 * N + ti + 16 * a]; cu[a][p2] = cb[p2 * N + tj + 16 * a]; }
      }
      // pivot block: P[q][p] = column k0+q at row k0+p (symmetric)
      double L4[4][4], dd[4], di[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int p2 = 0; p2 <= q; ++p2) L4[q][p2] = cb[p2 * N + k0 + q];   // A[k0+q][k0+p2] = column k0+p2 at row k0+q
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double pv = L4[j][j];
#pragma unroll
        for (int p2 = 0; p2 < j; ++p2) pv = fma(-L4[j][p2] * dd[p2], L4[j][p2], pv);
        double rc = __builtin_amdgcn_rcp(pv);
        rc = fma(fma(-pv, rc, 1.0), rc, rc);
        rc = fma(fma(-pv, rc, 1.0), rc, rc);
        dd[j] = pv; di[j] = rc;
        if (!(fabs(pv) > piv_tol * fmax(1.0, dg0[k0 + j]))) tiny = 1;
        if (pv < 0.0) ++nneg;
#pragma unroll
        for (int i = j + 1; i < 4; ++i) {
          double v = L4[i][j];
#pragma unroll
          for (int p2 = 0; p2 < j; ++p2) v = fma(-L4[i][p2] * dd[p2], L4[j][p2], v);
          L4[i][j] = v * rc;
        }
      }
      if (tid == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { d[k0 + j] = dd[j]; dinv[k0 + j] = di[j]; }
      }
      // y (rows) and u (columns): y_p = A[i][k0+p] - sum_{s<p} y_s L4[p][s]; columns that are not beyond this block get u = 0
      // (their entries are final or handled below), so the rank-4 update itself is unconditional
      double yr[4][4], lr[4][4], yc[4][4];
#pragma unroll
      for (int a = kb; a < 4; ++a) {
        const bool later = tj + 16 * a > k0 + 3;
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) {
          double v = cv[a][p2], u = cu[a][p2];
#pragma unroll
          for (int s2 = 0; s2 < p2; ++s2) { v = fma(-yr[a][s2], L4[p2][s2], v); u = fma(-yc[a][s2], L4[p2][s2], u); }
          yr[a][p2] = v; yc[a][p2] = u;
        }
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) { lr[a][p2] = yr[a][p2] * di[p2]; yc[a][p2] = later ? yc[a][p2] : 0.0; }
      }
      // only the lower blocks of the comb (a >= b) are ever written back
#pragma unroll
      for (int b = kb; b < 4; ++b) {
#pragma unroll
        for (int a = b; a < 4; ++a) {
          double upd = r[a][b];
#pragma unroll
          for (int p2 = 0; p2 < 4; ++p2) upd = fma(-lr[a][p2], yc[b][p2], upd);
          r[a][b] = upd;
        }
      }
      // the four columns of this block keep y (their L D): comb block column kb, threads tj in [kk, kk + 3]
      if ((tj >> 2) == kq) {
#pragma unroll
        for (int a = kb; a < 4; ++a) {
          // (selects kept apart by empty asm: as a chain of ifs the compiler turns them into yr[a][tj & 3], i.e. the array goes to
          // scratch memory and every step pays scratch round trips)
          double own = yr[a][0];
          own = ((tj & 3) == 1) ? yr[a][1] : own; asm("" : "+v"(own));
          own = ((tj & 3) == 2) ? yr[a][2] : own; asm("" : "+v"(own));
          own = ((tj & 3) == 3) ? yr[a][3] : own; asm("" : "+v"(own));
          r[a][kb] = own;
        }
      }
    }
  };
  block_col(std::integral_constant<int, 0>{});
  block_col(std::integral_constant<int, 1>{});
  block_col(std::integral_constant<int, 2>{});
  block_col(std::integral_constant<int, 3>{});
  lds_barrier();
  DTO_LDL_TICK(21);
  // L[i][j] = A_j[i][j] / d_j for i > j (column j of the thread's comb stopped changing after step j)
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = ti + 16 * a, j = tj + 16 * b;
      // (diagonal and upper part zeroed: the single-right-hand-side solves then need no masks, trsv_lower / trsv_lower_t)
      Mx[i * LD + j] = (i > j) ? r[a][b] * dinv[j] : 0.0;
    }
  }
  if (tid == 0) { cnt[0] += nneg; cnt[1] |= tiny; }
  lds_barrier();
  DTO_LDL_TICK(22);
  // inverses of the four unit-lower diagonal tiles, one wavefront each: lane c (< 16) builds column c of X = L_kk^-1 row by row
  {
    const int w = wave_id(), l = lane_id(), o = w * TB;
    lds_double* LIk = LI + w * TB * D::LI_LD;
    double X[TB];
    X[0] = (l == 0) ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j + 1 < TB; ++j) {
      double sacc = 0.0;
#pragma unroll
      for (int k = 0; k <= j; ++k) sacc += Mx[(o + j + 1) * LD + o + k] * X[k];
      X[j + 1] = (l == j + 1) ? 1.0 : (l < j + 1 ? -sacc : 0.0);
      __builtin_amdgcn_sched_barrier(0);   // one row at a time: without it all 136 LDS reads are hoisted to the top (272 registers)
    }
    if (l < TB) {
#pragma unroll
      for (int rr = 0; rr < TB; ++rr) LIk[rr * D::LI_LD + l] = X[rr];
    }
  }
  lds_barrier();
  DTO_LDL_TICK(23);
#undef DTO_LDL_TICK
}

// chain of KS MFMA k-steps on one accumulator with the operands of the next four steps requested while the current four are
// issued (fa(step), fb(step) read LDS); KS is a compile-time constant: everything unrolls
template <int KS, class FA, class FB>
__device__ __forceinline__ d4 mm_steps(d4 c, FA fa, FB fb) {
  static_assert(KS % 4 == 0 && KS >= 4, "chunks of four k-steps");
  double an[4], bn[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { an[i] = fa(i); bn[i] = fb(i); }
#pragma unroll
  for (int c0 = 0; c0 < KS; c0 += 4) {
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = an[i]; b[i] = bn[i]; }
    if (c0 + 4 < KS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { an[i] = fa(c0 + 4 + i); bn[i] = fb(c0 + 4 + i); }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], c, 0, 0, 0);
  }
  return c;
}

// X <- X * L^-T for row-tile `ib` of X (one wavefront; tiles of one row depend only on each other)
template <int N>
__device__ __forceinline__ void trsm_right_rowtile(double* X, const double* Lm, const double* LI, int ib) {
  using D = Dims<N>;
  constexpr int LD = D::LD, LL = D::LI_LD;
  const int r = lane_id() & 15, q = lane_id() >> 4;
  const double* xr = X + (ib * TB + r) * LD + q;     // A operand: X[ib rows][k]
  auto tile = [&](auto jbc) {
    constexpr int jb = decltype(jbc)::value;
    d4 c = tile_load(X, LD, ib * TB, jb * TB);
    if constexpr (jb > 0) {
      const double* lr = Lm + (jb * TB + r) * LD + q;  // B operand ("NT"): L[jb rows][k]
      c = mm_steps<4 * jb>(c, [&](int st) { return -xr[4 * st]; }, [&](int st) { return lr[4 * st]; });
    }
    tile_store(X, LD, ib * TB, jb * TB, c);
    const double* li = LI + (jb * TB + r) * LL + q;
    d4 c2 = {0.0, 0.0, 0.0, 0.0};
    c2 = mm_steps<4>(c2, [&](int st) { return xr[jb * TB + 4 * st]; }, [&](int st) { return li[4 * st]; });
    tile_store(X, LD, ib * TB, jb * TB, c2);
  };
  tile(std::integral_constant<int, 0>{});
  tile(std::integral_constant<int, 1>{});
  tile(std::integral_constant<int, 2>{});
  tile(std::integral_constant<int, 3>{});
}

// X <- L^-1 X for column-tile `jb` of X (one wavefront)
template <int N>
__device__ __forceinline__ void trsm_left_coltile(double* X, const double* Lm, const double* LI, int jb) {
  using D = Dims<N>;
  constexpr int LD = D::LD, LL = D::LI_LD;
  const int r = lane_id() & 15, q = lane_id() >> 4;
  const double* xc = X + q * LD + jb * TB + r;       // B operand ("NN"): X[k][jb columns]
  auto tile = [&](auto ibc) {
    constexpr int ib = decltype(ibc)::value;
    d4 c = tile_load(X, LD, ib * TB, jb * TB);
    if constexpr (ib > 0) {
      const double* lr = Lm + (ib * TB + r) * LD + q;  // A operand: L[ib rows][k]
      c = mm_steps<4 * ib>(c, [&](int st) { return -lr[4 * st]; }, [&](int st) { return xc[4 * st * LD]; });
    }
    tile_store(X, LD, ib * TB, jb * TB, c);
    const double* li = LI + (ib * TB + r) * LL + q;
    d4 c2 = {0.0, 0.0, 0.0, 0.0};
    c2 = mm_steps<4>(c2, [&](int st) { return li[4 * st]; }, [&](int st) { return xc[(ib * TB + 4 * st) * LD]; });
    tile_store(X, LD, ib * TB, jb * TB, c2);
  };
  tile(std::integral_constant<int, 0>{});
  tile(std::integral_constant<int, 1>{});
  tile(std::integral_constant<int, 2>{});
  tile(std::integral_constant<int, 3>{});
}

// both triangular solves of phase 7 for one wavefront, tile by tile in turn: two independent chains in one instruction stream
// (the LDS round trips between the tiles of one hide behind the MFMAs of the other)
template <int N>
__device__ __forceinline__ void trsm_pair(double* __restrict__ XF, double* __restrict__ XV, const double* __restrict__ Lm,
                                          const double* __restrict__ LI, int wv) {
  using D = Dims<N>;
  constexpr int LD = D::LD, LL = D::LI_LD;
  const int r = lane_id() & 15, q = lane_id() >> 4;
  const double* xr = XF + (wv * TB + r) * LD + q;
  const double* xc = XV + q * LD + wv * TB + r;
  auto tile = [&](auto tc) {
    constexpr int t = decltype(tc)::value;
    d4 cf = tile_load(XF, LD, wv * TB, t * TB);
    d4 cv = tile_load(XV, LD, t * TB, wv * TB);
    const double* lr = Lm + (t * TB + r) * LD + q;
    if constexpr (t > 0) {
      cf = mm_steps<4 * t>(cf, [&](int st) { return -xr[4 * st]; }, [&](int st) { return lr[4 * st]; });
      cv = mm_steps<4 * t>(cv, [&](int st) { return -lr[4 * st]; }, [&](int st) { return xc[4 * st * LD]; });
    }
    tile_store(XF, LD, wv * TB, t * TB, cf);
    tile_store(XV, LD, t * TB, wv * TB, cv);
    const double* li = LI + (t * TB + r) * LL + q;
    d4 c2 = {0.0, 0.0, 0.0, 0.0}, c3 = {0.0, 0.0, 0.0, 0.0};
    c2 = mm_steps<4>(c2, [&](int st) { return xr[t * TB + 4 * st]; }, [&](int st) { return li[4 * st]; });
    c3 = mm_steps<4>(c3, [&](int st) { return li[4 * st]; }, [&](int st) { return xc[(t * TB + 4 * st) * LD]; });
    tile_store(XF, LD, wv * TB, t * TB, c2);
    tile_store(XV, LD, t * TB, wv * TB, c3);
  };
  tile(std::integral_constant<int, 0>{});
  tile(std::integral_constant<int, 1>{});
  tile(std::integral_constant<int, 2>{});
  tile(std::integral_constant<int, 3>{});
}

// broadcast of one lane's double through the scalar registers (lane is a compile-time constant after unrolling): two
// v_readlane_b32 instead of the ds_bpermute pair __shfl costs
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// v <- L^-1 v (unit lower, strict lower part of Lm), one wavefront, lanes = rows.  Sixteen columns of the lane's row are loaded
// (ldl_rank1 leaves zeros on and above the diagonal; with the blocked factorisation they are masked to the strict lower part:
// 126 lane masks that the compiler hoists out of the stage loop and spills from the scalar registers) ahead of the sixteen dependent steps that use them: a step is two v_readlane and one fma
// (rounds 1-4: one LDS load, a ds_bpermute broadcast and a predicated fma per step, ~190 cycles each at one wavefront per SIMD).
template <int N>
__device__ __forceinline__ void trsv_lower(const double* Lm, double* v) {
  static_assert(N == 64, "one wavefront of rows");
  constexpr int LD = Dims<N>::LD;
  const int l = lane_id();
  double mine = v[l];
#pragma unroll
  for (int kb = 0; kb < N; kb += 16) {
    double col[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) col[j] = (DTO_WIDE_LDL_RANK1 || l > kb + j) ? Lm[l * LD + kb + j] : 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (kb + j < N - 1) mine = __builtin_fma(-col[j], readlane_f64(mine, kb + j), mine);
    }
  }
  v[l] = mine;
}
// v <- L^-T v, one wavefront
template <int N>
__device__ __forceinline__ void trsv_lower_t(const double* Lm, double* v) {
  static_assert(N == 64, "one wavefront of rows");
  constexpr int LD = Dims<N>::LD;
  const int l = lane_id();
  double mine = v[l];
#pragma unroll
  for (int kb = N - 16; kb >= 0; kb -= 16) {
    double row[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) row[j] = (DTO_WIDE_LDL_RANK1 || l < kb + j) ? Lm[(kb + j) * LD + l] : 0.0;
#pragma unroll
    for (int j = 15; j >= 0; --j) {
      if (kb + j >= 1) mine = __builtin_fma(-row[j], readlane_f64(mine, kb + j), mine);
    }
  }
  v[l] = mine;
}

// whole-matrix copies between LDS and the factor record in HBM, 16 bytes per lane (MAT is even and every matrix starts
// on a 16-byte boundary: MAT * 8 = 33 280 bytes)
template <int MAT>
__device__ __forceinline__ void copy_mat(double* dst, const double* src) {
  static_assert(MAT % 2 == 0, "matrix size must be even for 16-byte copies");
  // (left to the compiler's own pipelining: all nine pieces of a thread in registers at once made this kernel spill, three at a
  // time with the loop kept rolled was slower than this plain form -- 65 k against 38 k cycles for the four loads of a stage of
  // the backward sweep, profiles/r04/wide_phase_cycles.txt)
  const double2* s2 = reinterpret_cast<const double2*>(src);
  double2* d2 = reinterpret_cast<double2*>(dst);
  for (int i = threadIdx.x; i < MAT / 2; i += WG) d2[i] = s2[i];
}

// LDS -> factor record: three pieces per thread in flight (an LDS round trip per piece otherwise: ~130 cycles x 9 x 5 matrices)
template <int MAT>
__device__ __forceinline__ void store_fac(double* dst, const double* src) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  const v2d* s2 = reinterpret_cast<const v2d*>(src);
  v2d* d2 = reinterpret_cast<v2d*>(dst);
  constexpr int NP = (MAT / 2 + WG - 1) / WG;
  static_assert(NP % 3 == 0, "pieces per thread");
#pragma unroll 1
  for (int k = 0; k < NP; k += 3) {
    const int i0 = threadIdx.x + k * WG, i1 = i0 + WG, i2 = i1 + WG;
    const v2d a0 = s2[i0 < MAT / 2 ? i0 : 0], a1 = s2[i1 < MAT / 2 ? i1 : 0], a2 = s2[i2 < MAT / 2 ? i2 : 0];
    if (i0 < MAT / 2) d2[i0] = a0;
    if (i1 < MAT / 2) d2[i1] = a1;
    if (i2 < MAT / 2) d2[i2] = a2;
  }
}

// ---- the lower 16 x 16 tiles of a triangular factor (ten of sixteen), tile-major in the record: piece p of 1280 = (tile p >> 7,
//      row (p >> 3) & 15, column pair p & 7), five pieces per thread
template <int N>
__device__ __forceinline__ int ltile_elem(int p) {
  static_assert(N == 64, "four tiles per edge");
  const int tile = p >> 7, r = (p >> 3) & 15, c2 = p & 7;
  const int ib = tile >= 6 ? 3 : (tile >= 3 ? 2 : (tile >= 1 ? 1 : 0));
  const int jb = tile - ib * (ib + 1) / 2;
  return (ib * TB + r) * Dims<N>::LD + jb * TB + 2 * c2;
}
template <int N>
__device__ __forceinline__ void store_ltiles(double* dst, const double* Mx) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  v2d* d2 = reinterpret_cast<v2d*>(dst);
  constexpr int NPL = Dims<N>::LTILES * TB * TB / 2 / WG;
  static_assert(NPL * WG * 2 == Dims<N>::LTILES * TB * TB, "whole pieces per thread");
  v2d v[NPL];
#pragma unroll
  for (int k = 0; k < NPL; ++k) { const int e = ltile_elem<N>(threadIdx.x + k * WG); v[k] = v2d{Mx[e], Mx[e + 1]}; }
#pragma unroll
  for (int k = 0; k < NPL; ++k) d2[threadIdx.x + k * WG] = v[k];
}

// dot products over N terms, fully unrolled with four independent accumulators (LDS loads all in flight)
template <int N>
__device__ __forceinline__ double dot_rr(const double* a, const double* v) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
  for (int c = 0; c + 3 < N; c += 4) {
    s0 += a[c] * v[c]; s1 += a[c + 1] * v[c + 1]; s2 += a[c + 2] * v[c + 2]; s3 += a[c + 3] * v[c + 3];
  }
#pragma unroll
  for (int c = N & ~3; c < N; ++c) s0 += a[c] * v[c];   // (evaluator plugins of 17 .. 63 states: any N)
  return (s0 + s1) + (s2 + s3);
}
template <int N>
__device__ __forceinline__ double dot_rrs(const double* a, const double* v, const double* sc) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
  for (int c = 0; c < N; c += 4) {
    s0 += a[c] * v[c] * sc[c]; s1 += a[c + 1] * v[c + 1] * sc[c + 1];
    s2 += a[c + 2] * v[c + 2] * sc[c + 2]; s3 += a[c + 3] * v[c + 3] * sc[c + 3];
  }
  return (s0 + s1) + (s2 + s3);
}
// column of a row-major matrix (stride ld) against a vector
template <int N>
__device__ __forceinline__ double dot_cr(const double* a, int ld, const double* v) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
  for (int c = 0; c < N; c += 4) {
    s0 += a[c * ld] * v[c]; s1 += a[(c + 1) * ld] * v[c + 1]; s2 += a[(c + 2) * ld] * v[c + 2]; s3 += a[(c + 3) * ld] * v[c + 3];
  }
  return (s0 + s1) + (s2 + s3);
}
template <int N>
__device__ __forceinline__ double dot_crs(const double* a, int ld, const double* v, const double* sc) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
  for (int c = 0; c < N; c += 4) {
    s0 += a[c * ld] * v[c] * sc[c]; s1 += a[(c + 1) * ld] * v[c + 1] * sc[c + 1];
    s2 += a[(c + 2) * ld] * v[c + 2] * sc[c + 2]; s3 += a[(c + 3) * ld] * v[c + 3] * sc[c + 3];
  }
  return (s0 + s1) + (s2 + s3);
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft);
  return v;
}

// ---- 64-term dot products spread over the whole workgroup: thread = (row or column tid >> 2, quarter tid & 3), sixteen
//      interleaved terms per thread, the quarters summed across the four lanes by DPP.  (As `if (tid < N) dot_rr<N>(...)` a
//      product kept ONE wavefront busy for ~5 k cycles -- 64 dependent-latency LDS round trips -- while three waited.)
__device__ __forceinline__ double quad_sum(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  double o = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true));
  v += o;
  lo = __double2loint(v); hi = __double2hiint(v);
  o = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true));
  return v + o;
}
#ifndef DTO_WIDE_SPARSE_U
#define DTO_WIDE_SPARSE_U 1   // 1: rank-one terms of the action elimination restricted to the states the actions couple to (phase 5)
#endif
#ifndef DTO_WIDE_RMW_UNROLL
#define DTO_WIDE_RMW_UNROLL 2   // element-wise passes over the LDS matrices (rank-one terms of the actions): iterations in flight
#endif
#ifndef DTO_WIDE_DOTQ
#define DTO_WIDE_DOTQ 15   // bit per site of the forward sweep (phases 2, 4, 8, 11) that uses the workgroup-wide dot products
#endif
#ifndef DTO_DOTQ_UNROLL
#define DTO_DOTQ_UNROLL 2
#endif
// quarter of sum_c A[row][c] v[c] (row-major A, row = tid >> 2): call quad_sum on the result (or on a sum of several quarters)
template <int N>
__device__ __forceinline__ double dotq_r(const double* A, int ld, const double* v) {
  const int row = threadIdx.x >> 2, q = threadIdx.x & 3;
  const double* ar = A + row * ld + q;
  double s0 = 0.0, s1 = 0.0;
#pragma unroll DTO_DOTQ_UNROLL
  for (int j = 0; j < N / 4; j += 2) { s0 += ar[4 * j] * v[q + 4 * j]; s1 += ar[4 * j + 4] * v[q + 4 * j + 4]; }
  return s0 + s1;
}
template <int N>
__device__ __forceinline__ double dotq_rs(const double* A, int ld, const double* v, const double* sc) {
  const int row = threadIdx.x >> 2, q = threadIdx.x & 3;
  const double* ar = A + row * ld + q;
  double s0 = 0.0, s1 = 0.0;
#pragma unroll DTO_DOTQ_UNROLL
  for (int j = 0; j < N / 4; j += 2) {
    s0 += ar[4 * j] * v[q + 4 * j] * sc[q + 4 * j]; s1 += ar[4 * j + 4] * v[q + 4 * j + 4] * sc[q + 4 * j + 4];
  }
  return s0 + s1;
}
// quarter of sum_r A[r][col] v[r] (col = tid >> 2)
template <int N>
__device__ __forceinline__ double dotq_c(const double* A, int ld, const double* v) {
  const int col = threadIdx.x >> 2, q = threadIdx.x & 3;
  const double* ac = A + q * ld + col;
  double s0 = 0.0, s1 = 0.0;
#pragma unroll DTO_DOTQ_UNROLL
  for (int j = 0; j < N / 4; j += 2) { s0 += ac[4 * j * ld] * v[q + 4 * j]; s1 += ac[(4 * j + 4) * ld] * v[q + 4 * j + 4]; }
  return s0 + s1;
}
template <int N>
__device__ __forceinline__ double dotq_cs(const double* A, int ld, const double* v, const double* sc) {
  const int col = threadIdx.x >> 2, q = threadIdx.x & 3;
  const double* ac = A + q * ld + col;
  double s0 = 0.0, s1 = 0.0;
#pragma unroll DTO_DOTQ_UNROLL
  for (int j = 0; j < N / 4; j += 2) {
    s0 += ac[4 * j * ld] * v[q + 4 * j] * sc[q + 4 * j]; s1 += ac[(4 * j + 4) * ld] * v[q + 4 * j + 4] * sc[q + 4 * j + 4];
  }
  return s0 + s1;
}

// Cycle stamps of workgroup 0 (tools/wide_profile.py): compiled in only with -DDTO_WIDE_PROFILE=1.  A stamp is a never-taken
// DIVERGENT branch (threadIdx.x == 0) around memory instructions inside the stage loop -- the shape that made this compiler emit
// wrong code at -O3 in the lane-per-instance sweeps; the product's stage loop carries none.
#if DTO_WIDE_PROFILE
#define DTO_WIDE_TICK(slot)                                                       \
  do {                                                                             \
    if (a.prof && blockIdx.x == 0 && threadIdx.x == 0) {                           \
      const long long now_ = clock64();                                            \
      a.prof[slot] += now_ - tick_;                                                \
      tick_ = now_;                                                                \
    }                                                                              \
  } while (0)
#else
#define DTO_WIDE_TICK(slot) do { (void)tick_; } while (0)
#endif

// barrier terms of one variable with bounds lo < hi (at least one finite): sig = z_L/(x-lo) + z_U/(hi-x),
// br = mu/(x-lo) - mu/(hi-x); dz_L = mu/(x-lo) - z_L - z_L/(x-lo) dx and likewise for the upper bound are formed from the same
// pieces after the back substitution (wide_bar_step)
struct WideBar { double sig, br, sz_max, isz_max, sum_z, logb, zdiff; };
__device__ __forceinline__ WideBar wide_bar(double x, double lo, double hi, double zl, double zu, double mu) {
  WideBar o{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (lo == hi) return o;
  if (lo > -1e300) {
    const double g = x - lo;
    o.sig += zl / g; o.br += mu / g; o.sz_max = fmax(o.sz_max, g * zl); o.isz_max = fmax(o.isz_max, 1.0 / (g * zl));
    o.sum_z += fabs(zl); o.logb += log(g); o.zdiff -= zl;
  }
  if (hi < 1e300) {
    const double g = hi - x;
    o.sig += zu / g; o.br -= mu / g; o.sz_max = fmax(o.sz_max, g * zu); o.isz_max = fmax(o.isz_max, 1.0 / (g * zu));
    o.sum_z += fabs(zu); o.logb += log(g); o.zdiff += zu;
  }
  return o;
}
// step-length bounds of one variable given its primal step dx: ap <= tau (x-lo)/(-dx) ..., ad from z + ad dz >= (1 - tau) z
__device__ __forceinline__ void wide_bar_step(double x, double dx, double lo, double hi, double zl, double zu, double mu, double tau,
                                              double& ap, double& ad) {
  if (lo == hi) return;
  if (lo > -1e300) {
    const double g = x - lo, dzl = mu / g - zl - (zl / g) * dx;
    if (dx < 0.0) ap = fmin(ap, -tau * g / dx);
    if (dzl < 0.0) ad = fmin(ad, -tau * zl / dzl);
  }
  if (hi < 1e300) {
    const double g = hi - x, dzu = mu / g - zu + (zu / g) * dx;
    if (dx > 0.0) ap = fmin(ap, tau * g / dx);
    if (dzu < 0.0) ad = fmin(ad, -tau * zu / dzu);
  }
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) v = fmax(v, __shfl_xor(v, sft));
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) v = fmin(v, __shfl_xor(v, sft));
  return v;
}

// LDS of k_wide_step in doubles: four matrices, tile inverses, 15 + 3 NU vectors, gradient pad, model outputs, the action
// block, counters, statistics, fixed mask, column exchange (8 N), dg0, barrier terms
template <class M>
struct StepLds {
  static constexpr int N = M::WIDE_N, NU = M::WIDE_NU;
  static constexpr int SC = (3 * NU + NU * NU + 7) & ~7;
  static constexpr int DOUBLES = 4 * Dims<N>::MAT + Dims<N>::LI + (15 + 3 * NU) * N + 8 + M::MAX_NH + M::MAX_SNH + M::MAX_NJV + SC + 4 + 16
                                 + N + 8 * N + N + N + 8 * NU;
  static constexpr int BYTES = DOUBLES * (int)sizeof(double);
};

template <class M, int WKI>
struct WK {
  using KD = typename M::template WKind<WKI>;
};

// ---------------------------------------------------------------------------------------------------
// the kernel: forward factor/solve sweep, terminal solve, backward sweep.  grid = B, block = 256.
// ---------------------------------------------------------------------------------------------------
// BAR: instantiation with the barrier terms of finite variable bounds (launch_wide picks it when the caller passes bound
// multipliers); the plain KKT step carries none of that code
template <class M, bool BAR>
__global__ __launch_bounds__(WG) void k_wide_step(dto_wide_args a) {
  constexpr int N = M::WIDE_N, NU = M::WIDE_NU;
  static_assert(N == 64, "wide path is built for 64 states (one wavefront of rows, 4 x 4 tiles)");
  static_assert(NU >= 1 && NU <= 4, "wide path: one to four actions per stage");
  using D = Dims<N, NU>;
  using SL = StepLds<M>;
  static_assert(SL::BYTES <= 160 * 1024, "wide path: this model's stage data does not fit the 160 KB of LDS of one workgroup");
  constexpr int LD = D::LD, MAT = D::MAT, NT = D::NT;
  extern __shared__ double sm[];
  double* MA = sm;
  double* MF = MA + MAT;
  double* MV = MF + MAT;
  double* ME = MV + MAT;
  double* LI = ME + MAT;
  double* vec = LI + D::LI;
  double* xv = vec;            // [N]
  double* yv = xv + N;         // [N]
  double* lamv = yv + N;       // [N]
  double* au = lamv + N;       // A_xu   [NU][N]: row j = coupling of action j
  double* fu = au + NU * N;    // F_u    [NU][N]
  double* vu = fu + NU * N;    // V_u    [NU][N] (u-y coupling)
  double* bx = vu + NU * N;
  double* bd = bx + N;
  double* byc = bd + N;        // carried right-hand side for the next x
  double* byn = byc + N;
  double* gyp = byn + N;       // E_{t-1}' lam_{t-1}
  double* gyn = gyp + N;
  double* dA = gyn + N;
  double* dAi = dA + N;
  double* dM = dAi + N;
  double* dMi = dM + N;
  double* gc = dMi + N;        // cost gradient [N + NU]
  double* tmp = gc + N + 8;    // [N]
  double* hv = tmp + N;        // dynamics Hessian values [MAX_NH]
  double* chv = hv + M::MAX_NH;  // cost Hessian values [MAX_SNH]
  double* jvv = chv + M::MAX_SNH;  // variable Jacobian entries [MAX_NJV]
  double* sc = jvv + M::MAX_NJV;   // the action block: u [NU], W_uu [NU][NU] (LDL^T in place), right-hand side [NU], 1 / pivot [NU]
  double* auu = sc + NU;
  double* buv = auu + NU * NU;
  double* ipv = buv + NU;
  int* cnt = (int*)(sc + SL::SC);  // 0 nneg, 1 tiny

  const int tid = threadIdx.x, w = wave_id(), l = lane_id();
  const int64_t b = blockIdx.x;
  const double* z = a.z + b * a.ldz;
  const double* mu = a.mu + b * a.ldmu;
  double* facb = a.fac + b * (int64_t)a.T * D::FAC;
  if (a.active && !a.active[b]) return;
  const double dw = a.dw_inst ? a.dw_inst[b] : a.delta_w, dc = a.delta_c;
  const double gam = a.gam_inst ? a.gam_inst[b] : 1.0;
  double* stat = sc + SL::SC + 4;  // f, th1, thinf, dinf (LDS scalars)
  double* fxm = stat + 16;  // [N] 1.0 where x_t is fixed by equal bounds
  double* colb = fxm + N;   // [2][4][N] column exchange of ldl_rank1 (four columns per step, ping-pong)
  double* dg0 = colb + 8 * N;  // [N] |diagonal| before the factorisation (tiny-pivot test)
  double* nlf = dg0;           // nonlinear remainder of the residual, scattered to rows (phases 0-2 only: shares dg0, phases 6 and 9)
  double* brx = dg0 + N;       // [N + 8 NU] barrier part of the right-hand side of x, then 8 numbers per action; 0 without finite bounds
  if (tid < 16) stat[tid] = (tid == DTO_WIDE_APMAX || tid == DTO_WIDE_ADMAX) ? 1.0 : 0.0;
  const bool barrier = BAR && a.zl != nullptr;
  const double mub = (barrier && a.mu_inst) ? a.mu_inst[b] : 0.0;
  const double* zlb = barrier ? a.zl + b * a.ldz : nullptr;
  const double* zub = barrier ? a.zu + b * a.ldz : nullptr;
  const double taub = fmax(a.tau_min, 1.0 - mub);

  long long tick_ = clock64();
#if DTO_WIDE_DBG_POISON
  // (before anything else of this kernel writes LDS: the statistics block above is written again below)
  __syncthreads();
  for (int i = tid; i < SL::DOUBLES; i += WG) sm[i] = __longlong_as_double(0x7ff4dead0000beefLL);
  __syncthreads();
  if (tid < 16) stat[tid] = (tid == DTO_WIDE_APMAX || tid == DTO_WIDE_ADMAX) ? 1.0 : 0.0;
#endif
  for (int i = tid; i < MAT; i += WG) MA[i] = 0.0;
  if (tid < N) { byc[tid] = 0.0; gyp[tid] = 0.0; }
  if (tid == 0) { cnt[0] = 0; cnt[1] = 0; }
  __syncthreads();

  for (int t = 0; t < a.T - 1; ++t) {
    double* fac = facb + (int64_t)t * D::FAC;
    const int wk = M::wk_of_kind(a.kind[t]);
    M::dispatch_wk(wk, [&](auto wkc) {
      constexpr int WKI = decltype(wkc)::value;
      using KD = typename M::template WKind<WKI>;
      if constexpr (KD::DYN >= 0) {
        using DY = typename M::template Dyn<KD::DYN>;
        using CO = typename M::template Cost<KD::COST>;
        static_assert(DY::NX == N && DY::NY == N && DY::NU == NU, "uniform wide stages expected");
        const double* wp = a.params + b * a.ldw + a.woff[t];
        // ---- phase 0: the point, constant Jacobian part
        if (tid < N) {
          if (t == 0) {   // later stages: requested during phase 8 of the stage before, in LDS since its end
            xv[tid] = z[a.zoff[t] + tid];
            yv[tid] = z[a.zoff[t + 1] + tid];
            lamv[tid] = mu[a.cdoff[t] + tid];
            fxm[tid] = (a.fixed_lo && a.fixed_lo[a.zoff[t] + tid] == a.fixed_hi[a.zoff[t] + tid]) ? 1.0 : 0.0;
            const unsigned long long anyf = __ballot(fxm[tid] != 0.0);   // (tid < N is exactly wavefront 0)
            if (tid == 0) cnt[6] = anyf != 0ull;
          }
          if (t > 0 && tid == 0) cnt[6] = cnt[7];
#pragma unroll
          for (int j = 0; j < NU; ++j) { au[j * N + tid] = 0.0; vu[j * N + tid] = 0.0; }
          nlf[tid] = 0.0;
          double sig = 0.0;
          if (BAR) brx[tid] = 0.0;
          if (barrier) {
            const int gi = a.zoff[t] + tid;
            const WideBar wb = wide_bar(xv[tid], a.fixed_lo[gi], a.fixed_hi[gi], zlb[gi], zub[gi], mub);
            sig = wb.sig; brx[tid] = wb.br;
            if (a.stats) {   // tid < N is exactly wavefront 0: complementarity / barrier statistics of this knot's states
              const double m1 = wave_max(wb.sz_max), m2 = wave_max(wb.isz_max), s1 = wave_sum(wb.sum_z), s2 = wave_sum(wb.logb);
              if (tid == 0) {
                stat[DTO_WIDE_SZMAX] = fmax(stat[DTO_WIDE_SZMAX], m1); stat[DTO_WIDE_ISZMAX] = fmax(stat[DTO_WIDE_ISZMAX], m2);
                stat[DTO_WIDE_SUMZ] += s1; stat[DTO_WIDE_LOGBAR] += s2;
              }
            }
          }
          MA[tid * LD + tid] += dw + sig;
        }
        if (t == 0 && tid < NU) sc[tid] = z[a.zoff[t] + N + tid];
        if (tid < NU * NU) auu[tid] = 0.0;
        if (BAR && tid >= 64 && tid < 64 + NU) {   // the actions of this knot (lanes of wavefront 1): ubar = {br, sig, z_U - z_L, max s z, max 1/(s z), sum z, sum log s}
          double* ubar = brx + N + 8 * (tid - 64);
          WideBar wb{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
          if (barrier) {
            const int gi = a.zoff[t] + N + (tid - 64);
            wb = wide_bar(z[gi], a.fixed_lo[gi], a.fixed_hi[gi], zlb[gi], zub[gi], mub);
          }
          ubar[0] = wb.br; ubar[1] = wb.sig; ubar[2] = wb.zdiff; ubar[3] = wb.sz_max; ubar[4] = wb.isz_max; ubar[5] = wb.sum_z; ubar[6] = wb.logb;
        }
        {
          // (sixteen rows per wavefront.  Written as `for (r = w; r < N; r += 4)` the loop stayed rolled -- its trip count
          //  depends on w -- with a vmcnt(0) after every load: 21 k cycles per stage for 66 KB out of the L2.  Fully unrolled
          //  with eight or sixteen loads in flight it was faster still but pushed this kernel into 30-60 spilled registers,
          //  and the SOLVER's use of it -- statistics, fixed components -- then returned NaN steps while the plain KKT step
          //  stayed correct: not understood, not kept)
          constexpr int NC = 2 * N + NU;
          const double* fe = DY::fe_const();
#pragma unroll 1
          for (int part = 0; part < N / 16; ++part) {   // four rows of both matrices per pass: eight loads in flight
            const int r0 = w + 16 * part;
            double fr[4], er[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { fr[i] = fe[(r0 + 4 * i) * NC + l]; er[i] = fe[(r0 + 4 * i) * NC + N + NU + l]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) { MF[(r0 + 4 * i) * LD + l] = fr[i]; ME[(r0 + 4 * i) * LD + l] = er[i]; }
          }
          for (int i = tid; i < N * NU; i += WG) fu[i] = fe[(i % N) * NC + N + i / N];
          for (int i = tid; i < MAT; i += WG) MV[i] = 0.0;
        }
        lds_barrier();
        DTO_WIDE_TICK(0);
        // ---- phase 1: model code (wave-uniform values, one wavefront per function)
        if (w == 0) {
          DY::eval_nl(xv, sc, yv, wp, tmp);
          if (l < DY::NNL) nlf[DY::nl_row(l)] = tmp[l];
        } else if (w == 1) {
          DY::jac_var(xv, sc, yv, wp, jvv);
        } else if (w == 2) {
          if constexpr (DY::NH > 0) {
            DY::hess(xv, sc, yv, wp, lamv, hv);
            if (gam != 1.0 && l < DY::NH) hv[l] *= gam;
          }
        } else {
          CO::grad(xv, sc, wp, gc);
          if constexpr (CO::SNH > 0) CO::shess(xv, sc, wp, chv);
          if (a.stats) {
            CO::eval(xv, sc, wp, stat + 6);
            if (l == 0) stat[0] += stat[6];
          }
        }
        lds_barrier();
        DTO_WIDE_TICK(1);
        // ---- phase 2: residual from the constant part (variable Jacobian entries are still zero in MF/ME/fu)
#if DTO_WIDE_DOTQ & 1
        {
          const double part = quad_sum(dotq_r<N>(MF, LD, xv) + dotq_r<N>(ME, LD, yv));
          if ((tid & 3) == 0) {
            const int row = tid >> 2;
            double acc = nlf[row] + fu[row] * sc[0] + part;
#pragma unroll
            for (int j = 1; j < NU; ++j) acc += fu[j * N + row] * sc[j];
            bd[row] = -acc;
          }
        }
#else
        if (tid < N) {
          double acc = nlf[tid] + fu[tid] * sc[0] + dot_rr<N>(MF + tid * LD, xv) + dot_rr<N>(ME + tid * LD, yv);
#pragma unroll
          for (int j = 1; j < NU; ++j) acc += fu[j * N + tid] * sc[j];
          bd[tid] = -acc;
        }
#endif
        lds_barrier();
        DTO_WIDE_TICK(2);
        // ---- phase 3: variable Jacobian entries, Hessian blocks
        if (a.stats && w == 2) {
          const double sl_ = wave_sum(fabs(lamv[l]));
          if (l == 0) stat[5] += sl_;
        }
        if (a.stats && w == 3) {
          const double v = fabs(bd[l]);
          const double sm_ = wave_sum(v);
          double mx_ = v;
#pragma unroll
          for (int sft = 32; sft >= 1; sft >>= 1) mx_ = fmax(mx_, __shfl_xor(mx_, sft));
          if (l == 0) { stat[1] += sm_; stat[2] = fmax(stat[2], mx_); }
        }
        if (tid < DY::NJV) {
          const int r = DY::jv_row(tid), c = DY::jv_col(tid);
          const double v = jvv[tid];
          if (c < N) MF[r * LD + c] = v;
          else if (c < N + NU) fu[(c - N) * N + r] = v;
          else ME[r * LD + c - N - NU] = v;
        }
        if constexpr (CO::SNH > 0) {
          if (tid < CO::SNH) {
            const int r = CO::sh_row(tid), c = CO::sh_col(tid);
            const double v = chv[tid];
            if (r < N && c < N) MA[r * LD + c] += v;
            else if (r < N && c >= N) au[(c - N) * N + r] += v;
            else if (r >= N && c >= N) auu[(r - N) * NU + c - N] += v;
          }
        }
        lds_barrier();
        if constexpr (DY::NH > 0) {
          if (tid < DY::NH) {
            const int r = DY::h_row(tid), c = DY::h_col(tid);
            const double v = hv[tid];
            if (r < N) {
              if (c < N) MA[r * LD + c] += v;
              else if (c < N + NU) au[(c - N) * N + r] += v;
              else MV[r * LD + c - N - NU] += v;
            } else if (r < N + NU) {
              if (c >= N && c < N + NU) auu[(r - N) * NU + c - N] += v;
              else if (c >= N + NU) vu[(r - N) * N + c - N - NU] += v;
            }
          }
        }
        lds_barrier();
        DTO_WIDE_TICK(3);
        // ---- phase 4: gradient of the Lagrangian -> right-hand sides
#if DTO_WIDE_DOTQ & 2
        {
          const double pf_ = quad_sum(dotq_c<N>(MF, LD, lamv)), pe_ = quad_sum(dotq_c<N>(ME, LD, lamv));
          if ((tid & 3) == 0) {
            const int col = tid >> 2;
            bx[col] = -(gc[col] + gyp[col] + pf_) + byc[col] + (BAR ? brx[col] : 0.0);
            gyn[col] = pe_;
          }
        }
#else
        if (w == 0) {
          bx[l] = -(gc[l] + gyp[l] + dot_cr<N>(MF + l, LD, lamv)) + byc[l] + (BAR ? brx[l] : 0.0);
        } else if (w == 1) {
          gyn[l] = dot_cr<N>(ME + l, LD, lamv);
        }
#endif
        if (w == 2) {
#pragma unroll
          for (int j = 0; j < NU; ++j) {
            const double part = wave_sum(fu[j * N + l] * lamv[l]);
            if (l == 0) {
              const double* ubar = brx + N + 8 * j;
              buv[j] = -(gc[N + j] + part) + (BAR ? ubar[0] : 0.0);
              auu[j * NU + j] += dw + (BAR ? ubar[1] : 0.0);
              if (barrier && a.stats) {
                stat[DTO_WIDE_SZMAX] = fmax(stat[DTO_WIDE_SZMAX], ubar[3]); stat[DTO_WIDE_ISZMAX] = fmax(stat[DTO_WIDE_ISZMAX], ubar[4]);
                stat[DTO_WIDE_SUMZ] += ubar[5]; stat[DTO_WIDE_LOGBAR] += ubar[6];
              }
            }
          }
        }
        lds_barrier();
        DTO_WIDE_TICK(4);
        // ---- solver use: dual infeasibility of the free variables (grad L - z_L + z_U with bounds); variables fixed by equal
        //      bounds become identity rows
        if (a.stats && w == 3) {
          double zd = 0.0;
          if (barrier) {
            const int gi = a.zoff[t] + l;
            const double lo = a.fixed_lo[gi], hi = a.fixed_hi[gi];
            if (lo != hi) zd = (hi < 1e300 ? zub[gi] : 0.0) - (lo > -1e300 ? zlb[gi] : 0.0);
          }
          // bx = -grad L + byc + brx
          double v = (fxm[l] != 0.0) ? 0.0 : fabs(-(bx[l] - byc[l] - (BAR ? brx[l] : 0.0)) + zd);
          if (l < NU) v = fmax(v, fabs(-(buv[l] - (BAR ? brx[N + 8 * l] : 0.0)) + (BAR ? brx[N + 8 * l + 2] : 0.0)));
#pragma unroll
          for (int sft = 32; sft >= 1; sft >>= 1) v = fmax(v, __shfl_xor(v, sft));
          if (l == 0) stat[3] = fmax(stat[3], v);
        }
        // (the pass itself is skipped when no state of this knot is fixed -- it costs ~3 k cycles a stage -- but its two barriers
        //  stay: they also separate the statistics above, which read bx / buv on wavefront 3, from phase 5, where wavefront 0
        //  changes them)
        if (a.fixed_lo) {
          lds_barrier();
          if (cnt[6]) {
            for (int i = tid; i < N * N; i += WG) {
              const int r = i >> 6, c = i & 63;
              if (fxm[r] != 0.0 || fxm[c] != 0.0) MA[r * LD + c] = (r == c) ? 1.0 : 0.0;
              if (fxm[c] != 0.0) MF[r * LD + c] = 0.0;
              if (fxm[r] != 0.0) MV[r * LD + c] = 0.0;
            }
            if (tid < N && fxm[tid] != 0.0) {
#pragma unroll
              for (int j = 0; j < NU; ++j) au[j * N + tid] = 0.0;
              bx[tid] = 0.0;
            }
          }
          lds_barrier();
        }
        // ---- phase 5: eliminate u.  Several actions: W_uu = L_u D_u L_u' in place (one thread, NU <= 4), the coupling rows
        //      and the right-hand side go through L_u^-1, after which every action is a rank-one term of its own
        if constexpr (NU > 1) {
          if (tid == 0) {
#pragma unroll
            for (int j = 0; j < NU; ++j) {
              const double d = auu[j * NU + j], id = 1.0 / d;
              double tcol[NU];
#pragma unroll
              for (int k = j + 1; k < NU; ++k) tcol[k] = auu[k * NU + j];
#pragma unroll
              for (int k = j + 1; k < NU; ++k) {
#pragma unroll
                for (int k2 = j + 1; k2 <= k; ++k2) auu[k * NU + k2] -= tcol[k] * tcol[k2] * id;
                auu[k * NU + j] = tcol[k] * id;
                buv[k] -= tcol[k] * id * buv[j];
              }
              ipv[j] = id;
            }
          }
          lds_barrier();
          if (tid < 3 * N) {
            double* arr = tid < N ? au + tid : (tid < 2 * N ? fu + tid - N : vu + tid - 2 * N);
            double cj[NU];
#pragma unroll
            for (int j = 0; j < NU; ++j) cj[j] = arr[j * N];
#pragma unroll
            for (int j = 0; j < NU; ++j) {
#pragma unroll
              for (int k = j + 1; k < NU; ++k) cj[k] -= auu[k * NU + j] * cj[j];
            }
#pragma unroll
            for (int j = 1; j < NU; ++j) arr[j * N] = cj[j];
          }
          lds_barrier();
        }
        double ip[NU], bu[NU];
        if constexpr (NU == 1) { ip[0] = 1.0 / auu[0]; bu[0] = buv[0]; }
        else {
#pragma unroll
          for (int j = 0; j < NU; ++j) { ip[j] = ipv[j]; bu[j] = buv[j]; }
        }
        // A_xu and V_u are nonzero only at the states the actions couple to through second derivatives (KD::AU_N / VU_N of them,
        // listed by the generator: 1 + 1 of 64 + 64 for the acrobot embedding): the rank-one terms then touch AU_N (AU_N + VU_N + N)
        // + VU_N N entries instead of all 4 N^2 -- a few short loops instead of a read-modify-write pass over the four LDS matrices
        // (phase 5: 7.9 k -> 1.4 k cycles).  Written in round 4 and parked: with it the solver-mode use returned NaN steps at -O3 --
        // the exec-mask fault of the terminal stage ("root cause"), which this edit merely moved into view.
        constexpr bool SPARSE_U = DTO_WIDE_SPARSE_U && (KD::AU_N + KD::VU_N) <= 24;
        if constexpr (SPARSE_U) {
          constexpr int NA = KD::AU_N, NV = KD::VU_N;
          // (four loops, one per matrix: selecting the matrix per thread inside one loop makes the compiler address LDS through
          //  flat pointers)
          auto rank1 = [&](double* Mx, const double* ur, const double* vc, int r, int c) {
            double dsum = 0.0;
#pragma unroll
            for (int j = 0; j < NU; ++j) dsum += ur[j * N + r] * ip[j] * vc[j * N + c];
            Mx[r * LD + c] -= dsum;
          };
          if constexpr (NA > 0) {
            for (int i = tid; i < N * NA; i += WG) rank1(MF, fu, au, i / NA, KD::au_s(i % NA));
            for (int i = tid; i < NA * NA; i += WG) rank1(MA, au, au, KD::au_s(i / NA), KD::au_s(i % NA));
          }
          if constexpr (NV > 0) {
            for (int i = tid; i < N * NV; i += WG) rank1(ME, fu, vu, i / NV, KD::vu_s(i % NV));
          }
          if constexpr (NA > 0 && NV > 0) {
            for (int i = tid; i < NA * NV; i += WG) rank1(MV, au, vu, KD::au_s(i / NV), KD::vu_s(i % NV));
          }
        } else {
#pragma unroll DTO_WIDE_RMW_UNROLL
        for (int i = tid; i < N * N; i += WG) {
          const int r = i >> 6, c = i & 63;
          double da = 0.0, df = 0.0, dv = 0.0, de = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) {
            const double ar = au[j * N + r] * ip[j], fr = fu[j * N + r] * ip[j], ac = au[j * N + c], vc = vu[j * N + c];
            da += ar * ac; df += fr * ac; dv += ar * vc; de += fr * vc;
          }
          MA[r * LD + c] -= da;
          MF[r * LD + c] -= df;
          MV[r * LD + c] -= dv;
          ME[r * LD + c] -= de;
        }
        }
        if (tid < N) {
          double sx = 0.0, sd = 0.0, sy = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) {
            const double bj = bu[j] * ip[j];
            sx += au[j * N + tid] * bj; sd += fu[j * N + tid] * bj; sy += vu[j * N + tid] * bj;
          }
          bx[tid] -= sx;
          bd[tid] -= sd;
          byn[tid] = -sy;
        }
        if (tid == 0) {
#pragma unroll
          for (int j = 0; j < NU; ++j) {
            if (ip[j] < 0.0) cnt[0] += 1;
            if (!(fabs(1.0 / ip[j]) > a.piv_tol)) cnt[1] |= 1;
          }
        }
        lds_barrier();
        DTO_WIDE_TICK(5);
        // ---- phase 6: A = L_A D_A L_A'
        if (DTO_WIDE_LDL_RANK1) ldl_rank1<N>((lds_double*)MA, (lds_double*)dA, (lds_double*)dAi, (lds_double*)LI, (lds_double*)colb, (lds_double*)dg0, a.piv_tol, (lds_int*)cnt, (DTO_WIDE_PROFILE && blockIdx.x == 0) ? a.prof : nullptr);
        else ldl_blocked<N>(MA, dA, dAi, LI, a.piv_tol, cnt, (DTO_WIDE_PROFILE && blockIdx.x == 0) ? a.prof : nullptr);
        DTO_WIDE_TICK(6);
        // ---- phase 7: F~ = F L_A^-T (row tiles), V~ = L_A^-1 V (column tiles), bx~ = L_A^-1 bx
        if (w == 0) trsv_lower<N>(MA, bx);
        trsm_pair<N>(MF, MV, MA, LI, w);
        lds_barrier();
        DTO_WIDE_TICK(7);
        // ---- phase 8: M = D + F~ D_A^-1 F~' (registers), E'' = E - F~ D_A^-1 V~ (in place), bd~
        // (the factor record leaves as soon as its pieces are final -- L_A, F~, V~ here, L_M before phase 10, E~ before
        //  phase 11 -- so that the stores drain behind the matrix products: issued together at the end of the stage, the 170 KB
        //  of all 256 workgroups hit HBM at once and the first loads of the next stage waited ~25 k cycles behind them)
        // the point of the next stage (x, y, lam, fixed mask, u -- all dead here since phase 5): requested now, written to LDS
        // at the end of this phase, so that phase 0 of the next stage does not start with a round trip to HBM
        double nx_ = 0.0, ny_ = 0.0, nl_ = 0.0, nf_ = 0.0, nu_ = 0.0;
        const bool pre_ = t + 1 < a.T - 1;
        if (pre_ && tid < N) {
          nx_ = z[a.zoff[t + 1] + tid];
          ny_ = z[a.zoff[t + 2] + tid];
          nl_ = mu[a.cdoff[t + 1] + tid];
          nf_ = (a.fixed_lo && a.fixed_lo[a.zoff[t + 1] + tid] == a.fixed_hi[a.zoff[t + 1] + tid]) ? 1.0 : 0.0;
          if (tid < NU) nu_ = z[a.zoff[t + 1] + N + tid];
        }
        if (DTO_WIDE_PACK_L) store_ltiles<N>(fac + D::F_LA, MA); else store_fac<MAT>(fac + D::F_LA, MA);
        store_fac<MAT>(fac + D::F_FT, MF);
        store_fac<MAT>(fac + D::F_VT, MV);
        DTO_WIDE_TICK(24);
        d4 macc[NT];
#pragma unroll
        for (int jb = 0; jb < NT; ++jb) {
          const int r = l & 15, q = l >> 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int row = w * TB + q + 4 * j, col = jb * TB + r;
            double fuu = 0.0;
#pragma unroll
            for (int ju = 0; ju < NU; ++ju) fuu += fu[ju * N + row] * fu[ju * N + col] * ip[ju];
            macc[jb][j] = (row == col ? dc : 0.0) + fuu;
          }
        }
        mm_row4<0, N>(macc, MF, w * TB, MF, dAi, 1.0);
        DTO_WIDE_TICK(25);
        {
          d4 eacc[NT];
#pragma unroll
          for (int jb = 0; jb < NT; ++jb) eacc[jb] = tile_load(ME, LD, w * TB, jb * TB);
          mm_row4<1, N>(eacc, MF, w * TB, MV, dAi, -1.0);
#pragma unroll
          for (int jb = 0; jb < NT; ++jb) tile_store(ME, LD, w * TB, jb * TB, eacc[jb]);
        }
        DTO_WIDE_TICK(26);
#if DTO_WIDE_DOTQ & 4
        {
          const double part = quad_sum(dotq_rs<N>(MF, LD, bx, dAi));
          if ((tid & 3) == 0) tmp[tid >> 2] = bd[tid >> 2] - part;
        }
#else
        if (tid < N) tmp[tid] = bd[tid] - dot_rrs<N>(MF + tid * LD, bx, dAi);
#endif
        if (pre_ && tid < N) {
          xv[tid] = nx_; yv[tid] = ny_; lamv[tid] = nl_; fxm[tid] = nf_;
          if (tid < NU) sc[tid] = nu_;
          const unsigned long long anyf = __ballot(nf_ != 0.0);
          if (tid == 0) cnt[7] = anyf != 0ull;   // (cnt[6] of the NEXT stage: copied at its phase 0, this stage still reads cnt[6])
        }
        lds_barrier();
#pragma unroll
        for (int jb = 0; jb < NT; ++jb) tile_store(MA, LD, w * TB, jb * TB, macc[jb]);
        if (tid < N) bd[tid] = tmp[tid];
        lds_barrier();
        DTO_WIDE_TICK(8);
        // ---- phase 9: M = L_M D_M L_M'   (the KKT pivots of this block are -D_M)
        if (tid == 0) cnt[0] += N;  // N negative pivots if every D_M entry is positive; corrected below
        lds_barrier();
        {
          int* cm = cnt + 2;  // scratch counters for M
          if (tid == 0) { cm[0] = 0; cm[1] = 0; }
          lds_barrier();
          if (DTO_WIDE_LDL_RANK1) ldl_rank1<N>((lds_double*)MA, (lds_double*)dM, (lds_double*)dMi, (lds_double*)LI, (lds_double*)colb, (lds_double*)dg0, a.piv_tol, (lds_int*)cm);
          else ldl_blocked<N>(MA, dM, dMi, LI, a.piv_tol, cm);
          if (tid == 0) { cnt[0] -= cm[0]; cnt[1] |= cm[1]; }
        }
        DTO_WIDE_TICK(9);
        // ---- phase 10: E~ = L_M^-1 E'', bd^ = L_M^-1 bd~
        if (DTO_WIDE_PACK_L) store_ltiles<N>(fac + D::F_LM, MA); else store_fac<MAT>(fac + D::F_LM, MA);
        if (w == 0) trsv_lower<N>(MA, bd);
        trsm_left_coltile<N>(ME, MA, LI, w);
        lds_barrier();
        DTO_WIDE_TICK(10);
        // ---- phase 11: P' = -V~' D_A^-1 V~ + E~' D_M^-1 E~ (registers), carried right-hand side
        store_fac<MAT>(fac + D::F_ET, ME);
        if (tid < N) {
          double* fv = fac + D::F_VEC;
          fv[D::V_DA + tid] = dAi[tid];
          fv[D::V_DM + tid] = dMi[tid];
          fv[D::V_BX + tid] = bx[tid];
          fv[D::V_BD + tid] = bd[tid];
#pragma unroll
          for (int j = 0; j < NU; ++j) {
            fv[D::V_AU + j * N + tid] = au[j * N + tid];
            fv[D::V_FU + j * N + tid] = fu[j * N + tid];
            fv[D::V_VU + j * N + tid] = vu[j * N + tid];
          }
          fv[D::V_GC + tid] = gc[tid];
        }
        if (tid < NU) {
          double* fs = fac + D::F_VEC;
          fs[D::V_GC + N + tid] = gc[N + tid];
          fs[D::V_SC + tid] = (NU == 1) ? ip[0] : ipv[tid];
          fs[D::V_SC + NU + tid] = buv[tid];
        }
        if (NU > 1 && tid < NU * NU) fac[D::F_VEC + D::V_SC + 2 * NU + tid] = auu[tid];
#pragma unroll
        for (int jb = 0; jb < NT; ++jb) macc[jb] = d4{0.0, 0.0, 0.0, 0.0};
        mm_row4<2, N>(macc, MV, w * TB, MV, dAi, -1.0);
        mm_row4<2, N>(macc, ME, w * TB, ME, dMi, 1.0);
#if DTO_WIDE_DOTQ & 8
        {
          const double part = quad_sum(dotq_cs<N>(ME, LD, bd, dMi) - dotq_cs<N>(MV, LD, bx, dAi));
          if ((tid & 3) == 0) tmp[tid >> 2] = byn[tid >> 2] + part;
        }
#else
        if (tid < N) tmp[tid] = byn[tid] - dot_crs<N>(MV + tid, LD, bx, dAi) + dot_crs<N>(ME + tid, LD, bd, dMi);
#endif
        lds_barrier();
#pragma unroll
        for (int jb = 0; jb < NT; ++jb) tile_store(MA, LD, w * TB, jb * TB, macc[jb]);
        if (tid < N) { byc[tid] = tmp[tid]; gyp[tid] = gyn[tid]; }
        lds_barrier();
        DTO_WIDE_TICK(11);
        // the y-y part of this stage's Hessian and the u rank-one term complete P'
        if constexpr (SPARSE_U) {
          constexpr int NV = KD::VU_N;
          for (int i = tid; i < NV * NV; i += WG) {
            const int r = KD::vu_s(i / (NV > 0 ? NV : 1)), c = KD::vu_s(i % (NV > 0 ? NV : 1));
            double dvv = 0.0;
#pragma unroll
            for (int j = 0; j < NU; ++j) dvv += vu[j * N + r] * vu[j * N + c] * ip[j];
            MA[r * LD + c] -= dvv;
          }
        } else {
#pragma unroll DTO_WIDE_RMW_UNROLL
        for (int i = tid; i < N * N; i += WG) {
          const int r = i >> 6, c = i & 63;
          double dvv = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) dvv += vu[j * N + r] * vu[j * N + c] * ip[j];
          MA[r * LD + c] -= dvv;
        }
        }
        lds_barrier();
        if constexpr (DY::NH > 0) {
          if (tid < DY::NH) {
            const int r = DY::h_row(tid), c = DY::h_col(tid);
            if (r >= N + NU && c >= N + NU) MA[(r - N - NU) * LD + c - N - NU] += hv[tid];
          }
        }
        lds_barrier();
      }
    });
  }
  DTO_WIDE_TICK(12);
  // ---- terminal stage: (W_T + dw I + P') x = -(grad + E' lam) + by
  {
    const int t = a.T - 1;
    const int wk = M::wk_of_kind(a.kind[t]);
    M::dispatch_wk(wk, [&](auto wkc) {
      constexpr int WKI = decltype(wkc)::value;
      using KD = typename M::template WKind<WKI>;
      if constexpr (KD::DYN < 0) {
        using CO = typename M::template Cost<KD::COST>;
        const double* wp = a.params + b * a.ldw + a.woff[t];
        if (tid < N) {
          xv[tid] = z[a.zoff[t] + tid];
          fxm[tid] = (a.fixed_lo && a.fixed_lo[a.zoff[t] + tid] == a.fixed_hi[a.zoff[t] + tid]) ? 1.0 : 0.0;
        }
        __syncthreads();
        if (w == 0) {
          CO::grad(xv, sc, wp, gc);
          if constexpr (CO::SNH > 0) CO::shess(xv, sc, wp, chv);
          if (a.stats) {
            CO::eval(xv, sc, wp, stat + 6);
            if (l == 0) stat[0] += stat[6];
          }
        }
        __syncthreads();
        if constexpr (CO::SNH > 0) {
          if (tid < CO::SNH) MA[CO::sh_row(tid) * LD + CO::sh_col(tid)] += chv[tid];
        }
        __syncthreads();
        if (tid < N) {
          double sig = 0.0;
          if (BAR) brx[tid] = 0.0;
          double zd = 0.0;
          if (barrier) {
            const int gi = a.zoff[t] + tid;
            const WideBar wb = wide_bar(xv[tid], a.fixed_lo[gi], a.fixed_hi[gi], zlb[gi], zub[gi], mub);
            sig = wb.sig; brx[tid] = wb.br; zd = wb.zdiff;
            if (a.stats) {
              const double m1 = wave_max(wb.sz_max), m2 = wave_max(wb.isz_max), s1 = wave_sum(wb.sum_z), s2 = wave_sum(wb.logb);
              if (tid == 0) {
                stat[DTO_WIDE_SZMAX] = fmax(stat[DTO_WIDE_SZMAX], m1); stat[DTO_WIDE_ISZMAX] = fmax(stat[DTO_WIDE_ISZMAX], m2);
                stat[DTO_WIDE_SUMZ] += s1; stat[DTO_WIDE_LOGBAR] += s2;
              }
            }
          }
          MA[tid * LD + tid] += dw + sig;
          bx[tid] = -(gc[tid] + gyp[tid]) + byc[tid] + (BAR ? brx[tid] : 0.0);
          if (a.stats) {
            double v = (fxm[tid] != 0.0) ? 0.0 : fabs(gc[tid] + gyp[tid] + zd);
            v = wave_max(v);
            if (tid == 0) stat[3] = fmax(stat[3], v);
          }
        }
        __syncthreads();
        if (a.fixed_lo) {
          for (int i = tid; i < N * N; i += WG) {
            const int r = i >> 6, c = i & 63;
            if (fxm[r] != 0.0 || fxm[c] != 0.0) MA[r * LD + c] = (r == c) ? 1.0 : 0.0;
          }
          if (tid < N && fxm[tid] != 0.0) bx[tid] = 0.0;
          __syncthreads();
        }
        if (DTO_WIDE_LDL_RANK1) ldl_rank1<N>((lds_double*)MA, (lds_double*)dA, (lds_double*)dAi, (lds_double*)LI, (lds_double*)colb, (lds_double*)dg0, a.piv_tol, (lds_int*)cnt);
        else ldl_blocked<N>(MA, dA, dAi, LI, a.piv_tol, cnt);
        if (w == 0) {
          trsv_lower<N>(MA, bx);
          if (l < N) bx[l] *= dAi[l];
          trsv_lower_t<N>(MA, bx);
          if (l < N) {
            yv[l] = bx[l];
            a.dz[b * a.lddz + a.zoff[t] + l] = bx[l];
          }
          if (a.stats) {
            const double part = wave_sum((gc[l] - (BAR ? brx[l] : 0.0)) * bx[l]);     // gradient of the barrier objective along the step
            if (l == 0) stat[4] += part;
            if (barrier) {
              const int gi = a.zoff[t] + l;
              double ap = 1.0, ad = 1.0;
              wide_bar_step(xv[l], bx[l], a.fixed_lo[gi], a.fixed_hi[gi], zlb[gi], zub[gi], mub, taub, ap, ad);
              ap = wave_min(ap); ad = wave_min(ad);
              if (l == 0) { stat[DTO_WIDE_APMAX] = fmin(stat[DTO_WIDE_APMAX], ap); stat[DTO_WIDE_ADMAX] = fmin(stat[DTO_WIDE_ADMAX], ad); }
            }
          }
        }
        __syncthreads();
      }
    });
  }
  DTO_WIDE_TICK(13);
  if (tid == 0) a.flags[b] = (cnt[0] == (int)a.Nc && cnt[1] == 0) ? 1 : 0;
#if DTO_WIDE_SPLIT_BWD
  // the backward sweep is k_wide_bwd (next launch on the same stream): it picks up the statistics from here
  if (a.stats && tid < DTO_WIDE_NSTAT) a.stats[b * DTO_WIDE_NSTAT + tid] = stat[tid];
  return;
#endif
  // ---- backward sweep: y = x_{t+1} is in yv.  The factor records were written by all threads of this workgroup.
  __threadfence_block();
  __syncthreads();
  for (int t = a.T - 2; t >= 0; --t) {
    const double* fac = facb + (int64_t)t * D::FAC;
    const double* fv = fac + D::F_VEC;
    copy_mat<MAT>(ME, fac + D::F_ET);
    static_assert(DTO_WIDE_SPLIT_BWD || !DTO_WIDE_PACK_L, "the in-kernel backward sweep reads unpacked factors");
    copy_mat<MAT>(MA, fac + D::F_LM);
    copy_mat<MAT>(MF, fac + D::F_FT);
    copy_mat<MAT>(MV, fac + D::F_VT);
    if (tid < N) {
      dAi[tid] = fv[D::V_DA + tid];
      dMi[tid] = fv[D::V_DM + tid];
      bx[tid] = fv[D::V_BX + tid];
      bd[tid] = fv[D::V_BD + tid];
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        au[j * N + tid] = fv[D::V_AU + j * N + tid];
        fu[j * N + tid] = fv[D::V_FU + j * N + tid];
        vu[j * N + tid] = fv[D::V_VU + j * N + tid];
      }
    }
    __syncthreads();
    DTO_WIDE_TICK(14);
    // lam = L_M^-T D_M^-1 (E~ y - bd^)
    if (tid < N) {
      lamv[tid] = (dot_rr<N>(ME + tid * LD, yv) - bd[tid]) * dMi[tid];
    }
    __syncthreads();
    if (w == 0) trsv_lower_t<N>(MA, lamv);
    __syncthreads();
    DTO_WIDE_TICK(15);
    // x = L_A^-T D_A^-1 (bx~ - F~' lam - V~ y)
    if (tid < N) {
      xv[tid] = (bx[tid] - dot_cr<N>(MF + tid, LD, lamv) - dot_rr<N>(MV + tid * LD, yv)) * dAi[tid];
    }
    __syncthreads();
    copy_mat<MAT>(MA, fac + D::F_LA);
    __syncthreads();
    if (w == 0) trsv_lower_t<N>(MA, xv);
    __syncthreads();
    if (tid < N) {
      a.dz[b * a.lddz + a.zoff[t] + tid] = xv[tid];
      a.dmu[b * a.lddmu + a.cdoff[t] + tid] = lamv[tid];
    }
    if (w == 1) {
      // u_j = (bu_j - au_j'x - fu_j'lam - vu_j'y) / piv_j - sum_{k > j} L_u[k][j] u_k, last action first
      double duv[NU];
#pragma unroll
      for (int j = NU - 1; j >= 0; --j) {
        const double part = wave_sum(au[j * N + l] * xv[l] + fu[j * N + l] * lamv[l] + vu[j * N + l] * yv[l]);
        double uj = (fv[D::V_SC + NU + j] - part) * fv[D::V_SC + j];
#pragma unroll
        for (int k = j + 1; k < NU; ++k) uj -= fv[D::V_SC + 2 * NU + k * NU + j] * duv[k];
        duv[j] = uj;
      }
      double du = duv[0];   // lane j < NU: its own action
#pragma unroll
      for (int j = 1; j < NU; ++j) du = (l == j) ? duv[j] : du;
      if (l < NU) a.dz[b * a.lddz + a.zoff[t] + N + l] = du;
      if (a.stats) {
        // gradient of the barrier objective along the step, fraction-to-the-boundary limits of this knot (x: lane, u: lane 0)
        double brl = 0.0, ap = 1.0, ad = 1.0;
        if (barrier) {
          const int gi = a.zoff[t] + l;
          const double xo = z[gi], lo = a.fixed_lo[gi], hi = a.fixed_hi[gi];
          brl = wide_bar(xo, lo, hi, zlb[gi], zub[gi], mub).br;
          wide_bar_step(xo, xv[l], lo, hi, zlb[gi], zub[gi], mub, taub, ap, ad);
          if (l < NU) {
            const int gu = a.zoff[t] + N + l;
            const double uo = z[gu], ulo = a.fixed_lo[gu], uhi = a.fixed_hi[gu];
            wide_bar_step(uo, du, ulo, uhi, zlb[gu], zub[gu], mub, taub, ap, ad);
          }
          ap = wave_min(ap); ad = wave_min(ad);
        }
        double gl = (fv[D::V_GC + l] - brl) * xv[l];
        if (l < NU) {
          double bru = 0.0;
          if (barrier) {
            const int gu = a.zoff[t] + N + l;
            bru = wide_bar(z[gu], a.fixed_lo[gu], a.fixed_hi[gu], zlb[gu], zub[gu], mub).br;
          }
          gl += (fv[D::V_GC + N + l] - bru) * du;
        }
        const double gpart = wave_sum(gl);
        if (l == 0) {
          if (barrier) { stat[DTO_WIDE_APMAX] = fmin(stat[DTO_WIDE_APMAX], ap); stat[DTO_WIDE_ADMAX] = fmin(stat[DTO_WIDE_ADMAX], ad); }
          stat[4] += gpart;
        }
      }
    }
    __syncthreads();
    if (tid < N) yv[tid] = xv[tid];
    __syncthreads();
    DTO_WIDE_TICK(16);
  }
  if (a.stats && tid < DTO_WIDE_NSTAT) a.stats[b * DTO_WIDE_NSTAT + tid] = stat[tid];
}

// ---------------------------------------------------------------------------------------------------
// backward sweep as a kernel of its own (DTO_WIDE_SPLIT_BWD): same arithmetic as the tail of k_wide_step, but the factor
// record of stage t-1 travels from HBM into REGISTERS while stage t is worked on from LDS (45 x 16 bytes per thread: five
// matrices; inside k_wide_step there were no registers for that -- 256 + 240 in use -- and no LDS for a second stage).
// grid = B, block = 256.
// ---------------------------------------------------------------------------------------------------
template <class M>
struct BwdLds {
  static constexpr int N = M::WIDE_N, NU = M::WIDE_NU;
  static constexpr int DOUBLES = 4 * Dims<N>::MAT + (9 + 3 * NU) * N + 8 + 32 + 16;
  static constexpr int BYTES = DOUBLES * (int)sizeof(double);
};

template <class M, bool BAR>
__global__ __launch_bounds__(WG) void k_wide_bwd(dto_wide_args a) {
  constexpr int N = M::WIDE_N, NU = M::WIDE_NU;
  using D = Dims<N, NU>;
  constexpr int LD = D::LD, MAT = D::MAT;
  constexpr int NP = (MAT / 2 + WG - 1) / WG;   // 16-byte pieces of one matrix per thread
  extern __shared__ double sm[];
  double* MA = sm;
  double* MF = MA + MAT;
  double* MV = MF + MAT;
  double* ME = MV + MAT;
  double* xv = ME + MAT;
  double* yv = xv + N;
  double* lamv = yv + N;
  double* au = lamv + N;
  double* fu = au + NU * N;
  double* vu = fu + NU * N;
  double* bx = vu + NU * N;
  double* bd = bx + N;
  double* dAi = bd + N;
  double* dMi = dAi + N;
  double* gcv = dMi + N;       // cost gradient [N + 8]
  double* scv = gcv + N + 8;   // scalars of the action block [32]
  double* stat = scv + 32;     // [16]
  const int tid = threadIdx.x, w = wave_id(), l = lane_id();
  const int64_t b = blockIdx.x;
  if (a.active && !a.active[b]) return;
  const double* z = a.z + b * a.ldz;
  const double* facb = a.fac + b * (int64_t)a.T * D::FAC;
  const bool barrier = BAR && a.zl != nullptr;
  const double mub = (barrier && a.mu_inst) ? a.mu_inst[b] : 0.0;
  const double* zlb = barrier ? a.zl + b * a.ldz : nullptr;
  const double* zub = barrier ? a.zu + b * a.ldz : nullptr;
  const double taub = fmax(a.tau_min, 1.0 - mub);
  long long tick_ = clock64();
  if (tid < N) yv[tid] = a.dz[b * a.lddz + a.zoff[a.T - 1] + tid];
  if (tid < DTO_WIDE_NSTAT) stat[tid] = a.stats ? a.stats[b * DTO_WIDE_NSTAT + tid] : 0.0;

  typedef double v2d __attribute__((ext_vector_type(2)));   // (HIP's double2 is a class: arrays of it stay in scratch)
  constexpr int NPL = DTO_WIDE_PACK_L ? D::LTILES * TB * TB / 2 / WG : NP;   // pieces of a triangular factor
  v2d pe[NP], pm[NPL], pf[NP], pv[NP], pa[NPL];
  double pvec[5 + 3 * NU], psc = 0.0;
  auto load_mat = [&](v2d (&r)[NP], const double* src) {
    const v2d* s2 = reinterpret_cast<const v2d*>(src);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int i = tid + k * WG;
      if (i < MAT / 2) r[k] = s2[i];
    }
  };
  auto store_mat = [&](double* dst, const v2d (&r)[NP]) {
    v2d* d2 = reinterpret_cast<v2d*>(dst);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int i = tid + k * WG;
      if (i < MAT / 2) d2[i] = r[k];
    }
  };
  // triangular factors: the ten lower tiles straight from the record, written to their places in LDS (the upper tiles of MA
  // are zeroed once below and never written)
  auto load_l = [&](v2d (&r)[NPL], const double* src) {
    const v2d* s2 = reinterpret_cast<const v2d*>(src);
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const int i = tid + k * WG;
      if (DTO_WIDE_PACK_L || i < MAT / 2) r[k] = s2[i];
    }
  };
  auto store_l = [&](double* dst, const v2d (&r)[NPL]) {
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const int i = tid + k * WG;
      if (DTO_WIDE_PACK_L) { const int e = ltile_elem<N>(i); dst[e] = r[k].x; dst[e + 1] = r[k].y; }
      else if (i < MAT / 2) reinterpret_cast<v2d*>(dst)[i] = r[k];
    }
  };
  if (DTO_WIDE_PACK_L) {
    for (int i = tid; i < MAT; i += WG) MA[i] = 0.0;
  }
  auto issue = [&](int t) {
    const double* fac = facb + (int64_t)t * D::FAC;
    const double* fv = fac + D::F_VEC;
    load_mat(pe, fac + D::F_ET);
    load_l(pm, fac + D::F_LM);
    load_mat(pf, fac + D::F_FT);
    load_mat(pv, fac + D::F_VT);
    if (tid < N) {
      pvec[0] = fv[D::V_DA + tid]; pvec[1] = fv[D::V_DM + tid]; pvec[2] = fv[D::V_BX + tid]; pvec[3] = fv[D::V_BD + tid];
      pvec[4] = fv[D::V_GC + tid];
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        pvec[5 + 3 * j] = fv[D::V_AU + j * N + tid]; pvec[6 + 3 * j] = fv[D::V_FU + j * N + tid]; pvec[7 + 3 * j] = fv[D::V_VU + j * N + tid];
      }
    } else if (tid < N + 32) {
      // [0, 8): cost gradient of the actions; [8, 8 + NU (NU + 2)): 1 / pivots, reduced right-hand sides, L_u
      const int q = tid - N;
      psc = (q < 8) ? fv[D::V_GC + N + q] : (q - 8 < ((NU * (NU + 2) + 7) & ~7) ? fv[D::V_SC + q - 8] : 0.0);
    }
  };
  issue(a.T - 2);
  load_l(pa, facb + (int64_t)(a.T - 2) * D::FAC + D::F_LA);
  for (int t = a.T - 2; t >= 0; --t) {
    __syncthreads();
    store_mat(ME, pe);
    store_l(MA, pm);
    store_mat(MF, pf);
    store_mat(MV, pv);
    if (tid < N) {
      dAi[tid] = pvec[0]; dMi[tid] = pvec[1]; bx[tid] = pvec[2]; bd[tid] = pvec[3]; gcv[tid] = pvec[4];
#pragma unroll
      for (int j = 0; j < NU; ++j) { au[j * N + tid] = pvec[5 + 3 * j]; fu[j * N + tid] = pvec[6 + 3 * j]; vu[j * N + tid] = pvec[7 + 3 * j]; }
    } else if (tid < N + 32) {
      const int q = tid - N;
      if (q < 8) gcv[N + q] = psc; else scv[q - 8] = psc;
    }
    __syncthreads();
    if (t > 0) issue(t - 1);
    DTO_WIDE_TICK(14);
    // lam = L_M^-T D_M^-1 (E~ y - bd^)
    {
      const double part = quad_sum(dotq_r<N>(ME, LD, yv));
      if ((tid & 3) == 0) lamv[tid >> 2] = (part - bd[tid >> 2]) * dMi[tid >> 2];
    }
    __syncthreads();
    if (w == 0) trsv_lower_t<N>(MA, lamv);
    __syncthreads();
    DTO_WIDE_TICK(15);
    // x = L_A^-T D_A^-1 (bx~ - F~' lam - V~ y)
    {
      const double part = quad_sum(dotq_c<N>(MF, LD, lamv) + dotq_r<N>(MV, LD, yv));
      if ((tid & 3) == 0) xv[tid >> 2] = (bx[tid >> 2] - part) * dAi[tid >> 2];
    }
    store_l(MA, pa);   // L_M is done with (the trsv above ended at the last barrier)
    if (t > 0) load_l(pa, facb + (int64_t)(t - 1) * D::FAC + D::F_LA);
    __syncthreads();
    if (w == 0) trsv_lower_t<N>(MA, xv);
    __syncthreads();
    if (tid < N) {
      a.dz[b * a.lddz + a.zoff[t] + tid] = xv[tid];
      a.dmu[b * a.lddmu + a.cdoff[t] + tid] = lamv[tid];
    }
    if (w == 1) {
      // u_j = (bu_j - au_j'x - fu_j'lam - vu_j'y) / piv_j - sum_{k > j} L_u[k][j] u_k, last action first
      double duv[NU];
#pragma unroll
      for (int j = NU - 1; j >= 0; --j) {
        const double part = wave_sum(au[j * N + l] * xv[l] + fu[j * N + l] * lamv[l] + vu[j * N + l] * yv[l]);
        double uj = (scv[NU + j] - part) * scv[j];
#pragma unroll
        for (int k = j + 1; k < NU; ++k) uj -= scv[2 * NU + k * NU + j] * duv[k];
        duv[j] = uj;
      }
      double du = duv[0];   // lane j < NU: its own action
#pragma unroll
      for (int j = 1; j < NU; ++j) du = (l == j) ? duv[j] : du;
      if (l < NU) a.dz[b * a.lddz + a.zoff[t] + N + l] = du;
      if (a.stats) {
        // gradient of the barrier objective along the step, fraction-to-the-boundary limits of this knot (x: lane, u: lanes < NU)
        double brl = 0.0, ap = 1.0, ad = 1.0;
        if (barrier) {
          const int gi = a.zoff[t] + l;
          const double xo = z[gi], lo = a.fixed_lo[gi], hi = a.fixed_hi[gi];
          brl = wide_bar(xo, lo, hi, zlb[gi], zub[gi], mub).br;
          wide_bar_step(xo, xv[l], lo, hi, zlb[gi], zub[gi], mub, taub, ap, ad);
          if (l < NU) {
            const int gu = a.zoff[t] + N + l;
            const double uo = z[gu], ulo = a.fixed_lo[gu], uhi = a.fixed_hi[gu];
            wide_bar_step(uo, du, ulo, uhi, zlb[gu], zub[gu], mub, taub, ap, ad);
          }
          ap = wave_min(ap); ad = wave_min(ad);
        }
        double gl = (gcv[l] - brl) * xv[l];
        if (l < NU) {
          double bru = 0.0;
          if (barrier) {
            const int gu = a.zoff[t] + N + l;
            bru = wide_bar(z[gu], a.fixed_lo[gu], a.fixed_hi[gu], zlb[gu], zub[gu], mub).br;
          }
          gl += (gcv[N + l] - bru) * du;
        }
        const double gpart = wave_sum(gl);
        if (l == 0) {
          if (barrier) { stat[DTO_WIDE_APMAX] = fmin(stat[DTO_WIDE_APMAX], ap); stat[DTO_WIDE_ADMAX] = fmin(stat[DTO_WIDE_ADMAX], ad); }
          stat[4] += gpart;
        }
      }
    }
    __syncthreads();
    if (tid < N) yv[tid] = xv[tid];
    DTO_WIDE_TICK(16);
  }
  __syncthreads();
  if (a.stats && tid < DTO_WIDE_NSTAT) a.stats[b * DTO_WIDE_NSTAT + tid] = stat[tid];
}

// one wavefront per instance: out[b] = sum_t rows[b][t] in a fixed order (lane-strided partials, then a tree)
static __global__ __launch_bounds__(64) void k_wide_sum_rows(const double* rows, int64_t ld, int n, double* out) {
  const int64_t b = blockIdx.x;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) acc += rows[b * ld + i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (threadIdx.x == 0) out[b] = acc;
}

// ---------------------------------------------------------------------------------------------------
// evaluator callbacks for wide models (the five MOI methods, src/moi.jl:1-120): one wavefront per knot, the
// model code runs wave-uniform into LDS, results leave in full-line coalesced stores.  The stage Jacobian is
// the constant nonzero table (L2 resident) copied to the output plus the few state-dependent entries.
// ---------------------------------------------------------------------------------------------------
// The constant part of the stage Jacobian lives in one table per dynamics class.  k_wide_eval<CON> and k_wide_merit stage ONE
// class's table in LDS -- that of the middle stage of the horizon: the class nearly every stage has -- and read the table of any
// other class (time-varying dynamics; the embedding of stage constraints gives the first and the last stage classes of their
// own, solver.py: pad_to_wide) from global memory.  (Rounds 1-5 staged Dyn<0>'s table and used it for EVERY stage.)
template <class M>
__device__ __forceinline__ int staged_dyn_class(const int* kind, int T) {
  int dyn = 0;
  M::dispatch_wk(M::wk_of_kind(kind[T > 1 ? (T - 1) / 2 : 0]), [&](auto wkc) {
    using KD = typename M::template WKind<decltype(wkc)::value>;
    dyn = KD::DYN >= 0 ? KD::DYN : 0;
  });
  return dyn;
}
template <class M, int C = 0>
__device__ __forceinline__ const double* fe_const_of(int dyn) {
  if constexpr (C + 1 < M::N_DYN) {
    if (dyn != C) return fe_const_of<M, C + 1>(dyn);
  }
  return M::template Dyn<C>::fe_const();
}

constexpr int EV_WAVES = 4;

template <class M>
struct EvalLds {
  static constexpr int N = M::WIDE_N, NU = M::WIDE_NU;
  static constexpr int PT = 3 * N + NU + 3;                                   // x, u, y, lam
  static constexpr int OUT = (M::MAX_KEY > N + NU + 8 ? M::MAX_KEY : N + NU + 8) + M::MAX_NH + M::MAX_SNH + M::MAX_NJV + 8 + 2 * M::MAX_CON;
  static constexpr int PER_WAVE = PT + OUT;
};

template <class M, int OP>
__global__ __launch_bounds__(EV_WAVES * 64) void k_wide_eval(dto_eval_args a) {
  constexpr int N = M::WIDE_N, NU = M::WIDE_NU, NC = 2 * N + NU;
  using EL = EvalLds<M>;
  extern __shared__ double sm[];
  const int w = wave_id(), l = lane_id();
  double* mine = sm + w * EL::PER_WAVE;
  double* xv = mine;            // [N]
  double* uv = xv + N;          // [NU] (+ pad)
  double* yv = uv + NU + 1;     // [N]
  double* lamv = yv + N + 1;    // [N]
  double* ov = mine + EL::PT;   // outputs / key image
  double* hv = ov + (M::MAX_KEY > N + NU + 8 ? M::MAX_KEY : N + NU + 8);
  double* chv = hv + M::MAX_NH;
  double* jvv = chv + M::MAX_SNH;
  double* cvv = jvv + M::MAX_NJV + 8;           // [MAX_CON] values / Jacobian / Hessian nonzeros of the knot's stage constraint
  double* nuv = cvv + M::MAX_CON;               // [MAX_CON] its multipliers
  double* fe_s = sm + EV_WAVES * EL::PER_WAVE;  // [N][NC] constant Jacobian, OP == CON only
  int staged = 0;
  if constexpr (OP == DTO_OP_CON) {
    // the constant Jacobian table of the horizon's main dynamics class, staged once per workgroup (fe_const_of)
    staged = staged_dyn_class<M>(a.kind, a.T);
    const double* fe = fe_const_of<M>(staged);
    for (int i = threadIdx.x; i < N * NC; i += EV_WAVES * 64) fe_s[i] = fe[i];
    __syncthreads();
  }
  const int64_t nknot = a.B * (int64_t)a.T;
  for (int64_t kn = (int64_t)blockIdx.x * EV_WAVES + w; kn < nknot; kn += (int64_t)gridDim.x * EV_WAVES) {
    const int64_t b = kn / a.T;
    const int t = (int)(kn - b * a.T);
    const double* z = a.z + b * a.ldz;
    const int kind = a.kind[t];
    const int wk = M::wk_of_kind(kind);
    const double* wp = a.w + b * a.ldw + a.woff[t];
    M::dispatch_wk(wk, [&](auto wkc) {
      using KD = typename M::template WKind<decltype(wkc)::value>;
      using CO = typename M::template Cost<KD::COST>;
      constexpr bool HAS_DYN = KD::DYN >= 0;
      constexpr int NUK = HAS_DYN ? NU : 0;
      // lane = state index; evaluator plugins exist for any uniform N <= 64 (the KKT kernels for N == 64 only)
      const bool ln = l < N;
      if (ln) xv[l] = z[a.zoff[t] + l];
      if (l < NUK) uv[l] = z[a.zoff[t] + N + l];
      if constexpr (HAS_DYN) { if (ln) yv[l] = z[a.zoff[t + 1] + l]; }
      wave_lds_fence();   // every lane's entry of the point is in LDS before the wave-uniform model code reads them
      if constexpr (OP == DTO_OP_OBJ) {
        CO::eval(xv, uv, wp, ov);
        if (l == 0) a.scratch[b * a.T + t] = ov[0];
      } else if constexpr (OP == DTO_OP_GRAD) {
        CO::grad(xv, uv, wp, ov);
        double* g = a.out + b * a.ldout + a.zoff[t];
        if (ln) g[l] = ov[l];
        if (l < NUK) g[N + l] = ov[N + l];
      } else if constexpr (OP == DTO_OP_CON) {
        if constexpr (HAS_DYN) {
          using DY = typename M::template Dyn<KD::DYN>;
          if (ln) ov[l] = 0.0;
          DY::eval_nl(xv, uv, yv, wp, hv);
          wave_lds_fence();
          if (l < DY::NNL) ov[DY::nl_row(l)] = hv[l];
          wave_lds_fence();   // lane nl_row(q) reads what lane q stored
          if (ln) {
            // (one dynamics class: the LDS copy, through LDS addressing as in rounds 1-5; several: a generic pointer)
            const double* row = fe_s + l * NC;
            if constexpr (M::N_DYN > 1) { if (KD::DYN != staged) row = DY::fe_const() + l * NC; }
            double acc = ov[l] + dot_rr<N>(row, xv) + dot_rr<N>(row + N + NU, yv);
#pragma unroll
            for (int j = 0; j < NU; ++j) acc += row[N + j] * uv[j];
            a.out[b * a.ldout + a.cdoff[t] + l] = acc;
          }
        }
        // stage-constraint rows of this knot (src/constraints.jl:80-88): behind all dynamics rows (src/data.jl:68-69)
        if constexpr (KD::CON >= 0) {
          using CN = typename M::template Con<KD::CON>;
          CN::eval(xv, uv, wp, cvv);
          wave_lds_fence();
          for (int j = l; j < CN::NC; j += 64) a.out[b * a.ldout + a.ccoff[t] + j] = cvv[j];
        }
      } else if constexpr (OP == DTO_OP_JAC) {
        if constexpr (HAS_DYN) {
          using DY = typename M::template Dyn<KD::DYN>;
          DY::jac_var(xv, uv, yv, wp, jvv);
          double* o = a.out + b * a.ldout + a.jdoff[t];
          const double* jc = DY::jc_const();
          if ((((uintptr_t)o | (uintptr_t)jc) & 15) == 0) {
            // 16 B per lane: one full 1 KiB line group per wavefront store
            const double2* src = reinterpret_cast<const double2*>(jc);
            double2* dst = reinterpret_cast<double2*>(o);
#pragma unroll 4
            for (int i = l; i < DY::NJ / 2; i += 64) dst[i] = src[i];
            if ((DY::NJ & 1) && l == 0) o[DY::NJ - 1] = jc[DY::NJ - 1];
          } else {
            for (int i = l; i < DY::NJ; i += 64) o[i] = jc[i];
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          if (l < DY::NJV) o[DY::jv_k(l)] = jvv[l];
        }
        if constexpr (KD::CON >= 0) {   // src/constraints.jl:90-96: the knot's nonzeros in the class's CSC order
          using CN = typename M::template Con<KD::CON>;
          if constexpr (CN::NJ > 0) {
            CN::jac(xv, uv, wp, cvv);
            wave_lds_fence();
            for (int k = l; k < CN::NJ; k += 64) a.out[b * a.ldout + a.jcoff[t] + k] = cvv[k];
          }
        }
      } else if constexpr (OP == DTO_OP_HESS) {
        const double* mu = a.mu + b * a.ldmu;
        const int h0 = a.hoff[t], hlen = a.hoff[t + 1] - h0;
        for (int i = l; i < hlen; i += 64) ov[i] = 0.0;
        wave_lds_fence();   // (the key image is updated by different lanes in turn: every hand-over is fenced)
        if constexpr (CO::SNH > 0) {
          CO::shess(xv, uv, wp, chv);
          const int* mrow = a.hmap_cost + kind * a.hmap_stride;
          for (int i = l; i < CO::SNH; i += 64) ov[mrow[i]] += a.sigma * chv[i];
          wave_lds_fence();
        }
        if constexpr (HAS_DYN) {
          using DY = typename M::template Dyn<KD::DYN>;
          if constexpr (DY::NH > 0) {
            if (ln) lamv[l] = mu[a.cdoff[t] + l];
            wave_lds_fence();
            DY::hess(xv, uv, yv, wp, lamv, hv);
            const int* mrow = a.hmap_dyn_own + kind * a.hmap_stride;
            for (int i = l; i < DY::NH; i += 64) {
              const int m = mrow[i];
              if (m >= 0) ov[m] += hv[i];
            }
            wave_lds_fence();
          }
        }
        if constexpr (KD::CON >= 0) {   // src/constraints.jl:98-104: nu_t' c_t''
          using CN = typename M::template Con<KD::CON>;
          if constexpr (CN::NH > 0) {
            for (int j = l; j < CN::NC; j += 64) nuv[j] = mu[a.ccoff[t] + j];
            wave_lds_fence();
            CN::hess(xv, uv, wp, nuv, cvv);
            wave_lds_fence();
            const int* mrow = a.hmap_con + kind * a.hmap_stride;
            for (int i = l; i < CN::NH; i += 64) {
              const int m = mrow[i];
              if (m >= 0) ov[m] += cvv[i];
            }
            wave_lds_fence();
          }
        }
        // rows of this stage also receive the y-rows of the previous stage's dynamics Hessian
        if (t > 0) {
          const int wkp = M::wk_of_kind(a.kind[t - 1]);
          M::dispatch_wk(wkp, [&](auto wkp_c) {
            using KP = typename M::template WKind<decltype(wkp_c)::value>;
            if constexpr (KP::DYN >= 0) {
              using DP = typename M::template Dyn<KP::DYN>;
              if constexpr (DP::NH > 0) {
                // previous point: x_{t-1}, u_{t-1}, y = x_t
                if (ln) {
                  yv[l] = xv[l];
                  xv[l] = z[a.zoff[t - 1] + l];
                  lamv[l] = mu[a.cdoff[t - 1] + l];
                }
                if (l < NU) uv[l] = z[a.zoff[t - 1] + N + l];
                wave_lds_fence();
                DP::hess(xv, uv, yv, a.w + b * a.ldw + a.woff[t - 1], lamv, hv);
                const int* mrow = a.hmap_dyn_next + kind * a.hmap_stride;
                for (int i = l; i < DP::NH; i += 64) {
                  const int m = mrow[i];
                  if (m >= 0) ov[m] += hv[i];
                }
              }
            }
          });
        }
        wave_lds_fence();
        double* o = a.out + b * a.ldout + h0;
        for (int i = l; i < hlen; i += 64) o[i] = ov[i];
      }
    });
    wave_lds_fence();   // the next knot of this wavefront rewrites the point and the images
  }
}

// GeneralConstraint rows of a wide model (src/general_constraint.jl:73-83: values and Jacobian nonzeros behind the dynamics and
// stage blocks): one thread per instance, as dto_eval_kernels.hpp does it for the lane family
template <class M>
__global__ void k_wide_general(dto_eval_args a, int jac) {
  const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  if constexpr (M::HAS_GENERAL) {
    constexpr int NO = M::General::NC > M::General::NJ ? M::General::NC : M::General::NJ;
    double o[NO > 0 ? NO : 1];
    if (jac) {
      M::General::jac(a.z + b * a.ldz, a.w + b * a.ldw, o);
      for (int i = 0; i < M::General::NJ; ++i) a.out[b * a.ldout + a.general_jac0 + i] = o[i];
    } else {
      M::General::eval(a.z + b * a.ldz, a.w + b * a.ldw, o);
      for (int i = 0; i < M::General::NC; ++i) a.out[b * a.ldout + a.general_row0 + i] = o[i];
    }
  }
}

template <class M>
int launch_wide_eval(int op, const dto_eval_args* a, void* stream) {
  using EL = EvalLds<M>;
  constexpr int N = M::WIDE_N, NC = 2 * N + M::WIDE_NU;
  const int64_t nknot = a->B * (int64_t)a->T;
  const unsigned grid = (unsigned)((nknot + EV_WAVES - 1) / EV_WAVES < 4096 ? (nknot + EV_WAVES - 1) / EV_WAVES : 4096);
  const int lds = (int)sizeof(double) * EV_WAVES * EL::PER_WAVE;
  const int lds_con = lds + (int)sizeof(double) * N * NC;
  hipStream_t st = (hipStream_t)stream;
  switch (op) {
    case DTO_OP_OBJ:
      hipLaunchKernelGGL((k_wide_eval<M, DTO_OP_OBJ>), dim3(grid), dim3(EV_WAVES * 64), lds, st, *a);
      hipLaunchKernelGGL(k_wide_sum_rows, dim3((unsigned)a->B), dim3(64), 0, st, (const double*)a->scratch, (int64_t)a->T, a->T, a->out);
      break;
    case DTO_OP_GRAD: hipLaunchKernelGGL((k_wide_eval<M, DTO_OP_GRAD>), dim3(grid), dim3(EV_WAVES * 64), lds, st, *a); break;
    case DTO_OP_CON: {
      hipError_t e = hipFuncSetAttribute((const void*)k_wide_eval<M, DTO_OP_CON>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_con);
      if (e != hipSuccess) return (int)e;
      const unsigned gc = grid < 1024 ? grid : 1024;  // the table staging is per workgroup: fewer, longer-lived workgroups
      hipLaunchKernelGGL((k_wide_eval<M, DTO_OP_CON>), dim3(gc), dim3(EV_WAVES * 64), lds_con, st, *a);
      break;
    }
    case DTO_OP_JAC: hipLaunchKernelGGL((k_wide_eval<M, DTO_OP_JAC>), dim3(grid), dim3(EV_WAVES * 64), lds, st, *a); break;
    case DTO_OP_HESS: hipLaunchKernelGGL((k_wide_eval<M, DTO_OP_HESS>), dim3(grid), dim3(EV_WAVES * 64), lds, st, *a); break;
    case DTO_OP_GENERAL_CON:
    case DTO_OP_GENERAL_JAC:
      hipLaunchKernelGGL(k_wide_general<M>, dim3((unsigned)((a->B + 63) / 64)), dim3(64), 0, st, *a, op == DTO_OP_GENERAL_JAC ? 1 : 0);
      break;
    default: return (int)hipErrorInvalidValue;
  }
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// line search support for the solver on wide models: objective and ||c||_1 at the trial points z + 2^-k dz.
// One workgroup per instance, one wavefront per knot (strided); the residual is linear in alpha except for the few
// nonlinear rows, so the constant-table products are formed once per knot.
// ---------------------------------------------------------------------------------------------------
template <class M>
__global__ __launch_bounds__(WG) void k_wide_merit(dto_wide_args a) {
  constexpr int N = M::WIDE_N, NU = M::WIDE_NU, NC = 2 * N + NU;
  extern __shared__ double sm[];
  double* fe_s = sm;                       // [N][NC]
  double* per = sm + N * NC;               // per wave: p(N+NU+1) y(N) dp(N+NU+1) dy(N) pk(N+NU+1) yk(N) nl(8)
  constexpr int PW = 3 * (N + NU + 1) + 3 * N + 16;
  double* red = per + 4 * PW;              // [4][2*TRIALS]
  const int w = wave_id(), l = lane_id();
  const int64_t b = blockIdx.x;
  if (a.active && !a.active[b]) return;
  const int staged = staged_dyn_class<M>(a.kind, a.T);
  {
    const double* fe = fe_const_of<M>(staged);
    for (int i = threadIdx.x; i < N * NC; i += WG) fe_s[i] = fe[i];
  }
  __syncthreads();
  double* pv = per + w * PW; double* yv = pv + N + NU + 1; double* dp = yv + N; double* dy = dp + N + NU + 1;
  double* pk = dy + N; double* yk = pk + N + NU + 1; double* nl = yk + N;
  const double* z = a.z + b * a.ldz;
  const double* dz = a.dz + b * a.lddz;
  double facc[DTO_WIDE_TRIALS], tacc[DTO_WIDE_TRIALS];
#pragma unroll
  for (int k = 0; k < DTO_WIDE_TRIALS; ++k) facc[k] = tacc[k] = 0.0;
  for (int t = w; t < a.T; t += 4) {
    const int wk = M::wk_of_kind(a.kind[t]);
    const double* wp = a.params + b * a.ldw + a.woff[t];
    M::dispatch_wk(wk, [&](auto wkc) {
      using KD = typename M::template WKind<decltype(wkc)::value>;
      using CO = typename M::template Cost<KD::COST>;
      constexpr bool HAS_DYN = KD::DYN >= 0;
      constexpr int NUK = HAS_DYN ? NU : 0;
      pv[l] = z[a.zoff[t] + l]; dp[l] = dz[a.zoff[t] + l];
      if (l < NUK) { pv[N + l] = z[a.zoff[t] + N + l]; dp[N + l] = dz[a.zoff[t] + N + l]; }
      double lin0 = 0.0, lind = 0.0;
      if constexpr (HAS_DYN) {
        yv[l] = z[a.zoff[t + 1] + l]; dy[l] = dz[a.zoff[t + 1] + l];
        wave_lds_fence();   // the row products below read every lane's entries
        using DYc = typename M::template Dyn<KD::DYN>;
        const double* row = fe_s + l * NC;
        if constexpr (M::N_DYN > 1) { if (KD::DYN != staged) row = DYc::fe_const() + l * NC; }
        // (the four 64-term products in one pass over the row, eight terms in flight: as four fully unrolled dot_rr calls this
        //  kernel spilled 360 registers)
        double l0a = 0.0, l0b = 0.0, lda_ = 0.0, ldb_ = 0.0;
#pragma unroll 8
        for (int c = 0; c < N; ++c) {
          const double rx = row[c], ry = row[N + NU + c];
          l0a += rx * pv[c]; l0b += ry * yv[c]; lda_ += rx * dp[c]; ldb_ += ry * dy[c];
        }
        lin0 = l0a + l0b;
        lind = lda_ + ldb_;
#pragma unroll
        for (int j = 0; j < NU; ++j) { lin0 += row[N + j] * pv[N + j]; lind += row[N + j] * dp[N + j]; }
      }
      // trial steps alpha_pmax 2^-k (alpha_pmax = 1 without finite bounds); with bounds phi is the barrier objective
      const bool barrier = a.zl != nullptr;
      const double mub = (barrier && a.mu_inst) ? a.mu_inst[b] : 0.0;
      double alpha = barrier ? a.stats[b * DTO_WIDE_NSTAT + DTO_WIDE_APMAX] : 1.0;
      double lo_l = 0.0, hi_l = 0.0, lo_u = 0.0, hi_u = 0.0;
      if (barrier) {
        lo_l = a.fixed_lo[a.zoff[t] + l]; hi_l = a.fixed_hi[a.zoff[t] + l];
        if (l < NUK) { lo_u = a.fixed_lo[a.zoff[t] + N + l]; hi_u = a.fixed_hi[a.zoff[t] + N + l]; }
      }
#pragma unroll 1
      for (int k = 0; k < DTO_WIDE_TRIALS; ++k) {
        pk[l] = pv[l] + alpha * dp[l];
        if (l < NUK) pk[N + l] = pv[N + l] + alpha * dp[N + l];
        if constexpr (HAS_DYN) yk[l] = yv[l] + alpha * dy[l];
        wave_lds_fence();   // the trial point is complete in LDS before the model code reads it (see wave_lds_fence)
        CO::eval(pk, pk + N, wp, nl + 8);
        if (l == 0) facc[k] += nl[8];
        if (barrier) {
          double lb = 0.0;
          const double xk = pv[l] + alpha * dp[l];
          if (lo_l != hi_l) {
            if (lo_l > -1e300) lb += log(xk - lo_l);
            if (hi_l < 1e300) lb += log(hi_l - xk);
          }
          if (l < NUK && lo_u != hi_u) {
            const double uk = pv[N + l] + alpha * dp[N + l];
            if (lo_u > -1e300) lb += log(uk - lo_u);
            if (hi_u < 1e300) lb += log(hi_u - uk);
          }
          lb = wave_sum(lb);
          if (l == 0) facc[k] -= mub * lb;
        }
        if constexpr (HAS_DYN) {
          using DY = typename M::template Dyn<KD::DYN>;
          DY::eval_nl(pk, pk + N, yk, wp, nl);
          double r = lin0 + alpha * lind;
#pragma unroll
          for (int q = 0; q < DY::NNL; ++q)
            if (DY::nl_row(q) == l) r += nl[q];
          const double sum = wave_sum(fabs(r));
          if (l == 0) tacc[k] += sum;
        }
        wave_lds_fence();   // the next trial rewrites the point
        alpha *= 0.5;
      }
    });
  }
  if (l == 0) {
#pragma unroll
    for (int k = 0; k < DTO_WIDE_TRIALS; ++k) { red[w * 2 * DTO_WIDE_TRIALS + 2 * k] = facc[k]; red[w * 2 * DTO_WIDE_TRIALS + 2 * k + 1] = tacc[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * DTO_WIDE_TRIALS) {
    double v = 0.0;
    for (int ww = 0; ww < 4; ++ww) v += red[ww * 2 * DTO_WIDE_TRIALS + threadIdx.x];
    a.merit[b * 2 * DTO_WIDE_TRIALS + threadIdx.x] = v;
  }
}

template <class M>
int wide_info(dto_wide_info* out) {
  using D = Dims<M::WIDE_N, M::WIDE_NU>;
  out->supported = 1;
  out->n = M::WIDE_N;
  out->nu = M::WIDE_NU;
  out->fac_stage = D::FAC;
  out->lds_bytes = StepLds<M>::BYTES;
  return 0;
}

template <class M>
int launch_wide(int op, const dto_wide_args* a, void* stream) {
  if (op == DTO_WIDE_MERIT) {
    constexpr int N = M::WIDE_N, NU = M::WIDE_NU;
    const int lds = (int)sizeof(double) * (N * (2 * N + NU) + 4 * (3 * (N + NU + 1) + 3 * N + 16) + 4 * 2 * DTO_WIDE_TRIALS);
    hipError_t em = hipFuncSetAttribute((const void*)k_wide_merit<M>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (em != hipSuccess) return (int)em;
    hipLaunchKernelGGL(k_wide_merit<M>, dim3((unsigned)a->B), dim3(WG), lds, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
  }
  if (op != DTO_WIDE_STEP) return (int)hipErrorInvalidValue;
  dto_wide_info info;
  wide_info<M>(&info);
  if (a->zl) {
    hipError_t eb = hipFuncSetAttribute((const void*)k_wide_step<M, true>, hipFuncAttributeMaxDynamicSharedMemorySize, info.lds_bytes);
    if (eb != hipSuccess) return (int)eb;
    hipLaunchKernelGGL((k_wide_step<M, true>), dim3((unsigned)a->B), dim3(WG), info.lds_bytes, (hipStream_t)stream, *a);
#if DTO_WIDE_SPLIT_BWD
    eb = hipFuncSetAttribute((const void*)k_wide_bwd<M, true>, hipFuncAttributeMaxDynamicSharedMemorySize, BwdLds<M>::BYTES);
    if (eb != hipSuccess) return (int)eb;
    hipLaunchKernelGGL((k_wide_bwd<M, true>), dim3((unsigned)a->B), dim3(WG), BwdLds<M>::BYTES, (hipStream_t)stream, *a);
#endif
    return (int)hipGetLastError();
  }
  hipError_t e = hipFuncSetAttribute((const void*)k_wide_step<M, false>, hipFuncAttributeMaxDynamicSharedMemorySize, info.lds_bytes);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((k_wide_step<M, false>), dim3((unsigned)a->B), dim3(WG), info.lds_bytes, (hipStream_t)stream, *a);
#if DTO_WIDE_SPLIT_BWD
  e = hipFuncSetAttribute((const void*)k_wide_bwd<M, false>, hipFuncAttributeMaxDynamicSharedMemorySize, BwdLds<M>::BYTES);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((k_wide_bwd<M, false>), dim3((unsigned)a->B), dim3(WG), BwdLds<M>::BYTES, (hipStream_t)stream, *a);
#endif
  return (int)hipGetLastError();
}

}  // namespace wide
}  // namespace dto
