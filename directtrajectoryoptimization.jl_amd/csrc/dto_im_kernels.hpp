// Instance-major ("IM") engine of the interior-point / KKT path: the same iteration as dto_kkt_kernels.hpp (whose block
// algebra, convergence test, inertia ladder and filter line search it calls), on a data layout and a schedule in which an
// instance pays only for its own work.
//
// Why a second engine.  In the SoA-tile engine 64 instances share a wavefront for good: the sequential forward sweep of a
// tile is repeated until its SLOWEST lane has a factorisation with the right inertia (measured, acrobot T = 1000: 3.7
// sweep rounds per tile for 2.0 factorisations per instance; 45-60 % of the dominant kernel was lanes waiting), and a tile
// keeps all its wavefronts until its last instance has converged.  Here
//   * every per-instance vector is stored instance-major as one record per stage (reference order x_1,u_1,x_2,...:
//     src/dynamics.jl:188-195), so ANY 64 instances can share a wavefront: lane -> instance goes through a work list;
//   * an instance is a small state machine (phase EVAL -> FACT -> STEP -> EVAL ...).  One pass of the host loop =
//     evaluate the instances that took a step, ONE factorisation attempt for every instance that needs one (first
//     attempt or next rung of the delta_w ladder), and back substitution + line search + update for exactly the instances
//     whose attempt succeeded.  Work lists are rebuilt on the device before each group of kernels; there is no host
//     round trip and no lock step between instances: the forward sweep costs (attempts) x (one sweep), not (deepest
//     ladder of the tile) x (one sweep);
//   * stage-parallel kernels (residuals, line search, update) map lane = knot, wavefront = 64 consecutive knots of one
//     instance, exactly like the callback kernels of dto_eval_kernels.hpp; the sweeps map lane = instance.
// Scope: exact-Hessian models (the sweeps re-evaluate the derivatives: KindDims::FUSED), the sequential sweep (large
// batches).  Quasi-Newton models, time-partitioned sweeps for small batches and the linear-solver entry points stay on the
// SoA engine.
//
// Records of stage t of one instance (kind K; doubles; every record padded to an even count = 16 bytes):
//   A  [ p (NP) | lam (NY) | nu (Q) ]                 iterate and multipliers
//   R  [ r_p (NP) | d (NY) | c (Q) ]                  residuals (same offsets as A)
//   D  [ dp (NP) | dlam (NY) | dnu (Q) | ds (QI) ]    step
//   C  [ P (NX(NX+1)/2) | py (NX) ]                   carry-in of the forward recursion
//   Bd [ zl (NP) | zu (NP) | s (QI) | zs (QI) ]       bound multipliers and slacks (only allocated when there are any)
#pragma once

#include <algorithm>

#include "dto_kkt_kernels.hpp"

enum dto_im_op {
  DTO_IM_INIT = 0,        // guess (instance-major z of the C-ABI) -> A records, bound push, slacks, multipliers, scalars
  DTO_IM_UNPACK = 1,      // records -> instance-major vectors of the C-ABI
  DTO_IM_COMPACT = 2,     // work list of the instances in a given phase
  DTO_IM_EVAL = 3,
  DTO_IM_CONV = 4,
  DTO_IM_FWD = 5,
  DTO_IM_BWD = 6,
  DTO_IM_LINESEARCH = 7,
  DTO_IM_LS_REDUCE = 8,
  DTO_IM_UPDATE = 9,
  DTO_IM_COUNT = 10,      // how many instances are still running / still have work below the iteration target
  DTO_IM_OP_COUNT
};

enum dto_im_phase { DTO_IM_PH_EVAL = 0, DTO_IM_PH_FACT = 1, DTO_IM_PH_STEP = 2 };

struct dto_im_info {
  int supported;
  int n_kind;
  int a_size[16], d_size[16], c_size[16], b_size[16], n_ineq[16];  // doubles per stage record, by kind
  int npart, nscal, ls_trials, filter_cap;
  int own_eval, own_ls;  // stages a wavefront of the residual / line-search kernels owns
};

struct dto_im_args {
  int T;
  int64_t B;
  int64_t Nz, Nc, Ni, Nw;
  int64_t n_mult, n_bnd;
  const int* kind; const int* zoff; const int* woff; const int* cdoff; const int* ccoff; const int* ioff;
  const int* aoff; const int* doff; const int* coff; const int* boff;  // [T+1] record offsets inside one instance (doubles)
  int64_t a_total, d_total, c_total, b_total;                         // doubles per instance
  const double* lo; const double* hi;   // [Nz] shared variable bounds
  const double* params;                 // shared parameters
  const double* wpi; int64_t ldw;       // per-instance parameters [B][ldw] or NULL
  double* A; double* R; double* D; double* C; double* Bd;
  double* part; double* lspart; double* scal; double* filt;
  int* phase;                           // [B]
  int nwin_e, nwin_l;                   // wavefront windows per instance of the residual / line-search + update kernels
  const int* list; const int* count;    // work list of this launch: count[0] instances
  int* list_out; int* count_out; int phase_sel;   // DTO_IM_COMPACT
  int iter_target;                      // instances that have done this many iterations are not evaluated again (< 0: no limit)
  int* ticket;                          // sweeps: next chunk of 64 list entries (dynamic distribution over the resident wavefronts)
  int* running;                         // DTO_IM_COUNT: [0] running instances, [1] of them: work left below iter_target
  int sweep_occ;                        // wavefronts per SIMD of the sweep kernels (2 or 1: register budget 256 / 512)
  const double* aos_in; double* aos_out; int64_t ld_aos; int aos_which;
  dto_solver_opts opt;
};

namespace dto {
namespace im {

constexpr int OWN_EVAL = WAVE - 1;  // lane 0 of a residual wavefront is the halo stage t0 - 1 (hands E' lam to lane 1)
constexpr int OWN_LS = WAVE;

__host__ __device__ constexpr int even(int n) { return (n + 1) & ~1; }

template <class M, int K>
struct Rec {
  using D = KindDims<M, K>;
  static constexpr int NP = D::NP, NY = D::NY, Q = D::Q, QI = D::QI, NX = D::NX, BD = D::BD;
  static constexpr int A_LAM = NP, A_NU = NP + NY, AS = even(BD);
  static constexpr int D_DS = BD, DS = even(BD + QI);
  static constexpr int CS = even(NX * (NX + 1) / 2 + NX);
  static constexpr int B_ZU = NP, B_S = 2 * NP, B_ZS = 2 * NP + QI, BS = even(2 * NP + 2 * QI);
};

template <class M, int K = 0>
void fill_info(dto_im_info* o) {
  if constexpr (K < M::N_KIND) {
    using RC = Rec<M, K>;
    o->a_size[K] = RC::AS;
    o->d_size[K] = RC::DS;
    o->c_size[K] = RC::CS;
    o->b_size[K] = RC::BS;
    o->n_ineq[K] = RC::QI;
    fill_info<M, K + 1>(o);
  }
}

template <class M>
int im_info(dto_im_info* out) {
  out->supported = (M::N_KIND <= 16 && !M::HAS_GENERAL && M::EVALUATE_HESSIAN != 0) ? 1 : 0;
  out->n_kind = M::N_KIND;
  for (int i = 0; i < 16; ++i) out->a_size[i] = out->d_size[i] = out->c_size[i] = out->b_size[i] = out->n_ineq[i] = 0;
  fill_info<M>(out);
  out->npart = DTO_NPART;
  out->nscal = SC_COUNT;
  out->ls_trials = DTO_LS_TRIALS;
  out->filter_cap = DTO_FILTER_CAP;
  out->own_eval = OWN_EVAL;
  out->own_ls = OWN_LS;
  return 0;
}

template <class M, int K = 0>
constexpr int max_as() {
  if constexpr (K < M::N_KIND) {
    constexpr int rest = max_as<M, K + 1>();
    return Rec<M, K>::AS > rest ? Rec<M, K>::AS : rest;
  } else {
    return 2;
  }
}

template <class M, int K = 0>
constexpr int max_ds() {
  if constexpr (K < M::N_KIND) {
    constexpr int rest = max_ds<M, K + 1>();
    return Rec<M, K>::DS > rest ? Rec<M, K>::DS : rest;
  } else {
    return 2;
  }
}

// parameters of stage t of an instance
template <int N>
__device__ __forceinline__ void im_params(arr<N>& w, const dto_im_args& a, int64_t inst, int t) {
  const double* src = a.wpi ? a.wpi + inst * a.ldw + a.woff[t] : a.params + a.woff[t];
#pragma unroll
  for (int i = 0; i < N; ++i) w[i] = src[i];
}

// ------------------------------------------------------------------------------------------------
// stage data of the sweeps (lane = instance): the IO interface of stage_factor / stage_forward / stage_backward over the
// instance-major records, every lane addressing its own instance
// ------------------------------------------------------------------------------------------------
template <class M, int K>
struct DirectIO {
  using D = KindDims<M, K>;
  using RC = Rec<M, K>;
  const dto_im_args& a;
  const int64_t inst;
  const int t, z0;
  const double* Ap;
  const double* An;   // A record of stage t + 1 (its leading NX entries are y = x_{t+1})
  const double* Rp;
  double* Cp;
  double* Dp;
  const double* Bp;
  __device__ __forceinline__ DirectIO(const dto_im_args& a_, int64_t inst_, int t_)
      : a(a_), inst(inst_), t(t_), z0(a_.zoff[t_]), Ap(a_.A + inst_ * a_.a_total + a_.aoff[t_]),
        An(a_.A + inst_ * a_.a_total + a_.aoff[t_ + 1]), Rp(a_.R + inst_ * a_.a_total + a_.aoff[t_]),
        Cp(a_.C + inst_ * a_.c_total + a_.coff[t_]), Dp(a_.D + inst_ * a_.d_total + a_.doff[t_]),
        Bp(a_.Bd ? a_.Bd + inst_ * a_.b_total + a_.boff[t_] : nullptr) {}
  __device__ __forceinline__ double rec(int e) const { return Rp[e]; }   // exact-Hessian records: [r_p | d | c]
  __device__ __forceinline__ double p(int i) const { return Ap[i]; }
  __device__ __forceinline__ double y(int i) const { return An[i]; }
  __device__ __forceinline__ double lam(int k) const { return Ap[RC::A_LAM + k]; }
  __device__ __forceinline__ double nu(int j) const { return Ap[RC::A_NU + j]; }
  template <int N>
  __device__ __forceinline__ void params(arr<N>& w) const { im_params(w, a, inst, t); }
  __device__ __forceinline__ void bounds(StageBounds<D::NP>& b) const {
    const bool duals = Bp != nullptr;
#pragma unroll
    for (int i = 0; i < D::NP; ++i) {
      b.lo[i] = a.lo[z0 + i];
      b.hi[i] = a.hi[z0 + i];
      b.p[i] = Ap[i];
      b.zl[i] = duals ? Bp[i] : 0.0;
      b.zu[i] = duals ? Bp[RC::B_ZU + i] : 0.0;
    }
  }
  __device__ __forceinline__ bool has_sigx() const { return false; }
  __device__ __forceinline__ bool has_sigc() const { return false; }
  __device__ __forceinline__ double sigx(int) const { return 0.0; }
  __device__ __forceinline__ double sigc_con(int) const { return 0.0; }
  __device__ __forceinline__ double sigc_dyn(int) const { return 0.0; }
  __device__ __forceinline__ double slack(int j) const { return Bp[RC::B_S + D::slack(j)]; }
  __device__ __forceinline__ double slack_mult(int j) const { return Bp[RC::B_ZS + D::slack(j)]; }
  static constexpr bool PAIR_CARRY = false;
  __device__ __forceinline__ void put_carry(int i, double v) const { Cp[i] = v; }
  __device__ __forceinline__ double carry(int i) const { return Cp[i]; }
  __device__ __forceinline__ void put_dp(int i, double v) const { Dp[i] = v; }
  __device__ __forceinline__ void put_dlam(int k, double v) const { Dp[RC::A_LAM + k] = v; }
  __device__ __forceinline__ void put_dnu(int j, double v) const { Dp[RC::A_NU + j] = v; }
  __device__ __forceinline__ void put_ds(int j, double v) const { Dp[RC::D_DS + D::slack(j)] = v; }
  __device__ __forceinline__ long long* prof() const { return nullptr; }
};

// out-of-line forms for the heavy stage kinds (see stage_forward_cold in dto_kkt_kernels.hpp).  TAG: one copy per calling
// kernel variant -- a callee shared by kernels with different register budgets is compiled for the laxest of them, and its
// register count then lowers the occupancy of the stricter kernel
template <class M, int K, int TAG>
__device__ __attribute__((noinline)) void im_forward_cold(const dto_im_args& a, int64_t inst, int t, double mu, double dw, double gam,
                                                          bool need, Carry<M>* cy, int* okneg) {
  bool ok = okneg[0] != 0;
  int nneg = okneg[1];
  Spike<M> sp;
  stage_forward<M, K, false>(a.opt, DirectIO<M, K>(a, inst, t), mu, dw, gam, false, need, *cy, sp, ok, nneg, okneg[2] != 0);
  okneg[0] = ok ? 1 : 0;
  okneg[1] = nneg;
}
template <class M, int K, int TAG>
__device__ __attribute__((noinline)) void im_backward_cold(const dto_im_args& a, int64_t inst, int t, double mu, double tau, double dw,
                                                           double gam, double* xn, StepAcc* acc) {
  StepAcc ac = *acc;
  double xl[M::MAX_NX], xv[M::MAX_NX];
#pragma unroll
  for (int i = 0; i < M::MAX_NX; ++i) {
    xl[i] = 0.0;
    xv[i] = xn[i];
  }
  stage_backward<M, K, false>(a.opt, DirectIO<M, K>(a, inst, t), mu, tau, dw, gam, false, xl, xv, ac);
#pragma unroll
  for (int i = 0; i < M::MAX_NX; ++i) xn[i] = xv[i];
  *acc = ac;
}

// ------------------------------------------------------------------------------------------------
// work lists.  grid: ceil(B / 256) blocks of 256; an instance is listed when it is running, in the selected phase and
// (phase EVAL only) still below the iteration target.  Order inside the list is not deterministic (one atomic per
// wavefront) and does not matter: instances never interact.
// ------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void kim_compact(dto_im_args a) {
  const int64_t inst = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool take = false;
  if (inst < a.B) {
    const double* sc = a.scal + inst * SC_COUNT;
    take = sc[SC_STATUS] == 0.0 && a.phase[inst] == a.phase_sel;
    if (take && a.phase_sel == DTO_IM_PH_EVAL && a.iter_target >= 0) take = sc[SC_ITER] < (double)a.iter_target;
  }
  const unsigned long long m = __ballot(take);
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == 0 && m) base = atomicAdd(a.count_out, __popcll(m));
  base = __shfl(base, 0, WAVE);
  if (take) a.list_out[base + __popcll(m & ((1ull << lane) - 1ull))] = (int)inst;
}

static __global__ __launch_bounds__(256) void kim_count(dto_im_args a) {
  const int64_t inst = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool run = false, work = false;
  if (inst < a.B) {
    const double* sc = a.scal + inst * SC_COUNT;
    run = sc[SC_STATUS] == 0.0;
    work = run && (a.phase[inst] != DTO_IM_PH_EVAL || a.iter_target < 0 || sc[SC_ITER] < (double)a.iter_target);
  }
  const unsigned long long mr = __ballot(run), mw = __ballot(work);
  if ((threadIdx.x & 63) == 0) {
    if (mr) atomicAdd(a.running, __popcll(mr));
    if (mw) atomicAdd(a.running + 1, __popcll(mw));
  }
}

// ------------------------------------------------------------------------------------------------
// initialisation (as k_init): lane = knot.  grid = B * nwin_l single-wave blocks.
// ------------------------------------------------------------------------------------------------
template <class M>
__global__ __launch_bounds__(WAVE) void kim_init(dto_im_args a) {
  const int64_t inst = blockIdx.x / a.nwin_l;
  const int t = (blockIdx.x % a.nwin_l) * OWN_LS + threadIdx.x;
  const dto_solver_opts& o = a.opt;
  double* sc = a.scal + inst * SC_COUNT;
  const double mu_prev = sc[SC_MU];
  const double mu0 = !o.warm ? o.mu_init : (o.mu_warm > 0.0 ? o.mu_warm : (mu_prev > 0.0 ? mu_prev : o.mu_init));
  if (t < a.T) {
    dispatch_kind<M>(a.kind[t], [&](auto kc) {
      constexpr int K = decltype(kc)::value;
      using D = KindDims<M, K>;
      using KD = typename D::KD;
      using RC = Rec<M, K>;
      const int z0 = a.zoff[t];
      double* Ap = a.A + inst * a.a_total + a.aoff[t];
      double* Bp = a.Bd ? a.Bd + inst * a.b_total + a.boff[t] : nullptr;
      arr<D::NP> p;
#pragma unroll
      for (int i = 0; i < D::NP; ++i) {
        double v = a.aos_in ? a.aos_in[inst * a.ld_aos + z0 + i] : Ap[i];
        const double lo = a.lo[z0 + i], hi = a.hi[z0 + i];
        double zl = 0.0, zu = 0.0;
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
          if (o.warm && Bp) {
            const double pl = Bp[i], pu = Bp[RC::B_ZU + i];
            if (fl && pl > 0.0) zl = pl;
            if (fh && pu > 0.0) zu = pu;
          }
        }
        p[i] = v;
        Ap[i] = v;
        if (Bp) {
          Bp[i] = zl;
          Bp[RC::B_ZU + i] = zu;
        }
      }
      if constexpr (KD::DYN >= 0) {
        if (!o.warm) {
#pragma unroll
          for (int i = 0; i < D::NY; ++i) Ap[RC::A_LAM + i] = 0.0;
        }
      }
      if constexpr (KD::CON >= 0) {
        using C = typename M::template Con<KD::CON>;
        arr<C::NW> w; arr<C::NC> c;
        im_params(w, a, inst, t);
        C::eval(p.data(), p.data() + C::NX, w.data(), c.data());
#pragma unroll
        for (int j = 0; j < C::NC; ++j) {
          double nu = o.warm ? Ap[RC::A_NU + j] : 0.0;
          if (D::ineq(j)) {
            double sv = fmax(-c[j], o.bound_push * fmax(1.0, fabs(c[j])));
            double zv = mu0 / sv;
            if (o.warm) {
              const double ps = Bp[RC::B_S + D::slack(j)], pz = Bp[RC::B_ZS + D::slack(j)];
              if (ps > 0.0 && pz > 0.0) { sv = ps; zv = pz; } else nu = zv;
            } else {
              nu = zv;
            }
            Bp[RC::B_S + D::slack(j)] = sv;
            Bp[RC::B_ZS + D::slack(j)] = zv;
          }
          Ap[RC::A_NU + j] = nu;
        }
      }
    });
  }
  if (blockIdx.x % a.nwin_l == 0 && threadIdx.x == 0) {
    sc[SC_STATUS] = 0.0; sc[SC_ITER] = 0.0; sc[SC_MU] = mu0; sc[SC_PENALTY] = 0.0; sc[SC_LS_MODE] = o.ls_penalty ? 1.0 : 2.0; sc[SC_ASCALE] = 1.0; sc[SC_DELTA_W] = 0.0;
    sc[SC_DELTA_LAST] = 0.0; sc[SC_LS_FAIL] = 0.0; sc[SC_NFACT] = 0.0; sc[SC_ALPHA] = 0.0;
    sc[SC_THETA_MAX] = -1.0; sc[SC_THETA_MIN] = -1.0; sc[SC_FILTER_N] = 0.0; sc[SC_LS_KIND] = 0.0; sc[SC_QN_RESET] = 1.0;
    sc[SC_FULL_STREAK] = 0.0; sc[SC_SHORT_STREAK] = 0.0; sc[SC_WATCHDOG] = 0.0; sc[SC_ACC_COUNT] = 0.0; sc[SC_F_LAST] = 1e300;
    sc[SC_XMAX] = 0.0; sc[SC_NNEG] = 0.0; sc[SC_NEED] = 0.0; sc[SC_GAMMA] = 1.0;
    a.phase[inst] = DTO_IM_PH_EVAL;
  }
}

// ------------------------------------------------------------------------------------------------
// records -> instance-major vectors of the C-ABI (reference order: z = [x_1;u_1;...;x_T], multipliers [dynamics rows;
// stage rows], src/data.jl:64-75).  lane = knot; grid = B * nwin_l.
// which: 0 z, 1 lam, 2 dz, 3 dlam, 5 zl, 6 zu, 7 s, 8 zs, 9 ds
// ------------------------------------------------------------------------------------------------
template <class M>
__global__ __launch_bounds__(WAVE) void kim_unpack(dto_im_args a) {
  const int64_t inst = blockIdx.x / a.nwin_l;
  const int t = (blockIdx.x % a.nwin_l) * OWN_LS + threadIdx.x;
  if (t >= a.T) return;
  double* out = a.aos_out + inst * a.ld_aos;
  dispatch_kind<M>(a.kind[t], [&](auto kc) {
    constexpr int K = decltype(kc)::value;
    using D = KindDims<M, K>;
    using RC = Rec<M, K>;
    const double* Ap = a.A + inst * a.a_total + a.aoff[t];
    const double* Dp = a.D + inst * a.d_total + a.doff[t];
    const double* Bp = a.Bd ? a.Bd + inst * a.b_total + a.boff[t] : nullptr;
    const int w = a.aos_which;
    if (w == 0 || w == 2) {
      const double* src = (w == 0) ? Ap : Dp;
#pragma unroll
      for (int i = 0; i < D::NP; ++i) out[a.zoff[t] + i] = src[i];
    } else if (w == 1 || w == 3) {
      const double* src = (w == 1) ? Ap : Dp;
#pragma unroll
      for (int k = 0; k < D::NY; ++k) out[a.cdoff[t] + k] = src[RC::A_LAM + k];
#pragma unroll
      for (int j = 0; j < D::Q; ++j) out[a.ccoff[t] + j] = src[RC::A_NU + j];
    } else if (w == 5 || w == 6) {
#pragma unroll
      for (int i = 0; i < D::NP; ++i) out[a.zoff[t] + i] = Bp ? Bp[(w == 5 ? 0 : RC::B_ZU) + i] : 0.0;
    } else if (w == 7 || w == 8) {
#pragma unroll
      for (int j = 0; j < D::QI; ++j) out[a.ioff[t] + j] = Bp[(w == 7 ? RC::B_S : RC::B_ZS) + j];
    } else if (w == 9) {
#pragma unroll
      for (int j = 0; j < D::QI; ++j) out[a.ioff[t] + j] = Dp[RC::D_DS + j];
    }
  });
}

// butterfly reductions over the 64 lanes of a wavefront (fixed order: deterministic)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 1; off < WAVE; off <<= 1) v += __shfl_xor(v, off, WAVE);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 1; off < WAVE; off <<= 1) v = fmax(v, __shfl_xor(v, off, WAVE));
  return v;
}

// ------------------------------------------------------------------------------------------------
// residuals of the current iterate (as k_stage_eval, exact-Hessian form): lane = knot.  A wavefront owns OWN_EVAL
// consecutive knots of one instance; lane 0 is the halo knot t0 - 1, which only evaluates its dynamics Jacobian to hand
// E_{t0-1}' lam_{t0-1} to lane 1 (inside the window the hand-over is a lane shift).  The R records of the owned knots are
// contiguous: they are deposited in an LDS image and streamed out with full-line stores.
// grid = n_grid * nwin_e single-wave blocks.
// ------------------------------------------------------------------------------------------------
template <class M>
__global__ __launch_bounds__(WAVE) void kim_eval(dto_im_args a) {
  __shared__ __attribute__((aligned(16))) double s_r[OWN_EVAL * max_as<M>() + 2];
  __shared__ __attribute__((aligned(16))) double s_a[(OWN_EVAL + 2) * max_as<M>() + 2];
  const int64_t n_item = (int64_t)(*a.count) * a.nwin_e;
  for (int64_t item = blockIdx.x; item < n_item; item += gridDim.x) {
  const int e = (int)(item / a.nwin_e);
  const int win = (int)(item % a.nwin_e);
  const int64_t inst = a.list[e];
  const int lane = threadIdx.x;
  const int t0 = win * OWN_EVAL;
  const int tend = min(t0 + OWN_EVAL, a.T);
  const int s = t0 - 1 + lane;
  const bool live = s >= 0 && s < tend;
  const bool own = live && s >= t0;
  const int r0 = a.aoff[t0];
  // the A records of the halo, the owned knots and the knot after them (its head is y of the last owned knot) are one
  // contiguous range of the instance: one coalesced copy into LDS, lanes read their records from there
  const int a0 = a.aoff[t0 > 0 ? t0 - 1 : 0];
  wave_load(s_a, a.A + inst * a.a_total + a0, a.aoff[min(tend + 1, a.T)] - a0);
  __syncthreads();
  double f = 0.0, th1 = 0.0, thinf = 0.0, dinf = 0.0, szmax = 0.0, iszmax = 0.0, sumlam = 0.0, sumz = 0.0, logbar = 0.0, xmax = 0.0;
  double enext[M::MAX_NX];
#pragma unroll
  for (int i = 0; i < M::MAX_NX; ++i) enext[i] = 0.0;
  arr<M::MAX_NXU> rp;
  const int kind = live ? a.kind[s] : -1;
  if (live) {
    dispatch_kind<M>(kind, [&](auto kc) {
      constexpr int K = decltype(kc)::value;
      using D = KindDims<M, K>;
      using KD = typename D::KD;
      using RC = Rec<M, K>;
      using CO = typename M::template Cost<KD::COST>;
      const double* Ap = s_a + (a.aoff[s] - a0);
      arr<D::NP> p;
#pragma unroll
      for (int i = 0; i < D::NP; ++i) p[i] = Ap[i];
      if (own) {
        arr<CO::NW> wc;
        im_params(wc, a, inst, s);
        double o1[1];
        CO::eval(p.data(), p.data() + CO::NX, wc.data(), o1);
        f = o1[0];
        CO::grad(p.data(), p.data() + CO::NX, wc.data(), rp.data());
      }
      if constexpr (KD::DYN >= 0) {
        using DY = typename M::template Dyn<KD::DYN>;
        const double* An = s_a + (a.aoff[s + 1] - a0);
        arr<DY::NY> y, lam, d;
        arr<DY::NW> w;
        im_params(w, a, inst, s);
#pragma unroll
        for (int i = 0; i < DY::NY; ++i) {
          y[i] = An[i];
          lam[i] = Ap[RC::A_LAM + i];
        }
        arr<DY::NJ> jv;
        DY::eval_jac(p.data(), p.data() + DY::NX, y.data(), w.data(), d.data(), jv.data());
        DY::etlam(jv.data(), lam.data(), enext);
        if (own) {
          DY::jtlam(jv.data(), lam.data(), rp.data());
#pragma unroll
          for (int i = 0; i < DY::NY; ++i) {
            s_r[a.aoff[s] - r0 + RC::A_LAM + i] = d[i];
            th1 += fabs(d[i]);
            thinf = fmax(thinf, fabs(d[i]));
            sumlam += fabs(lam[i]);
          }
        }
      }
      if constexpr (KD::CON >= 0) {
        if (own) {
          using C = typename M::template Con<KD::CON>;
          const double* Bp = a.Bd ? a.Bd + inst * a.b_total + a.boff[s] : nullptr;
          arr<C::NW> w; arr<C::NC> c, nu; arr<C::NJ> jv;
          im_params(w, a, inst, s);
#pragma unroll
          for (int j = 0; j < C::NC; ++j) nu[j] = Ap[RC::A_NU + j];
          C::eval(p.data(), p.data() + C::NX, w.data(), c.data());
          C::jac(p.data(), p.data() + C::NX, w.data(), jv.data());
          C::jtlam(jv.data(), nu.data(), rp.data());
#pragma unroll
          for (int j = 0; j < C::NC; ++j) {
            double r = c[j];
            if (D::ineq(j)) {
              const double sv = Bp[RC::B_S + D::slack(j)], zv = Bp[RC::B_ZS + D::slack(j)];
              r = c[j] + sv;
              dinf = fmax(dinf, fabs(nu[j] - zv));
              szmax = fmax(szmax, sv * zv);
              iszmax = fmax(iszmax, 1.0 / (sv * zv));
              sumz += fabs(zv);
              logbar += log(sv);
            }
            s_r[a.aoff[s] - r0 + RC::A_NU + j] = r;
            th1 += fabs(r);
            thinf = fmax(thinf, fabs(r));
            sumlam += fabs(nu[j]);
          }
        }
      }
    });
  }
  // E_{s-1}' lam_{s-1} comes from the lane below (the halo lane for the first owned knot)
  double ecarry[M::MAX_NX];
#pragma unroll
  for (int i = 0; i < M::MAX_NX; ++i) ecarry[i] = __shfl_up(enext[i], 1, WAVE);
  if (own) {
    dispatch_kind<M>(kind, [&](auto kc) {
      constexpr int K = decltype(kc)::value;
      using D = KindDims<M, K>;
      using KD = typename D::KD;
      using RC = Rec<M, K>;
      const double* Ap = s_a + (a.aoff[s] - a0);
      const double* Bp = a.Bd ? a.Bd + inst * a.b_total + a.boff[s] : nullptr;
      if constexpr (KD::PREV >= 0) {
        using DP = typename M::template Dyn<KD::PREV>;
#pragma unroll
        for (int i = 0; i < DP::NY; ++i) rp[i] += ecarry[i];
      }
      const int z0 = a.zoff[s];
#pragma unroll
      for (int i = 0; i < D::NP; ++i) {
        s_r[a.aoff[s] - r0 + i] = rp[i];
        const double pi = Ap[i];
        xmax = fmax(xmax, fabs(pi));
        const double lo = a.lo[z0 + i], hi = a.hi[z0 + i];
        if (lo != hi) {
          const double zl = Bp ? Bp[i] : 0.0, zu = Bp ? Bp[RC::B_ZU + i] : 0.0;
          dinf = fmax(dinf, fabs(rp[i] - zl + zu));
          if (finite_lo(lo)) {
            szmax = fmax(szmax, (pi - lo) * zl);
            iszmax = fmax(iszmax, 1.0 / ((pi - lo) * zl));
            sumz += fabs(zl);
            logbar += log(pi - lo);
          }
          if (finite_hi(hi)) {
            szmax = fmax(szmax, (hi - pi) * zu);
            iszmax = fmax(iszmax, 1.0 / ((hi - pi) * zu));
            sumz += fabs(zu);
            logbar += log(hi - pi);
          }
        }
      }
      if constexpr (RC::AS > RC::BD) s_r[a.aoff[s] - r0 + RC::BD] = 0.0;  // pad
    });
  }
  __syncthreads();
  wave_store_image(a.R + inst * a.a_total + r0, s_r, a.aoff[tend] - r0, lane);
  // partial sums / maxima of this window, fixed-order butterfly
  f = wave_sum(own ? f : 0.0);
  th1 = wave_sum(th1);
  thinf = wave_max(thinf);
  dinf = wave_max(dinf);
  szmax = wave_max(szmax);
  iszmax = wave_max(iszmax);
  sumlam = wave_sum(sumlam);
  sumz = wave_sum(sumz);
  logbar = wave_sum(logbar);
  xmax = wave_max(xmax);
  if (lane == 0) {
    double* part = a.part + (inst * a.nwin_e + win) * DTO_NPART;
    part[0] = f; part[1] = th1; part[2] = thinf; part[3] = dinf; part[4] = szmax; part[5] = iszmax; part[6] = sumlam;
    part[7] = sumz; part[8] = logbar; part[9] = xmax;
  }
  __syncthreads();   // the image is rewritten by the next item
  }
}

// convergence test + barrier update + factorisation request: lane = list entry.  grid = ceil(n_grid / 64).
static __global__ __launch_bounds__(WAVE) void kim_conv(dto_im_args a) {
  const int n = *a.count;
  for (int e = blockIdx.x * WAVE + threadIdx.x; e < n; e += gridDim.x * WAVE) {
  const int64_t inst = a.list[e];
  double* sc = a.scal + inst * SC_COUNT;
  ConvSums cs{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int w = 0; w < a.nwin_e; ++w) {
    const double* part = a.part + (inst * a.nwin_e + w) * DTO_NPART;
    cs.f += part[0];
    cs.th1 += part[1];
    cs.thinf = fmax(cs.thinf, part[2]);
    cs.dinf = fmax(cs.dinf, part[3]);
    cs.szmax = fmax(cs.szmax, part[4]);
    cs.iszmax = fmax(cs.iszmax, part[5]);
    cs.slam += part[6];
    cs.sz += part[7];
    cs.lb += part[8];
    cs.xmax = fmax(cs.xmax, part[9]);
  }
  conv_body<0>(a.opt, sc, cs, a.n_mult, a.n_bnd);
  if (sc[SC_STATUS] == 0.0) a.phase[inst] = DTO_IM_PH_FACT;
  }
}

// ------------------------------------------------------------------------------------------------
// Record staging of the sweeps.  A lane's stage record is a few contiguous 16-byte pieces somewhere in HBM (its instance);
// 64 lanes reading "their own" record one double at a time means 64 partly used sectors per load and read-modify-write
// stores (measured with exactly that form: the backward sweep 4x, the forward sweep 1.5x slower per sweep than the SoA
// engine's).  Instead the WAVE moves the 64 records together: wave-instruction k moves the pieces q = 64 k + lane, piece q
// being piece (q mod NPC) of the record of lane (q div NPC) -- neighbouring lanes move neighbouring 16 bytes of one record,
// every fetched sector is used whole -- by LDS-DMA (global_load_lds_dwordx4: no destination registers, the source address
// is per lane, the LDS image is piece-linear = record-contiguous with pitch NPC * 16 bytes) one stage AHEAD of its use, so
// that the copy of stage t+1's records runs under the arithmetic of stage t.  A lane then reads its record out of LDS.
// Stores go the other way: lanes deposit their record in LDS, the wave stores the pieces with full 16-byte lanes.
// Stage kinds with outsized blocks (heavy_kind: the end stages that carry the pin constraints) are not staged: they run
// out of line on direct loads (DirectIO) -- sizing the LDS slots for them would cost the hot stages their occupancy.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// NPC: 16-byte pieces per record (record doubles / 2).  `inst`: the lane's instance; `slot`: wave-uniform LDS address.
template <int NPC>
__device__ __forceinline__ void gather_records(double* slot, const double* base, int64_t stride, int off, int inst) {
#pragma unroll
  for (int k = 0; k < NPC; ++k) {
    const int q = k * WAVE + (int)threadIdx.x;
    const int r = q / NPC, pc = q - r * NPC;
    const int64_t ir = __shfl(inst, r, WAVE);
    const double* src = base + ir * stride + off + 2 * pc;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(slot + k * (2 * WAVE)), 16, 0, 0);
  }
}
// the reverse: records deposited in `slot` (lane l at slot + l * 2 * NPC doubles) go to HBM; `mask`: lanes whose record is stored
template <int NPC>
__device__ __forceinline__ void scatter_records(const double* slot, double* base, int64_t stride, int off, int inst,
                                                unsigned long long mask) {
#pragma unroll
  for (int k = 0; k < NPC; ++k) {
    const int q = k * WAVE + (int)threadIdx.x;
    const int r = q / NPC, pc = q - r * NPC;
    const int64_t ir = __shfl(inst, r, WAVE);
    const double2 v = *reinterpret_cast<const double2*>(slot + 2 * q);
    if ((mask >> r) & 1ull) *reinterpret_cast<double2*>(base + ir * stride + off + 2 * pc) = v;
  }
}

// slot sizes: the largest record among the kinds that are staged
template <class M, int K = 0>
constexpr int hot_max(int which) {
  if constexpr (K < M::N_KIND) {
    using RC = Rec<M, K>;
    const int rest = hot_max<M, K + 1>(which);
    const int mine = heavy_kind<M, K>() ? 0 : (which == 0 ? RC::AS : (which == 1 ? RC::CS : RC::DS));
    return mine > rest ? mine : rest;
  } else {
    return 2;
  }
}
// A-record size of a kind if it is staged, 0 if it is a heavy kind (wave-uniform run-time lookup)
template <class M, int K = 0>
__device__ __forceinline__ int staged_as(int kind) {
  if constexpr (K < M::N_KIND) return (kind == K) ? (heavy_kind<M, K>() ? 0 : Rec<M, K>::AS) : staged_as<M, K + 1>(kind);
  else return 0;
}

// stage data out of LDS (register copies made at construction: the slots are refilled right afterwards)
template <class M, int K>
struct LdsIO {
  using D = KindDims<M, K>;
  using RC = Rec<M, K>;
  const dto_im_args& a;
  const int64_t inst;
  const int t, z0;
  double av[RC::AS], rv[RC::AS], yv[D::NY > 0 ? D::NY : 1], cv[RC::CS];
  double* dstage;      // the lane's D record in the LDS staging slot (backward sweep), else NULL
  const double* Bp;
  // la / lr / lc: the lane's records in LDS (lc NULL in the forward sweep); ly: x_{t+1} in LDS or registers
  __device__ __forceinline__ LdsIO(const dto_im_args& a_, int64_t inst_, int t_, const double* la, const double* lr, const double* lc,
                                   const double* ly, double* dstage_)
      : a(a_), inst(inst_), t(t_), z0(a_.zoff[t_]), dstage(dstage_),
        Bp(a_.Bd ? a_.Bd + inst_ * a_.b_total + a_.boff[t_] : nullptr) {
#pragma unroll
    for (int i = 0; i < RC::AS; ++i) av[i] = la[i];
#pragma unroll
    for (int i = 0; i < RC::AS; ++i) rv[i] = lr[i];
    if (lc) {
#pragma unroll
      for (int i = 0; i < RC::CS; ++i) cv[i] = lc[i];
    }
    if constexpr (D::NY > 0) {
#pragma unroll
      for (int i = 0; i < D::NY; ++i) yv[i] = ly[i];
    }
  }
  __device__ __forceinline__ double rec(int e) const { return rv[e]; }
  __device__ __forceinline__ double p(int i) const { return av[i]; }
  __device__ __forceinline__ double y(int i) const { return yv[i]; }
  __device__ __forceinline__ double lam(int k) const { return av[RC::A_LAM + k]; }
  __device__ __forceinline__ double nu(int j) const { return av[RC::A_NU + j]; }
  template <int N>
  __device__ __forceinline__ void params(arr<N>& w) const { im_params(w, a, inst, t); }
  __device__ __forceinline__ void bounds(StageBounds<D::NP>& b) const {
    const bool duals = Bp != nullptr;
#pragma unroll
    for (int i = 0; i < D::NP; ++i) {
      b.lo[i] = a.lo[z0 + i];
      b.hi[i] = a.hi[z0 + i];
      b.p[i] = av[i];
      b.zl[i] = duals ? Bp[i] : 0.0;
      b.zu[i] = duals ? Bp[RC::B_ZU + i] : 0.0;
    }
  }
  __device__ __forceinline__ bool has_sigx() const { return false; }
  __device__ __forceinline__ bool has_sigc() const { return false; }
  __device__ __forceinline__ double sigx(int) const { return 0.0; }
  __device__ __forceinline__ double sigc_con(int) const { return 0.0; }
  __device__ __forceinline__ double sigc_dyn(int) const { return 0.0; }
  __device__ __forceinline__ double slack(int j) const { return Bp[RC::B_S + D::slack(j)]; }
  __device__ __forceinline__ double slack_mult(int j) const { return Bp[RC::B_ZS + D::slack(j)]; }
  static constexpr bool PAIR_CARRY = false;
  __device__ __forceinline__ void put_carry(int, double) const {}   // the kernel stores the carry-in itself (staged)
  __device__ __forceinline__ double carry(int i) const { return cv[i]; }
  __device__ __forceinline__ void put_dp(int i, double v) const { dstage[i] = v; }
  __device__ __forceinline__ void put_dlam(int k, double v) const { dstage[RC::A_LAM + k] = v; }
  __device__ __forceinline__ void put_dnu(int j, double v) const { dstage[RC::A_NU + j] = v; }
  __device__ __forceinline__ void put_ds(int j, double v) const { dstage[RC::D_DS + D::slack(j)] = v; }
  __device__ __forceinline__ long long* prof() const { return nullptr; }
};

// issue the LDS-DMA of stage tt's A / R / C records (whichever slot pointers are non-NULL) unless tt is out of range or of a
// heavy kind (those stages load directly)
template <class M>
__device__ __forceinline__ void prefetch_stage(const dto_im_args& a, int tt, int inst, double* slot_a, double* slot_r, double* slot_c) {
  if (tt < 0 || tt >= a.T) return;
  dispatch_uniform<M>(a.kind[tt], [&](auto kc) {
    constexpr int K = decltype(kc)::value;
    using RC = Rec<M, K>;
    if constexpr (!heavy_kind<M, K>()) {
      if (slot_a) gather_records<RC::AS / 2>(slot_a, a.A, a.a_total, a.aoff[tt], inst);
      if (slot_r) gather_records<RC::AS / 2>(slot_r, a.R, a.a_total, a.aoff[tt], inst);
      if (slot_c) gather_records<RC::CS / 2>(slot_c, a.C, a.c_total, a.coff[tt], inst);
    }
  });
}

// ------------------------------------------------------------------------------------------------
// one factorisation attempt of every listed instance (lane = list entry): forward sweep with the (delta_w, gamma) the
// instance's ladder is at, inertia judged at the end; success moves the instance to phase STEP, failure leaves it in
// phase FACT with the next rung.  Chunks of 64 list entries are handed out by a ticket.
// LDS: two A slots (stages t and t+1: y = x_{t+1} is the head of the next record) and one slot that holds R(t) on arrival
// and then stages the carry-in C(t) on its way out.
// ------------------------------------------------------------------------------------------------
// WPS: wavefronts per SIMD the register allocation is asked to fit (2: 256 VGPRs with spills to scratch; 1: 512 registers)
template <class M, int WPS>
__global__ __launch_bounds__(WAVE, WPS) void kim_fwd(dto_im_args a) {
  constexpr int HA = hot_max<M>(0), HC = hot_max<M>(1), HRC = HA > HC ? HA : HC;
  __shared__ __attribute__((aligned(16))) double s_a[2][WAVE * HA];
  __shared__ __attribute__((aligned(16))) double s_rc[WAVE * HRC];
  const int n = *a.count;
  const int lane = threadIdx.x;
  for (;;) {
  // chunks of 64 list entries are handed out dynamically: attempts end early or late (a lane's attempt is lost at the first
  // stage with the wrong pivot signs), and the resident wavefronts should not wait for a static share
  int chunk = 0;
  if (lane == 0) chunk = atomicAdd(a.ticket, 1);
  chunk = __shfl(chunk, 0, WAVE);
  if ((int64_t)chunk * WAVE >= n) return;
  const int e = chunk * WAVE + lane;
  const bool need = e < n;
  const int inst32 = a.list[need ? e : chunk * WAVE];  // lanes beyond the list shadow a listed instance (reads only)
  const int64_t inst = inst32;
  double* sc = a.scal + inst * SC_COUNT;
  const double mu = sc[SC_MU];
  const double dw = sc[SC_TRY_DW], gam = sc[SC_TRY_GAM];
  const bool keep_lost = (int)sc[SC_ATTEMPT] >= a.opt.max_refactor;
  Carry<M> cy;
#pragma unroll
  for (int i = 0; i < M::MAX_NX * (M::MAX_NX + 1) / 2; ++i) cy.P[i] = 0.0;
#pragma unroll
  for (int i = 0; i < M::MAX_NX; ++i) cy.py[i] = 0.0;
  bool ok = true;
  int nneg = 0;
  wait_lds();   // the previous chunk's LDS reads are complete before its slots are refilled
  prefetch_stage<M>(a, 0, inst32, s_a[0], s_rc, nullptr);
  prefetch_stage<M>(a, 1, inst32, s_a[1], nullptr, nullptr);
  for (int t = 0; t < a.T; ++t) {
    wait_vm();   // what was issued a stage ago has landed (and that stage's stores have left)
    const int as_next = (t + 1 < a.T) ? staged_as<M>(a.kind[t + 1]) : 0;
    const unsigned long long store_mask = __ballot(need && (ok || keep_lost));
    dispatch_uniform<M>(a.kind[t], [&](auto kc) {
      constexpr int K = decltype(kc)::value;
      using D = KindDims<M, K>;
      using RC = Rec<M, K>;
      if constexpr (heavy_kind<M, K>()) {
        prefetch_stage<M>(a, t + 1, inst32, nullptr, s_rc, nullptr);
        prefetch_stage<M>(a, t + 2, inst32, s_a[t & 1], nullptr, nullptr);
        Carry<M> cyc = cy;
        int okneg[3] = {ok ? 1 : 0, nneg, keep_lost ? 1 : 0};
        const dto_im_args acold = a;
        im_forward_cold<M, K, WPS>(acold, inst, t, mu, dw, gam, need, &cyc, okneg);
        cy = cyc;
        ok = okneg[0] != 0;
        nneg = okneg[1];
      } else {
        // y = x_{t+1}: head of the next A record -- in LDS if that stage is staged, else straight from HBM
        const double* ly = as_next > 0 ? s_a[(t + 1) & 1] + lane * as_next : a.A + inst * a.a_total + a.aoff[t + 1];
        const LdsIO<M, K> io(a, inst, t, s_a[t & 1] + lane * RC::AS, s_rc + lane * RC::AS, nullptr, ly, nullptr);
        wait_lds();
        // carry-in of this stage (all the backward sweep needs besides the records): staged in the R slot, stored by the wave
        double* cst = s_rc + lane * RC::CS;
#pragma unroll
        for (int i = 0; i < D::NX * (D::NX + 1) / 2; ++i) cst[D::F_P + i] = cy.P[i];
#pragma unroll
        for (int i = 0; i < D::NX; ++i) cst[D::F_PY + i] = cy.py[i];
        if constexpr (RC::CS > D::NX * (D::NX + 1) / 2 + D::NX) cst[RC::CS - 1] = 0.0;
        wait_lds();
        scatter_records<RC::CS / 2>(s_rc, a.C, a.c_total, a.coff[t], inst32, store_mask);
        wait_lds();
        prefetch_stage<M>(a, t + 1, inst32, nullptr, s_rc, nullptr);
        prefetch_stage<M>(a, t + 2, inst32, s_a[t & 1], nullptr, nullptr);
        Spike<M> sp;
        stage_forward<M, K, false>(a.opt, io, mu, dw, gam, false, need, cy, sp, ok, nneg, keep_lost);
      }
    });
    if (!__any(need && (ok || keep_lost))) break;
  }
  wait_vm();    // prefetches of an abandoned sweep must not land in the next chunk's slots
  if (need) {
    retry_update_t<0>(a.opt, (int)a.Nc, sc, ok, nneg);
    if (sc[SC_NEED] == 0.0) a.phase[inst] = DTO_IM_PH_STEP;
  }
  }
}

// ------------------------------------------------------------------------------------------------
// back substitution of the listed instances (their last attempt was accepted): lane = list entry.  LDS: A, R and C slots
// (refilled for stage t-1 as soon as stage t has copied its records to registers; x_{t+1} stays in registers from the stage
// before) and a staging slot for the step record D(t) on its way out.
// ------------------------------------------------------------------------------------------------
template <class M, int WPS>
__global__ __launch_bounds__(WAVE, WPS) void kim_bwd(dto_im_args a) {
  constexpr int HA = hot_max<M>(0), HC = hot_max<M>(1), HD = hot_max<M>(2);
  __shared__ __attribute__((aligned(16))) double s_a[WAVE * HA];
  __shared__ __attribute__((aligned(16))) double s_r[WAVE * HA];
  __shared__ __attribute__((aligned(16))) double s_c[WAVE * HC];
  __shared__ __attribute__((aligned(16))) double s_d[WAVE * HD];
  const int n = *a.count;
  const dto_solver_opts& o = a.opt;
  const int lane = threadIdx.x;
  for (;;) {
  int chunk = 0;
  if (lane == 0) chunk = atomicAdd(a.ticket, 1);
  chunk = __shfl(chunk, 0, WAVE);
  if ((int64_t)chunk * WAVE >= n) return;
  const int e = chunk * WAVE + lane;
  const bool live = e < n;
  const int inst32 = a.list[live ? e : chunk * WAVE];   // lanes beyond the list shadow a listed instance; they store nothing
  const int64_t inst = inst32;
  const unsigned long long live_mask = __ballot(live);
  double* sc = a.scal + inst * SC_COUNT;
  constexpr int N = M::MAX_NX;
  const double mu = sc[SC_MU];
  const double tau = fmax(o.tau_min, 1.0 - mu);
  const double dw = sc[SC_DELTA_W], gam = sc[SC_GAMMA];
  double xL[N], xn[N], ycar[N];   // ycar: x_{t+1} itself (the y argument of stage t's dynamics)
#pragma unroll
  for (int i = 0; i < N; ++i) xL[i] = xn[i] = ycar[i] = 0.0;
  StepAcc acc{1.0, 1.0, 0.0, 0.0};
  wait_lds();
  prefetch_stage<M>(a, a.T - 1, inst32, s_a, s_r, s_c);
  for (int t = a.T - 1; t >= 0; --t) {
    wait_vm();
    dispatch_uniform<M>(a.kind[t], [&](auto kc) {
      constexpr int K = decltype(kc)::value;
      using D = KindDims<M, K>;
      using RC = Rec<M, K>;
      if constexpr (heavy_kind<M, K>()) {
        prefetch_stage<M>(a, t - 1, inst32, s_a, s_r, s_c);
        if (live) {
          const dto_im_args acold = a;
          double xnc[N];
#pragma unroll
          for (int i = 0; i < N; ++i) xnc[i] = xn[i];
          StepAcc accc = acc;
          im_backward_cold<M, K, WPS>(acold, inst, t, mu, tau, dw, gam, xnc, &accc);
#pragma unroll
          for (int i = 0; i < N; ++i) xn[i] = xnc[i];
          acc = accc;
        }
        const double* Ap = a.A + inst * a.a_total + a.aoff[t];
#pragma unroll
        for (int i = 0; i < D::NX; ++i) ycar[i] = Ap[i];
      } else {
        double* dst = s_d + lane * RC::DS;
        const LdsIO<M, K> io(a, inst, t, s_a + lane * RC::AS, s_r + lane * RC::AS, s_c + lane * RC::CS, ycar, dst);
        wait_lds();
        prefetch_stage<M>(a, t - 1, inst32, s_a, s_r, s_c);
        stage_backward<M, K, false>(a.opt, io, mu, tau, dw, gam, false, xL, xn, acc);
        if constexpr (RC::DS > RC::BD + RC::QI) dst[RC::DS - 1] = 0.0;
#pragma unroll
        for (int i = 0; i < D::NX; ++i) ycar[i] = io.p(i);
        wait_lds();
        scatter_records<RC::DS / 2>(s_d, a.D, a.d_total, a.doff[t], inst32, live_mask);
        wait_lds();   // the staging slot is rewritten by the next stage
      }
    });
  }
  wait_vm();
  if (live) {
    sc[SC_DMERIT] = acc.gphid;
    sc[SC_ALPHA_PMAX] = acc.apmax * (sc[SC_LS_MODE] == 1.0 ? sc[SC_ASCALE] : 1.0);
    sc[SC_ALPHA_DMAX] = acc.admax;
  }
  }
}

// ------------------------------------------------------------------------------------------------
// merit partials of the trial step sizes (as k_linesearch): lane = knot, OWN_LS knots of one instance per wavefront.
// grid = n_grid * nwin_l.
// ------------------------------------------------------------------------------------------------
template <class M>
__global__ __launch_bounds__(WAVE) void kim_linesearch(dto_im_args a) {
  __shared__ double s_acc[2 * DTO_LS_TRIALS * WAVE];
  __shared__ __attribute__((aligned(16))) double s_a[(OWN_LS + 1) * max_as<M>() + 2];
  __shared__ __attribute__((aligned(16))) double s_d[(OWN_LS + 1) * max_ds<M>() + 2];
  const int64_t n_item = (int64_t)(*a.count) * a.nwin_l;
  for (int64_t item = blockIdx.x; item < n_item; item += gridDim.x) {
  const int e = (int)(item / a.nwin_l);
  const int win = (int)(item % a.nwin_l);
  const int64_t inst = a.list[e];
  const int lane = threadIdx.x;
  const double* sc = a.scal + inst * SC_COUNT;
  // iterate and step of the window's knots plus the knot after them (y and dy of the last one): coalesced copies into LDS
  const int t0w = win * OWN_LS, tendw = min(t0w + OWN_LS + 1, a.T);
  const int a0 = a.aoff[t0w], d0 = a.doff[t0w];
  __syncthreads();   // the previous item's reads of the images are complete
  wave_load(s_a, a.A + inst * a.a_total + a0, a.aoff[tendw] - a0);
  wave_load(s_d, a.D + inst * a.d_total + d0, a.doff[tendw] - d0);
  __syncthreads();
  const double mu = sc[SC_MU];
  const double amax = sc[SC_ALPHA_PMAX];
  double* acc = s_acc + lane;
#pragma unroll
  for (int k = 0; k < 2 * DTO_LS_TRIALS; ++k) acc[k * WAVE] = 0.0;
  const int t = win * OWN_LS + lane;
  if (t < a.T) {
    dispatch_kind<M>(a.kind[t], [&](auto kc) {
      constexpr int K = decltype(kc)::value;
      using D = KindDims<M, K>;
      using KD = typename D::KD;
      using RC = Rec<M, K>;
      using CO = typename M::template Cost<KD::COST>;
      const int z0 = a.zoff[t];
      const double* Ap = s_a + (a.aoff[t] - a0);
      const double* Dp = s_d + (a.doff[t] - d0);
      const double* Bp = a.Bd ? a.Bd + inst * a.b_total + a.boff[t] : nullptr;
      arr<D::NP> p, dp;
      arr<D::NY> y, dy;
#pragma unroll
      for (int i = 0; i < D::NP; ++i) {
        p[i] = Ap[i];
        dp[i] = Dp[i];
      }
      if constexpr (D::NY > 0) {
        const double* An = s_a + (a.aoff[t + 1] - a0);
        const double* Dn = s_d + (a.doff[t + 1] - d0);
#pragma unroll
        for (int i = 0; i < D::NY; ++i) {
          y[i] = An[i];
          dy[i] = Dn[i];
        }
      }
      arr<CO::NW> w;
      im_params(w, a, inst, t);
      double blo[D::NP > 0 ? D::NP : 1], bhi[D::NP > 0 ? D::NP : 1];
#pragma unroll
      for (int i = 0; i < D::NP; ++i) { blo[i] = a.lo[z0 + i]; bhi[i] = a.hi[z0 + i]; }
      double sl[D::QI > 0 ? D::QI : 1], dsl[D::QI > 0 ? D::QI : 1];
#pragma unroll
      for (int j = 0; j < D::QI; ++j) {
        sl[j] = Bp[RC::B_S + j];
        dsl[j] = Dp[RC::D_DS + j];
      }
      // trial points from the shortest step up; trigonometry by recurrence where the arguments are affine (k_linesearch)
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
#pragma unroll
        for (int i = 0; i < D::NP; ++i) {
          const double lo = blo[i], hi = bhi[i];
          if (lo != hi) {
            if (finite_lo(lo)) phi -= mu * log(pk[i] - lo);
            if (finite_hi(hi)) phi -= mu * log(hi - pk[i]);
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
            for (int j = 0; j < NTRIG; ++j) {
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
            if (D::ineq(j)) {
              const double sk = sl[D::slack(j)] + alpha * dsl[D::slack(j)];
              r += sk;
              phi -= mu * log(sk);
            }
            th += fabs(r);
          }
        }
        acc[(2 * k) * WAVE] = phi;
        acc[(2 * k + 1) * WAVE] = th;
        alpha *= 2.0;
      }
    });
  }
  // sum over the knots of the window (fixed-order butterfly), one row of 2 * DTO_LS_TRIALS sums per window
  double* out = a.lspart + (inst * a.nwin_l + win) * (2 * DTO_LS_TRIALS);
#pragma unroll
  for (int k = 0; k < 2 * DTO_LS_TRIALS; ++k) {
    const double v = wave_sum(acc[k * WAVE]);
    if (lane == 0) out[k] = v;
  }
  }
}

// the filter decides: lane = list entry
static __global__ __launch_bounds__(WAVE) void kim_ls_reduce(dto_im_args a) {
  const int n = *a.count;
  for (int e = blockIdx.x * WAVE + threadIdx.x; e < n; e += gridDim.x * WAVE) {
  const int64_t inst = a.list[e];
  double* sc = a.scal + inst * SC_COUNT;
  double phi[DTO_LS_TRIALS], th[DTO_LS_TRIALS];
#pragma unroll
  for (int k = 0; k < DTO_LS_TRIALS; ++k) phi[k] = th[k] = 0.0;
  for (int w = 0; w < a.nwin_l; ++w) {
    const double* in = a.lspart + (inst * a.nwin_l + w) * (2 * DTO_LS_TRIALS);
#pragma unroll
    for (int k = 0; k < DTO_LS_TRIALS; ++k) {
      phi[k] += in[2 * k];
      th[k] += in[2 * k + 1];
    }
  }
  ls_reduce_body<0>(a.opt, sc, a.filt + inst * (2 * DTO_FILTER_CAP), phi, th);
  }
}

// ------------------------------------------------------------------------------------------------
// take the step (as k_update): lane = knot.  The instance goes back to phase EVAL.  grid = n_grid * nwin_l.
// ------------------------------------------------------------------------------------------------
template <class M>
__global__ __launch_bounds__(WAVE) void kim_update(dto_im_args a) {
  // the A and D records of the window's knots are contiguous ranges of the instance: coalesced copies into LDS images, the
  // lanes update their record in the A image, the image goes back with full-line stores
  __shared__ __attribute__((aligned(16))) double s_a[OWN_LS * max_as<M>() + 2];
  __shared__ __attribute__((aligned(16))) double s_d[OWN_LS * max_ds<M>() + 2];
  const int64_t n_item = (int64_t)(*a.count) * a.nwin_l;
  for (int64_t item = blockIdx.x; item < n_item; item += gridDim.x) {
  const int e = (int)(item / a.nwin_l);
  const int win = (int)(item % a.nwin_l);
  const int64_t inst = a.list[e];
  double* sc = a.scal + inst * SC_COUNT;
  const double mu = sc[SC_MU];
  const double al = sc[SC_ALPHA];
  const double ad = sc[SC_ALPHA_DMAX];
  constexpr double KSIG = 1e10;
  const int t0 = win * OWN_LS, tend = min(t0 + OWN_LS, a.T);
  const int t = t0 + threadIdx.x;
  const int a0 = a.aoff[t0], d0 = a.doff[t0];
  wave_load(s_a, a.A + inst * a.a_total + a0, a.aoff[tend] - a0);
  wave_load(s_d, a.D + inst * a.d_total + d0, a.doff[tend] - d0);
  __syncthreads();
  if (t < a.T) {
    dispatch_kind<M>(a.kind[t], [&](auto kc) {
      constexpr int K = decltype(kc)::value;
      using D = KindDims<M, K>;
      using RC = Rec<M, K>;
      const int z0 = a.zoff[t];
      double* Ap = s_a + (a.aoff[t] - a0);
      const double* Dp = s_d + (a.doff[t] - d0);
      double* Bp = a.Bd ? a.Bd + inst * a.b_total + a.boff[t] : nullptr;
      double v[RC::BD], dv[RC::BD];
#pragma unroll
      for (int i = 0; i < RC::BD; ++i) {
        v[i] = Ap[i];
        dv[i] = Dp[i];
      }
#pragma unroll
      for (int i = 0; i < D::NP; ++i) {
        const double p = v[i], dp = dv[i];
        const double pn = p + al * dp;
        const double lo = a.lo[z0 + i], hi = a.hi[z0 + i];
        if (lo != hi && Bp) {
          if (finite_lo(lo)) {
            const double zl = Bp[i];
            const double gap = p - lo;
            const double dzl = mu / gap - zl - (zl / gap) * dp;
            double zn = zl + ad * dzl;
            const double gn = pn - lo;
            zn = fmin(fmax(zn, mu / (KSIG * gn)), KSIG * mu / gn);
            Bp[i] = zn;
          }
          if (finite_hi(hi)) {
            const double zu = Bp[RC::B_ZU + i];
            const double gap = hi - p;
            const double dzu = mu / gap - zu + (zu / gap) * dp;
            double zn = zu + ad * dzu;
            const double gn = hi - pn;
            zn = fmin(fmax(zn, mu / (KSIG * gn)), KSIG * mu / gn);
            Bp[RC::B_ZU + i] = zn;
          }
        }
        Ap[i] = pn;
      }
#pragma unroll
      for (int j = 0; j < D::Q; ++j) {
        if (D::ineq(j)) {
          const double sv = Bp[RC::B_S + D::slack(j)], zv = Bp[RC::B_ZS + D::slack(j)];
          const double dsv = Dp[RC::D_DS + D::slack(j)];
          const double dzs = mu / sv - zv - (zv / sv) * dsv;
          const double sn = sv + al * dsv;
          double zn = zv + ad * dzs;
          zn = fmin(fmax(zn, mu / (KSIG * sn)), KSIG * mu / sn);
          Bp[RC::B_S + D::slack(j)] = sn;
          Bp[RC::B_ZS + D::slack(j)] = zn;
        }
      }
#pragma unroll
      for (int i = D::NP; i < RC::BD; ++i) Ap[i] = v[i] + al * dv[i];
    });
  }
  __syncthreads();
  wave_store_image(a.A + inst * a.a_total + a0, s_a, a.aoff[tend] - a0, threadIdx.x);
  if (win == 0 && threadIdx.x == 0) {
    sc[SC_ITER] += 1.0;
    a.phase[inst] = DTO_IM_PH_EVAL;
  }
  __syncthreads();   // the images are refilled by the next item
  }
}

// ------------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------------
template <class M>
int launch_im(int op, const dto_im_args* args, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  const dto_im_args& a = *args;
  if constexpr (M::HAS_GENERAL || M::EVALUATE_HESSIAN == 0 || M::N_KIND > 16) {
    return (int)hipErrorNotSupported;
  } else {
    // list-driven kernels size their grids for the hardware, not for the list: items are walked with a grid stride (stage
    // kernels) or handed out by a ticket (sweeps), so a launch costs nothing for instances that are not on its list
    const int64_t nwin = a.nwin_e > a.nwin_l ? a.nwin_e : a.nwin_l;
    if (a.B * nwin > 0x7fffffffll) return (int)hipErrorInvalidValue;
    const unsigned g_all = (unsigned)(a.B * a.nwin_l);
    const unsigned g_stage_e = (unsigned)std::min<int64_t>(a.B * a.nwin_e, 1 << 16);
    const unsigned g_stage_l = (unsigned)std::min<int64_t>(a.B * a.nwin_l, 1 << 16);
    const unsigned g_entry = (unsigned)std::min<int64_t>((a.B + WAVE - 1) / WAVE, 1 << 14);
    const unsigned g_sweep = (unsigned)std::min<int64_t>((a.B + WAVE - 1) / WAVE, 2048);   // resident wavefronts: 2 / 1 per SIMD
    const unsigned g_sweep1 = (unsigned)std::min<int64_t>((a.B + WAVE - 1) / WAVE, 1024);
    switch (op) {
      case DTO_IM_INIT: hipLaunchKernelGGL(kim_init<M>, dim3(g_all), dim3(WAVE), 0, st, a); break;
      case DTO_IM_UNPACK: hipLaunchKernelGGL(kim_unpack<M>, dim3(g_all), dim3(WAVE), 0, st, a); break;
      case DTO_IM_COMPACT: hipLaunchKernelGGL(kim_compact, dim3((unsigned)((a.B + 255) / 256)), dim3(256), 0, st, a); break;
      case DTO_IM_COUNT: hipLaunchKernelGGL(kim_count, dim3((unsigned)((a.B + 255) / 256)), dim3(256), 0, st, a); break;
      case DTO_IM_EVAL: hipLaunchKernelGGL(kim_eval<M>, dim3(g_stage_e), dim3(WAVE), 0, st, a); break;
      case DTO_IM_CONV: hipLaunchKernelGGL(kim_conv, dim3(g_entry), dim3(WAVE), 0, st, a); break;
      case DTO_IM_FWD:
        if (a.sweep_occ == 1) hipLaunchKernelGGL((kim_fwd<M, 1>), dim3(g_sweep1), dim3(WAVE), 0, st, a);
        else hipLaunchKernelGGL((kim_fwd<M, 2>), dim3(g_sweep), dim3(WAVE), 0, st, a);
        break;
      case DTO_IM_BWD:
        if (a.sweep_occ == 1) hipLaunchKernelGGL((kim_bwd<M, 1>), dim3(g_sweep1), dim3(WAVE), 0, st, a);
        else hipLaunchKernelGGL((kim_bwd<M, 2>), dim3(g_sweep), dim3(WAVE), 0, st, a);
        break;
      case DTO_IM_LINESEARCH: hipLaunchKernelGGL(kim_linesearch<M>, dim3(g_stage_l), dim3(WAVE), 0, st, a); break;
      case DTO_IM_LS_REDUCE: hipLaunchKernelGGL(kim_ls_reduce, dim3(g_entry), dim3(WAVE), 0, st, a); break;
      case DTO_IM_UPDATE: hipLaunchKernelGGL(kim_update<M>, dim3(g_stage_l), dim3(WAVE), 0, st, a); break;
      default: return -1;
    }
    return (int)hipGetLastError();
  }
}

}  // namespace im
}  // namespace dto
