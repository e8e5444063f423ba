// Hand-written gfx950 stage kernels for the five NLP callbacks, instance-major ("AoS") layout.
//
// These replace the serial stage loops of the reference:
//   cost / gradient!          src/costs.jl:49-64
//   constraints! / jacobian!  src/dynamics.jl:103-117, src/constraints.jl:80-94
//   hessian!/hessian_lagrangian!  src/costs.jl:66-73, src/dynamics.jl:119-127, src/constraints.jl:96-104
// and the unpack copies trajectory!/duals! (src/data.jl:258-278), which disappear: a lane reads its
// knot straight out of the staged z.
//
// Mapping (MI355X-first, not a translation of the loop):
//   * one 64-lane wavefront = 64 consecutive knots of ONE instance; lane = knot t.  The model's
//     expression DAG (generated, straight-line, all VGPR) is evaluated per lane.
//   * inputs: a lane reads its (x_t, u_t, x_{t+1}) straight from global memory (contiguous per lane, overlapping the
//     neighbour's; the vector L1 serves the stride).  Staging the inputs through LDS was measured and dropped: it
//     costs a barrier and occupancy for nothing.
//   * outputs: a knot's values (rows of c, nonzeros of J, key slots of H) are contiguous per stage
//     and stages are contiguous per wave, so lanes deposit into an LDS image of the wave's output
//     range and the wave streams that image to HBM fully coalesced (no scattered 8-byte stores).
//   * the Hessian's `.+=` overlap (yy block of stage t-1 lands in the xx block of stage t,
//     src/dynamics.jl:123) is resolved inside LDS in two phases instead of with f64 atomics; the wave
//     owns 63 stages plus one halo lane that only contributes the overlap.
//   * blocks are 2-4 waves: >= B*T/256 workgroups, LDS <= ~24 KiB per wave so 6+
//     waves share a CU; nothing is reused across workgroups, so the block->XCD mapping is left to
//     the round-robin dispatcher (tables are < 64 KiB and live in every XCD's L2).
#pragma once

#include <hip/hip_runtime.h>
#include <type_traits>

#include "dto_math.hpp"
#include "dto_model_plugin.h"

namespace dto {

constexpr int WAVE = 64;

template <int N>
struct arr {
  double v[N > 0 ? N : 1];
  __device__ __forceinline__ double& operator[](int i) { return v[i]; }
  __device__ __forceinline__ const double& operator[](int i) const { return v[i]; }
  __device__ __forceinline__ double* data() { return v; }
  __device__ __forceinline__ const double* data() const { return v; }
};

// compile-time dispatch on the (wave-divergent but almost always uniform) stage kind
template <class M, int K = 0, class F>
__device__ __forceinline__ void dispatch_kind(int kind, F&& f) {
  if constexpr (K < M::N_KIND) {
    if (kind == K)
      f(std::integral_constant<int, K>{});
    else
      dispatch_kind<M, K + 1>(kind, static_cast<F&&>(f));
  }
}

template <int N>
__device__ __forceinline__ void lds_load(arr<N>& dst, const double* src) {
#pragma unroll
  for (int i = 0; i < N; ++i) dst[i] = src[i];
}

template <int N>
__device__ __forceinline__ void gmem_load(arr<N>& dst, const double* src) {
#pragma unroll
  for (int i = 0; i < N; ++i) dst[i] = src[i];
}

// cooperative, coalesced copies between HBM and the wave's LDS image: 16 bytes per lane (1 KiB per
// wave-instruction) on the 16-byte-aligned body, scalar head/tail.  `lds` must be 16-byte aligned.
__device__ __forceinline__ void wave_load(double* lds, const double* g, int count) {
  // shift so that the GLOBAL address of the vector body is 16-byte aligned; the LDS image keeps the same
  // element offsets, so lds + head is 16-byte aligned only if head is even -- handled by the (lds) parity test
  const int head = (int)((reinterpret_cast<uintptr_t>(g) >> 3) & 1);
  if (head == 0 && (count >= 2)) {
    const int nv = count >> 1;
    const double2* gv = reinterpret_cast<const double2*>(g);
    double2* lv = reinterpret_cast<double2*>(lds);
    for (int i = threadIdx.x; i < nv; i += WAVE) lv[i] = gv[i];
    if ((count & 1) && threadIdx.x == 0) lds[count - 1] = g[count - 1];
  } else {
    for (int i = threadIdx.x; i < count; i += WAVE) lds[i] = g[i];
  }
}
__device__ __forceinline__ void wave_store(double* g, const double* lds, int count) {
  const int head = (int)((reinterpret_cast<uintptr_t>(g) >> 3) & 1);
  if (head == 0 && (count >= 2)) {
    const int nv = count >> 1;
    double2* gv = reinterpret_cast<double2*>(g);
    const double2* lv = reinterpret_cast<const double2*>(lds);
    for (int i = threadIdx.x; i < nv; i += WAVE) gv[i] = lv[i];
    if ((count & 1) && threadIdx.x == 0) g[count - 1] = lds[count - 1];
  } else {
    for (int i = threadIdx.x; i < count; i += WAVE) g[i] = lds[i];
  }
}

// stream a wave's LDS image to HBM: 16 bytes per lane where the destination is 16-byte aligned
__device__ __forceinline__ void wave_store_image(double* g, const double* l, int count, int lane) {
  if (((reinterpret_cast<uintptr_t>(g) >> 3) & 1) == 0) {
    const int nv = count >> 1;
    double2* gv = reinterpret_cast<double2*>(g);
    const double2* lv = reinterpret_cast<const double2*>(l);
    for (int i = lane; i < nv; i += WAVE) gv[i] = lv[i];
    if ((count & 1) && lane == 0) g[count - 1] = l[count - 1];
  } else {
    for (int i = lane; i < count; i += WAVE) g[i] = l[i];
  }
}

// LDS needed to stage z for 64(+1) knots
template <class M>
constexpr int z_stage_len() { return (WAVE + 1) * M::MAX_NXU + M::MAX_NX; }

// ------------------------------------------------------------------------------------------------
// objective: per-stage cost values -> scratch[B][T]   (summed in stage order by k_sum_rows)
// ------------------------------------------------------------------------------------------------
template <class M>
__global__ __launch_bounds__(WAVE) void k_obj(dto_eval_args a) {
  __shared__ __attribute__((aligned(16))) double s_z[z_stage_len<M>()];
  const int wpi = (a.T + WAVE - 1) / WAVE;
  const int64_t b = blockIdx.x / wpi;
  const int t0 = (blockIdx.x % wpi) * WAVE;
  const int tend = min(t0 + WAVE, a.T);
  const int z0 = a.zoff[t0];
  wave_load(s_z, a.z + b * a.ldz + z0, a.zoff[tend] - z0);
  __syncthreads();
  const int t = t0 + threadIdx.x;
  if (t < a.T) {
    double val = 0.0;
    dispatch_kind<M>(a.kind[t], [&](auto kc) {
      using KD = typename M::template Kind<decltype(kc)::value>;
      using C = typename M::template Cost<KD::COST>;
      arr<C::NX> x; arr<C::NU> u; arr<C::NW> w;
      const double* zs = s_z + (a.zoff[t] - z0);
      lds_load(x, zs); lds_load(u, zs + C::NX);
      gmem_load(w, a.w + b * a.ldw + a.woff[t]);
      double o[1];
      C::eval(x.data(), u.data(), w.data(), o);
      val = o[0];
    });
    a.scratch[b * (int64_t)a.T + t] = val;
  }
}

// one wave per row: out[b] = sum_t rows[b][t], fixed order (lane-strided partials, then a tree)
static __global__ __launch_bounds__(WAVE) void k_sum_rows(const double* rows, int64_t ld, int n, double* out) {
  const int64_t b = blockIdx.x;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += WAVE) acc += rows[b * ld + i];
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, WAVE);
  if (threadIdx.x == 0) out[b] = acc;
}

// ------------------------------------------------------------------------------------------------
// gradient: dense [x_t;u_t] gradient per stage, contiguous over the wave
// ------------------------------------------------------------------------------------------------
template <class M>
__global__ __launch_bounds__(WAVE) void k_grad(dto_eval_args a) {
  __shared__ __attribute__((aligned(16))) double s_z[z_stage_len<M>()];
  __shared__ __attribute__((aligned(16))) double s_o[WAVE * M::MAX_NXU];
  const int wpi = (a.T + WAVE - 1) / WAVE;
  const int64_t b = blockIdx.x / wpi;
  const int t0 = (blockIdx.x % wpi) * WAVE;
  const int tend = min(t0 + WAVE, a.T);
  const int z0 = a.zoff[t0];
  const int zlen = a.zoff[tend] - z0;
  wave_load(s_z, a.z + b * a.ldz + z0, zlen);
  __syncthreads();
  const int t = t0 + threadIdx.x;
  if (t < a.T) {
    dispatch_kind<M>(a.kind[t], [&](auto kc) {
      using KD = typename M::template Kind<decltype(kc)::value>;
      using C = typename M::template Cost<KD::COST>;
      arr<C::NX> x; arr<C::NU> u; arr<C::NW> w;
      const int zo = a.zoff[t] - z0;
      lds_load(x, s_z + zo); lds_load(u, s_z + zo + C::NX);
      gmem_load(w, a.w + b * a.ldw + a.woff[t]);
      arr<C::NX + C::NU> g;
      C::grad(x.data(), u.data(), w.data(), g.data());
#pragma unroll
      for (int i = 0; i < C::NX + C::NU; ++i) s_o[zo + i] = g[i];
    });
  }
  __syncthreads();
  wave_store(a.out + b * a.ldout + z0, s_o, zlen);
}

// ------------------------------------------------------------------------------------------------
// constraints: dynamics rows then stage rows
// ------------------------------------------------------------------------------------------------
template <class M>
__global__ __launch_bounds__(4 * WAVE) void k_con(dto_eval_args a) {
  // same mapping as k_jac: direct reads, LDS-transposed full-line stores for the dynamics rows, 4 waves/workgroup
  __shared__ __attribute__((aligned(16))) double s_d[4][WAVE * M::MAX_DYN_NC + 2];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpi = (a.T + WAVE - 1) / WAVE;
  const int bpi = (wpi + 3) / 4;
  const int64_t b = blockIdx.x / bpi;
  const int t0 = ((blockIdx.x % bpi) * 4 + wv) * WAVE;
  const bool live_wave = t0 < a.T;
  const int tend = live_wave ? min(t0 + WAVE, a.T) : 0;
  const int t = t0 + lane;
  double* ob = a.out + b * a.ldout;
  const int d0 = live_wave ? a.cdoff[t0] : 0;
  if (live_wave && t < a.T) {
    const double* zs = a.z + b * a.ldz + a.zoff[t];
    const double* wp = a.w + b * a.ldw + a.woff[t];
    dispatch_kind<M>(a.kind[t], [&](auto kc) {
      using KD = typename M::template Kind<decltype(kc)::value>;
      if constexpr (KD::DYN >= 0) {
        using D = typename M::template Dyn<KD::DYN>;
        arr<D::NX> x; arr<D::NU> u; arr<D::NY> y; arr<D::NW> w; arr<D::NY> o;
        gmem_load(x, zs); gmem_load(u, zs + D::NX); gmem_load(y, zs + D::NX + D::NU);
        gmem_load(w, wp);
        D::eval(x.data(), u.data(), y.data(), w.data(), o.data());
        const int off = a.cdoff[t] - d0;
#pragma unroll
        for (int i = 0; i < D::NY; ++i) s_d[wv][off + i] = o[i];
      }
      if constexpr (KD::CON >= 0) {
        using C = typename M::template Con<KD::CON>;
        arr<C::NX> x; arr<C::NU> u; arr<C::NW> w; arr<C::NC> o;
        gmem_load(x, zs); gmem_load(u, zs + C::NX);
        gmem_load(w, wp);
        C::eval(x.data(), u.data(), w.data(), o.data());
        double* dst = ob + a.ccoff[t];
#pragma unroll
        for (int i = 0; i < C::NC; ++i) dst[i] = o[i];
      }
    });
  }
  __syncthreads();
  if (live_wave) wave_store_image(ob + d0, s_d[wv], a.cdoff[tend] - d0, lane);
}

// ------------------------------------------------------------------------------------------------
// constraint Jacobian nonzeros, reference COO order (dynamics block then stage block)
// ------------------------------------------------------------------------------------------------
constexpr int JAC_WAVES = 4;  // wavefronts per workgroup (each owns 64 consecutive knots of one instance)

template <class M>
__global__ __launch_bounds__(JAC_WAVES * WAVE) void k_jac(dto_eval_args a) {
  // wave = 64 consecutive knots of ONE instance (so its dynamics nonzeros are one contiguous output range).
  //  * reads: straight from global -- a lane's (x_t,u_t,x_{t+1}) are contiguous and overlap the neighbour's, the
  //    vector L1 serves the 40-byte-stride pattern; no input staging, no barrier before the arithmetic;
  //  * writes: lanes deposit their NJ values into the wave's LDS image, the wave streams the image out with
  //    16-byte-per-lane stores (1 KiB per wave-instruction, full lines).  Measured on MI355X: full-line stores
  //    reach 5.2-5.5 TB/s, the direct "26 contiguous doubles per lane" pattern tops out at 2.7-3.5 TB/s even
  //    without any arithmetic (tools/micro/store_ceiling.hip), and non-temporal stores are 7x slower;
  //  * only the output image lives in LDS (13 KiB per wave for the acrobot): 12 waves per CU instead of 8;
  //    the (rare) stage-constraint nonzeros are written directly.
  __shared__ __attribute__((aligned(16))) double s_d[JAC_WAVES][WAVE * M::MAX_DYN_NJ + 2];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpi = (a.T + WAVE - 1) / WAVE;                    // wave tiles per instance
  const int bpi = (wpi + JAC_WAVES - 1) / JAC_WAVES;          // workgroups per instance
  const int64_t b = blockIdx.x / bpi;
  const int tile = (blockIdx.x % bpi) * JAC_WAVES + wv;
  const int t0 = tile * WAVE;
  const bool live_wave = t0 < a.T;
  const int tend = live_wave ? min(t0 + WAVE, a.T) : 0;
  const int t = t0 + lane;
  double* ob = a.out + b * a.ldout;
  const int d0 = live_wave ? a.jdoff[t0] : 0;
  if (live_wave && t < a.T) {
    const double* zs = a.z + b * a.ldz + a.zoff[t];
    const double* wp = a.w + b * a.ldw + a.woff[t];
    dispatch_kind<M>(a.kind[t], [&](auto kc) {
      using KD = typename M::template Kind<decltype(kc)::value>;
      if constexpr (KD::DYN >= 0) {
        using D = typename M::template Dyn<KD::DYN>;
        arr<D::NX> x; arr<D::NU> u; arr<D::NY> y; arr<D::NW> w; arr<D::NJ> o;
        gmem_load(x, zs); gmem_load(u, zs + D::NX); gmem_load(y, zs + D::NX + D::NU);
        gmem_load(w, wp);
        D::jac(x.data(), u.data(), y.data(), w.data(), o.data());
        const int off = a.jdoff[t] - d0;
#pragma unroll
        for (int i = 0; i < D::NJ; ++i) s_d[wv][off + i] = o[i];
      }
      if constexpr (KD::CON >= 0) {
        using C = typename M::template Con<KD::CON>;
        arr<C::NX> x; arr<C::NU> u; arr<C::NW> w; arr<C::NJ> o;
        gmem_load(x, zs); gmem_load(u, zs + C::NX);
        gmem_load(w, wp);
        C::jac(x.data(), u.data(), w.data(), o.data());
        double* dst = ob + a.jcoff[t];
#pragma unroll
        for (int i = 0; i < C::NJ; ++i) dst[i] = o[i];
      }
    });
  }
  __syncthreads();
  if (live_wave) wave_store_image(ob + d0, s_d[wv], a.jdoff[tend] - d0, lane);
}

// ------------------------------------------------------------------------------------------------
// Hessian of the Lagrangian, reference key order (row-major sorted unique, both triangles)
// ------------------------------------------------------------------------------------------------
// How the key image of a stage is filled (measured on MI355X: 107 ds_add_f64 per lane made the kernel LDS-atomic bound,
// 64 cycles per wave instruction; read-modify-write of every entry exposed ~120 cycles of latency each):
//   1. zero fill, 2. the dynamics Hessian entries of the stage's own rows are PLAIN STORES (within one stage every such
//   entry has its own slot), 3. the few cost / constraint entries and, after a barrier, the previous stage's y-row
//   entries are added.  Fixed order per slot -> deterministic.
constexpr int HOWN = WAVE - 1;  // stages owned by one wave; lane 0 is the halo (stage t0-1)
constexpr int HESS_WAVES = 1;   // one wavefront per workgroup: the barriers between the deposit phases cost nothing
// Stages per output image: as many as fit ~10 KiB of LDS (32 for the acrobot's 41 keys per stage -> the 63 stages of a
// wave go out in two images).  Measured, acrobot T=1000, 8192 instances: 20.7 KiB (one image, 7 waves per CU) 0.98 ms,
// 14 KiB 0.94 ms, 10 KiB (4 waves per SIMD at 116 VGPRs) 0.89 ms = 46 % of the HBM peak.  Round 1 form (two-wave
// workgroups, per-entry "-1 -> trash slot" selects in both deposit passes, 207 VGPRs): 1.20 ms = 34 %.
template <class M>
constexpr int hess_span() {
  constexpr int fit = (10 * 1024 / 8 - 2) / (M::MAX_KEY > 0 ? M::MAX_KEY : 1);
  return fit >= HOWN ? HOWN : (fit < 4 ? 4 : fit);
}

template <class M>
__global__ __launch_bounds__(HESS_WAVES * WAVE) void k_hess(dto_eval_args a) {
  // which of a dynamics' Hessian nonzeros go to the rows of its own stage and which to the next stage's is a literal
  // table of the generated class (Dyn::hess_row_own), so the unrolled deposit loops touch exactly their entries
  constexpr int HHALF = hess_span<M>();
  constexpr int IMG = HHALF * M::MAX_KEY + 2;
  __shared__ __attribute__((aligned(16))) double s_img[HESS_WAVES][IMG];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* s_o = s_img[wv];
  const int wpi = (a.T + HOWN - 1) / HOWN;
  const int bpi = (wpi + HESS_WAVES - 1) / HESS_WAVES;
  const int64_t b = blockIdx.x / bpi;
  const int tile = (blockIdx.x % bpi) * HESS_WAVES + wv;
  const int t0 = tile * HOWN;                    // first owned stage
  const bool live_wave = t0 < a.T;
  const int tend = live_wave ? min(t0 + HOWN, a.T) : 0;  // one past the last owned stage
  const int s = t0 - 1 + lane;                   // this lane's stage (lane 0 is the halo)
  const bool live = live_wave && (s >= 0) && (s < tend);
  const bool own = live && (s >= t0);
  const double* mu = a.mu + b * a.ldmu;
  const int kind = live ? a.kind[s] : -1;
  // ---- evaluate: all Hessian nonzeros of this lane's stage stay in registers
  arr<M::MAX_DYN_NH> hv;
  arr<M::MAX_COST_NH> cv;
  arr<M::MAX_CON_NH> kv;
  if (live) {
    dispatch_kind<M>(kind, [&](auto kc) {
      using KD = typename M::template Kind<decltype(kc)::value>;
      const double* zs = a.z + b * a.ldz + a.zoff[s];  // direct reads: contiguous per lane, L1-served overlap
      const double* wp = a.w + b * a.ldw + a.woff[s];
      if constexpr (M::template Cost<KD::COST>::NH > 0) {
        using C = typename M::template Cost<KD::COST>;
        if (own) {
          arr<C::NX> x; arr<C::NU> u; arr<C::NW> w; arr<C::NH> o;
          gmem_load(x, zs); gmem_load(u, zs + C::NX); gmem_load(w, wp);
          C::hess(x.data(), u.data(), w.data(), o.data());
#pragma unroll
          for (int i = 0; i < C::NH; ++i) cv[i] = a.sigma * o[i];
        }
      }
      if constexpr (KD::DYN >= 0) {
        using D = typename M::template Dyn<KD::DYN>;
        if constexpr (D::NH > 0) {
          arr<D::NX> x; arr<D::NU> u; arr<D::NY> y; arr<D::NW> w; arr<D::NY> lam; arr<D::NH> o;
          gmem_load(x, zs); gmem_load(u, zs + D::NX); gmem_load(y, zs + D::NX + D::NU);
          gmem_load(w, wp); gmem_load(lam, mu + a.cdoff[s]);
          D::hess(x.data(), u.data(), y.data(), w.data(), lam.data(), o.data());
#pragma unroll
          for (int i = 0; i < D::NH; ++i) hv[i] = o[i];
        }
      }
      if constexpr (KD::CON >= 0) {
        using C = typename M::template Con<KD::CON>;
        if constexpr (C::NH > 0) {
          if (own) {
            arr<C::NX> x; arr<C::NU> u; arr<C::NW> w; arr<C::NC> lam; arr<C::NH> o;
            gmem_load(x, zs); gmem_load(u, zs + C::NX); gmem_load(w, wp);
            gmem_load(lam, mu + a.ccoff[s]);
            C::hess(x.data(), u.data(), w.data(), lam.data(), o.data());
#pragma unroll
            for (int i = 0; i < C::NH; ++i) kv[i] = o[i];
          }
        }
      }
    });
  }
  // ---- deposit and stream out, HHALF stages at a time.  Per slot: dynamics (store), cost, constraint (own lane, program
  //      order), then the previous stage's y-rows (after the barrier).
  for (int half = 0; half < (HOWN + HHALF - 1) / HHALF; ++half) {
    const int ta = t0 + half * HHALF;                        // stages [ta, tb) are in this image
    const bool live_half = live_wave && ta < tend;
    const int tb = live_half ? min(ta + HHALF, tend) : ta;
    const int h0 = live_half ? a.hoff[ta] : 0;
    const int hlen = live_half ? a.hoff[tb] - h0 : 0;
    for (int i = lane; i < hlen; i += WAVE) s_o[i] = 0.0;
    __syncthreads();
    if (own && s >= ta && s < tb) {
      dispatch_kind<M>(kind, [&](auto kc) {
        using KD = typename M::template Kind<decltype(kc)::value>;
        const int base = a.hoff[s] - h0;
        if constexpr (KD::DYN >= 0) {
          using D = typename M::template Dyn<KD::DYN>;
          if constexpr (D::NH > 0) {
            const int* mrow = a.hmap_dyn_own + decltype(kc)::value * a.hmap_stride;
#pragma unroll
            for (int i = 0; i < D::NH; ++i)
              if (D::hess_row_own(i)) s_o[base + mrow[i]] = hv[i];
          }
        }
        if constexpr (M::template Cost<KD::COST>::NH > 0) {
          using C = typename M::template Cost<KD::COST>;
          const int* mrow = a.hmap_cost + decltype(kc)::value * a.hmap_stride;
#pragma unroll
          for (int i = 0; i < C::NH; ++i) s_o[base + mrow[i]] += cv[i];
        }
        if constexpr (KD::CON >= 0) {
          using C = typename M::template Con<KD::CON>;
          if constexpr (C::NH > 0) {
            const int* mrow = a.hmap_con + decltype(kc)::value * a.hmap_stride;
#pragma unroll
            for (int i = 0; i < C::NH; ++i) s_o[base + mrow[i]] += kv[i];
          }
        }
      });
    }
    __syncthreads();
    // yy (and y-row) entries of dyn(s) go to the rows of stage s+1 (owned by lane+1, possibly in the next image)
    if (live && s + 1 >= ta && s + 1 < tb) {
      dispatch_kind<M>(kind, [&](auto kc) {
        using KD = typename M::template Kind<decltype(kc)::value>;
        if constexpr (KD::DYN >= 0) {
          using D = typename M::template Dyn<KD::DYN>;
          if constexpr (D::NH > 0) {
            const int base = a.hoff[s + 1] - h0;
            const int* mrow = a.hmap_dyn_next + a.kind[s + 1] * a.hmap_stride;
            int mo[D::NH];
#pragma unroll
            for (int i = 0; i < D::NH; ++i)
              if (!D::hess_row_own(i)) mo[i] = base + mrow[i];
            double cur[D::NH];
#pragma unroll
            for (int i = 0; i < D::NH; ++i)
              if (!D::hess_row_own(i)) cur[i] = s_o[mo[i]];
#pragma unroll
            for (int i = 0; i < D::NH; ++i)
              if (!D::hess_row_own(i)) s_o[mo[i]] = cur[i] + hv[i];
          }
        }
      });
    }
    __syncthreads();
    if (live_half) wave_store_image(a.out + b * a.ldout + h0, s_o, hlen, lane);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// general constraint (src/general_constraint.jl:73-83): one lane per instance; rows/nonzeros are
// appended after the stage blocks (src/data.jl:72-75)
// ------------------------------------------------------------------------------------------------
template <class M>
__global__ void k_general_con(dto_eval_args a) {
  const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  if constexpr (M::HAS_GENERAL) {
    double o[M::General::NC > 0 ? M::General::NC : 1];
    M::General::eval(a.z + b * a.ldz, a.w + b * a.ldw, o);
    for (int i = 0; i < M::General::NC; ++i) a.out[b * a.ldout + a.general_row0 + i] = o[i];
  }
}

template <class M>
__global__ void k_general_jac(dto_eval_args a) {
  const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  if constexpr (M::HAS_GENERAL) {
    double o[M::General::NJ > 0 ? M::General::NJ : 1];
    M::General::jac(a.z + b * a.ldz, a.w + b * a.ldw, o);
    for (int i = 0; i < M::General::NJ; ++i) a.out[b * a.ldout + a.general_jac0 + i] = o[i];
  }
}

// ------------------------------------------------------------------------------------------------
// launcher used by the generated plugin
// ------------------------------------------------------------------------------------------------
template <class M>
int launch_eval(int op, const dto_eval_args* args, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const dto_eval_args& a = *args;
  const int wpi = (a.T + WAVE - 1) / WAVE;
  const unsigned grid = (unsigned)(a.B * wpi);
  switch (op) {
    case DTO_OP_OBJ:
      hipLaunchKernelGGL(k_obj<M>, dim3(grid), dim3(WAVE), 0, stream, a);
      hipLaunchKernelGGL(k_sum_rows, dim3((unsigned)a.B), dim3(WAVE), 0, stream, (const double*)a.scratch, (int64_t)a.T, a.T, a.out);
      break;
    case DTO_OP_GRAD: hipLaunchKernelGGL(k_grad<M>, dim3(grid), dim3(WAVE), 0, stream, a); break;
    case DTO_OP_CON: hipLaunchKernelGGL(k_con<M>, dim3((unsigned)(a.B * ((wpi + 3) / 4))), dim3(4 * WAVE), 0, stream, a); break;
    case DTO_OP_JAC: {
      const int bpi = (wpi + JAC_WAVES - 1) / JAC_WAVES;
      hipLaunchKernelGGL(k_jac<M>, dim3((unsigned)(a.B * bpi)), dim3(JAC_WAVES * WAVE), 0, stream, a);
      break;
    }
    case DTO_OP_HESS: {
      const int wph = (a.T + HOWN - 1) / HOWN;
      const int bph = (wph + HESS_WAVES - 1) / HESS_WAVES;
      hipLaunchKernelGGL(k_hess<M>, dim3((unsigned)(a.B * bph)), dim3(HESS_WAVES * WAVE), 0, stream, a);
      break;
    }
    case DTO_OP_GENERAL_CON:
      hipLaunchKernelGGL(k_general_con<M>, dim3((unsigned)((a.B + 63) / 64)), dim3(64), 0, stream, a);
      break;
    case DTO_OP_GENERAL_JAC:
      hipLaunchKernelGGL(k_general_jac<M>, dim3((unsigned)((a.B + 63) / 64)), dim3(64), 0, stream, a);
      break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

}  // namespace dto
