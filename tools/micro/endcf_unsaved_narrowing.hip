// tools/micro/endcf_unsaved_narrowing.hip -- the PRECONDITION of the code-generation fault of DESIGN.md section 4.3, standalone
// (something that could go upstream with the `-mllvm -amdgpu-remove-redundant-endcf=0` workaround).
//
//   hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S -o a.s tools/micro/endcf_unsaved_narrowing.hip
//   python tools/check_exec_merge.py a.s        ->  "... (1 unsaved narrowings in all)"
//   ... the same with  -mllvm -amdgpu-remove-redundant-endcf=0   ->  "(0 unsaved narrowings in all)"
//
// The shape: nested divergent regions that END TOGETHER --  if (wave == 0) { ...; if (stats) { ...; if (lane == 0) {...} } }  --
// in a kernel at the register limit.  AMD clang 22.0.0 (ROCm 7.2.0), SILowerControlFlow, removes the "redundant" restore of the
// innermost region's exec mask: the ISA narrows exec WITHOUT saving it,
//       s_and_b64 exec, exec, s[2:3]
//       s_cbranch_execz .LBB0_199
//       ...innermost body...
//     .LBB0_199:
//       s_or_b64 exec, exec, s[60:61]          <- only the ENCLOSING region's restore is left
// That is correct as long as nothing vector-valued is placed between the label and the s_or_b64.  The pass runs BEFORE register
// allocation; when live-range splitting later parks a register around the innermost block and puts the reload at the head of the
// merge block (`v_accvgpr_read_b32 v52, a34` in the product's k_wide_step: tests/golden/exec_merge_fault_excerpt.s, the committed
// excerpt of the real fault), the reload runs under the innermost mask -- one lane -- and 63 lanes keep a stale register.  This file
// reproduces the narrowing deterministically; whether a reload lands in the merge block depends on the allocator's split
// decisions (it did in two instantiations of the product kernel and in the fused lane-per-instance sweeps of rounds 3-4; four
// variants of this reduced kernel did not provoke it), which is why the build disables the transformation instead of hoping.
#include <hip/hip_runtime.h>
#define NV 48
#define NT 32
__global__ __launch_bounds__(256) void k(const double* __restrict__ in, double* __restrict__ out, double* stats, const double* fx, int n) {
  __shared__ double sh[256 * 4];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  double v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = in[tid + 256 * i];
  const int off = 8 * (tid & 63);          // a value hoisted out of the loop and used after the region
  for (int it = 0; it < n; ++it) {
    sh[tid] = v[it % NV];
    __syncthreads();
    if (w == 0) {
      double g = 0.0;
#pragma unroll
      for (int i = 0; i < NV; ++i) g += v[i] * sh[(l + i) & 255];
      sh[256 + l] = g;
      if (stats) {
        double e = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) e += sin(v[i]) * sh[256 + ((l + i) & 63)];
        sh[512 + l] = e;
        if (l == 0) {
          double t[NT];
#pragma unroll
          for (int i = 0; i < NT; ++i) t[i] = stats[8 + i] * sh[512 + i];
          double q = 0.0;
#pragma unroll
          for (int i = 0; i < NT; ++i) q += t[i] * t[(i * 7 + 3) % NT];
          stats[0] += q + sh[512 + 6];
        }
      }
    }
    __syncthreads();
    if (fx) {
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] += fx[off + (i & 7)] * sh[256 + ((off + i) & 63)];
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) out[tid + 256 * i] = v[i];
}
