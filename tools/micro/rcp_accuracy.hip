// Accuracy of v_rcp_f64 + Newton steps against IEEE division (the pivots' reciprocals in ldl_inplace):
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/rcp_accuracy tools/micro/rcp_accuracy.hip && /tmp/rcp_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, double* rd, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r = __builtin_amdgcn_rcp(v);
  r0[i] = r;
  r = fma(fma(-v, r, 1.0), r, r);
  r1[i] = r;
  r = fma(fma(-v, r, 1.0), r, r);
  r2[i] = r;
  rd[i] = 1.0 / v;
}
static double ulps(double a, double ref) {
  int64_t ia, ib; memcpy(&ia, &a, 8); memcpy(&ib, &ref, 8);
  return (double)llabs(ia - ib);
}
int main() {
  const int n = 1 << 22;
  std::vector<double> x(n);
  uint64_t s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double m = 1.0 + (double)(s >> 11) / 9007199254740992.0;
    const int e = (int)((s >> 3) % 240) - 120;
    x[i] = ((s & 1) ? -1.0 : 1.0) * ldexp(m, e);
  }
  double *dx, *d0, *d1, *d2, *dd;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&dd, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, d0, d1, d2, dd, n);
  std::vector<double> r0(n), r1(n), r2(n), rd(n);
  hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(rd.data(), dd, n * 8, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, m2 = 0, md = 0; long ne2 = 0;
  for (int i = 0; i < n; ++i) {
    const double ref = 1.0 / x[i];
    m0 = fmax(m0, ulps(r0[i], ref)); m1 = fmax(m1, ulps(r1[i], ref)); m2 = fmax(m2, ulps(r2[i], ref)); md = fmax(md, ulps(rd[i], ref));
    if (r2[i] != ref) ++ne2;
  }
  printf("max ulp error vs host 1/x over %d values: v_rcp_f64 %.0f, +1 Newton %.0f, +2 Newton %.0f (%.3f %% not bit-equal), device division %.0f\n",
         n, m0, m1, m2, 100.0 * ne2 / n, md);
  return 0;
}
