// Microbenchmark: HBM store ceiling for the Jacobian write pattern (851 MB per launch).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)
constexpr int NJ = 26;
__global__ void fill16(double2* out, size_t n2) {            // fully coalesced, 16 B per lane
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) out[i] = make_double2(1.0, 2.0);
}
__global__ void fill8(double* out, size_t n) {                // fully coalesced, 8 B per lane
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = 1.0;
}
__global__ void fill_seg(double* out, size_t nknots) {       // k_jac pattern: lane owns 26 contiguous doubles
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nknots) return;
  double* d = out + k * NJ;
#pragma unroll
  for (int i = 0; i < NJ; ++i) d[i] = (double)i;
}
__global__ void copy_seg(const double* in, double* out, size_t nknots) {  // + the 40-byte-stride reads
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nknots) return;
  const double* s = in + k * 5;
  double acc = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) acc += s[i];
  double* d = out + k * NJ;
#pragma unroll
  for (int i = 0; i < NJ; ++i) d[i] = acc + i;
}
int main() {
  const size_t nknots = 4096ull * 1000, n = nknots * NJ;
  double *out, *in;
  CK(hipMalloc(&out, n * 8)); CK(hipMalloc(&in, (nknots * 5 + 16) * 8)); CK(hipMemset(in, 0, (nknots * 5 + 16) * 8));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch, double bytes) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-28s %8.3f ms  %8.1f GB/s\n", name, ms, bytes / (ms * 1e-3) / 1e9);
  };
  run("fill16 (grid 2048x256)", [&] { hipLaunchKernelGGL(fill16, dim3(2048), dim3(256), 0, 0, (double2*)out, n / 2); }, n * 8.0);
  run("fill16 (grid 16384x256)", [&] { hipLaunchKernelGGL(fill16, dim3(16384), dim3(256), 0, 0, (double2*)out, n / 2); }, n * 8.0);
  run("fill8  (grid 4096x256)", [&] { hipLaunchKernelGGL(fill8, dim3(4096), dim3(256), 0, 0, out, n); }, n * 8.0);
  run("fill_seg 26 doubles/lane", [&] { hipLaunchKernelGGL(fill_seg, dim3((nknots + 255) / 256), dim3(256), 0, 0, out, nknots); }, n * 8.0);
  run("copy_seg (+9 strided reads)", [&] { hipLaunchKernelGGL(copy_seg, dim3((nknots + 255) / 256), dim3(256), 0, 0, in, out, nknots); }, n * 8.0 + nknots * 40.0);
  return 0;
}
