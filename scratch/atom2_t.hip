// cost of fire-and-forget per-channel atomics at the end of a kernel (no ticket, no fence)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int MODE>   // 0 rows, 1 f32 atomics, 2 f64 atomics, 3 nothing
__global__ __launch_bounds__(256) void k(float* rows, float* accf, double* accd, float* dummy, int C, int spin, int nc) {
  float v = threadIdx.x * 0.001f;
  for (int i = 0; i < spin; ++i) v = fmaf(v, 1.0001f, 0.5f);
  dummy[(size_t)blockIdx.x * 256 + threadIdx.x] = v;
  if (threadIdx.x < 2 * C) {
    if (MODE == 0) rows[(size_t)blockIdx.x * 2 * C + threadIdx.x] = v;
    if (MODE == 1) __hip_atomic_fetch_add(accf + (blockIdx.x % nc) * 2 * C + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 2) __hip_atomic_fetch_add(accd + (blockIdx.x % nc) * 2 * C + threadIdx.x, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
int main() {
  float *rows, *dummy, *accf; double* accd;
  CK(hipMalloc(&rows, 4096 * 512 * 4)); CK(hipMalloc(&dummy, 4096 * 256 * 4)); CK(hipMalloc(&accf, 64 * 512 * 4)); CK(hipMalloc(&accd, 64 * 512 * 8));
  CK(hipMemset(accf, 0, 64 * 512 * 4)); CK(hipMemset(accd, 0, 64 * 512 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int C : {32, 128}) for (int G : {256, 512}) for (int nc : {1, 8}) {
    float ms[4];
    for (int mode = 0; mode < 4; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int it = 0; it < 200; ++it) {
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(G), dim3(256), 0, 0, rows, accf, accd, dummy, C, 1000, nc);
          if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(G), dim3(256), 0, 0, rows, accf, accd, dummy, C, 1000, nc);
          if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(G), dim3(256), 0, 0, rows, accf, accd, dummy, C, 1000, nc);
          if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(G), dim3(256), 0, 0, rows, accf, accd, dummy, C, 1000, nc);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[mode], e0, e1));
      }
    }
    printf("C=%d G=%d copies=%d: rows %.2f us, f32 atomics %.2f, f64 atomics %.2f, none %.2f\n", C, G, nc, ms[0] * 5, ms[1] * 5, ms[2] * 5, ms[3] * 5);
  }
  return 0;
}
