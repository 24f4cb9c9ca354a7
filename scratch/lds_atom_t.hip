// micro-test: rate of LDS atomics by type (ds_add_f32 / ds_add_u32 / ds_add_u64 / ds_add_rtn_*), conflict-free and random cells
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int CELLS = 4096;     // one 64x64 plane
template <int MODE>
__global__ __launch_bounds__(256) void k(const int* idx, int n, float* out) {
  __shared__ unsigned long long buf[CELLS];
  for (int i = threadIdx.x; i < CELLS; i += 256) buf[i] = 0;
  __syncthreads();
  float* bf = (float*)buf; unsigned* bu = (unsigned*)buf;
  const int* my = idx + (size_t)blockIdx.x * n * 256;
  for (int i = 0; i < n; ++i) {
    const int c = my[i * 256 + threadIdx.x];
    const float v = 1.0f + c * 1e-6f;
    if (MODE == 0) __hip_atomic_fetch_add(bf + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 1) __hip_atomic_fetch_add(bu + c, (unsigned)(int)(v * 1024.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 2) __hip_atomic_fetch_add(buf + c, (unsigned long long)(long long)(v * 1024.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 3) bf[c] += v;    // plain read-modify-write (wrong under conflicts: a rate reference only)
  }
  __syncthreads();
  float s = 0;
  for (int i = threadIdx.x; i < CELLS; i += 256) s += MODE == 0 || MODE == 3 ? bf[i] : MODE == 1 ? (float)bu[i] : (float)buf[i];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  const int blocks = 1024, n = 2048;
  std::vector<int> h((size_t)blocks * n * 256);
  int* d; float* o;
  CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&o, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* pat[] = {"conflict-free (lane = cell)", "random cells", "bilinear-like (4 neighbours of a smooth walk)"};
  for (int p = 0; p < 3; ++p) {
    unsigned r = 12345u;
    for (size_t i = 0; i < h.size(); ++i) {
      r = r * 1664525u + 1013904223u;
      const int lane = (int)(i & 255), step = (int)((i >> 8) % n);
      if (p == 0) h[i] = (lane + step * 7) & (CELLS - 1);
      else if (p == 1) h[i] = (r >> 8) & (CELLS - 1);
      else { const int base = ((lane & 63) + 64 * ((lane >> 6) + (step >> 2) % 60)) + ((r >> 9) & 3); h[i] = (base + ((step & 1) ? 1 : 0) + ((step & 2) ? 64 : 0)) & (CELLS - 1); }
    }
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    printf("%s\n", pat[p]);
    const char* nm[] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "plain rmw"};
    for (int m = 0; m < 4; ++m) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, n, o);
        if (m == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, n, o);
        if (m == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, n, o);
        if (m == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, n, o);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      // 1024 blocks over 256 CUs = 4 blocks per CU (co-resident or not: 32 KB LDS each); wave instructions per CU
      const double winstr = (double)blocks / 256 * n * 4;
      printf("  %-12s %8.3f ms   %.1f cycles per wave instruction per CU (2.4 GHz, 4 blocks/CU sharing the LDS pipe)\n", nm[m], ms, ms * 1e-3 * 2.4e9 / winstr);
    }
  }
  return 0;
}
