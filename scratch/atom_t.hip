// micro-test: cost of "f64 atomics + release ticket + last-block finalize" vs "rows + separate kernel"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_rows(float* rows, float* dummy, int C, int spin) {
  float v = threadIdx.x * 0.001f;
  for (int i = 0; i < spin; ++i) v = fmaf(v, 1.0001f, 0.5f);
  dummy[(size_t)blockIdx.x * 256 + threadIdx.x] = v;   // a dirty line per thread, like a conv tile
  if (threadIdx.x < 2 * C) rows[(size_t)blockIdx.x * 2 * C + threadIdx.x] = v;
}
__global__ __launch_bounds__(1024) void k_fin(const float* rows, int n, int C, float* out) {
  __shared__ double red[32][32];
  const int cl = threadIdx.x & 31, tl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
  double s = 0;
  for (int t = tl; t < n; t += 32) s += rows[(size_t)t * 2 * C + c];
  red[tl][cl] = s; __syncthreads();
  for (int k = 16; k > 0; k >>= 1) { if (tl < k) red[tl][cl] += red[tl + k][cl]; __syncthreads(); }
  if (tl == 0) out[c] = (float)red[0][cl];
}
template <int MODE, int NC>
__global__ __launch_bounds__(256) void k_atom(double* acc, unsigned* ticket, float* dummy, float* out, int C, int spin) {
  float v = threadIdx.x * 0.001f;
  for (int i = 0; i < spin; ++i) v = fmaf(v, 1.0001f, 0.5f);
  dummy[(size_t)blockIdx.x * 256 + threadIdx.x] = v;
  if (threadIdx.x < 2 * C) __hip_atomic_fetch_add(acc + (blockIdx.x % NC) * 2 * C + threadIdx.x, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __shared__ unsigned last;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t;
    if (MODE == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    else { __builtin_amdgcn_s_waitcnt(0x0F70); t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    last = (t == gridDim.x - 1);
  }
  __syncthreads();
  if (last) {
    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (threadIdx.x < 2 * C) {
      double s = 0;
      for (int q = 0; q < NC; ++q) { s += __hip_atomic_load(acc + q * 2 * C + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); acc[q * 2 * C + threadIdx.x] = 0.0; }
      out[threadIdx.x] = (float)s;
    }
    if (threadIdx.x == 0) *ticket = 0;
  }
}
int main() {
  const int C = 64, spin = 2000;
  float *rows, *dummy, *out; double* acc; unsigned* ticket;
  CK(hipMalloc(&rows, 4096 * 2 * C * 4)); CK(hipMalloc(&dummy, 4096 * 256 * 4)); CK(hipMalloc(&out, 2 * C * 4));
  CK(hipMalloc(&acc, 64 * 2 * C * 8)); CK(hipMalloc(&ticket, 4));
  CK(hipMemset(acc, 0, 64 * 2 * C * 8)); CK(hipMemset(ticket, 0, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int G : {256, 768, 2048}) {
    std::vector<float> ref(2 * C), got(2 * C);
    float ms[5];
    for (int mode = 0; mode < 5; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int it = 0; it < 200; ++it) {
          if (mode == 0) { hipLaunchKernelGGL(k_rows, dim3(G), dim3(256), 0, 0, rows, dummy, C, spin);
                           hipLaunchKernelGGL(k_fin, dim3(2 * C / 32), dim3(1024), 0, 0, rows, G, C, out); }
          else if (mode == 1) hipLaunchKernelGGL((k_atom<0, 1>), dim3(G), dim3(256), 0, 0, acc, ticket, dummy, out, C, spin);
          else if (mode == 2) hipLaunchKernelGGL((k_atom<0, 8>), dim3(G), dim3(256), 0, 0, acc, ticket, dummy, out, C, spin);
          else if (mode == 3) hipLaunchKernelGGL((k_atom<0, 32>), dim3(G), dim3(256), 0, 0, acc, ticket, dummy, out, C, spin);
          else hipLaunchKernelGGL(k_rows, dim3(G), dim3(256), 0, 0, rows, dummy, C, spin);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[mode], e0, e1));
      }
      CK(hipMemcpy(mode == 0 ? ref.data() : got.data(), out, 2 * C * 4, hipMemcpyDeviceToHost));
      if (mode && mode < 4) { double d = 0; for (int i = 0; i < 2 * C; ++i) d = fmax(d, fabs((double)got[i] - ref[i]) / fabs(ref[i])); printf("  mode %d max rel diff %.3g\n", mode, d); }
    }
    printf("G=%d: rows+fin %.2f us, atom x1 %.2f, x8 %.2f, x32 %.2f, rows only %.2f us per iteration\n", G, ms[0] * 5, ms[1] * 5, ms[2] * 5, ms[3] * 5, ms[4] * 5);
  }
  return 0;
}
