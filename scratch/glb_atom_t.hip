// micro-test: rate of GLOBAL atomics by type in the pattern of the weight-gradient epilogue: `splits` workgroups per
// output block add their 18 432-element tile (64 consecutive elements per wave instruction) into the same buffer
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int TILE = 18432;
template <int MODE>
__global__ __launch_bounds__(256) void k(void* buf, float* dummy) {
  const int blk = blockIdx.y;                 // output block
  float* bf = (float*)buf + (size_t)blk * TILE;
  unsigned* bu = (unsigned*)buf + (size_t)blk * TILE;
  unsigned long long* bl = (unsigned long long*)buf + (size_t)blk * TILE;
  const float v = 1.0f + threadIdx.x * 1e-6f + blockIdx.x;
  for (int i = threadIdx.x; i < TILE; i += 256) {
    if (MODE == 0) atomicAdd(bf + i, v);
    if (MODE == 1) atomicAdd(bu + i, (unsigned)(int)(v * 1024.f));
    if (MODE == 2) atomicAdd(bl + i, (unsigned long long)(long long)(v * 1024.f));
    if (MODE == 3) bf[i + (size_t)blockIdx.x * 8 * TILE] = v;       // plain store of a slab (reference: bytes only)
  }
  if (v == 12345.f) dummy[0] = v;
}
int main() {
  void* buf; float* dummy;
  CK(hipMalloc(&buf, (size_t)512 * 8 * TILE * 8)); CK(hipMalloc(&dummy, 4));
  CK(hipMemset(buf, 0, (size_t)512 * 8 * TILE * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* nm[] = {"global_atomic_add_f32", "global_atomic_add_u32", "global_atomic_add_u64", "plain slab store"};
  for (int splits : {16, 32, 64, 128}) {
    printf("%d splits x 8 output blocks = %d workgroups, %d adds per workgroup\n", splits, splits * 8, TILE);
    for (int m = 0; m < 4; ++m) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        if (m == 0) hipLaunchKernelGGL(k<0>, dim3(splits, 8), dim3(256), 0, 0, buf, dummy);
        if (m == 1) hipLaunchKernelGGL(k<1>, dim3(splits, 8), dim3(256), 0, 0, buf, dummy);
        if (m == 2) hipLaunchKernelGGL(k<2>, dim3(splits, 8), dim3(256), 0, 0, buf, dummy);
        if (m == 3) hipLaunchKernelGGL(k<3>, dim3(splits, 8), dim3(256), 0, 0, buf, dummy);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("  %-24s %7.1f us\n", nm[m], best * 1e3);
    }
  }
  return 0;
}
