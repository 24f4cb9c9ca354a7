// does hipExtAnyOrderLaunch let back-to-back launches on ONE stream overlap on gfx950?  (hipcc --offload-arch=gfx950 -O2)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long cycles, int* sink) {
  long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < cycles) {}
  if (sink && threadIdx.x == 9999) *sink = 1;
}
int main() {
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int* sink; hipMalloc(&sink, 4);
  for (int flags = 0; flags < 2; ++flags)
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, s);
      for (int k = 0; k < 8; ++k)
        hipExtLaunchKernelGGL(spin, dim3(32), dim3(256), 0, s, nullptr, nullptr, (k % 4 == 0) ? 0 : flags, 5000L, sink);   // 5000 ticks of 100 MHz = 50 us
      hipEventRecord(e1, s);
      hipError_t err = hipStreamSynchronize(s);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("flags=%d rep=%d: 8 launches of 32 WGs x 50us: %.1f us (%s)\n", flags, rep, ms * 1e3, hipGetErrorString(err));
    }
  return 0;
}
