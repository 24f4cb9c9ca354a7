#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  int l = threadIdx.x; int i = l & 15, g = l >> 4;
  // A[i][k] = 100*i + k ; B[k][j] = (k==2) ? (j+1) : 0
  float a = 100.f * i + g;
  float b = (g == 2) ? (float)(i + 1) : 0.f;
  f32x4 c = {0,0,0,0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  // expect D[i][j] = A[i][2]*B[2][j] = (100 i + 2)(j+1); lane l: col j=l&15, row 4*(l>>4)+r
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    int j = l & 15, i = 4 * (l >> 4) + r;
    float e = (100.f * i + 2) * (j + 1);
    if (h[l*4+r] != e) { if (bad < 5) printf("l=%d r=%d got %f exp %f\n", l, r, h[l*4+r], e); bad++; }
  }
  printf("bad=%d\n", bad);
  return 0;
}
