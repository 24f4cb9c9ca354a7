// The one GEMM-shaped layer of the network: the 1x1 convolution of the head that maps the 480-channel concatenated
// features onto 480 channels (pose_hrnet.py:334-340, last_layer.0) - 262144 pixels x 480 x 480 at batch 64, 121 GFLOP
// forward and the same again for its input gradient. The tile-walking conv body runs it at 330-520 TFLOP/s with
// 128-wide output-channel blocks (the pixel operand is re-read once per block); here ONE workgroup keeps all output
// channels of a 128-pixel block in its accumulators, so the pixel operand is read exactly once and the weight
// matrix streams from L2:
//
//   y[p, n] = sum_k x[p, k] * w[n, k]  (+ bias[n]),  per-channel (sum y, sum y^2) added into sums[8][2][N]
//
//   512 threads = 8 waves as 4 (channel blocks of 128) x 2 (pixel blocks of 64); MFMA 16x16x32 bf16 with the
//   channel dimension as rows (a lane ends up with 8 contiguous channels of one pixel: 16-byte stores);
//   K in chunks of 32: the x chunk (128 x 32) and the w chunk (512 x 32, rows beyond N zero) are double-buffered
//   in LDS (2 x 50 KB), the global loads of chunk k+1 are issued before the MFMAs of chunk k.
//
// bf16, Cin a multiple of 32, 256 <= Cout <= 512. Serves the forward launch (bias, statistics by float atomics)
// and the input-gradient launch (transposed packed weights, no epilogue).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

struct GemmArgs {
  const char* x;      // [P][K] bf16
  const char* w;      // [N][K] bf16 (packed weights of the 1x1 conv, or its transposed copy for the input gradient)
  const float* bias;  // [N] or NULL
  char* y;            // [P][N] bf16
  float* sums;        // [8][2][N] f32 (atomic) or NULL
  long long P;
  int K, N;
};

constexpr int G_BM = 128, G_BN = 512, G_BK = 32, G_NT = 512;
constexpr int G_ROWB = G_BK * 2 + 16;              // padded LDS row: 80 bytes (conflict-free 16-byte operand reads)
constexpr int G_XB = G_BM * G_ROWB, G_WB = G_BN * G_ROWB;

__global__ __launch_bounds__(G_NT) void gemm_pw_kernel(GemmArgs a) {
  typedef bf16_t T;
  __shared__ __attribute__((aligned(16))) char lds[2 * (G_XB + G_WB)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wn = wave & 3, wp = wave >> 2;         // channel block (128) / pixel block (64) of this wave
  const long long p0 = (long long)blockIdx.x * G_BM;
  const int nk = a.K / G_BK;

  // staging: thread -> one 16-byte vector of the x chunk (128 rows x 4 vectors) and four of the w chunk
  const int xrow = tid >> 2, xv = tid & 3;
  const bool xok = p0 + xrow < a.P;
  const char* xsrc = a.x + ((size_t)(xok ? p0 + xrow : 0) * a.K + xv * 8) * 2;
  // LDS row q of the w chunk holds the output channel the MFMA row order needs (8 contiguous channels per lane)
  int wrow[4];
  const char* wsrc[4];
  bool wok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = (tid >> 2) + j * 128;
    const int n = (q & ~31) + ((q & 15) >> 2) * 8 + ((q >> 4) & 1) * 4 + (q & 3);
    wrow[j] = q;
    wok[j] = n < a.N;
    wsrc[j] = a.w + ((size_t)(wok[j] ? n : 0) * a.K + xv * 8) * 2;
  }
  // (a second register set with the loads two chunks ahead was measured slower: 317 vs 290 us per forward launch -
  // the kernel is bound by LDS operand traffic and the per-chunk barrier, not by load latency)
  V16 rx, rw[4];
  auto load_chunk = [&](int k) {
    rx = *(const V16*)(xsrc + (size_t)k * G_BK * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) rw[j] = *(const V16*)(wsrc[j] + (size_t)k * G_BK * 2);
  };
  auto store_chunk = [&](int buf) {
    char* xl = lds + buf * (G_XB + G_WB);
    char* wl = xl + G_XB;
    *(V16*)(xl + xrow * G_ROWB + xv * 16) = xok ? rx : v16_zero();
#pragma unroll
    for (int j = 0; j < 4; ++j) *(V16*)(wl + wrow[j] * G_ROWB + xv * 16) = wok[j] ? rw[j] : v16_zero();
  };

  f32x4 acc[8][4];      // [channel fragment][pixel fragment]
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[f][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int aoff = (wn * 128 + li) * G_ROWB + lg * 16;     // + f * 16 rows
  const int boff = (wp * 64 + li) * G_ROWB + lg * 16;      // + g * 16 rows

  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int k = 0; k < nk; ++k) {
    const int buf = k & 1;
    if (k + 1 < nk) load_chunk(k + 1);                     // in flight while the MFMAs below run
    const char* xl = lds + buf * (G_XB + G_WB);
    const char* wl = xl + G_XB;
    V16 bf[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bf[g] = *(const V16*)(xl + boff + g * 16 * G_ROWB);
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const V16 af = *(const V16*)(wl + aoff + f * 16 * G_ROWB);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[f][g] = mma16<T>(af, bf[g], acc[f][g]);
    }
    if (k + 1 < nk) store_chunk(buf ^ 1);                  // the other buffer was last read in iteration k-1
    __syncthreads();
  }

  // ---- epilogue: bias, store (8 contiguous channels per lane and 32-channel group), statistics ----
  const bool stats = a.sums != nullptr;
  float* sl = (float*)lds;            // [2 pixel blocks][2][512] partial sums (the chunk buffers are free)
#pragma unroll
  for (int j = 0; j < 4; ++j) {       // 32-channel group j of this wave's 128 channels
    const int n0 = wn * 128 + j * 32 + lg * 8;
    float b8[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) b8[c] = (a.bias && n0 + c < a.N) ? a.bias[n0 + c] : 0.f;
    float s1[8], s2[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) s1[c] = s2[c] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const long long p = p0 + wp * 64 + g * 16 + li;
      float v[8];
      v[0] = acc[2 * j][g].x; v[1] = acc[2 * j][g].y; v[2] = acc[2 * j][g].z; v[3] = acc[2 * j][g].w;
      v[4] = acc[2 * j + 1][g].x; v[5] = acc[2 * j + 1][g].y; v[6] = acc[2 * j + 1][g].z; v[7] = acc[2 * j + 1][g].w;
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] += b8[c];
      if (p < a.P && n0 < a.N) {
        *(V16*)(a.y + ((size_t)p * a.N + n0) * 2) = v16_pack<T>(v);
        if (stats) {
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            s1[c] += v[c];
            s2[c] = fmaf(v[c], v[c], s2[c]);
          }
        }
      }
    }
    if (stats) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        s1[c] = wave_sum16(s1[c]);
        s2[c] = wave_sum16(s2[c]);
      }
      if (li == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          sl[(wp * 2 + 0) * G_BN + n0 + c] = s1[c];
          sl[(wp * 2 + 1) * G_BN + n0 + c] = s2[c];
        }
      }
    }
  }
  if (stats) {
    __syncthreads();
    for (int o = tid; o < 2 * G_BN; o += G_NT) {
      const int which = o / G_BN, n = o % G_BN;
      if (n < a.N)
        atomicAdd(a.sums + ((size_t)(blockIdx.x & (HR_BN_COPIES - 1)) * 2 + which) * a.N + n,
                  sl[(0 * 2 + which) * G_BN + n] + sl[(1 * 2 + which) * G_BN + n]);
    }
  }
}

}  // namespace

// 1 if hr_gemm_pw serves this 1x1 stride-1 conv launch (bf16, the head's 480 -> 480 shape class)
int hr_gemm_pw_supported(int dtype, int Cin, int Cout) {
  static const bool off = hr_knob("HRNET_GEMM_PW", 1) == 0;
  return !off && dtype == HR_BF16 && Cin % 32 == 0 && Cin >= 256 && Cout >= 256 && Cout <= G_BN && Cout % 8 == 0;
}

int hr_gemm_pw(const void* x, const void* w, const float* bias, void* y, float* sums, long long pixels, int Cin,
               int Cout, hipStream_t s) {
  HR_REQUIRE(x && w && y && pixels > 0, "gemm_pw: null pointer / empty shape");
  GemmArgs a;
  a.x = (const char*)x; a.w = (const char*)w; a.bias = bias; a.y = (char*)y; a.sums = sums;
  a.P = pixels; a.K = Cin; a.N = Cout;
  const long long blocks = (pixels + G_BM - 1) / G_BM;
  HR_REQUIRE(blocks < (1ll << 31), "gemm_pw: pixel count");
  hipLaunchKernelGGL(gemm_pw_kernel, dim3((unsigned)blocks), dim3(G_NT), 0, s, a);
  return hr_check_launch("gemm_pw");
}
