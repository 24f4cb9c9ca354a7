// The head without its 480-channel concat (pose_hrnet.py:560-566).
//
//   last_layer.0( cat(x0, up(x1), up(x2), up(x3)) ) = W0 x0 + up(W1 x1) + up(W2 x2) + up(W3 x3) + b
//
// (a 1x1 convolution commutes with bilinear upsampling). The three low-resolution products t_j = W_j x_j are plain
// 1x1 conv launches at THEIR resolution; hrnet_head_mix forms W0 x0 on the full-resolution grid (K = C0), adds the
// bias and the bilinearly upsampled t_j in its epilogue, stores the raw output once and gathers its batch statistics:
// 1/8 of the FLOPs of the concat form, and neither the concat nor its gradient ever exist. Backward: the gradient
// of t_j is the TRANSPOSE of the upsampling applied to all channels of G = d(raw output) - hrnet_upsample_bilinear_t
// here (tile form, up to three scales in one pass over G) and in eltwise.hip (streamed form, any scale).
//
// Both tile kernels are built around one fact measured on this chip: a lane-level gather from L2 costs a wave ~100
// cycles per instruction at the occupancy these kernels reach (the first version of the forward kernel issued its 12
// taps per output vector straight from global memory and ran 449 us for 252 MB of output), so the low-resolution
// operands of a 16x16-pixel tile are staged in LDS once (a few KB per scale) and every tap is a ds_read_b128.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int HM_T = 16;        // tile edge in pixels
constexpr int HM_CH = 96;       // output channels per workgroup (3 sub-blocks of 32): 42 KB of LDS, 3 workgroups per CU
constexpr int HM_TROW = HM_CH * 2 + 16;   // LDS bytes per staged low-resolution pixel (bf16, padded)

struct HeadMixArgs {
  const char* x;      // [N][H][W][K]
  const char* w;      // packed [Cout][K]
  const float* bias;  // [Cout] or NULL
  char* y;            // [N][H][W][Cout]
  float* sums;        // [8][2][Cout] (float atomics) or NULL
  float* rows;        // [N * tiles][2][Cout], one row per pixel tile (deterministic) or NULL
  const char* up[3];  // [N][uh][uw][Cout]
  int uh[3], uw[3];
  int cap[3];         // staged rows / columns per scale (capacity)
  int toff[3];        // LDS byte offset of each scale's staging area
  int nup, N, H, W, K, Cout, align;
  int tiles_x, tiles_y, chunks;
  int wrow;           // LDS bytes per weight row (K * 2 + 16)
  int sl_off;         // LDS byte offset of the per-wave statistics [4][2][HM_CH] floats
  // backward of the layer BEHIND y (hrnet_head_bwd; MODE 1 / 2 of the kernels): x = dY of that layer [N][H][W][K],
  // w = its transposed packed weight [Cout][K], so the product is d(ReLU output) of this layer - never stored
  const char* yin;        // raw y [N][H][W][Cout]
  const float* bn_scale;  // [Cout] BatchNorm affine of y (the ReLU mask is [scale*y + shift > 0]) or NULL: mask [y > 0]
  const float* bn_shift;
  const float* coef;      // MODE 2: [3][Cout] A, B, C of hrnet_bn_bwd_finalize
  int inner_relu;
};

// bf16. 256 threads = 4 waves; wave v owns tile rows 4v .. 4v+3 (64 pixels = 4 MFMA pixel fragments; lane (li, lg)
// holds column li of those rows) and walks the workgroup's channels in sub-blocks of 32 (two A fragments whose rows are
// permuted so that a lane ends up with 8 CONTIGUOUS channels of each of its 4 pixels: 16-byte taps, 16-byte stores).
// The upsampling is evaluated separably per lane: the rows of a scale the wave's 4 pixel rows touch (<= HM_MAXR, the
// same for all lanes: scalar registers) are interpolated along x once (2 taps), and each pixel row adds its two
// y-weighted copies - 18 LDS reads and ~480 VALU operations per 32 outputs of a lane instead of 48 and ~1080.
constexpr int HM_MAXR = 5;
// MODE 0: the forward mix. MODE 1 / 2 (hrnet_head_bwd): the same K-step product is the gradient dz of the ReLU output
// of y's BatchNorm (x = the gradient of the NEXT layer's output, w = that layer's transposed weight); it is masked
// by the ReLU and either reduced to the BatchNorm-backward sums (1: rows of (sum dz, sum dz*y), one per pixel tile)
// or turned into G = A*dz + B*y + C and stored (2) - dz itself (252 MB at batch 64) is never written or re-read.
template <int MODE>
__global__ __launch_bounds__(256, 3) void head_mix_tile_kernel(HeadMixArgs a) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  // (workgroup ids go round-robin over the 8 XCDs: logical index L gives every XCD a contiguous run of tiles, so
  // neighbouring tiles - which stage overlapping low-resolution pixels - and the channel chunks of a tile share an L2)
  int b = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
  if (b >= a.N * a.tiles_y * a.tiles_x * a.chunks) return;
  const int chunk = b % a.chunks; b /= a.chunks;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int img = b / a.tiles_y;
  const int tile_id = (img * a.tiles_y + ty) * a.tiles_x + tx;
  const int c_base = chunk * HM_CH;
  const int Y0 = ty * HM_T, X0 = tx * HM_T;
  const int Ylast = min(Y0 + HM_T, a.H) - 1, Xlast = min(X0 + HM_T, a.W) - 1;

  // ---- staging: the weights of this channel chunk (LDS row q holds the output channel the MFMA row order needs) and
  // rows [ylo, ..] x columns [xlo, ..] of each t_u for this chunk's channels. ALL global loads are issued before the
  // first LDS store (<= 2 + 5 + 5 + 5 16-byte loads per thread): a workgroup pays one memory round trip, not one per
  // staging loop iteration.
  const int kv = a.K / 8;      // 16-byte vectors per weight row
  constexpr int cvn = HM_CH / 8;
  constexpr int WB = 2, TB = 5;        // loads per thread and batch: weights / one low-resolution tile
  int ylo[3], xlo[3], rw[3], tot[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    ylo[u] = xlo[u] = 0; rw[u] = 1; tot[u] = 0;
    if (u < a.nup) {
      int i0, i1, j0, j1;
      float l;
      bilin_src(Y0, a.uh[u], a.H, a.align, i0, i1, l);
      bilin_src(Ylast, a.uh[u], a.H, a.align, j0, j1, l);
      ylo[u] = i0;
      const int rh = min(j1 - i0 + 1, a.cap[u]);
      bilin_src(X0, a.uw[u], a.W, a.align, i0, i1, l);
      bilin_src(Xlast, a.uw[u], a.W, a.align, j0, j1, l);
      xlo[u] = i0;
      rw[u] = min(j1 - i0 + 1, a.cap[u]);
      tot[u] = rh * rw[u] * cvn;
    }
  }
  const int wtot = HM_CH * kv;
  auto wload = [&](int i) -> V16 {
    const int q = i / kv, v = i - q * kv;
    const int n = c_base + (q & ~31) + ((q & 15) >> 2) * 8 + ((q >> 4) & 1) * 4 + (q & 3);
    return (i < wtot && n < a.Cout) ? *(const V16*)(a.w + ((size_t)n * a.K + v * 8) * 2) : v16_zero();
  };
  auto tload = [&](int u, int i) -> V16 {
    const int cv = i % cvn, px = i / cvn;
    const int r = px / rw[u], c = px - r * rw[u];
    const char* src = a.up[u] + ((size_t)img * a.uh[u] * a.uw[u] * a.Cout + c_base) * 2;
    return (i < tot[u] && c_base + cv * 8 < a.Cout)
               ? *(const V16*)(src + ((size_t)(ylo[u] + r) * a.uw[u] + xlo[u] + c) * a.Cout * 2 + cv * 16)
               : v16_zero();
  };
  {
    V16 wv[WB], tv[3][TB];
#pragma unroll
    for (int q = 0; q < WB; ++q) wv[q] = wload(q * 256 + tid);
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int q = 0; q < TB; ++q) tv[u][q] = u < a.nup ? tload(u, q * 256 + tid) : v16_zero();
#pragma unroll
    for (int q = 0; q < WB; ++q) {
      const int i = q * 256 + tid;
      if (i < wtot) *(V16*)(lds + (i / kv) * a.wrow + (i % kv) * 16) = wv[q];
    }
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int q = 0; q < TB; ++q) {
        const int i = q * 256 + tid;
        if (i < tot[u]) *(V16*)(lds + a.toff[u] + (i / cvn) * HM_TROW + (i % cvn) * 16) = tv[u][q];
      }
  }
  // (larger problems: the rest, batch by batch)
  for (int i0 = WB * 256; i0 < wtot; i0 += 256) {
    const int i = i0 + tid;
    const V16 v = wload(i);
    if (i < wtot) *(V16*)(lds + (i / kv) * a.wrow + (i % kv) * 16) = v;
  }
#pragma unroll
  for (int u = 0; u < 3; ++u)
    for (int i0 = TB * 256; i0 < tot[u]; i0 += 256) {
      const int i = i0 + tid;
      const V16 v = tload(u, i);
      if (i < tot[u]) *(V16*)(lds + a.toff[u] + (i / cvn) * HM_TROW + (i % cvn) * 16) = v;
    }
  // ---- this lane's column: x taps per scale (vector registers) ----
  const int ox = X0 + li;
  const bool xok = ox < a.W;
  int xo0[3], xdx[3];
  float xl[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    xo0[u] = xdx[u] = 0; xl[u] = 0.f;
    if (u < a.nup && xok) {
      int x0, x1;
      float lx;
      bilin_src(ox, a.uw[u], a.W, a.align, x0, x1, lx);
      xo0[u] = a.toff[u] + (x0 - xlo[u]) * HM_TROW;
      xdx[u] = (x1 - x0) * HM_TROW;
      xl[u] = lx;
    }
  }
  // ---- this wave's 4 pixel rows: y taps per scale (the same for every lane: scalar registers) ----
  const int oy0 = Y0 + wave * 4;
  // (float arithmetic runs on the vector ALU: the results are moved to scalar registers explicitly)
  int rb[3], nr[3], ry0[3][4], ry1[3][4];
  float rwa[3][4], rwb[3][4];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    rb[u] = 0; nr[u] = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) { ry0[u][g] = -1; ry1[u][g] = -1; rwa[u][g] = 0.f; rwb[u][g] = 0.f; }
    if (u < a.nup) {
      int last = 0, first = 0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int oy = min(oy0 + g, a.H - 1);
        int y0, y1;
        float ly;
        bilin_src(oy, a.uh[u], a.H, a.align, y0, y1, ly);
        if (g == 0) first = y0;
        ry0[u][g] = __builtin_amdgcn_readfirstlane(y0 - first);
        ry1[u][g] = __builtin_amdgcn_readfirstlane(y1 - first);
        rwa[u][g] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.f - ly)));
        rwb[u][g] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ly)));
        last = y1;
      }
      nr[u] = __builtin_amdgcn_readfirstlane(min(last - first + 1, HM_MAXR));
      rb[u] = __builtin_amdgcn_readfirstlane(first - ylo[u]);
    }
  }
  // B fragments of the 4 pixel rows (K <= 128: up to 4 K steps stay in registers only for K = 32; reloaded otherwise)
  const int nk = (a.K + 31) / 32;     // (K = 48: the second K step is half empty - its upper lane groups feed zeros)
  bool pok[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) pok[g] = oy0 + g < a.H && xok;
  const char* xp0 = a.x + (((size_t)img * a.H + min(oy0, a.H - 1)) * a.W + min(ox, a.W - 1)) * a.K * 2 + lg * 16;
  const size_t xrow = (size_t)a.W * a.K * 2;          // bytes between pixel rows
  V16 bf0[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bf0[g] = (pok[g] && lg * 8 < a.K) ? *(const V16*)(xp0 + g * xrow) : v16_zero();
  __syncthreads();

  float* sl = (float*)(lds + a.sl_off);      // [4 waves][2][HM_CH]
  const bool stats = MODE == 1 || (MODE == 0 && (a.sums != nullptr || a.rows != nullptr));
#pragma unroll 1
  for (int sb = 0; sb < HM_CH / 32; ++sb) {
    const int n0 = c_base + sb * 32 + lg * 8;       // this lane's 8 output channels
    if (c_base + sb * 32 >= a.Cout) break;          // (uniform)
    f32x4 acc[2][4];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[f][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int k = 0; k < nk; ++k) {
      V16 bf[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
        bf[g] = k == 0 ? bf0[g] : ((pok[g] && k * 32 + lg * 8 < a.K) ? *(const V16*)(xp0 + g * xrow + k * 64) : v16_zero());
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const V16 af = (k * 32 + lg * 8 < a.K) ? *(const V16*)(lds + (sb * 32 + f * 16 + li) * a.wrow + k * 64 + lg * 16)
                                               : v16_zero();
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[f][g] = mma16<T>(af, bf[g], acc[f][g]);
      }
    }
    float v[4][8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      v[g][0] = acc[0][g].x; v[g][1] = acc[0][g].y; v[g][2] = acc[0][g].z; v[g][3] = acc[0][g].w;
      v[g][4] = acc[1][g].x; v[g][5] = acc[1][g].y; v[g][6] = acc[1][g].z; v[g][7] = acc[1][g].w;
    }
    {
      float b8[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) b8[c] = (a.bias && n0 + c < a.Cout) ? a.bias[n0 + c] : 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int c = 0; c < 8; ++c) v[g][c] += b8[c];
    }
    const int coff = (sb * 32 + lg * 8) * 2;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      if (u < a.nup) {
        const float lx = xl[u], hx = 1.f - lx;
        const char* p = lds + xo0[u] + coff + rb[u] * rw[u] * HM_TROW;
#pragma unroll
        for (int r = 0; r < HM_MAXR; ++r) {
          if (r < nr[u]) {                 // (scalar)
            float f0[8], f1[8], xr[8];
            v16_unpack<T>(*(const V16*)(p + r * rw[u] * HM_TROW), f0);
            v16_unpack<T>(*(const V16*)(p + r * rw[u] * HM_TROW + xdx[u]), f1);
#pragma unroll
            for (int c = 0; c < 8; ++c) xr[c] = hx * f0[c] + lx * f1[c];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              // staged row r in pixel row g: weight (1 - ly) if it is the row's first tap, ly if its second (scalar)
              if (ry0[u][g] == r) {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[g][c] = fmaf(rwa[u][g], xr[c], v[g][c]);
              }
              if (ry1[u][g] == r) {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[g][c] = fmaf(rwb[u][g], xr[c], v[g][c]);
              }
            }
          }
        }
      }
    }
    float s1[8], s2[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) s1[c] = s2[c] = 0.f;
    if constexpr (MODE == 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (pok[g] && n0 < a.Cout) {
          *(V16*)(a.y + ((((size_t)img * a.H + oy0 + g) * a.W + ox) * a.Cout + n0) * 2) = v16_pack<T>(v[g]);
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            s1[c] += v[g][c];
            s2[c] = fmaf(v[g][c], v[g][c], s2[c]);
          }
        }
      }
    } else {
      const bool cok = n0 < a.Cout;
      V16 yv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
        yv[g] = (pok[g] && cok) ? *(const V16*)(a.yin + ((((size_t)img * a.H + oy0 + g) * a.W + ox) * a.Cout + n0) * 2)
                                : v16_zero();
      float sc[8], sf[8], cA[8], cB[8], cC[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        sc[c] = (a.bn_scale && cok) ? a.bn_scale[n0 + c] : 1.f;
        sf[c] = (a.bn_shift && cok) ? a.bn_shift[n0 + c] : 0.f;
        if constexpr (MODE == 2) {
          cA[c] = cok ? a.coef[n0 + c] : 0.f;
          cB[c] = cok ? a.coef[a.Cout + n0 + c] : 0.f;
          cC[c] = cok ? a.coef[2 * a.Cout + n0 + c] : 0.f;
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float yf[8], o[8];
        v16_unpack<T>(yv[g], yf);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float z = fmaf(yf[c], sc[c], sf[c]);
          const float dz = (!a.inner_relu || z > 0.f) ? v[g][c] : 0.f;
          if constexpr (MODE == 1) {
            s1[c] += dz;
            hr_fma_acc(s2[c], dz, yf[c]);
          } else {
            o[c] = fmaf(cA[c], dz, fmaf(cB[c], yf[c], cC[c]));
          }
        }
        if constexpr (MODE == 2) {
          if (pok[g] && cok)
            *(V16*)(a.y + ((((size_t)img * a.H + oy0 + g) * a.W + ox) * a.Cout + n0) * 2) = v16_pack<T>(o);
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
          sl[(wave * 2 + 0) * HM_CH + sb * 32 + lg * 8 + c] = s1[c];
          sl[(wave * 2 + 1) * HM_CH + sb * 32 + lg * 8 + c] = s2[c];
        }
      }
    }
  }
  if (stats) {
    __syncthreads();
    for (int o = tid; o < 2 * HM_CH; o += 256) {
      const int which = o / HM_CH, c = o % HM_CH, n = c_base + c;
      if (n < a.Cout) {
        const float v = ((sl[(0 * 2 + which) * HM_CH + c] + sl[(1 * 2 + which) * HM_CH + c]) +
                         sl[(2 * 2 + which) * HM_CH + c]) + sl[(3 * 2 + which) * HM_CH + c];
        if (a.rows) a.rows[((size_t)tile_id * 2 + which) * a.Cout + n] = v;
        else atomicAdd(a.sums + ((size_t)(tile_id & (HR_BN_COPIES - 1)) * 2 + which) * a.Cout + n, v);
      }
    }
  }
}

// fp32 form (the validation path: plain FMAs, no MFMA, taps from global memory). One workgroup per pixel tile, as
// above (so the rows of the deterministic statistics mean the same thing); thread (half, cv) owns the 4-channel
// vector cv over the tile's even or odd rows, in pixel order - its sums are formed in a fixed order.
template <int MODE>
__global__ __launch_bounds__(256) void head_mix_f32_kernel(HeadMixArgs a) {
  __shared__ float sl[2][2][512];
  const int tid = threadIdx.x, half = tid >> 7, cv = tid & 127;
  const int ncv = a.Cout / 4, n0 = cv * 4;
  int b = blockIdx.x;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int img = b / a.tiles_y;
  const bool stats = MODE == 1 || (MODE == 0 && (a.sums != nullptr || a.rows != nullptr));
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (cv < ncv) {
    const float* w = (const float*)a.w;
    float b4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) b4[c] = a.bias ? a.bias[n0 + c] : 0.f;
    // (K <= 32, the head's shapes: this thread's 4 weight rows stay in registers for the whole tile)
    const bool wreg = a.K <= 32;
    f32x4 wr[4][8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        wr[c][k] = (wreg && k * 4 < a.K) ? *(const f32x4*)(w + (size_t)(n0 + c) * a.K + k * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int q = half; q < HM_T * HM_T; q += 2) {
      const int oy = ty * HM_T + q / HM_T, ox = tx * HM_T + q % HM_T;
      if (oy >= a.H || ox >= a.W) continue;
      const size_t p = ((size_t)img * a.H + oy) * a.W + ox;
      const float* x = (const float*)a.x + p * a.K;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (wreg) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (k * 4 < a.K) {
            const f32x4 xv = *(const f32x4*)(x + k * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              v[c] = fmaf(xv.x, wr[c][k].x, v[c]); v[c] = fmaf(xv.y, wr[c][k].y, v[c]);
              v[c] = fmaf(xv.z, wr[c][k].z, v[c]); v[c] = fmaf(xv.w, wr[c][k].w, v[c]);
            }
          }
        }
      } else {
        for (int k = 0; k < a.K; k += 4) {
          const f32x4 xv = *(const f32x4*)(x + k);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x4 wv = *(const f32x4*)(w + (size_t)(n0 + c) * a.K + k);
            v[c] = fmaf(xv.x, wv.x, v[c]); v[c] = fmaf(xv.y, wv.y, v[c]);
            v[c] = fmaf(xv.z, wv.z, v[c]); v[c] = fmaf(xv.w, wv.w, v[c]);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] += b4[c];
      for (int u = 0; u < a.nup; ++u) {
        const int hs = a.uh[u], ws = a.uw[u];
        int y0, y1, x0, x1;
        float ly, lx;
        bilin_src(oy, hs, a.H, a.align, y0, y1, ly);
        bilin_src(ox, ws, a.W, a.align, x0, x1, lx);
        const size_t base = (size_t)img * hs * ws;
        const float* src = (const float*)a.up[u] + n0;
        const f32x4 f00 = *(const f32x4*)(src + (base + (size_t)y0 * ws + x0) * a.Cout);
        const f32x4 f01 = *(const f32x4*)(src + (base + (size_t)y0 * ws + x1) * a.Cout);
        const f32x4 f10 = *(const f32x4*)(src + (base + (size_t)y1 * ws + x0) * a.Cout);
        const f32x4 f11 = *(const f32x4*)(src + (base + (size_t)y1 * ws + x1) * a.Cout);
        const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] += hy * (hx * f00[c] + lx * f01[c]) + ly * (hx * f10[c] + lx * f11[c]);
      }
      if constexpr (MODE == 0) {
        *(f32x4*)((float*)a.y + p * a.Cout + n0) = f32x4{v[0], v[1], v[2], v[3]};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          s1[c] += v[c];
          s2[c] = fmaf(v[c], v[c], s2[c]);
        }
      } else {
        const f32x4 yv = *(const f32x4*)((const float*)a.yin + p * a.Cout + n0);
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float z = a.bn_scale ? fmaf(yv[c], a.bn_scale[n0 + c], a.bn_shift[n0 + c]) : yv[c];
          const float dz = (!a.inner_relu || z > 0.f) ? v[c] : 0.f;
          if constexpr (MODE == 1) {
            s1[c] += dz;
            hr_fma_acc(s2[c], dz, yv[c]);
          } else {
            o[c] = fmaf(a.coef[n0 + c], dz, fmaf(a.coef[a.Cout + n0 + c], yv[c], a.coef[2 * a.Cout + n0 + c]));
          }
        }
        if constexpr (MODE == 2) *(f32x4*)((float*)a.y + p * a.Cout + n0) = f32x4{o[0], o[1], o[2], o[3]};
      }
    }
  }
  if (stats) {
    if (cv < ncv) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        sl[half][0][n0 + c] = s1[c];
        sl[half][1][n0 + c] = s2[c];
      }
    }
    __syncthreads();
    for (int o = tid; o < 2 * a.Cout; o += 256) {
      const int which = o / a.Cout, n = o % a.Cout;
      const float v = sl[0][which][n] + sl[1][which][n];
      if (a.rows) a.rows[((size_t)blockIdx.x * 2 + which) * a.Cout + n] = v;
      else atomicAdd(a.sums + ((size_t)(blockIdx.x & (HR_BN_COPIES - 1)) * 2 + which) * a.Cout + n, v);
    }
  }
}

// ---- transpose of the upsampling, tile form: up to three integer scales (2, 4, 8) in ONE pass over G ----
// A workgroup stages a (16 + 2*halo)^2-pixel region of G (halo = half the largest scale) for 4 channel vectors in
// LDS and forms, for every scale s, the (16/s)^2 low-resolution pixels whose footprints (2s x 2s pixels, clipped at
// the borders) lie inside it (upt_scale).
struct UpTileArgs {
  const char* g;      // [N][H][W][C]
  char* out[3];       // [N][H/s][W/s][C]
  int sc[3];          // scales
  int ns, N, H, W, C, halo;
  int tiles_x, tiles_y, chunks;
};

constexpr int UT_PXB = 4 * 16 + 16;          // LDS bytes per staged pixel: 4 channel vectors + pad

// one scale S of the tile kernel below. 256 threads = (4 channel vectors) x (2S footprint columns) x (32/S outputs):
// a thread walks the 2S footprint rows of its column (consecutive lanes read consecutive staged pixels: no bank
// conflicts), the 2S columns of an output are then added across lanes (xor shuffles inside the wave).
// Weights: those of the forward op - full-resolution index s*h - s/2 + r (r < 2s) enters low-resolution pixel h with
// (r + 0.5)/s for r < s and 1 - (r - s + 0.5)/s above (exact: s is a power of two); at the borders the clamped source
// index of the forward op sends the whole weight of the outer half to the border pixel.
template <typename T, int S>
__device__ __forceinline__ void upt_scale(const UpTileArgs& a, char* out, const char* lds, int R, int Yr, int Xr,
                                          int ty, int tx, int img, int chunk, int tid) {
  constexpr int VEC = TT<T>::VEC;
  constexpr int SIDE = HM_T / S, OUTS = SIDE * SIDE, NRX = 2 * S, OPP = 256 / (4 * NRX), PASSES = OUTS / OPP;
  constexpr float INV = 1.f / (float)S;
  const int cv = tid & 3, rx = (tid >> 2) % NRX, og = tid / (4 * NRX);
  const int hs = a.H / S, ws = a.W / S;
  const int c0 = (chunk * 4 + cv) * VEC;
  const int rpitch = R * UT_PXB + 16;
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
    const int o = pass * OPP + og;
    const int oh = ty * SIDE + o / SIDE, ow = tx * SIDE + o % SIDE;
    const int X = S * ow - S / 2 + rx;
    const bool live = oh < hs && ow < ws && X >= 0 && X < a.W;
    float wx = rx < S ? ((float)rx + 0.5f) * INV : 1.f - ((float)(rx - S) + 0.5f) * INV;
    if ((ow == 0 && rx < S) || (ow == ws - 1 && rx >= S)) wx = 1.f;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    if (live) {
      const int Y0 = S * oh - S / 2;
      const char* col = lds + (Y0 - Yr) * rpitch + (X - Xr) * UT_PXB + cv * 16;
#pragma unroll
      for (int ry = 0; ry < NRX; ++ry) {
        const int Y = Y0 + ry;
        float wy = ry < S ? ((float)ry + 0.5f) * INV : 1.f - ((float)(ry - S) + 0.5f) * INV;
        if ((oh == 0 && ry < S) || (oh == hs - 1 && ry >= S)) wy = 1.f;
        if (Y < 0 || Y >= a.H) wy = 0.f;           // (the staged region holds zeros there)
        float gv[VEC];
        v16_unpack<T>(*(const V16*)(col + ry * rpitch), gv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = fmaf(wy, gv[j], acc[j]);
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] *= wx;
    }
#pragma unroll
    for (int m = 4; m < 4 * NRX; m <<= 1)
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += __shfl_xor(acc[j], m);
    if (rx == 0 && oh < hs && ow < ws && c0 < a.C)
      *(V16*)(out + ((((size_t)img * hs + oh) * ws + ow) * a.C + c0) * sizeof(T)) = v16_pack<T>(acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample_t_tile_kernel(UpTileArgs a) {
  constexpr int VEC = TT<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  // (every XCD takes a contiguous run of tiles: the halos of neighbouring tiles overlap by half the region)
  int b = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
  if (b >= a.N * a.tiles_y * a.tiles_x * a.chunks) return;
  const int chunk = b % a.chunks; b /= a.chunks;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int img = b / a.tiles_y;
  const int R = HM_T + 2 * a.halo;             // staged region edge
  const int Yr = ty * HM_T - a.halo, Xr = tx * HM_T - a.halo;
  const int rpitch = R * UT_PXB + 16;
  // ---- stage G (zeros outside the image) ----
  // (loads in batches of 9 per thread - the whole 24x24 region at once: the compiler waits for every load of a
  // one-load-per-iteration loop)
  const int total = R * R * 4;
  for (int i0 = 0; i0 < total; i0 += 9 * 256) {
    V16 val[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int i = i0 + q * 256 + tid;
      const int v = i & 3, px = i >> 2;
      const int r = px / R, c = px - r * R;
      const int Y = Yr + r, X = Xr + c, ch = (chunk * 4 + v) * VEC;
      val[q] = (i < total && Y >= 0 && Y < a.H && X >= 0 && X < a.W && ch < a.C)
                   ? *(const V16*)(a.g + ((((size_t)img * a.H + Y) * a.W + X) * a.C + ch) * sizeof(T))
                   : v16_zero();
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int i = i0 + q * 256 + tid;
      const int px = i >> 2;
      const int r = px / R, c = px - r * R;
      if (i < total) *(V16*)(lds + r * rpitch + c * UT_PXB + (i & 3) * 16) = val[q];
    }
  }
  __syncthreads();
  for (int k = 0; k < a.ns; ++k) {
    switch (a.sc[k]) {
      case 2: upt_scale<T, 2>(a, a.out[k], lds, R, Yr, Xr, ty, tx, img, chunk, tid); break;
      case 4: upt_scale<T, 4>(a, a.out[k], lds, R, Yr, Xr, ty, tx, img, chunk, tid); break;
      default: upt_scale<T, 8>(a, a.out[k], lds, R, Yr, Xr, ty, tx, img, chunk, tid); break;
    }
  }
}

int mix_cap(int hs, int H) { return (15 * hs) / H + 3; }

}  // namespace

// slots: i = {dtype, N, H, W, C0, Cout, nup, align, h1, w1, h2, w2, h3, w3, rows mode}
//        p = {x0 [N][H][W][C0], w0 packed [Cout][C0], bias f32 [Cout] or NULL, y [N][H][W][Cout],
//             statistics (i[14] = 0: sums[8][2][Cout], float atomics; 1: rows[hrnet_head_mix_rows()][2][Cout]) or NULL,
//             t1, t2, t3 [N][h][w][Cout]}
extern "C" int hrnet_head_mix_rows(int N, int H, int W) {
  return N * ((H + HM_T - 1) / HM_T) * ((W + HM_T - 1) / HM_T);
}

extern "C" int hrnet_head_mix_supported(int dtype, int C0, int Cout) {
  if (dtype == HR_F32) return C0 % 4 == 0 && C0 >= 4 && Cout % 4 == 0 && Cout >= 4 && Cout <= 512;
  return dtype == HR_BF16 && C0 % 16 == 0 && C0 >= 16 && C0 <= 128 && Cout >= 8 && Cout % 8 == 0;
}

static int head_launch(HeadMixArgs& a, int dtype, int mode, hipStream_t s) {
  a.tiles_y = (a.H + HM_T - 1) / HM_T; a.tiles_x = (a.W + HM_T - 1) / HM_T;
  a.chunks = (a.Cout + HM_CH - 1) / HM_CH;
  a.wrow = a.K * 2 + 16;
  int off = HM_CH * a.wrow;
  for (int u = 0; u < a.nup; ++u) {
    const int ch = mix_cap(a.uh[u], a.H), cw = mix_cap(a.uw[u], a.W);
    a.cap[u] = ch > cw ? ch : cw;
    a.toff[u] = off;
    off += a.cap[u] * a.cap[u] * HM_TROW;
  }
  a.sl_off = off;
  off += 4 * 2 * HM_CH * (int)sizeof(float);
  const long long tiles = (long long)a.N * a.tiles_y * a.tiles_x;
  if (dtype == HR_F32) {
    HR_REQUIRE(tiles < (1ll << 31), "head_mix: grid");
    if (mode == 0) hipLaunchKernelGGL(head_mix_f32_kernel<0>, dim3((unsigned)tiles), dim3(256), 0, s, a);
    else if (mode == 1) hipLaunchKernelGGL(head_mix_f32_kernel<1>, dim3((unsigned)tiles), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(head_mix_f32_kernel<2>, dim3((unsigned)tiles), dim3(256), 0, s, a);
    return hr_check_launch("head_mix");
  }
  HR_REQUIRE(off <= 160 * 1024, "head_mix: %d bytes of LDS staging (scales too close to 1)", off);
  HR_REQUIRE(tiles * a.chunks < (1ll << 31), "head_mix: grid");
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)head_mix_tile_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)head_mix_tile_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)head_mix_tile_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      hr_set_error("head_mix: hipFuncSetAttribute failed");
      return HR_E_BADARG;
    }
    attr_set = true;
  }
  const dim3 grid((unsigned)((tiles * a.chunks + 7) / 8 * 8));
  if (mode == 0) hipLaunchKernelGGL(head_mix_tile_kernel<0>, grid, dim3(256), (size_t)off, s, a);
  else if (mode == 1) hipLaunchKernelGGL(head_mix_tile_kernel<1>, grid, dim3(256), (size_t)off, s, a);
  else hipLaunchKernelGGL(head_mix_tile_kernel<2>, grid, dim3(256), (size_t)off, s, a);
  return hr_check_launch("head_mix");
}

int hr_launch_head_mix(const HrOp& op, hipStream_t s) {
  const int dtype = op.i[0], N = op.i[1], H = op.i[2], W = op.i[3], C0 = op.i[4], Cout = op.i[5], nup = op.i[6];
  HR_REQUIRE(hrnet_head_mix_supported(dtype, C0, Cout),
             "head_mix: bf16 with C0 %% 16 == 0 (<= 128) or f32 with C0 %% 4 == 0 and Cout <= 512 (got dtype %d, %d, %d)",
             dtype, C0, Cout);
  HR_REQUIRE(N > 0 && H > 0 && W > 0 && nup >= 0 && nup <= 3, "head_mix: shape");
  HR_REQUIRE(op.p[0] && op.p[1] && op.p[3], "head_mix: null pointer");
  HeadMixArgs a = {};
  a.x = (const char*)op.p[0]; a.w = (const char*)op.p[1]; a.bias = (const float*)op.p[2]; a.y = (char*)op.p[3];
  if (op.i[14]) a.rows = (float*)op.p[4]; else a.sums = (float*)op.p[4];
  a.N = N; a.H = H; a.W = W; a.K = C0; a.Cout = Cout; a.nup = nup; a.align = op.i[7];
  for (int u = 0; u < nup; ++u) {
    a.up[u] = (const char*)op.p[5 + u];
    a.uh[u] = op.i[8 + 2 * u]; a.uw[u] = op.i[9 + 2 * u];
    HR_REQUIRE(a.up[u] && a.uh[u] > 0 && a.uw[u] > 0 && a.uh[u] <= H && a.uw[u] <= W,
               "head_mix: low-resolution term %d (%dx%d)", u, a.uh[u], a.uw[u]);
  }
  return head_launch(a, dtype, 0, s);
}

// Backward of the layer behind the head's BatchNorm (last_layer.3, pose_hrnet.py:341-346) fused with that BatchNorm's
// backward: dz = W3^T dHM is a K = 32 product per pixel - cheaper to form twice than to store and re-read (252 MB each
// way at batch 64). mode 1: rows[hrnet_head_mix_rows()][2][Cout] = (sum dz*mask, sum dz*mask*y) per pixel tile;
// mode 2: out = A*(dz*mask) + B*y + C, the gradient of the raw y (coef from hrnet_bn_bwd_finalize on those rows).
// slots: i = {dtype, N, H, W, K, Cout, mode, inner_relu}, p = {dY [N][H][W][K], wT packed [Cout][K] (hrnet_pack_weights
//        mode 1 of the 1x1 layer), y raw [N][H][W][Cout], out (rows f32 | G), bn scale, bn shift, coef [3][Cout]}
int hr_launch_head_bwd(const HrOp& op, hipStream_t s) {
  const int dtype = op.i[0], N = op.i[1], H = op.i[2], W = op.i[3], K = op.i[4], Cout = op.i[5], mode = op.i[6];
  HR_REQUIRE(hrnet_head_mix_supported(dtype, K, Cout), "head_bwd: dtype %d, K %d, Cout %d not served", dtype, K, Cout);
  HR_REQUIRE(N > 0 && H > 0 && W > 0 && (mode == 1 || mode == 2), "head_bwd: shape / mode");
  HR_REQUIRE(op.p[0] && op.p[1] && op.p[2] && op.p[3], "head_bwd: null pointer");
  HR_REQUIRE((op.p[4] == nullptr) == (op.p[5] == nullptr), "head_bwd: scale/shift must come together");
  HR_REQUIRE(mode == 1 || op.p[6], "head_bwd: mode 2 needs the coefficients");
  HeadMixArgs a = {};
  a.x = (const char*)op.p[0]; a.w = (const char*)op.p[1]; a.yin = (const char*)op.p[2];
  if (mode == 1) a.rows = (float*)op.p[3]; else a.y = (char*)op.p[3];
  a.bn_scale = (const float*)op.p[4]; a.bn_shift = (const float*)op.p[5]; a.coef = (const float*)op.p[6];
  a.inner_relu = op.i[7];
  a.N = N; a.H = H; a.W = W; a.K = K; a.Cout = Cout;
  return head_launch(a, dtype, mode, s);
}

// 1 if the tile form serves these output sizes (align_corners=False, one integer scale of 2, 4 or 8 per output on both
// axes); the caller (eltwise.hip: hr_launch_upsample_t) otherwise runs the streamed form, one output at a time
int hr_upsample_t_tile(int dtype, const void* g, void* const* outs, const int* hs, const int* ws, int nout, int N, int H,
                       int W, int C, int align, hipStream_t s) {
  if (align || nout < 1 || nout > 3) return 1;
  UpTileArgs a = {};
  int halo = 0;
  for (int k = 0; k < nout; ++k) {
    if (hs[k] <= 0 || ws[k] <= 0 || H % hs[k] || W % ws[k] || H / hs[k] != W / ws[k]) return 1;
    const int sc = H / hs[k];
    if (sc != 2 && sc != 4 && sc != 8) return 1;
    a.sc[k] = sc;
    a.out[k] = (char*)outs[k];
    if (sc / 2 > halo) halo = sc / 2;
  }
  const int vec = dtype == HR_F32 ? 4 : 8;
  a.g = (const char*)g; a.ns = nout; a.N = N; a.H = H; a.W = W; a.C = C; a.halo = halo;
  a.tiles_y = (H + HM_T - 1) / HM_T; a.tiles_x = (W + HM_T - 1) / HM_T;
  a.chunks = (C / vec + 3) / 4;
  const int R = HM_T + 2 * halo;
  const size_t lds = (size_t)R * (R * UT_PXB + 16);
  const long long blocks = ((long long)N * a.tiles_y * a.tiles_x * a.chunks + 7) / 8 * 8;
  if (blocks >= (1ll << 31)) return 1;
  if (dtype == HR_F32)
    hipLaunchKernelGGL(upsample_t_tile_kernel<float>, dim3((unsigned)blocks), dim3(256), lds, s, a);
  else
    hipLaunchKernelGGL(upsample_t_tile_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), lds, s, a);
  return 0;
}
