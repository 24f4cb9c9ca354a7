// Fused backward of a 1x1 (pointwise) convolution whose output feeds a BatchNorm, for gfx950: the three
// pointwise layers of a Bottleneck (pose_hrnet.py:60-105: conv1 Cin->64, conv3 64->256, downsample 64->256) on
// the 64x64 maps of layer1, where every 256-channel tensor of a 64-image batch is 134 MB and the unfused
// sequence (grad_term, wgrad, conv, grad_term) is eleven passes over such tensors.
//
// Same contract as hrnet_conv3x3_bwd_fused (bwd_fused.hip), with pixels as a flat index (a 1x1 conv has no halo):
//
//   g[p,co]  = A[co]*dz[p,co] + B[co]*y[p,co] + C[co]      BatchNorm backward applied while staging
//   dW[co,ci] += sum_p g[p,co] * a[p,ci]                    weight gradient (pixels = K, transposing LDS reads)
//   v[p,ci]  = sum_co Wt[ci,co] * g[p,co] (+ addend[p,ci])  input gradient (+ the residual stream)
//   dx[p,ci] = v * [a[p,ci] > 0]                            masked by the ReLU in front of the conv
//   rows     = (sum dx, sum dx*yb)                          statistics of the next BatchNorm backward
//
// The kernel is an HBM stream (17 GFLOP against 330-470 MB per launch at batch 64): 64-pixel tiles, 4 waves,
// ~78 KB of LDS (g tile, a tile, the whole weight matrix resident) so that TWO workgroups share a CU - one
// loads its tile while the other is in its matrix phases; every global access of the staging pass is a full
// 16-byte lane vector of a contiguous pixel row. One f32 slab [Cout][Cin] per workgroup, summed by
// hrnet_wgrad_reduce(_table). bf16 only (the fp32 device path keeps the unfused kernels).
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace {

struct PwArgs {
  const char* dz;        // [P,Cout] upstream gradient w.r.t. the BatchNorm output, ReLU mask applied
  const char* y;         // [P,Cout] raw conv output
  const float* coef;     // [3][Cout] A,B,C of hrnet_bn_bwd_finalize, or NULL: g = dz
  HrBnBwdRef ref;        // ref.rows != NULL: the coefficients are built here from the partial rows (coef unused)
  const char* x;         // [P,Cin] conv input as stored
  const float* in_scale; // optional per-Cin affine (+ReLU) the forward applied on load
  const float* in_shift;
  const char* wT;        // packed [Cin][Cout] (hrnet_pack_weights mode 1 of a 1x1 kernel)
  char* dx;              // [P,Cin] out
  const char* addend;    // optional [P,Cin] added before the mask (may alias dx)
  const char* bs_y;      // optional [P,Cin]: rows get sum(dx*bs_y)
  float* rows;           // optional [nsplit][2][Cin]
  float* slabs;          // [nsplit][Cout][Cin] f32
  long long P;
  int total_tiles, nsplit;
  int atomic;            // `slabs` IS the [Cout][Cin] f32 gradient: the workgroups ADD their tiles into it (float atomics)
  int in_relu, mask_out;
};

// transposing LDS read (ds_read_b64_tr_b16): 16 lanes x 4 pixels -> each lane gets 8 consecutive pixels (K) of
// one channel (row of the MFMA operand); see bwd_fused.hip
__device__ __forceinline__ V16 trl16(const char* base, int r0, int rstep, int choff, int lane) {
  const int q = (lane & 15) >> 2, p4 = lane & 3;
  const int addr = r0 + q * rstep + choff + p4 * 8;
  const LDS_AS char* l = (const LDS_AS char*)base;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(l + addr));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(l + addr + 4 * rstep));
  const bf16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return __builtin_bit_cast(V16, v);
}

// (the 256-channel input side keeps 64 + 64 accumulator registers, the residual and statistics operands: one
// wave per SIMD with the full register file; it is launched one workgroup per CU)
template <int CO, int CI>
__global__ __launch_bounds__(256, (CI > 64 ? 1 : 2)) void bwd_pw_kernel(PwArgs a) {
  typedef bf16_t T;
  constexpr int VEC = 8, ES = 2, NT = 256, TP = 64;
  constexpr int GPIX = CO * ES + 16, APIX = CI * ES + 16, WROW = CO * ES + 16;
  constexpr int GBYTES = TP * GPIX, ABYTES = TP * APIX, WBYTES = CI * WROW;
  constexpr int VPG = CO / VEC, VPA = CI / VEC;
  constexpr int XG = TP * VPG / NT, XA = TP * VPA / NT;         // 16-byte vectors per thread and tile
  constexpr int PSG = NT / VPG, PSA = NT / VPA;                 // pixel step between a thread's vectors
  constexpr int FCI = CI / 16, FCO = CO / 16, NG = CI / 32;     // fragments; 32-channel groups of the input side
  constexpr int COSPLIT = CO == 256 ? 4 : 1, CISPLIT = 4 / COSPLIT;
  constexpr int FCOW = FCO / COSPLIT, FCIW = FCI / CISPLIT;     // weight-gradient fragments per wave
  constexpr bool ROWS = CI <= 64;                               // statistics in registers: 2 x CI/4 per lane
  constexpr bool ROWS2 = !ROWS;                                 // 256 channels: through the LDS tile, per staging thread
  constexpr bool AFF = CI <= 64;                                // input affine (a 256-channel input is a stored activation)
  static_assert(TP * VPG % NT == 0 && TP * VPA % NT == 0 && FCO % COSPLIT == 0 && FCI % CISPLIT == 0, "geometry");
  static_assert(GBYTES + ABYTES + WBYTES <= 80 * 1024, "two workgroups per CU");
  __shared__ __attribute__((aligned(16))) char lds[GBYTES + ABYTES + WBYTES];
  char* gl = lds;
  char* al = lds + GBYTES;
  char* wl = lds + GBYTES + ABYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int split = blockIdx.x;

  // ---- the weight matrix, resident for the whole walk; LDS row q holds the channel the MFMA row order needs so
  // that a lane ends up with 8 contiguous channels of one pixel per 32-channel group ----
  for (int idx = tid; idx < CI * VPG; idx += NT) {
    const int q = idx / VPG, v = idx % VPG;
    const int ci = (q & ~31) + ((q & 15) >> 2) * 8 + ((q >> 4) & 1) * 4 + (q & 3);
    *(V16*)(wl + q * WROW + v * 16) = *(const V16*)(a.wT + ((size_t)ci * CO + v * VEC) * ES);
  }
  // a thread keeps one channel vector of each side for every tile: its coefficients live in registers
  const int vg = tid % VPG, va = tid % VPA;
  const bool from_rows = a.ref.rows != nullptr;
  const bool has_coef = a.coef != nullptr || from_rows, has_aff = AFF && a.in_scale != nullptr, in_relu = a.in_relu != 0;
  float cA[VEC], cB[VEC], cC[VEC], sc[AFF ? VEC : 1], sh[AFF ? VEC : 1];
  const float* ctab = a.coef;
  if (from_rows) {
    // scratch and table live in the (still unused) tile region; the helper ends with a barrier
    static_assert(NT * 2 * 8 + 3 * CO * 4 <= GBYTES + ABYTES, "row-sum scratch fits the tiles");
    float* tab = (float*)(lds + NT * 2 * 8);
    hr_bnbwd_coef_from_rows<NT, CO>(a.ref, CO, (double*)lds, tab, blockIdx.x == 0);
    ctab = tab;
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    cA[j] = has_coef ? ctab[vg * VEC + j] : 1.f;
    cB[j] = has_coef ? ctab[CO + vg * VEC + j] : 0.f;
    cC[j] = has_coef ? ctab[2 * CO + vg * VEC + j] : 0.f;
    if constexpr (AFF) {
      sc[j] = has_aff ? a.in_scale[va * VEC + j] : 1.f;
      sh[j] = has_aff ? a.in_shift[va * VEC + j] : 0.f;
    }
  }

  if (from_rows) __syncthreads();   // every thread has its coefficients: the table's LDS becomes tile space
  // weight gradient: this wave's co fragments x ci fragments; D: col (li) = ci, rows (lg*4+r) = co
  const int wco = wave % COSPLIT, wci = wave / COSPLIT;
  f32x4 accw[FCOW][FCIW];
#pragma unroll
  for (int f = 0; f < FCOW; ++f)
#pragma unroll
    for (int c = 0; c < FCIW; ++c) accw[f][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float s1[ROWS ? NG * 8 : 1], s2[ROWS ? NG * 8 : 1];
#pragma unroll
  for (int k = 0; k < (ROWS ? NG * 8 : 1); ++k) s1[k] = s2[k] = 0.f;
  // wide input side: the MFMA layout gives a lane 8 channels of EVERY 32-channel group (128 partial sums), so the
  // masked result goes back into the input tile's LDS (each lane overwrites exactly the mask values it read) and
  // the statistics are taken in the staging layout, where a thread keeps ONE channel vector: 16 sums per thread
  float t1[ROWS2 ? VEC : 1], t2[ROWS2 ? VEC : 1];
#pragma unroll
  for (int k = 0; k < (ROWS2 ? VEC : 1); ++k) t1[k] = t2[k] = 0.f;
  const bool rows2 = ROWS2 && a.rows != nullptr;

  // input gradient D[ci][pixel]: a wave owns 16 pixels x all CI channels
  const int boff = (wave * 16 + li) * GPIX + lg * 16;
  const int aoff = li * WROW + lg * 16;

  for (int t = split; t < a.total_tiles; t += a.nsplit) {
    const long long p0 = (long long)t * TP;
    // ---- stage: every load of the tile is issued before the first use ----
    V16 rz[XG], ry[XG], rx[XA];
#pragma unroll
    for (int k = 0; k < XG; ++k) {
      const long long p = p0 + tid / VPG + k * PSG;
      const bool ok = p < a.P;
      const size_t o = ok ? ((size_t)p * CO + vg * VEC) * ES : 0;
      rz[k] = *(const V16*)(a.dz + o);
      if (has_coef) ry[k] = *(const V16*)(a.y + o);
    }
#pragma unroll
    for (int k = 0; k < XA; ++k) {
      const long long p = p0 + tid / VPA + k * PSA;
      const size_t o = p < a.P ? ((size_t)p * CI + va * VEC) * ES : 0;
      rx[k] = *(const V16*)(a.x + o);
    }
    // epilogue operands of this wave's 16 pixels (MFMA layout: 8 channels per 32-channel group)
    const long long pe = p0 + wave * 16 + li;
    const bool pok = pe < a.P;
    const size_t eo = pok ? ((size_t)pe * CI + lg * 8) * ES : 0;
    V16 pa[NG], pb[ROWS ? NG : 1];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      pa[j] = v16_zero();
      if (a.addend) pa[j] = *(const V16*)(a.addend + eo + j * 32 * ES);
      if constexpr (ROWS) {
        pb[j] = v16_zero();
        if (a.bs_y) pb[j] = *(const V16*)(a.bs_y + eo + j * 32 * ES);
      }
    }
#pragma unroll
    for (int k = 0; k < XG; ++k) {
      const int pl = tid / VPG + k * PSG;
      const bool ok = p0 + pl < a.P;
      V16 g = rz[k];
      if (has_coef) {
        float fz[VEC], fy[VEC];
        v16_unpack<T>(rz[k], fz);
        v16_unpack<T>(ry[k], fy);
#pragma unroll
        for (int j = 0; j < VEC; ++j) fz[j] = fmaf(cA[j], fz[j], fmaf(cB[j], fy[j], cC[j]));
        g = v16_pack<T>(fz);
      }
      *(V16*)(gl + pl * GPIX + vg * 16) = ok ? g : v16_zero();     // beyond the last pixel the gradient is zero
    }
#pragma unroll
    for (int k = 0; k < XA; ++k) {
      const int pl = tid / VPA + k * PSA;
      const bool ok = p0 + pl < a.P;
      V16 v = rx[k];
      if (has_aff || in_relu) {
        float f[VEC];
        v16_unpack<T>(rx[k], f);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          if constexpr (AFF) f[j] = fmaf(f[j], sc[j], sh[j]);
          if (in_relu) f[j] = f[j] > 0.f ? f[j] : 0.f;
        }
        v = v16_pack<T>(f);
      }
      *(V16*)(al + pl * APIX + va * 16) = ok ? v : v16_zero();
    }
    __syncthreads();

    // ---- input gradient ----
    f32x4 accd[FCI];
#pragma unroll
    for (int f = 0; f < FCI; ++f) accd[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < CO / 32; ++kk) {
      const V16 bf = *(const V16*)(gl + boff + kk * 64);
#pragma unroll
      for (int f = 0; f < FCI; ++f) {
        const V16 af = *(const V16*)(wl + aoff + f * 16 * WROW + kk * 64);
        accd[f] = mma16<T>(af, bf, accd[f]);
      }
    }

    // ---- weight gradient: both operands come transposed out of the two tiles ----
#pragma unroll
    for (int ks = 0; ks < TP / 32; ++ks) {
      const int pr = ks * 32 + lg * VEC;
      V16 af[FCOW], bfr[FCIW];
#pragma unroll
      for (int f = 0; f < FCOW; ++f) af[f] = trl16(gl, pr * GPIX, GPIX, (wco * FCOW + f) * 16 * ES, lane);
#pragma unroll
      for (int c = 0; c < FCIW; ++c) bfr[c] = trl16(al, pr * APIX, APIX, (wci * FCIW + c) * 16 * ES, lane);
#pragma unroll
      for (int f = 0; f < FCOW; ++f)
#pragma unroll
        for (int c = 0; c < FCIW; ++c) accw[f][c] = mma16<T>(af[f], bfr[c], accw[f][c]);
    }

    // ---- tile epilogue: residual addend, ReLU mask from the staged input tile, store, statistics ----
    V16 qb[ROWS2 ? XA : 1];
    if constexpr (ROWS2) {
      if (rows2 && a.bs_y) {
#pragma unroll
        for (int k = 0; k < XA; ++k) {
          const long long p = p0 + tid / VPA + k * PSA;
          qb[k] = *(const V16*)(a.bs_y + (p < a.P ? ((size_t)p * CI + va * VEC) * ES : 0));
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      float v[8];
      v[0] = accd[2 * j].x; v[1] = accd[2 * j].y; v[2] = accd[2 * j].z; v[3] = accd[2 * j].w;
      v[4] = accd[2 * j + 1].x; v[5] = accd[2 * j + 1].y; v[6] = accd[2 * j + 1].z; v[7] = accd[2 * j + 1].w;
      if (pok) {
        if (a.addend) {
          float ad[8];
          v16_unpack<T>(pa[j], ad);
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] += ad[k];
        }
        if (a.mask_out) {
          float am[8];
          v16_unpack<T>(*(const V16*)(al + (wave * 16 + li) * APIX + (j * 32 + lg * 8) * ES), am);
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = am[k] > 0.f ? v[k] : 0.f;
        }
        const V16 packed = v16_pack<T>(v);
        *(V16*)(a.dx + eo + j * 32 * ES) = packed;
        if constexpr (ROWS2) {
          if (rows2) *(V16*)(al + (wave * 16 + li) * APIX + (j * 32 + lg * 8) * ES) = packed;
        }
        if constexpr (ROWS) {
          if (a.rows) {
            float yb[8];
            v16_unpack<T>(pb[j], yb);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              s1[j * 8 + k] += v[k];
              hr_fma_acc(s2[j * 8 + k], v[k], yb[k]);
            }
          }
        }
      } else {
        if constexpr (ROWS2) {
          if (rows2) *(V16*)(al + (wave * 16 + li) * APIX + (j * 32 + lg * 8) * ES) = v16_zero();
        }
      }
    }
    if constexpr (ROWS2) {
      if (rows2) {
        __syncthreads();   // the input tile now holds the stored gradient
#pragma unroll
        for (int k = 0; k < XA; ++k) {
          float dv[VEC], yb[VEC];
          v16_unpack<T>(*(const V16*)(al + (tid / VPA + k * PSA) * APIX + va * 16), dv);
          if (a.bs_y) {
            v16_unpack<T>(qb[k], yb);
#pragma unroll
            for (int j = 0; j < VEC; ++j) hr_fma_acc(t2[j], dv[j], yb[j]);
          }
#pragma unroll
          for (int j = 0; j < VEC; ++j) t1[j] += dv[j];
        }
      }
    }
    __syncthreads();   // the tiles are free again
  }

  // ---- backward statistics: lanes -> waves -> one row per workgroup (deterministic) ----
  if constexpr (ROWS) {
    if (a.rows) {
      float* sl = (float*)lds;   // [4 waves][2][CI]
#pragma unroll
      for (int k = 0; k < NG * 8; ++k) {
        s1[k] = wave_sum16(s1[k]);
        s2[k] = wave_sum16(s2[k]);
      }
      if (li == 0) {
#pragma unroll
        for (int j = 0; j < NG; ++j)
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            sl[(wave * 2 + 0) * CI + j * 32 + lg * 8 + k] = s1[j * 8 + k];
            sl[(wave * 2 + 1) * CI + j * 32 + lg * 8 + k] = s2[j * 8 + k];
          }
      }
      __syncthreads();
      if (tid < 2 * CI) {
        const int which = tid / CI, cl = tid % CI;
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) s += sl[(q * 2 + which) * CI + cl];
        a.rows[((size_t)split * 2 + which) * CI + cl] = s;
      }
    }
  }

  if constexpr (ROWS2) {
    if (rows2) {
      float* sl = (float*)lds;   // [NT / VPA thread groups][2][CI]
      constexpr int NGRP = NT / VPA;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        sl[((tid / VPA) * 2 + 0) * CI + va * VEC + j] = t1[j];
        sl[((tid / VPA) * 2 + 1) * CI + va * VEC + j] = t2[j];
      }
      __syncthreads();
      for (int o = tid; o < 2 * CI; o += NT) {
        const int which = o / CI, cl = o % CI;
        float sacc = 0.f;
#pragma unroll
        for (int q = 0; q < NGRP; ++q) sacc += sl[(q * 2 + which) * CI + cl];
        a.rows[((size_t)split * 2 + which) * CI + cl] = sacc;
      }
    }
  }

  // ---- weight-gradient slab of this workgroup: slab[split][co][ci] ----
  float* slab = a.slabs + (a.atomic ? (size_t)0 : (size_t)split * CO * CI);   // (no channel padding at these widths)
#pragma unroll
  for (int f = 0; f < FCOW; ++f)
#pragma unroll
    for (int c = 0; c < FCIW; ++c) {
      const int co = (wco * FCOW + f) * 16 + lg * 4, ci = (wci * FCIW + c) * 16 + li;
      const float v4[4] = {accw[f][c].x, accw[f][c].y, accw[f][c].z, accw[f][c].w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* dst = slab + (size_t)(co + r) * CI + ci;
        if (a.atomic) atomicAdd(dst, v4[r]); else *dst = v4[r];
      }
    }
}

inline int pw_shape(int Cin, int Cout) {
  if (Cout == 256 && Cin == 64) return 1;
  if (Cout == 64 && Cin == 256) return 2;
  if (Cout == 64 && Cin == 64) return 3;
  return 0;
}

}  // namespace

// 1 if hrnet_conv1x1_bwd_fused serves this layer (bf16; the Bottleneck shapes 64->256, 256->64, 64->64)
extern "C" int hrnet_bwd_pw_supported(int dtype, int Cin, int Cout) {
  return dtype == HR_BF16 && pw_shape(Cin, Cout) != 0 ? 1 : 0;
}

// 1 if the launch can also gather the next BatchNorm's backward sums (`rows`): every served shape (the wide
// input side takes them through LDS)
extern "C" int hrnet_bwd_pw_rows_supported(int dtype, int Cin, int Cout) {
  return hrnet_bwd_pw_supported(dtype, Cin, Cout);
}

// number of slabs / statistics rows = workgroups, every one walking the same number of 64-pixel tiles when
// possible. Two workgroups fit a CU, but inside the training step one per CU is as fast (19.96 vs 19.97 ms/step:
// the launch is HBM-bound either way) and writes half the slabs
extern "C" int hrnet_bwd_pw_splits(int dtype, long long pixels, int Cin, int Cout) {
  if (!hrnet_bwd_pw_supported(dtype, Cin, Cout) || pixels <= 0) return 0;
  const long long tiles = (pixels + 63) / 64;
  static const int wgs = hr_knob("HRNET_PW_WGS", 256);
  long long ns = wgs < tiles ? wgs : tiles;
  long long even = ns;
  while (even > 1 && tiles % even != 0) --even;
  if (even * 2 > ns) ns = even;
  return (int)ns;
}

extern "C" int hrnet_bwd_pw_kernel_name(int dtype, int Cin, int Cout, char* buf, int buflen) {
  (void)dtype;
  return snprintf(buf, buflen, "bwd_pw_kernel<%d, %d>", Cout, Cin);
}

extern "C" int hrnet_conv1x1_bwd_fused(int dtype, const void* dz, const void* y, const float* coef, const void* x,
                                       const float* in_scale, const float* in_shift, int in_relu, const void* wT,
                                       void* dx, const void* addend, int mask_out, float* rows, const void* bs_y,
                                       float* slabs, long long pixels, int Cin, int Cout, hr_stream_t stream) {
  return hrnet_conv1x1_bwd_fused_bnref(dtype, dz, y, coef, nullptr, x, in_scale, in_shift, in_relu, wT, dx, addend,
                                       mask_out, rows, bs_y, slabs, pixels, Cin, Cout, stream);
}

static int bwd_pw_launch(int dtype, const void* dz, const void* y, const float* coef, const HrBnBwdRef* ref,
                         const void* x, const float* in_scale, const float* in_shift, int in_relu, const void* wT,
                         void* dx, const void* addend, int mask_out, float* rows, const void* bs_y, float* slabs,
                         long long pixels, int Cin, int Cout, int atomic, hr_stream_t stream);

extern "C" int hrnet_conv1x1_bwd_fused_bnref(int dtype, const void* dz, const void* y, const float* coef,
                                             const HrBnBwdRef* ref, const void* x, const float* in_scale,
                                             const float* in_shift, int in_relu, const void* wT, void* dx,
                                             const void* addend, int mask_out, float* rows, const void* bs_y,
                                             float* slabs, long long pixels, int Cin, int Cout, hr_stream_t stream) {
  return bwd_pw_launch(dtype, dz, y, coef, ref, x, in_scale, in_shift, in_relu, wT, dx, addend, mask_out, rows, bs_y,
                       slabs, pixels, Cin, Cout, 0, stream);
}

static int bwd_pw_launch(int dtype, const void* dz, const void* y, const float* coef, const HrBnBwdRef* ref,
                         const void* x, const float* in_scale, const float* in_shift, int in_relu, const void* wT,
                         void* dx, const void* addend, int mask_out, float* rows, const void* bs_y, float* slabs,
                         long long pixels, int Cin, int Cout, int atomic, hr_stream_t stream) {
  HR_REQUIRE(hrnet_bwd_pw_supported(dtype, Cin, Cout), "bwd_pw: dtype %d Cin %d Cout %d not served", dtype, Cin, Cout);
  HR_REQUIRE(dz && x && wT && dx && slabs, "bwd_pw: null pointer");
  HR_REQUIRE((!coef && !ref) || y, "bwd_pw: coef needs y");
  HR_REQUIRE(!ref || (ref->rows && ref->gamma && ref->save_mean && ref->save_invstd && ref->dgamma && ref->dbeta &&
                      ref->nrows >= 1 && (long long)ref->nrows * Cout <= 8192 && ref->count > 0.f),
             "bwd_pw: incomplete HrBnBwdRef (or nrows * Cout > 8192)");
  HR_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "bwd_pw: scale/shift must come together");
  HR_REQUIRE(!bs_y || rows, "bwd_pw: bs_y needs rows");
  HR_REQUIRE(!in_scale || Cin <= 64, "bwd_pw: no input affine for Cin %d", Cin);
  HR_REQUIRE(!rows || hrnet_bwd_pw_rows_supported(dtype, Cin, Cout), "bwd_pw: no statistics rows for Cin %d", Cin);
  HR_REQUIRE(pixels > 0 && pixels < (1ll << 31) * 64, "bwd_pw: pixel count");
  PwArgs a;
  a.dz = (const char*)dz; a.y = (const char*)y; a.coef = ref ? nullptr : coef; a.x = (const char*)x;
  if (ref) a.ref = *ref; else a.ref = HrBnBwdRef{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0, 0, 0};
  a.in_scale = in_scale; a.in_shift = in_shift; a.wT = (const char*)wT; a.dx = (char*)dx;
  a.addend = (const char*)addend; a.bs_y = (const char*)bs_y; a.rows = rows; a.slabs = slabs;
  a.P = pixels;
  a.total_tiles = (int)((pixels + 63) / 64);
  a.nsplit = hrnet_bwd_pw_splits(dtype, pixels, Cin, Cout);
  a.in_relu = in_relu; a.mask_out = mask_out;
  HR_REQUIRE(atomic >= 0, "bwd_pw: the atomic form serves unpadded layers (real channel counts = tensor channel counts)");
  a.atomic = atomic;
  hipStream_t s = (hipStream_t)stream;
  switch (pw_shape(Cin, Cout)) {
    case 1: hipLaunchKernelGGL((bwd_pw_kernel<256, 64>), dim3(a.nsplit), dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL((bwd_pw_kernel<64, 256>), dim3(a.nsplit), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((bwd_pw_kernel<64, 64>), dim3(a.nsplit), dim3(256), 0, s, a); break;
  }
  return hr_check_launch("conv1x1_bwd_fused");
}

// op slots as OP_BWD_FUSED (p[0..11] = dz,y,coef,x,scale,shift,wT,dx,addend,rows,bs_y,slabs;
// i[0..7] = dtype,N,H,W,Cin,Cout,in_relu,mask_out)
int hr_launch_bwd_pw(const HrOp& op, hipStream_t s) {
  // p[12]: HOST pointer to a HrBnBwdRef (kept alive by the plan), or NULL; i[8] = 1: p[11] is the [Cout][Cin] gradient
  // the weight-gradient tiles are ADDED to (float atomics; i[9], i[10] = its real Cout, Cin: must equal the tensors')
  return bwd_pw_launch(op.i[0], op.p[0], op.p[1], (const float*)op.p[2], (const HrBnBwdRef*)op.p[12], op.p[3],
                       (const float*)op.p[4], (const float*)op.p[5], op.i[6], op.p[6], op.p[7], op.p[8], op.i[7],
                       (float*)op.p[9], op.p[10], (float*)op.p[11], (long long)op.i[1] * op.i[2] * op.i[3], op.i[4],
                       op.i[5], (op.i[8] && op.i[9] == op.i[5] && op.i[10] == op.i[4]) ? 1 : (op.i[8] ? -1 : 0), (hr_stream_t)s);
}
