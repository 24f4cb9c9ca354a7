// Weight gradient on MFMA for gfx950.
//
//   dW[co][tap][ci] = sum_{n, oy, ox} dY[n,oy,ox,co] * Xa[n, oy*S - pad + r, ox*S - pad + s, ci]
//
// The contraction index is the PIXEL, which is the slow axis of both NHWC operands. bf16 uses
// ds_read_b64_tr_b16 (gfx950's transposing LDS read: 4 pixels x 16 channels delivered
// channel-per-lane) so the same NHWC halo image the forward conv uses feeds the MFMA without a
// transposed copy; f32 uses the one-value-per-lane 16x16x4 operand form (4 ds_read_b32).
//
// A workgroup owns BCO output channels x (all taps) x KC input channels and loops over its
// share of the pixel tiles, accumulating in registers; each of the 4 waves owns a fixed subset
// of the (tap, ci-fragment) x co-fragment outputs, so there is no cross-wave reduction. It then
// writes one f32 slab; hrnet_wgrad_reduce sums the slabs in a fixed order (bitwise
// reproducible, no float atomics).
#include <stdio.h>

#include <type_traits>
#include <stdlib.h>

#include "common.h"

namespace {

struct WgradArgs {
  const char* x;
  const char* dy;
  const float* in_scale;
  const float* in_shift;
  float* slabs;
  int N, H, W, Cin, Ho, Wo, Cout;
  unsigned x_bytes, dy_bytes;   // bytes of the two operand tensors (buffer descriptors: 32-bit offsets)
  int tiles_y, tiles_x, total_tiles;
  int in_relu;
  // atomic = 1: `slabs` IS the OIHW f32 gradient [Cout_real][Cin_real][ks][ks]; every workgroup ADDS its tile into it
  // with float atomics (no slabs, no reduce launch). 3x3: the tile goes through LDS so that a wave instruction adds
  // 64 consecutive floats of one output channel's [ci][tap] run
  int atomic, Cout_real, Cin_real;
  int ld;   // floats between consecutive output-channel rows of the gradient (1x1 launches into a column slice)
};

// lane's KSTEP-deep operand fragment for 16 channels starting at byte offset `choff` of each
// pixel row; r0 = LDS byte address of the lane group's first pixel, rstep = byte distance between
// consecutive pixels of the group.
template <typename T>
__device__ __forceinline__ V16 tr_load(const char* base, int r0, int rstep, int choff, int lane);

template <>
__device__ __forceinline__ V16 tr_load<bf16_t>(const char* base, int r0, int rstep, int choff,
                                               int lane) {
  // ds_read_b64_tr_b16: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3; lane i
  // receives column i of the 4 rows. Two reads cover the group's 8 pixels.
  const int q = (lane & 15) >> 2, p4 = lane & 3;
  const int addr = r0 + q * rstep + choff + p4 * 8;
  const LDS_AS char* l = (const LDS_AS char*)base;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(l + addr));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(l + addr + 4 * rstep));
  const bf16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return __builtin_bit_cast(V16, v);
}

template <>
__device__ __forceinline__ V16 tr_load<float>(const char* base, int r0, int rstep, int choff,
                                              int lane) {
  const char* p = base + r0 + choff + (lane & 15) * 4;
  return V16{*(const uint32_t*)p, *(const uint32_t*)(p + rstep), *(const uint32_t*)(p + 2 * rstep),
             *(const uint32_t*)(p + 3 * rstep)};
}

// XOR swizzle of the 16-byte chunk index of a pixel-major LDS image with unpadded RB-byte rows, keyed on the pixel's
// COLUMN c in its image row (bwd_fused.hip: lds_swz): ds_read_b64_tr_b16 of eight consecutive pixels of a row - the
// 32-lane halves of a transposing read below - covers the 64 banks once. f32 (one-float transposing reads): rows padded
// by 16 bytes, no swizzle.
template <int RB, bool SWZ>
__device__ __forceinline__ int wg_swz(int c) {
  static_assert(!SWZ || RB == 64 || RB == 128 || RB == 256, "row bytes");
  if constexpr (!SWZ) return 0;
  return RB == 64 ? ((c >> 1) & 2) : RB == 128 ? (c & 6) : ((c & 7) << 1);
}

template <typename T, int KS, int STRIDE, int TH, int TW, int BCO, int KC, int WCO, int WN>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
  constexpr int VEC = TT<T>::VEC;
  constexpr int KSTEP = TT<T>::KSTEP;
  constexpr int ES = (int)sizeof(T);
  constexpr int TAPS = KS * KS;
  constexpr int HALO_H = (TH - 1) * STRIDE + KS;
  constexpr int HALO_W = (TW - 1) * STRIDE + KS;
  constexpr bool SWZ = ES == 2;
  constexpr int PIXB = KC * ES + (SWZ ? 0 : 16);      // row (pixel) bytes of the x image
  constexpr int DYB = BCO * ES + (SWZ ? 0 : 16);      // ... of the dY image
  constexpr int XBYTES = HALO_H * HALO_W * PIXB;
  constexpr int BM = TH * TW;
  constexpr int FCO = BCO / 16, FCI = KC / 16;
  constexpr int FCOW = FCO / WCO;            // co fragments per wave
  constexpr int NFR = TAPS * FCI;            // (tap, ci-fragment) outputs
  constexpr int NPW = (NFR + WN - 1) / WN;   // of which per wave
  constexpr int DVPP = BCO / VEC;            // 16-byte vectors per dY pixel
  static_assert(WCO * WN == 4 && FCO % WCO == 0, "wave grid");
  static_assert(BM % KSTEP == 0 && TW % VEC == 0, "pixel groups stay inside a tile row");
  static_assert(TW == 16 || TW == 8, "a k-step is two rows of 16 or four rows of 8 pixels");
  constexpr int LDSTOT = (XBYTES + BM * DYB) > 16 * KC * TAPS * 4 ? (XBYTES + BM * DYB) : 16 * KC * TAPS * 4;
  __shared__ __attribute__((aligned(16))) char lds[LDSTOT];
  char* xl = lds;
  char* dl = lds + XBYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave % WCO, wn = wave / WCO;
  const int li = lane & 15, lg = lane >> 4;
  const int co0 = blockIdx.y * BCO, c0 = blockIdx.z * KC;
  constexpr int PAD = KS / 2;

  f32x4 acc[NPW][FCOW];
#pragma unroll
  for (int j = 0; j < NPW; ++j)
#pragma unroll
    for (int f = 0; f < FCOW; ++f) acc[j][f] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging registers: the next tile's loads are issued before the MFMAs of the current one and written to LDS
  // after them (issue-early / write-late). Buffer loads: a vector outside the tensor (or a walk with no next tile)
  // gets an offset beyond the descriptor's range - zeros come back, nothing is fetched, and no branch cuts the loop ----
  constexpr int VPP = KC / VEC;                          // 16-byte vectors per halo pixel
  constexpr int XVECS = HALO_H * HALO_W * VPP, DVECS = BM * DVPP;
  constexpr int XV = (XVECS + 255) / 256, DV = (DVECS + 255) / 256;
  static_assert(256 % VPP == 0 && 256 % DVPP == 0 && XV <= 31, "staging layout");
  constexpr unsigned OOB = 0x80000000u;
  const auto rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
  const auto rdy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)a.dy_bytes, 0x00020000);
  V16 xr[XV], dr[DV];
  unsigned xok = 0;
  const int xv = tid % VPP, dv = tid % DVPP;
  const int xc = c0 + xv * VEC;
  const bool xcvalid = xc < a.Cin, dcvalid = co0 + dv * VEC < a.Cout;
  const bool has_affine = a.in_scale != nullptr;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sc[j] = (has_affine && xcvalid) ? a.in_scale[xc + j] : 1.f;
    sh[j] = (has_affine && xcvalid) ? a.in_shift[xc + j] : 0.f;
  }
  // per-thread staging geometry (the same for every tile): halo coordinates, global byte offset from the tile's first
  // halo pixel, LDS byte offset in the (swizzled) image
  int xhy[XV], xgo[XV], xlo[XV], dpy[DV], dgo[DV], dlo[DV];
  const int pixX = a.Cin * ES, rowX = a.W * pixX, pixD = a.Cout * ES, rowD = a.Wo * pixD;
#pragma unroll
  for (int k = 0; k < XV; ++k) {
    const int idx = tid + k * 256;
    const int pix = idx / VPP, hy = pix / HALO_W, hx = pix - hy * HALO_W;
    xhy[k] = (idx < XVECS && xcvalid) ? (hy << 16) | hx : (0x4000 << 16);
    xgo[k] = hy * rowX + hx * pixX + xc * ES;
    xlo[k] = pix * PIXB + ((xv ^ wg_swz<PIXB, SWZ>(hx)) << 4);
  }
#pragma unroll
  for (int k = 0; k < DV; ++k) {
    const int idx = tid + k * 256;
    const int p = idx / DVPP, py = p / TW, px = p - py * TW;
    dpy[k] = (idx < DVECS && dcvalid) ? (py << 16) | px : (0x4000 << 16);
    dgo[k] = py * rowD + px * pixD + (co0 + dv * VEC) * ES;
    dlo[k] = p * DYB + ((dv ^ wg_swz<DYB, SWZ>(px)) << 4);
  }

  auto load_tile = [&](int tile, bool valid) {
    int b = valid ? tile : 0;
    const int tx = b % a.tiles_x;
    b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    const int n = b / a.tiles_y;
    const int iy0 = ty * TH * STRIDE - PAD, ix0 = tx * TW * STRIDE - PAD;
    const int xb = (n * a.H + iy0) * rowX + ix0 * pixX;          // may be negative: valid lanes add at least its magnitude
    const int db = (n * a.Ho + ty * TH) * rowD + tx * TW * pixD;
    xok = 0;
#pragma unroll
    for (int k = 0; k < XV; ++k) {
      const int gy = iy0 + (xhy[k] >> 16), gx = ix0 + (xhy[k] & 0xffff);
      const bool ok = valid && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      xr[k] = __builtin_amdgcn_raw_buffer_load_b128(rx_, ok ? (unsigned)(xb + xgo[k]) : OOB, 0, 0);
      xok |= (ok ? 1u : 0u) << k;
    }
#pragma unroll
    for (int k = 0; k < DV; ++k) {
      const int oy = ty * TH + (dpy[k] >> 16), ox = tx * TW + (dpy[k] & 0xffff);
      const bool ok = valid && oy < a.Ho && ox < a.Wo;
      dr[k] = __builtin_amdgcn_raw_buffer_load_b128(rdy, ok ? (unsigned)(db + dgo[k]) : OOB, 0, 0);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int k = 0; k < XV; ++k) {
      const int idx = tid + k * 256;
      if (idx < XVECS) {
        V16 val = xr[k];
        if (has_affine || a.in_relu) {
          float f[VEC];
          v16_unpack<T>(val, f);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            f[j] = fmaf(f[j], sc[j], sh[j]);
            if (a.in_relu) f[j] = fmaxf(f[j], 0.f);
          }
          val = ((xok >> k) & 1u) ? v16_pack<T>(f) : v16_zero();   // zero padding is applied AFTER the transform
        }
        *(V16*)(xl + xlo[k]) = val;
      }
    }
#pragma unroll
    for (int k = 0; k < DV; ++k) {
      const int idx = tid + k * 256;
      if (idx < DVECS) *(V16*)(dl + dlo[k]) = dr[k];
    }
  };

  // ---- per-lane offsets of the transposing operand reads (bf16). Lane group lg of a k-step reads four consecutive
  // pixels of a tile row - columns 4 lg .. 4 lg + 3 of the k-step's first row (16-pixel rows; 8-pixel rows: row lg / 2,
  // columns 4 (lg & 1) ..) with its first read, the same columns ROWS2 rows below with its second: which pixel carries
  // which k index is free as long as both operands agree, and this pairing keeps the 32-lane halves of a read on eight
  // consecutive pixels of one row (wg_swz). Displacements by k-step, tap row and fragment are instruction offsets.
  constexpr int KROWS = KSTEP / TW;            // tile rows of a k-step: 2 (or 4, or 1 / 2 in f32)
  constexpr int ROWS2 = TW == 16 ? 1 : 2;      // rows between a lane's two reads
  const int wq = (lane & 15) >> 2, wp4 = lane & 3;
  const int lrow = TW == 16 ? 0 : (lg >> 1), lcol = (TW == 16 ? 4 * lg : 4 * (lg & 1)) + (SWZ ? wq : 0);
  int wao[FCOW], wbo[KS][FCI];
#pragma unroll
  for (int f = 0; f < FCOW; ++f)
    wao[f] = (lrow * TW + lcol) * DYB + ((((wco * FCOW + f) * 2 + (wp4 >> 1)) ^ wg_swz<DYB, SWZ>(lcol)) << 4) + (wp4 & 1) * 8;
#pragma unroll
  for (int dx = 0; dx < KS; ++dx)
#pragma unroll
    for (int cf = 0; cf < FCI; ++cf)
      wbo[dx][cf] = (lrow * STRIDE * HALO_W + lcol * STRIDE + dx) * PIXB +
                    (((cf * 2 + (wp4 >> 1)) ^ wg_swz<PIXB, SWZ>(lcol * STRIDE + dx)) << 4) + (wp4 & 1) * 8;
  // one 16-channel x KSTEP-pixel fragment; bf16: `off` = table entry + displacement, the second read `hi` bytes below;
  // f32: four one-float reads of pixels P0, P0 + pstep, ... (padded rows, no swizzle)
  auto tr_frag = [&](const char* img, int off, int hi, int f32_off, int f32_step) -> V16 {
    if constexpr (SWZ) {
      const LDS_AS char* l = (const LDS_AS char*)img;
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(l + off));
      const bf16x4 h2 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(l + off + hi));
      const bf16x8 v = {lo.x, lo.y, lo.z, lo.w, h2.x, h2.y, h2.z, h2.w};
      return __builtin_bit_cast(V16, v);
    } else {
      const char* p = img + f32_off + li * 4;
      return V16{*(const uint32_t*)p, *(const uint32_t*)(p + f32_step), *(const uint32_t*)(p + 2 * f32_step),
                 *(const uint32_t*)(p + 3 * f32_step)};
    }
  };
  constexpr int MF = ES == 2 ? 1 : 4, RDF = ES == 2 ? 2 : 4;   // MFMAs / LDS reads behind one fragment
  // this wave's (tap, ci-fragment) list: offsets of its B fragments at k-step 0 (an uneven list repeats the wave's first
  // fragment in the missing position; that accumulator is dropped)
  int boj[NPW], bfj[NPW];
#pragma unroll
  for (int j = 0; j < NPW; ++j) {
    int fr = wn + WN * j;
    if (fr >= NFR) fr = wn;
    const int t = fr / FCI, cf = fr % FCI, ty_ = t / KS, tx_ = t % KS;
    int base = 0;
#pragma unroll
    for (int dx = 0; dx < KS; ++dx)
#pragma unroll
      for (int c = 0; c < FCI; ++c)
        if (dx == tx_ && c == cf) base = wbo[dx][c];
    boj[j] = base + ty_ * HALO_W * PIXB;
    const int pl = lg * VEC;                         // f32: first pixel of the lane group within a k-step
    bfj[j] = (((pl / TW) * STRIDE + ty_) * HALO_W + (pl % TW) * STRIDE + tx_) * PIXB + cf * 16 * ES;
  }

  int tile = blockIdx.x;
  if (tile < a.total_tiles) load_tile(tile, true);
  for (; tile < a.total_tiles; tile += gridDim.x) {
    store_tile();
    __syncthreads();
    load_tile(tile + gridDim.x, tile + (int)gridDim.x < a.total_tiles);  // in flight during the MFMAs
    // ---- steps (k-step, fragment j of this wave's (tap, ci-fragment) list): the B fragment of step s + RB - 1 is read
    // before the MFMAs of step s are issued; the A fragments of the next k-step half a k-step ahead ----
    {
      constexpr int NKS = BM / KSTEP, NST = NKS * NPW, RB = NPW >= 4 ? 4 : NPW + 1;
      V16 waf[2][FCOW], wbf[RB];
      auto rd_a = [&](int ks, int buf) {
#pragma unroll
        for (int f = 0; f < FCOW; ++f)
          waf[buf][f] = tr_frag(dl, wao[f] + ks * (KSTEP * DYB), ROWS2 * TW * DYB,
                                (ks * KSTEP + lg * VEC) * DYB + (wco * FCOW + f) * 16 * ES, DYB);
      };
      auto rd_bb = [&](int s_) {
        const int ks = s_ / NPW, j = s_ % NPW;
        wbf[s_ % RB] = tr_frag(xl, boj[j] + ks * (KROWS * STRIDE * HALO_W * PIXB), ROWS2 * STRIDE * HALO_W * PIXB,
                              bfj[j] + ks * ((KSTEP / TW) * STRIDE * HALO_W * PIXB), STRIDE * PIXB);
      };
      rd_a(0, 0);
#pragma unroll
      for (int s_ = 0; s_ < RB - 1 && s_ < NST; ++s_) rd_bb(s_);
      __builtin_amdgcn_sched_group_barrier(0x100, (FCOW + RB - 1) * RDF, 0);
#pragma unroll
      for (int s_ = 0; s_ < NST; ++s_) {
        const int ks = s_ / NPW, j = s_ % NPW;
        if (s_ + RB - 1 < NST) rd_bb(s_ + RB - 1);
        if (j == NPW / 2 && ks + 1 < NKS) rd_a(ks + 1, (ks + 1) & 1);
#pragma unroll
        for (int f = 0; f < FCOW; ++f) acc[j][f] = mma16<T>(waf[ks & 1][f], wbf[s_ % RB], acc[j][f]);
        {
          const int nrd = (s_ + RB - 1 < NST ? RDF : 0) + ((j == NPW / 2 && ks + 1 < NKS) ? FCOW * RDF : 0);
          switch (nrd) {
            case 2: __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); break;
            case 4: __builtin_amdgcn_sched_group_barrier(0x100, 4, 0); break;
            case 6: __builtin_amdgcn_sched_group_barrier(0x100, 6, 0); break;
            case 8: __builtin_amdgcn_sched_group_barrier(0x100, 8, 0); break;
            case 10: __builtin_amdgcn_sched_group_barrier(0x100, 10, 0); break;
            case 12: __builtin_amdgcn_sched_group_barrier(0x100, 12, 0); break;
            case 16: __builtin_amdgcn_sched_group_barrier(0x100, 16, 0); break;
            case 18: __builtin_amdgcn_sched_group_barrier(0x100, 18, 0); break;
            case 20: __builtin_amdgcn_sched_group_barrier(0x100, 20, 0); break;
            default: break;
          }
          __builtin_amdgcn_sched_group_barrier(0x008, FCOW * MF, 0);
        }
      }
    }
    __syncthreads();
  }

  if (a.atomic) {
    if constexpr (TAPS == 1) {
      // [co][ci]: the 16 lanes of an accumulator row are 16 consecutive floats
#pragma unroll
      for (int j = 0; j < NPW; ++j) {
        const int fr = wn + WN * j;
        if (fr < NFR) {
          const int ci = c0 + (fr % FCI) * 16 + li;
#pragma unroll
          for (int f = 0; f < FCOW; ++f) {
            const int co = co0 + (wco * FCOW + f) * 16 + lg * 4;
            const float v4[4] = {acc[j][f].x, acc[j][f].y, acc[j][f].z, acc[j][f].w};
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (co + r < a.Cout_real && ci < a.Cin_real) atomicAdd(a.slabs + (size_t)(co + r) * a.ld + ci, v4[r]);
          }
        }
      }
    } else {
      // through LDS as [co][ci][tap] (the loop ended on a barrier: the images are free), 16 output channels at a time
      constexpr int HCO = 16, NH = BCO / HCO;
      static_assert(HCO * KC * TAPS * 4 <= LDSTOT, "weight-gradient tile does not fit the LDS images");
      float* tl = (float*)lds;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
          const int fr = wn + WN * j;
          if (fr < NFR) {
            const int t = fr / FCI, cil = (fr % FCI) * 16 + li;
#pragma unroll
            for (int f = 0; f < FCOW; ++f) {
              const int col = (wco * FCOW + f) * 16 + lg * 4 - h * HCO;
              const float v4[4] = {acc[j][f].x, acc[j][f].y, acc[j][f].z, acc[j][f].w};
              if (col >= 0 && col < HCO) {
#pragma unroll
                for (int r = 0; r < 4; ++r) tl[((col + r) * KC + cil) * TAPS + t] = v4[r];
              }
            }
          }
        }
        __syncthreads();
        const int nci = min(KC, a.Cin_real - c0);      // real input channels of this block (<= 0: nothing to add)
        // (round 4: starting every split at its own 64-float segment of the tile changes nothing, 25.3 us either way - the
        // ~10 us the atomic form adds to a launch is the CU's own issue rate for float atomics, ~1 lane per cycle for the
        // tile's 18 K floats whatever the split count, not a convoy on the addresses: scratch/wgrad_atomic_micro.py)
        for (int idx = tid; idx < HCO * KC * TAPS; idx += 256) {
          const int col = idx / (KC * TAPS), rem = idx - col * (KC * TAPS);
          const int co = co0 + h * HCO + col;
          if (co < a.Cout_real && rem < nci * TAPS)
            atomicAdd(a.slabs + ((size_t)co * a.Cin_real + c0) * TAPS + rem, tl[idx]);
        }
        if (h + 1 < NH) __syncthreads();
      }
    }
    return;
  }
  // slab[split][co][tap][ci]; D layout: col (lane&15) = ci, row 4*(lane>>4)+r = co
  float* slab = a.slabs + (size_t)blockIdx.x * a.Cout * TAPS * a.Cin;
#pragma unroll
  for (int j = 0; j < NPW; ++j) {
    const int fr = wn + WN * j;
    if (fr < NFR) {
      const int t = fr / FCI, ci = c0 + (fr % FCI) * 16 + li;
#pragma unroll
      for (int f = 0; f < FCOW; ++f) {
        const int co = co0 + (wco * FCOW + f) * 16 + lg * 4;
        const float v4[4] = {acc[j][f].x, acc[j][f].y, acc[j][f].z, acc[j][f].w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (co + r < a.Cout && ci < a.Cin) slab[((size_t)(co + r) * TAPS + t) * a.Cin + ci] = v4[r];
      }
    }
  }
}

struct WgCfg {
  int th, tw, bco, kc, id;
};

WgCfg choose_wg(int dtype, int Ho, int Wo, int Cout, int ks, int stride, int Cin = 0) {
  const bool f32 = dtype == HR_F32;
  // the GEMM-shaped head layer (480x480, 720x720): 128x128 output tiles halve the operand traffic per FLOP
  // (measured 508 -> 365 us); narrower layers lose with it (64->256: 34 -> 48 us)
  if (ks == 1 && !f32 && Cout >= 256 && Cin >= 256) return WgCfg{8, 16, 128, 128, 11};
  if (ks == 1) return f32 ? WgCfg{8, 16, 64, 32, 10} : WgCfg{8, 16, 64, 64, 10};
  bool big = Ho >= 16 && Wo >= 16 && stride == 1;
  if (big) {
    // (as choose_tile in conv_body.h: 8x8 tiles where they walk far fewer padded pixels than 16x16 ones - w48's 24x18 maps)
    static const int pct = hr_knob("HRNET_SMALL_TILE_PCT", 80);
    const long long p16 = (long long)((Ho + 15) / 16 * 16) * ((Wo + 15) / 16 * 16), p8 = (long long)((Ho + 7) / 8 * 8) * ((Wo + 7) / 8 * 8);
    if (p8 * 100 <= p16 * pct) big = false;
  }
  const int kc = f32 ? 16 : 32;
  if (Cout <= 32) return big ? WgCfg{16, 16, 32, kc, 0} : WgCfg{8, 8, 32, kc, 1};
  return big ? WgCfg{16, 16, 64, kc, 2} : WgCfg{8, 8, 64, kc, 3};
}

template <typename T, int KC3, int KC1>
int launch_wg(const WgradArgs& a, const WgCfg& c, int ks, int stride, dim3 grid, hipStream_t s) {
#define WG(KS_, ST_, TH_, TW_, BCO_, KC_, WCO_, WN_) \
  hipLaunchKernelGGL((wgrad_kernel<T, KS_, ST_, TH_, TW_, BCO_, KC_, WCO_, WN_>), grid, dim3(256), 0, s, a)
  if (ks == 1 && c.id == 11) {
    if constexpr (std::is_same<T, bf16_t>::value) WG(1, 1, 8, 16, 128, 128, 2, 2);
  } else if (ks == 1) {
    WG(1, 1, 8, 16, 64, KC1, 2, 2);
  } else if (stride == 1) {
    switch (c.id) {
      case 0: WG(3, 1, 16, 16, 32, KC3, 1, 4); break;
      case 1: WG(3, 1, 8, 8, 32, KC3, 1, 4); break;
      case 2: WG(3, 1, 16, 16, 64, KC3, 2, 2); break;
      default: WG(3, 1, 8, 8, 64, KC3, 2, 2); break;
    }
  } else {
    if (c.bco == 32) WG(3, 2, 8, 8, 32, KC3, 1, 4);
    else WG(3, 2, 8, 8, 64, KC3, 2, 2);
  }
#undef WG
  return hr_check_launch("conv2d_wgrad");
}

}  // namespace

extern "C" int hrnet_wgrad_splits(int dtype, int N, int Ho, int Wo, int Cout, int Cin, int ks,
                                  int stride) {
  const WgCfg c = choose_wg(dtype, Ho, Wo, Cout, ks, stride, Cin);
  const int tiles = N * ((Ho + c.th - 1) / c.th) * ((Wo + c.tw - 1) / c.tw);
  const int gy = (Cout + c.bco - 1) / c.bco;
  const int gz = (Cin + c.kc - 1) / c.kc;
  // measured on MI355X (scratch/wgrad_micro.py): fastest with ~512 workgroups in total for 3x3 and ~1024
  // for 1x1 tiles, and only when every split walks the same number of tiles (an uneven split costs
  // 20-30 %). Each split costs one f32 slab of Cout*taps*Cin written and re-read by hrnet_wgrad_reduce.
  static const int wdiv = hr_knob("HRNET_WGRAD_DIV", 1);   // (measurement override)
  int target = (ks == 1 ? 1024 : 512) / wdiv;
  // 3x3 launches with plenty of pixel tiles per output block (w48's 96-channel branch on 48x36 maps: 288 tiles x 6
  // blocks) take half as many workgroups: every workgroup ends with ~18 K float atomics into the gradient (10 us of a
  // 25 us launch at two tiles per workgroup, scratch/wgrad_atomic_micro.py), so twice the tiles per workgroup halve that
  // share - w48 29.92 -> 29.40 ms/step; w32's deferred launches (64 tiles x 8 blocks) stay where they are
  static const int bigt = hr_knob("HRNET_WGRAD_BIG_TILES", 1024);   // (measurement: 0 = off)
  // (bf16 only: the f32 launches - 16-channel input blocks, twice the blocks - lose with it, 55.4 against 53.4 ms/step)
  if (ks == 3 && dtype != HR_F32 && bigt > 0 && (long long)tiles * gy * gz >= bigt) target /= 2;
  int ns = target / (gy * gz);
  if (ns < 1) ns = 1;
  if (ns > 512) ns = 512;
  if (ns > tiles) ns = tiles;
  // slab traffic (written here, re-read by the reduce) is kept below max(20 MB, 3/4 of the operand bytes):
  // 64->64 @32x32 runs 18.5 us with 128 splits and 17.0 us with 256, but the extra 19 MB cost ~8 us of HBM time
  const double esz = dtype == HR_F32 ? 4.0 : 2.0;
  const double operand = ((double)N * Ho * Wo * Cout + (double)N * Ho * Wo * stride * stride * Cin) * esz;
  const double slab = (double)Cout * ks * ks * Cin * 4.0;
  double budget = 0.75 * operand;
  if (budget < 20e6) budget = 20e6;
  if (ns * slab > budget) ns = (int)(budget / slab);
  if (ns < 1) ns = 1;
  int even = ns;
  while (even > 1 && tiles % even != 0) --even;
  if (even * 2 > ns) ns = even;
  return ns;
}

// workgroups per split of a weight-gradient launch (output-channel blocks x input-channel chunks): a launch runs
// splits x this many workgroups
extern "C" int hrnet_wgrad_blocks_per_split(int dtype, int Ho, int Wo, int Cout, int Cin, int ks, int stride) {
  const WgCfg c = choose_wg(dtype, Ho, Wo, Cout, ks, stride, Cin);
  return ((Cout + c.bco - 1) / c.bco) * ((Cin + c.kc - 1) / c.kc);
}

// pixel tiles a weight-gradient launch walks (each of the hrnet_wgrad_splits() splits takes tiles/splits of them)
extern "C" int hrnet_wgrad_tiles(int dtype, int N, int Ho, int Wo, int Cout, int Cin, int ks, int stride) {
  const WgCfg c = choose_wg(dtype, Ho, Wo, Cout, ks, stride, Cin);
  return N * ((Ho + c.th - 1) / c.th) * ((Wo + c.tw - 1) / c.tw);
}

int hr_launch_wgrad(const HrOp& op, hipStream_t s) {
  const int dtype = op.i[0], N = op.i[1], H = op.i[2], W = op.i[3], Cin = op.i[4], Ho = op.i[5],
            Wo = op.i[6], Cout = op.i[7], ks = op.i[8], stride = op.i[9], nsplit = op.i[11];
  HR_REQUIRE(dtype == HR_F32 || dtype == HR_BF16, "wgrad: bad dtype %d", dtype);
  HR_REQUIRE(ks == 1 || ks == 3, "wgrad: kernel size %d", ks);
  HR_REQUIRE(stride == 1 || (stride == 2 && ks == 3), "wgrad: stride %d", stride);
  HR_REQUIRE(Cin % (dtype == HR_F32 ? 4 : 8) == 0 && Cout % (dtype == HR_F32 ? 4 : 8) == 0,
             "wgrad: channel counts must be 16-byte multiples (Cin=%d Cout=%d)", Cin, Cout);
  HR_REQUIRE(nsplit >= 1, "wgrad: nsplit=%d", nsplit);
  HR_REQUIRE(op.p[0] && op.p[1] && op.p[4], "wgrad: null pointer");
  const int pad = ks / 2;
  HR_REQUIRE((H + 2 * pad - ks) / stride + 1 == Ho && (W + 2 * pad - ks) / stride + 1 == Wo,
             "wgrad: shape mismatch");
  const WgCfg c = choose_wg(dtype, Ho, Wo, Cout, ks, stride, Cin);
  WgradArgs a;
  a.x = (const char*)op.p[0];
  a.dy = (const char*)op.p[1];
  a.in_scale = (const float*)op.p[2];
  a.in_shift = (const float*)op.p[3];
  a.slabs = (float*)op.p[4];
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
  const double es_ = dtype == HR_F32 ? 4.0 : 2.0;
  HR_REQUIRE((double)N * H * W * Cin * es_ < 2147483648.0 && (double)N * Ho * Wo * Cout * es_ < 2147483648.0,
             "wgrad: an operand tensor of 2 GiB or more (32-bit buffer offsets)");
  a.x_bytes = (unsigned)((double)N * H * W * Cin * es_);
  a.dy_bytes = (unsigned)((double)N * Ho * Wo * Cout * es_);
  a.tiles_y = (Ho + c.th - 1) / c.th;
  a.tiles_x = (Wo + c.tw - 1) / c.tw;
  a.total_tiles = N * a.tiles_y * a.tiles_x;
  a.in_relu = op.i[10];
  a.atomic = op.i[12]; a.Cout_real = op.i[13]; a.Cin_real = op.i[14];
  a.ld = op.i[15] ? op.i[15] : a.Cin_real;
  HR_REQUIRE(!op.i[15] || (a.atomic && ks == 1 && op.i[15] >= a.Cin_real),
             "wgrad: a row pitch (%d) belongs to an atomic 1x1 launch", op.i[15]);
  HR_REQUIRE(!a.atomic || (a.Cout_real >= 1 && a.Cout_real <= Cout && a.Cin_real >= 1 && a.Cin_real <= Cin),
             "wgrad: the atomic form needs the gradient's real channel counts (%d, %d)", a.Cout_real, a.Cin_real);
  dim3 grid((unsigned)nsplit, (unsigned)((Cout + c.bco - 1) / c.bco), (unsigned)((Cin + c.kc - 1) / c.kc));
  if (dtype == HR_F32) return launch_wg<float, 16, 32>(a, c, ks, stride, grid, s);
  return launch_wg<bf16_t, 32, 64>(a, c, ks, stride, grid, s);
}

extern "C" int hrnet_conv2d_wgrad(int dtype, const void* x, const void* dy, const float* in_scale,
                                  const float* in_shift, float* slabs, int N, int H, int W, int Cin,
                                  int Ho, int Wo, int Cout, int ks, int stride, int in_relu,
                                  int nsplit, hr_stream_t stream) {
  HrOp op = {};
  op.kind = HR_OP_WGRAD;
  const int iv[12] = {dtype, N, H, W, Cin, Ho, Wo, Cout, ks, stride, in_relu, nsplit};
  for (int k = 0; k < 12; ++k) op.i[k] = iv[k];
  op.p[0] = (void*)x; op.p[1] = (void*)dy; op.p[2] = (void*)in_scale; op.p[3] = (void*)in_shift;
  op.p[4] = slabs;
  return hr_launch_wgrad(op, (hipStream_t)stream);
}

extern "C" int hrnet_wgrad_kernel_name(int dtype, int Ho, int Wo, int Cout, int Cin, int ks, int stride, char* buf,
                                       int buflen) {
  const WgCfg c = choose_wg(dtype, Ho, Wo, Cout, ks, stride, Cin);
  int wco = 2, wn = 2;
  if (ks == 3 && c.bco == 32) { wco = 1; wn = 4; }
  return snprintf(buf, buflen, "wgrad_kernel<%s, %d, %d, %d, %d, %d, %d, %d, %d>", dtype == HR_F32 ? "float" : "__bf16",
                  ks, stride, c.th, c.tw, c.bco, c.kc, wco, wn);
}
