// (device body + launch templates; instantiated per MODE by conv.hip / conv_bs.hip / conv_fwd.hip / conv_dg.hip)
// Implicit-GEMM convolution on MFMA for gfx950: forward conv and (with transposed packed
// weights / zero-stuffed input) the input gradient.
//
//   D[cout][pixel] += sum_{tap, ci} Wp[cout][tap][ci] * Xa[pixel + tap][ci]
//
// A workgroup (4 waves) owns a TH x TW output-pixel tile of one image and BN output channels.
// Per K chunk of KC input channels it stages
//   - the input halo tile [(TH-1)*S+KS][(TW-1)*S+KS][KC] into LDS, applying the producer's
//     BatchNorm affine + ReLU on the way (so normalised activations never round-trip HBM) and
//     zero padding AFTER the transform;
//   - the weight slice [BN][KS*KS][KC];
// then every tap is a shifted LDS view of the same halo (9x reuse of each staged byte).
// MFMA operands: A = weights (rows = cout), B = activations (cols = pixels), so a lane ends up
// with 4*FC CONTIGUOUS output channels of one pixel (weight rows are permuted on the LDS read)
// and stores them as one 16-byte NHWC vector.
// BatchNorm batch statistics (sum, sum of squares per channel) are reduced from the f32
// accumulators in the epilogue: in-lane over pixel fragments, DPP over the 16 pixel lanes, LDS
// across waves, one row per workgroup to HBM (deterministic; finished by bn_finalize).
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {


struct ConvArgs {
  const char* x;
  const char* w;
  const float* in_scale;
  const float* in_shift;
  const float* bias;
  char* y;
  float* stats;
  // backward-statistics mode (dgrad launches): the statistics rows hold (sum dz, sum dz*yraw) of the
  // BatchNorm that produced this launch's output gradient, dz = v * [mask > 0], with the mask value
  // bs_mask (or bs_y) optionally mapped through bs_scale/bs_shift. Replaces a bn_bwd_reduce pass.
  const char* bs_y;
  const char* bs_mask;
  const float* bs_scale;
  const float* bs_shift;
  int N, H, W, Cin;  // stored input
  int Hz, Wz;        // logical input extent (== H, W unless upz)
  int Ho, Wo, Cout;
  int tiles_y, tiles_x, total_tiles, tpw;  // tpw = pixel tiles per workgroup
  int gx, gy;                              // pixel walks x output-channel blocks (1-D grid, see conv_body)
  int xcd_runs;                            // workgroup -> walk mapping: 1 = every XCD takes a contiguous run of walks
  int in_relu, upz, accumulate;
  // input BatchNorm given as batch sums (hr_bn_from_sums) instead of scale/shift arrays: no finalize launch between
  // the producer and this conv. stats_atomic: this launch's own statistics go to sums[8][2][Cout] by float atomics
  // (zeroed by the caller) instead of one deterministic row per workgroup.
  const float* in_sums;
  const float* in_gamma;
  const float* in_beta;
  float in_inv_count, in_eps;
  int stats_atomic;
  // CONV_FWDS: the input is the output of a residual sum that is formed HERE, a = relu(scale*x + shift + x2) (the
  // BasicBlock / Bottleneck tail relu(bn(y) + identity), pose_hrnet.py:54-55, :95-96), and the centre pixels of
  // every staged tile are also written to `side`: the sum tensor the next residual add and the backward pass read.
  const char* x2;
  char* side;
  // backward-statistics launches: store dz = v * [mask > 0] instead of v (the masked-gradient convention of the
  // fused backward launches: the buffer then needs no separate mask pass)
  int bs_store_masked;
  // the staged input window is displaced by (in_dy, in_dx) pixels (zero outside the image): one tap of a dilated
  // convolution run as a shifted 1x1 conv (hrnet_conv2d_dilated3x3)
  int in_dy, in_dx;
};

constexpr int HR_CONV_MAXC = 768;   // channels of the on-the-fly coefficient table (w48 head: 720)

template <typename T, int KS, int STRIDE, int TH, int TW, int BN, int WP, int WC, int KM>
struct ConvCfg {
  static constexpr int VEC = TT<T>::VEC;
  static constexpr int KSTEP = TT<T>::KSTEP;
  static constexpr int KC = KSTEP * KM;            // channels staged per chunk (KM fragment steps)
  static constexpr int VPP = KC / VEC;             // 16-byte vectors per pixel per chunk
  static constexpr int TAPS = KS * KS;
  // STRIDE == 4 is the four-parity input gradient of a 3x3 stride-2 conv (S2D, see conv_body): the staged
  // image is the (TH/2+1) x (TW/2+1) dY tile that the TH x TW output tile depends on
  static constexpr bool S2D = STRIDE == 4;
  static constexpr int HALO_H = S2D ? TH / 2 + 1 : (TH - 1) * STRIDE + KS;
  static constexpr int HALO_W = S2D ? TW / 2 + 1 : (TW - 1) * STRIDE + KS;
  static constexpr int PIXB = KC * (int)sizeof(T) + 16;          // padded pixel stride (bytes)
  static constexpr int WROWB = TAPS * KC * (int)sizeof(T) + 16;  // padded weight-row stride
  static constexpr int XBYTES = HALO_H * HALO_W * PIXB;
  static constexpr int WBYTES = BN * WROWB;
  static constexpr int XVECS = HALO_H * HALO_W * VPP;   // 16-byte vectors of one staged halo chunk
  static constexpr int WVECS = BN * TAPS * VPP;
  static constexpr int XV = (XVECS + 255) / 256;        // ... per thread
  static constexpr int WV = (WVECS + 255) / 256;
  static constexpr int BM = TH * TW;
  static constexpr int PM = BM / WP;   // pixels per wave
  static constexpr int FP = PM / 16;   // pixel fragments per wave
  static constexpr int CN = BN / WC;   // couts per wave
  static constexpr int FC = CN / 16;   // cout fragments per wave
  static constexpr int LANE_C = 4 * FC;  // contiguous couts per lane
  static constexpr int STATB = WP * BN * 2 * (int)sizeof(float);
  static constexpr int LDSB0 = (XBYTES + WBYTES) > STATB ? (XBYTES + WBYTES) : STATB;
  static constexpr int LDSB = (LDSB0 + 15) / 16 * 16;
  static_assert(WP * WC == 4, "4 waves");
  static_assert(PM % 16 == 0 && CN % 16 == 0, "fragment multiples");
  static_assert(256 % VPP == 0, "each thread keeps one channel vector");
  static_assert(XV <= 31, "validity mask");
  static_assert(!S2D || (KS == 3 && FP == 4 && (TH / 2) * (TW / 2) == WP * 16), "S2D: one fragment per parity");
};

// A workgroup walks `tpw` pixel tiles x `nch` K-chunks as a flat sequence of stages. The global
// loads of stage s+1 are issued into registers before the MFMAs of stage s and written to LDS
// after them (issue-early / write-late), so HBM/L2 latency hides under the matrix work even at
// one workgroup per CU. With a single K chunk (Cin <= KC) the weight slice is staged once and
// stays resident for every tile. BatchNorm statistics accumulate in registers across the tiles
// and are reduced across lanes / waves once per workgroup.
// MODE selects what is compiled in (run-time feature flags keep their operands live: the generic body
// needs ~20 more registers and 30 % more code than the specialised ones):
//   CONV_GENERIC  everything by run-time flag (bias, zero-stuffed input, accumulate, ...)
//   CONV_BS       input gradient + backward-statistics epilogue (raw dY in, no bias, no forward statistics)
//   CONV_FWD      forward conv feeding a BatchNorm: optional input affine/ReLU, statistics; no bias /
//                 zero-stuffing / accumulate
//   CONV_DG       plain input gradient: raw dY in, no bias, no statistics; zero-stuffing / accumulate by flag
//   CONV_FWDB     CONV_FWD with a bias (the head convs)
//   CONV_FWDS     CONV_FWD whose input is a residual sum formed in the prologue (two input tensors) and written out
enum { CONV_GENERIC = 0, CONV_BS = 1, CONV_FWD = 2, CONV_DG = 3, CONV_FWDB = 4, CONV_FWDS = 5 };

template <typename T, int KS, int STRIDE, int TH, int TW, int BN, int WP, int WC, int KM, int MODE>
__device__ __forceinline__ void conv_body(const ConvArgs& a) {
  constexpr bool BS = MODE == CONV_BS;
  constexpr bool RAW_IN = MODE == CONV_BS || MODE == CONV_DG;   // no input affine / ReLU
  constexpr bool NO_STATS = MODE == CONV_DG;
  using C = ConvCfg<T, KS, STRIDE, TH, TW, BN, WP, WC, KM>;
  constexpr int VEC = C::VEC;
  // modes that can read an input BatchNorm given as batch sums (training forward launches). The generic kernel
  // (eval-mode forward: precomputed scale/shift) stays without the table: its LDS and code cost 4-7 % there
  constexpr bool TAB = MODE == CONV_FWD || MODE == CONV_FWDB || MODE == CONV_FWDS;
  constexpr bool X2 = MODE == CONV_FWDS;
  __shared__ __attribute__((aligned(16))) char lds[C::LDSB + (TAB ? 2 * HR_CONV_MAXC * 4 : 0)];
  char* xl = lds;
  char* wl = lds + C::XBYTES;
  float* bntab = (float*)(lds + C::LDSB);     // [scale Cin][shift Cin] computed from the producer's batch sums

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wp = wave % WP;
  const int wc = wave / WP;
  const int li = lane & 15;
  const int lg = lane >> 4;
  // 1-D grid, XCD-aware: workgroup ids round-robin over the 8 XCDs, so the gy output-channel blocks of one
  // pixel walk get ids 8 apart - the same XCD (they share its L2 for the input tile) and adjacent in time
  // ... and (xcd_runs, round 3) every XCD takes ONE contiguous run of walks: logical index L = (id % 8) * (grid / 8) +
  // id / 8, so neighbouring tiles (which share halo rows) meet in one L2 as well - with consecutive walks on
  // consecutive XCDs every XCD fetched its neighbours' halo rows again (1.24 - 1.57x the algorithmic bytes)
  int wg_p, wg_nb;
  if (a.xcd_runs) {
    const int G8 = gridDim.x >> 3;
    const int L = ((int)blockIdx.x & 7) * G8 + ((int)blockIdx.x >> 3);
    wg_p = L / a.gy; wg_nb = L - wg_p * a.gy;
  } else if (a.gy == 1) {
    wg_p = blockIdx.x; wg_nb = 0;
  } else {
    const int grp = blockIdx.x / (8 * a.gy), r = blockIdx.x % (8 * a.gy);
    wg_nb = r / 8;
    wg_p = grp * 8 + (r & 7);
  }
  if (wg_p >= a.gx) return;
  const int n0 = wg_nb * BN;
  constexpr bool S2D = C::S2D;
  constexpr int PAD = S2D ? 0 : KS / 2;
  constexpr int SST = S2D ? 1 : STRIDE;     // stride of the staged image
  constexpr int TIH = S2D ? TH / 2 : TH, TIW = S2D ? TW / 2 : TW;   // tile extent in staged-image steps

  const int nch = (a.Cin + C::KC - 1) / C::KC;
  const int tile0 = wg_p * a.tpw;
  const int ntile = min(a.tpw, a.total_tiles - tile0);
  const int nstage = ntile * nch;
  const bool wres = nch == 1;  // weights stay resident in LDS
  const int v = tid % C::VPP;  // this thread's 16-byte vector within a pixel / weight tap (fixed)
  // a backward-statistics launch is an input-gradient conv: raw dY in, no bias (checked by the launcher)
  const bool from_sums = TAB ? a.in_sums != nullptr : false;
  const bool has_affine = RAW_IN ? false : (a.in_scale != nullptr || from_sums);
  // (the table is built AFTER the first stage's global loads are issued - its 16 loads per channel and the f64
  // arithmetic run under their latency - and read when that stage is written to LDS)
  auto build_bntab = [&]() {
    if constexpr (TAB) {
      if (from_sums) {
        for (int c = threadIdx.x; c < a.Cin; c += 256) {
          float sc_, sh_, m_, r_, v_;
          hr_bn_from_sums(a.in_sums, a.Cin, c, a.in_inv_count, a.in_eps, a.in_gamma[c], a.in_beta[c], sc_, sh_, m_, r_, v_);
          bntab[c] = sc_;
          bntab[a.Cin + c] = sh_;
        }
        __syncthreads();
      }
    }
  };
  const bool in_relu = RAW_IN ? false : a.in_relu != 0;
  constexpr bool FWDLIKE = MODE == CONV_FWD || MODE == CONV_FWDB || MODE == CONV_FWDS;
  const bool A_UPZ = FWDLIKE ? false : a.upz != 0;
  const bool A_ACC = FWDLIKE ? false : a.accumulate != 0;
  const bool A_BIAS = MODE == CONV_GENERIC ? a.bias != nullptr : MODE == CONV_FWDB;

  // per-lane LDS byte offsets of the MFMA operands
  int aoff[C::FC];
#pragma unroll
  for (int fc = 0; fc < C::FC; ++fc) {
    // weight rows are STORED in MFMA order (row q = fc*16 + li of the wave's block holds output channel
    // (li>>2)*LANE_C + fc*4 + (li&3)), so the 16 lanes of a fragment read 16 consecutive padded rows:
    // conflict-free, where reading permuted rows put rows r and r+16 on the same banks
    const int row = wc * C::CN + fc * 16 + li;
    aoff[fc] = row * C::WROWB + lg * 16;
  }
  int boff[C::FP];
#pragma unroll
  for (int fp = 0; fp < C::FP; ++fp) {
    if constexpr (S2D) {
      // fragment fp = output parity; lane li of wave wp = dY position q of the tile (same for every parity)
      const int q = wp * 16 + li;
      boff[fp] = ((q / TIW) * C::HALO_W + q % TIW) * C::PIXB + lg * 16;
    } else {
      const int p = wp * C::PM + fp * 16 + li;
      const int py = p / TW, px = p % TW;
      boff[fp] = ((py * STRIDE) * C::HALO_W + px * STRIDE) * C::PIXB + lg * 16;
    }
  }
  // output pixel of (fragment, lane) within the tile
  auto out_yx = [&](int fp, int& y, int& x) {
    if constexpr (S2D) {
      const int q = wp * 16 + li;
      y = 2 * (q / TIW) + (fp >> 1);
      x = 2 * (q % TIW) + (fp & 1);
    } else {
      const int p = wp * C::PM + fp * 16 + li;
      y = p / TW;
      x = p % TW;
    }
  };

  // staging registers of the stage in flight
  V16 xr[C::XV], wr[C::WV];
  V16 x2r[X2 ? C::XV : 1];        // the identity term of the residual sum (CONV_FWDS)
  char* side_base = nullptr;      // where the staged tile's first halo pixel lives in `side`
  unsigned xok = 0;
  float sc[VEC], sh[VEC];

  // Per-thread staging geometry is the same for every tile: halo coordinates, the byte offset of
  // the element relative to the tile's first halo pixel, and its LDS slot are computed ONCE; per
  // tile only two adds + two unsigned compares per element remain (the address is a wave-uniform
  // base + a constant per-lane offset).
  const int esz = (int)sizeof(T);
  const int rowB = a.W * a.Cin * esz;      // bytes per stored input row
  const int pixB = a.Cin * esz;
  int hyx[C::XV], goff[C::XV];
#pragma unroll
  for (int k = 0; k < C::XV; ++k) {
    const int pix = (tid + k * 256) / C::VPP;
    const int hy = pix / C::HALO_W, hx = pix % C::HALO_W;
    hyx[k] = (hy << 16) | hx;
    goff[k] = A_UPZ ? 0 : hy * rowB + hx * pixB + v * 16;
  }
  // CONV_FWDS: which of this thread's staged vectors are centre (non-halo) pixels of the tile - those are written
  // to `side`, by the workgroups of the first output-channel block only (every pixel exactly once)
  unsigned cmask = 0;
  if constexpr (X2) {
    constexpr int PADC = KS / 2;
#pragma unroll
    for (int k = 0; k < C::XV; ++k) {
      const int hy = hyx[k] >> 16, hx = hyx[k] & 0xffff;
      if (hy >= PADC && hy < PADC + TH && hx >= PADC && hx < PADC + TW) cmask |= 1u << k;
    }
    if (n0 != 0 || a.side == nullptr) cmask = 0;
  }
  const int ldsx = (tid / C::VPP) * C::PIXB + v * 16;   // + k * (256 / VPP) * PIXB

  // tile walk without divisions inside the loop
  int cur_n, cur_ty, cur_tx;      // tile whose loads are being issued
  {
    int bq = tile0;
    cur_tx = bq % a.tiles_x;
    bq /= a.tiles_x;
    cur_ty = bq % a.tiles_y;
    cur_n = bq / a.tiles_y;
  }
  int ep_n = cur_n, ep_ty = cur_ty, ep_tx = cur_tx;   // tile whose accumulators are being finished
  auto advance = [&](int& n, int& ty, int& tx) {
    if (++tx == a.tiles_x) {
      tx = 0;
      if (++ty == a.tiles_y) {
        ty = 0;
        ++n;
      }
    }
  };

  auto load_stage = [&](int s) {
    const int t = s / nch, ch = s - t * nch;
    const int n = __builtin_amdgcn_readfirstlane(cur_n), ty = __builtin_amdgcn_readfirstlane(cur_ty),
              tx = __builtin_amdgcn_readfirstlane(cur_tx);
    const int iy0 = ty * TIH * SST - PAD + a.in_dy, ix0 = tx * TIW * SST - PAD + a.in_dx;
    const int c = ch * C::KC + v * VEC;
    const bool cvalid = c < a.Cin;
    if (has_affine && cvalid && !from_sums) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        sc[j] = a.in_scale[c + j];
        sh[j] = a.in_shift[c + j];
      }
    }
    xok = 0;
    if (!A_UPZ) {
      // wave-uniform base of (image n, halo origin, channel chunk); may point before the image
      const char* base = a.x + (size_t)n * a.H * rowB + (ptrdiff_t)iy0 * rowB + (ptrdiff_t)ix0 * pixB +
                         ch * C::KC * esz;
#pragma unroll
      for (int k = 0; k < C::XV; ++k) {
        const int gy = iy0 + (hyx[k] >> 16), gx = ix0 + (hyx[k] & 0xffff);
        const bool ok = (tid + k * 256) < C::XVECS && cvalid && (unsigned)gy < (unsigned)a.Hz &&
                        (unsigned)gx < (unsigned)a.Wz;
        xr[k] = v16_zero();
        if (ok) {
          xr[k] = *(const V16*)(base + goff[k]);
          xok |= 1u << k;
        }
        if constexpr (X2) {
          x2r[k] = v16_zero();
          if (ok) x2r[k] = *(const V16*)(a.x2 + (base - a.x) + goff[k]);
        }
      }
      if constexpr (X2) side_base = a.side + (base - a.x);
    } else {
#pragma unroll
      for (int k = 0; k < C::XV; ++k) {
        const int gy = iy0 + (hyx[k] >> 16), gx = ix0 + (hyx[k] & 0xffff);
        bool ok = (tid + k * 256) < C::XVECS && cvalid && (unsigned)gy < (unsigned)a.Hz &&
                  (unsigned)gx < (unsigned)a.Wz && !((gy | gx) & 1);
        const int sy = gy >> 1, sx = gx >> 1;
        ok = ok && sy < a.H && sx < a.W;
        xr[k] = v16_zero();
        if (ok) {
          xr[k] = *(const V16*)(a.x + ((size_t)((n * a.H + sy) * a.W + sx) * a.Cin + c) * sizeof(T));
          xok |= 1u << k;
        }
      }
    }
    if (ch == nch - 1) advance(cur_n, cur_ty, cur_tx);
    if (!wres || s == 0) {
#pragma unroll
      for (int k = 0; k < C::WV; ++k) {
        const int idx = tid + k * 256;
        const int rt = idx / C::VPP;
        const int tp = rt % C::TAPS, r = rt / C::TAPS;
        const int q = r % C::CN;   // LDS row r holds the output channel the MFMA row order needs
        const int co = n0 + (r / C::CN) * C::CN + ((q & 15) >> 2) * C::LANE_C + (q >> 4) * 4 + (q & 3);
        wr[k] = v16_zero();
        if (idx < C::WVECS && cvalid && co < a.Cout)
          wr[k] = *(const V16*)(a.w + ((size_t)(co * C::TAPS + tp) * a.Cin + c) * sizeof(T));
      }
    }
  };

  // BatchNorm affine (+ReLU) of the staged vectors; the three variants are selected by ONE
  // wave-uniform branch (runtime flags inside the unrolled loop made the compiler evaluate both
  // sides and select per element: 56 VALU per vector instead of 28)
  auto xform = [&](auto aff_tag, auto relu_tag) {
    constexpr bool AFF = decltype(aff_tag)::value, RELU = decltype(relu_tag)::value;
#pragma unroll
    for (int k = 0; k < C::XV; ++k) {
      if ((xok >> k) & 1u) {
        float f[VEC];
        v16_unpack<T>(xr[k], f);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          if constexpr (AFF) f[j] = fmaf(f[j], sc[j], sh[j]);
          if constexpr (RELU) f[j] = f[j] > 0.f ? f[j] : 0.f;
        }
        xr[k] = v16_pack<T>(f);
      }
    }
  };

  auto store_stage = [&](int s) {
    if constexpr (TAB) {
      if (from_sums) {      // this stage's channel chunk out of the on-the-fly table
        const int c = (s % nch) * C::KC + v * VEC;
        if (c < a.Cin) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            sc[j] = bntab[c + j];
            sh[j] = bntab[a.Cin + c + j];
          }
        }
      }
    }
    if constexpr (X2) {
      // a = relu(scale*x + shift + x2), rounded once: what goes to LDS IS what goes to `side`
#pragma unroll
      for (int k = 0; k < C::XV; ++k) {
        if ((xok >> k) & 1u) {
          float f[VEC], g2[VEC];
          v16_unpack<T>(xr[k], f);
          v16_unpack<T>(x2r[k], g2);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            f[j] = fmaf(f[j], sc[j], sh[j]) + g2[j];
            f[j] = f[j] > 0.f ? f[j] : 0.f;
          }
          xr[k] = v16_pack<T>(f);
          if ((cmask >> k) & 1u) *(V16*)(side_base + goff[k]) = xr[k];
        }
      }
    } else
    if (has_affine) {
      if (in_relu) xform(std::true_type{}, std::true_type{});
      else xform(std::true_type{}, std::false_type{});
    } else if (in_relu) {
      xform(std::false_type{}, std::true_type{});
    }
#pragma unroll
    for (int k = 0; k < C::XV; ++k)
      if (tid + k * 256 < C::XVECS) *(V16*)(xl + ldsx + k * (256 / C::VPP) * C::PIXB) = xr[k];
    if (!wres || s == 0) {
#pragma unroll
      for (int k = 0; k < C::WV; ++k) {
        const int idx = tid + k * 256;
        if (idx < C::WVECS) {
          const int rt = idx / C::VPP;
          const int tp = rt % C::TAPS, r = rt / C::TAPS;
          *(V16*)(wl + r * C::WROWB + (tp * C::KC) * (int)sizeof(T) + v * 16) = wr[k];
        }
      }
    }
  };

  const int cbase = n0 + wc * C::CN + lg * C::LANE_C;
  const bool cok = cbase < a.Cout;
  float bias[C::LANE_C];
#pragma unroll
  for (int k = 0; k < C::LANE_C; ++k) bias[k] = (A_BIAS && cok) ? a.bias[cbase + k] : 0.f;
  float s1[C::LANE_C], s2[C::LANE_C];
#pragma unroll
  for (int k = 0; k < C::LANE_C; ++k) s1[k] = s2[k] = 0.f;
  // backward-statistics operands: one lane's LANE_C contiguous channels per pixel of tensors laid out
  // like the output, fetched before the tile's last K chunk so the epilogue does not wait for them
  constexpr int LB = C::LANE_C * (int)sizeof(T);        // bytes per lane per pixel: 8 .. 64
  constexpr int LV = LB >= 16 ? LB / 16 : 1;
  V16 pre_y[C::FP][LV], pre_m[C::FP][LV];
  auto lane_load = [&](const char* src, V16* out) {
    if constexpr (LB >= 16) {
#pragma unroll
      for (int q = 0; q < LV; ++q) out[q] = *(const V16*)(src + q * 16);
    } else {
      const uint2 q = *(const uint2*)src;
      out[0] = V16{q.x, q.y, 0u, 0u};
    }
  };
  auto lane_unpack = [&](const V16* in, float* out) {
    if constexpr (LB >= 16) {
#pragma unroll
      for (int q = 0; q < LV; ++q) v16_unpack<T>(in[q], out + q * VEC);
    } else {
      float f[VEC];
      v16_unpack<T>(in[0], f);
#pragma unroll
      for (int k = 0; k < C::LANE_C; ++k) out[k] = f[k];
    }
  };
  auto bs_prefetch = [&]() {
    const int n = __builtin_amdgcn_readfirstlane(ep_n), ty = __builtin_amdgcn_readfirstlane(ep_ty),
              tx = __builtin_amdgcn_readfirstlane(ep_tx);
#pragma unroll
    for (int fp = 0; fp < C::FP; ++fp) {
      int oy, ox;
      out_yx(fp, oy, ox);
      oy += ty * TH; ox += tx * TW;
      if (cok && oy < a.Ho && ox < a.Wo) {
        const size_t off = ((size_t)((n * a.Ho + oy) * a.Wo + ox) * a.Cout + cbase) * sizeof(T);
        lane_load(a.bs_y + off, pre_y[fp]);
        if (a.bs_mask) lane_load(a.bs_mask + off, pre_m[fp]);
      }
    }
  };
  float bsc[C::LANE_C], bsh[C::LANE_C];
#pragma unroll
  for (int k = 0; k < C::LANE_C; ++k) {
    bsc[k] = (a.bs_scale && cok) ? a.bs_scale[cbase + k] : 1.f;
    bsh[k] = (a.bs_scale && cok) ? a.bs_shift[cbase + k] : 0.f;
  }

#ifdef HR_STAMP
  unsigned long long* stamp_buf = (unsigned long long*)a.stats + (size_t)blockIdx.x * 32;
  int stamp_i = 0;
#define STAMP() do { if (tid == 0 && stamp_i < 32) stamp_buf[stamp_i++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP() do { } while (0)
#endif
  STAMP();
  if (nstage > 0) load_stage(0);
  build_bntab();
  STAMP();
  // tile loop outside, K-chunk loop inside: the accumulators live in one tile iteration (declared,
  // zeroed, accumulated in place, stored). A flat stage loop with a conditional reset made hipcc shuffle
  // every accumulator AGPR<->VGPR around each MFMA (12-16 v_accvgpr moves per MFMA).
  int s = 0;
  for (int t = 0; t < ntile; ++t) {
    f32x4 acc[C::FC][C::FP];
#pragma unroll
    for (int fc = 0; fc < C::FC; ++fc)
#pragma unroll
      for (int fp = 0; fp < C::FP; ++fp) acc[fc][fp] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < nch; ++ch, ++s) {
      store_stage(s);
      STAMP();
      __syncthreads();
      STAMP();
      if (s + 1 < nstage) load_stage(s + 1);  // in flight while the MFMAs below run
      if constexpr (BS) { if (ch + 1 == nch) bs_prefetch(); }
      if constexpr (S2D) {
        // dx[2y+py][2x+px] = sum over the taps of the flipped kernel whose zero-stuffed source is a real dY
        // element: tap index 1 <-> parity 0 (dY[y]); 0 <-> parity 1 (dY[y]); 2 <-> parity 1 (dY[y+1]).
        // 9 tap-steps feed the 4 parity fragments (1 + 2 + 2 + 4): a quarter of the zero-stuffed MFMAs.
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          constexpr int KB = C::KSTEP * (int)sizeof(T);
          const int tr = tp / 3, ts = tp % 3;
          const int fpq = ((tr != 1) ? 2 : 0) + ((ts != 1) ? 1 : 0);
          const int tapb = ((tr == 2 ? 1 : 0) * C::HALO_W + (ts == 2 ? 1 : 0)) * C::PIXB;
#pragma unroll
          for (int kk = 0; kk < KM; ++kk) {
            const V16 bfv = *(const V16*)(xl + boff[0] + tapb + kk * KB);
#pragma unroll
            for (int fc = 0; fc < C::FC; ++fc) {
              const V16 afv = *(const V16*)(wl + aoff[fc] + tp * C::KC * (int)sizeof(T) + kk * KB);
              acc[fc][fpq] = mma16<T>(afv, bfv, acc[fc][fpq]);
            }
          }
        }
      } else {
#pragma unroll
        for (int tp = 0; tp < C::TAPS; ++tp) {
          const int tapb = ((tp / KS) * C::HALO_W + (tp % KS)) * C::PIXB;
  #pragma unroll
          for (int kk = 0; kk < KM; ++kk) {
            constexpr int KB = C::KSTEP * (int)sizeof(T);   // bytes of one fragment step
            V16 af[C::FC], bf[C::FP];
  #pragma unroll
            for (int fc = 0; fc < C::FC; ++fc)
              af[fc] = *(const V16*)(wl + aoff[fc] + tp * C::KC * (int)sizeof(T) + kk * KB);
  #pragma unroll
            for (int fp = 0; fp < C::FP; ++fp) bf[fp] = *(const V16*)(xl + boff[fp] + tapb + kk * KB);
  #pragma unroll
            for (int fc = 0; fc < C::FC; ++fc)
  #pragma unroll
              for (int fp = 0; fp < C::FP; ++fp) acc[fc][fp] = mma16<T>(af[fc], bf[fp], acc[fc][fp]);
          }
        }
      }
      STAMP();
      if (ch + 1 < nch) {
        __syncthreads();  // every wave is done reading this chunk's LDS image
        STAMP();
      }
    }
    {
    // ---- tile epilogue: bias, (accumulate), 4*FC contiguous couts per pixel, BN statistics ----
    const int n = __builtin_amdgcn_readfirstlane(ep_n), ty = __builtin_amdgcn_readfirstlane(ep_ty),
              tx = __builtin_amdgcn_readfirstlane(ep_tx);
    advance(ep_n, ep_ty, ep_tx);
#pragma unroll
    for (int fp = 0; fp < C::FP; ++fp) {
      int oy, ox;
      out_yx(fp, oy, ox);
      oy += ty * TH; ox += tx * TW;
      const bool pok = cok && oy < a.Ho && ox < a.Wo;
      float vals[C::LANE_C];
#pragma unroll
      for (int fc = 0; fc < C::FC; ++fc) {
        vals[fc * 4 + 0] = acc[fc][fp].x + bias[fc * 4 + 0];
        vals[fc * 4 + 1] = acc[fc][fp].y + bias[fc * 4 + 1];
        vals[fc * 4 + 2] = acc[fc][fp].z + bias[fc * 4 + 2];
        vals[fc * 4 + 3] = acc[fc][fp].w + bias[fc * 4 + 3];
      }
      if (pok) {
        char* dst = a.y + ((size_t)((n * a.Ho + oy) * a.Wo + ox) * a.Cout + cbase) * sizeof(T);
        if constexpr (!BS && !NO_STATS) {
#pragma unroll
          for (int k = 0; k < C::LANE_C; ++k) {
            s1[k] += vals[k];
            s2[k] += vals[k] * vals[k];
          }
        }
        // (accumulate) -> statistics -> store: the backward-statistics mode may store the MASKED gradient
        if (A_ACC) {
#pragma unroll
          for (int k0 = 0; k0 < C::LANE_C; k0 += VEC) {
            if constexpr (C::LANE_C >= VEC) {
              float old[VEC];
              v16_unpack<T>(*(const V16*)(dst + k0 * sizeof(T)), old);
#pragma unroll
              for (int j = 0; j < VEC; ++j) vals[k0 + j] += old[j];
            } else {
              const bf16x4 old = *(const bf16x4*)dst;
              vals[0] += (float)old.x; vals[1] += (float)old.y; vals[2] += (float)old.z; vals[3] += (float)old.w;
            }
          }
        }
        if constexpr (BS) {
          // vals now hold the finished gradient of this output element
          float yv[C::LANE_C], mv[C::LANE_C];
          lane_unpack(pre_y[fp], yv);
          if (a.bs_mask) {
            lane_unpack(pre_m[fp], mv);
#pragma unroll
            for (int k = 0; k < C::LANE_C; ++k) mv[k] = fmaf(mv[k], bsc[k], bsh[k]);
          } else {
#pragma unroll
            for (int k = 0; k < C::LANE_C; ++k) mv[k] = fmaf(yv[k], bsc[k], bsh[k]);
          }
          const bool masked = a.bs_mask || a.bs_scale;
#pragma unroll
          for (int k = 0; k < C::LANE_C; ++k) {
            const float dz = (!masked || mv[k] > 0.f) ? vals[k] : 0.f;
            s1[k] += dz;
            hr_fma_acc(s2[k], dz, yv[k]);
            if (a.bs_store_masked) vals[k] = dz;     // what is stored IS the next BatchNorm backward's dz
          }
        }
#pragma unroll
        for (int k0 = 0; k0 < C::LANE_C; k0 += VEC) {
          if constexpr (C::LANE_C >= VEC) {
            *(V16*)(dst + k0 * sizeof(T)) = v16_pack<T>(vals + k0);
          } else {
            // LANE_C == 4 with bf16: one 8-byte store
            const bf16x4 o = {(bf16_t)vals[0], (bf16_t)vals[1], (bf16_t)vals[2], (bf16_t)vals[3]};
            *(bf16x4*)dst = o;
          }
        }
      }
    }
    }
    STAMP();
    __syncthreads();  // the last chunk's LDS image is free again
    STAMP();
  }

#ifdef HR_STAMP
  if (false) {
#else
  if (!NO_STATS && a.stats) {
#endif
    float* sl = (float*)lds;  // [WP][2][BN] (the loop ended on a barrier: LDS is free)
#pragma unroll
    for (int k = 0; k < C::LANE_C; ++k) {
      s1[k] = wave_sum16(s1[k]);
      s2[k] = wave_sum16(s2[k]);
    }
    if (li == 0) {
#pragma unroll
      for (int k = 0; k < C::LANE_C; ++k) {
        const int cl = wc * C::CN + lg * C::LANE_C + k;
        sl[(wp * 2 + 0) * BN + cl] = s1[k];
        sl[(wp * 2 + 1) * BN + cl] = s2[k];
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, cl = tid % BN;
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < WP; ++q) s += sl[(q * 2 + which) * BN + cl];
      if (n0 + cl < a.Cout) {
        if (a.stats_atomic) atomicAdd(a.stats + ((size_t)(wg_p & (HR_BN_COPIES - 1)) * 2 + which) * a.Cout + n0 + cl, s);
        else a.stats[((size_t)wg_p * 2 + which) * a.Cout + n0 + cl] = s;
      }
    }
  }
}

template <typename T, int KS, int STRIDE, int TH, int TW, int BN, int WP, int WC, int KM>
__global__ __launch_bounds__(256) void conv_kernel(ConvArgs a) {
  conv_body<T, KS, STRIDE, TH, TW, BN, WP, WC, KM, CONV_GENERIC>(a);
}

// the backward-statistics variant keeps two workgroups per CU (the extra operands would otherwise push
// the register count past 256 / 2)
template <typename T, int KS, int STRIDE, int TH, int TW, int BN, int WP, int WC, int KM>
__global__ __launch_bounds__(256, 2) void conv_bs_kernel(ConvArgs a) {
  conv_body<T, KS, STRIDE, TH, TW, BN, WP, WC, KM, CONV_BS>(a);
}

template <typename T, int KS, int STRIDE, int TH, int TW, int BN, int WP, int WC, int KM>
__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvArgs a) {
  conv_body<T, KS, STRIDE, TH, TW, BN, WP, WC, KM, CONV_FWD>(a);
}

template <typename T, int KS, int STRIDE, int TH, int TW, int BN, int WP, int WC, int KM>
__global__ __launch_bounds__(256) void conv_fwdb_kernel(ConvArgs a) {
  conv_body<T, KS, STRIDE, TH, TW, BN, WP, WC, KM, CONV_FWDB>(a);
}

template <typename T, int KS, int STRIDE, int TH, int TW, int BN, int WP, int WC, int KM>
__global__ __launch_bounds__(256) void conv_fwds_kernel(ConvArgs a) {
  conv_body<T, KS, STRIDE, TH, TW, BN, WP, WC, KM, CONV_FWDS>(a);
}

template <typename T, int KS, int STRIDE, int TH, int TW, int BN, int WP, int WC, int KM>
__global__ __launch_bounds__(256) void conv_dg_kernel(ConvArgs a) {
  conv_body<T, KS, STRIDE, TH, TW, BN, WP, WC, KM, CONV_DG>(a);
}

// ---- tile configuration choice (host) --------------------------------------------------
struct TileChoice {
  int th, tw, bn, id, tpw, gx;
};

TileChoice choose_tile(int N, int Ho, int Wo, int Cout, int ks, int stride, bool bs = false, bool s2d = false) {
  // id: 0 = 16x16/BN32 (4x1 waves), 1 = 8x16/BN64 (2x2), 2 = 8x8/BN64 (2x2), 3 = 8x8/BN32 (2x2),
  //     4 = 8x16/BN128 (2x2; 1x1 convs with many output channels: 64 FLOP per staged byte)
  //     6 = 16x16/BN32 (4x1), 7 = 8x16/BN64 (2x2): four-parity input gradient of a stride-2 conv (S2D)
  TileChoice tc;
  if (s2d) tc = Cout >= 64 ? TileChoice{8, 16, 64, 7, 1, 0} : TileChoice{16, 16, 32, 6, 1, 0};
  else if (stride == 2) tc = Cout >= 64 ? TileChoice{8, 8, 64, 2, 1, 0} : TileChoice{8, 8, 32, 3, 1, 0};
  else if (Cout <= 32) tc = (Ho >= 16 && Wo >= 16) ? TileChoice{16, 16, 32, 0, 1, 0} : TileChoice{8, 8, 32, 3, 1, 0};
  // (not for backward-statistics launches: their epilogue operands do not fit 128 accumulators' registers)
  else if (ks == 1 && Cout >= 128 && Wo >= 16 && Ho >= 16 && !bs) tc = TileChoice{8, 16, 128, 4, 1, 0};  // GEMM-like
  else if (Wo >= 16 && Ho >= 16) {
    tc = TileChoice{8, 16, 64, 1, 1, 0};
    // maps whose width is far from a multiple of 16 (w48's 24x18 maps: 8x16 tiles cover 24x32 pixels, 8x8 tiles 24x24):
    // the 8x8 tile when it walks at most HRNET_SMALL_TILE_PCT % of the pixels the 8x16 tile would
    static const int pct = hr_knob("HRNET_SMALL_TILE_PCT", 80);
    const long long p16 = (long long)((Ho + 7) / 8 * 8) * ((Wo + 15) / 16 * 16), p8 = (long long)((Ho + 7) / 8 * 8) * ((Wo + 7) / 8 * 8);
    if (p8 * 100 <= p16 * pct) tc = TileChoice{8, 8, 64, 2, 1, 0};
  } else tc = TileChoice{8, 8, 64, 2, 1, 0};
  const int tiles = N * ((Ho + tc.th - 1) / tc.th) * ((Wo + tc.tw - 1) / tc.tw);
  int gy = (Cout + tc.bn - 1) / tc.bn;
  // few workgroups on 8x8-tiled maps: halve BN for twice the workgroups (latency hiding beats reuse;
  // measured the other way round for the 8x16 tile: 128->128 @16x16 runs 15.5 us with BN64, 20.3 with BN32)
  if (tc.id == 2 && tiles * gy < 512) {
    tc.bn = 32;
    tc.id = 3;
    if (tc.th != 8 || tc.tw != 8) { tc.th = 8; tc.tw = 8; }
    gy = (Cout + 31) / 32;
  }
  const int tiles2 = N * ((Ho + tc.th - 1) / tc.th) * ((Wo + tc.tw - 1) / tc.tw);
  // 2 resident workgroups per CU (register-limited) x 256 CUs: a grid of <= 512 workgroups runs as one
  // wave of workgroups with no tail; the rest of the tiles are walked by the same workgroups
  // (measured: 64->64 3x3 @64x64 40.1 us with 512 workgroups, 47.2 us with 683)
  static const int wg_target = hr_knob("HRNET_CONV_WGS", 512);   // (measurement override)
  int tpw = (tiles2 * gy + wg_target - 1) / wg_target;
  if (tpw < 1) tpw = 1;
  if (tpw > 16) tpw = 16;
  tc.tpw = tpw;
  tc.gx = (tiles2 + tpw - 1) / tpw;
  return tc;
}

// K depth per stage (fragment steps staged per barrier pair). A 1x1 conv has one tap, so it stages 4
// steps (2 when Cin <= 2 steps: no zero-padded MFMAs); 8x8/BN32 3x3 tiles have LDS room for two.
inline int conv_km(int dtype, int ks, int Cin, int tile_id) {
  const int kstep = dtype == HR_F32 ? 16 : 32;
  if (ks == 1) return Cin <= 2 * kstep ? 2 : 4;
  return tile_id == 3 ? 2 : 1;   // (S2D tiles 6, 7: 1)
}

#define LAUNCH_CONV(...)                                                                              \
  do {                                                                                                \
    if constexpr (MODE == CONV_BS) hipLaunchKernelGGL((conv_bs_kernel<__VA_ARGS__>), grid, dim3(256), 0, s, a);        \
    else if constexpr (MODE == CONV_FWD) hipLaunchKernelGGL((conv_fwd_kernel<__VA_ARGS__>), grid, dim3(256), 0, s, a); \
    else if constexpr (MODE == CONV_DG) hipLaunchKernelGGL((conv_dg_kernel<__VA_ARGS__>), grid, dim3(256), 0, s, a);   \
    else if constexpr (MODE == CONV_FWDB) hipLaunchKernelGGL((conv_fwdb_kernel<__VA_ARGS__>), grid, dim3(256), 0, s, a); \
    else if constexpr (MODE == CONV_FWDS) hipLaunchKernelGGL((conv_fwds_kernel<__VA_ARGS__>), grid, dim3(256), 0, s, a); \
    else hipLaunchKernelGGL((conv_kernel<__VA_ARGS__>), grid, dim3(256), 0, s, a);                    \
  } while (0)

template <int MODE, typename T, int KS, int STRIDE, int KM>
int launch_km(const ConvArgs& a, const TileChoice& tc, hipStream_t s) {
  const int gy_ = (a.Cout + tc.bn - 1) / tc.bn;
  dim3 grid((unsigned)(a.xcd_runs ? (tc.gx * gy_ + 7) / 8 * 8 : (gy_ == 1 ? tc.gx : (tc.gx + 7) / 8 * 8 * gy_)));
  if constexpr (STRIDE == 4) {
    if (tc.id == 6) LAUNCH_CONV(T, 3, 4, 16, 16, 32, 4, 1, 1);
    else LAUNCH_CONV(T, 3, 4, 8, 16, 64, 2, 2, 1);
    return hr_check_launch("conv2d");
  } else
  switch (tc.id) {
    case 0:   // (stride-2 convs only use the 8x8 tiles: their halo is (2*T+1)^2)
      if constexpr (STRIDE == 1)
        LAUNCH_CONV(T, KS, STRIDE, 16, 16, 32, 4, 1, KM);
      break;
    case 1:
      if constexpr (STRIDE == 1)
        LAUNCH_CONV(T, KS, STRIDE, 8, 16, 64, 2, 2, KM);
      break;
    case 2:
      LAUNCH_CONV(T, KS, STRIDE, 8, 8, 64, 2, 2, KM);
      break;
    case 4:
      if constexpr (KS == 1)
        LAUNCH_CONV(T, 1, 1, 8, 16, 128, 2, 2, KM);
      break;
    default:
      LAUNCH_CONV(T, KS, STRIDE, 8, 8, 32, 2, 2, KM);
      break;
  }
  return hr_check_launch("conv2d");
}

template <int MODE, typename T, int KS, int STRIDE>
int launch_cfg(const ConvArgs& a, const TileChoice& tc, int N, hipStream_t s) {
  const int km = conv_km(TT<T>::ID, KS, a.Cin, tc.id);
  if constexpr (STRIDE == 4) {
    return launch_km<MODE, T, 3, 4, 1>(a, tc, s);
  } else
  if constexpr (KS == 1) {
    return km == 2 ? launch_km<MODE, T, 1, 1, 2>(a, tc, s) : launch_km<MODE, T, 1, 1, 4>(a, tc, s);
  } else {
    return km == 2 ? launch_km<MODE, T, KS, STRIDE, 2>(a, tc, s) : launch_km<MODE, T, KS, STRIDE, 1>(a, tc, s);
  }
}

template <int MODE, typename T>
int launch_t(const ConvArgs& a, const TileChoice& tc, int N, int ks, int stride, hipStream_t s) {
  if (stride == 4) {   // four-parity stride-2 input gradient: only the input-gradient bodies carry it
    if constexpr (MODE == CONV_BS || MODE == CONV_DG) return launch_cfg<MODE, T, 3, 4>(a, tc, N, s);
    else return HR_E_BADARG;
  }
  if (ks == 1) return launch_cfg<MODE, T, 1, 1>(a, tc, N, s);
  if (stride == 1) return launch_cfg<MODE, T, 3, 1>(a, tc, N, s);
  return launch_cfg<MODE, T, 3, 2>(a, tc, N, s);
}


}  // namespace

// one translation unit per mode (parallel builds): defined in conv.hip / conv_bs.hip / conv_fwd.hip / conv_dg.hip
struct ConvLaunch {
  ConvArgs a;
  TileChoice tc;
  int dtype, N, ks, stride;
};
int hr_conv_launch_generic(const ConvLaunch& l, hipStream_t s);
int hr_conv_launch_bs(const ConvLaunch& l, hipStream_t s);
int hr_conv_launch_fwd(const ConvLaunch& l, hipStream_t s);
int hr_conv_launch_dg(const ConvLaunch& l, hipStream_t s);
int hr_conv_launch_fwds(const ConvLaunch& l, hipStream_t s);
int hr_conv_launch_fwdb(const ConvLaunch& l, hipStream_t s);

#define HR_DEFINE_CONV_LAUNCH(NAME, MODE)                                              \
  int NAME(const ConvLaunch& l, hipStream_t s) {                                       \
    if (l.dtype == HR_F32) return launch_t<MODE, float>(l.a, l.tc, l.N, l.ks, l.stride, s); \
    return launch_t<MODE, bf16_t>(l.a, l.tc, l.N, l.ks, l.stride, s);                  \
  }
