// 3x3 stride-1 convolution (bf16) as an LDS-ring pipeline for gfx950: the branch convolutions of the
// HighResolutionModules (BasicBlock.conv1/conv2, pose_hrnet.py:41-57), which conv_body.h runs latency-bound
// (one register-staged tile in flight per workgroup).
//
//   D[cout][pixel] += sum_{tap, ci} Wp[cout][tap][ci] * a[pixel + tap][ci],   a = relu?(scale*x + shift)
//
// A stage = (pixel tile, 32 input channels): the halo tile [(TH+2)*(TW+2)][32 ch] (64 bytes per pixel) and - when
// the weights of the workgroup's output-channel block do not stay resident - the weight slice [9][NB][32 ch].
// Stages are fetched by DIRECT global->LDS loads (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction, no
// staging registers) into a ring of R slots, R-1 stages ahead of the matrix work; every vector-memory
// instruction of a wave is counted, and the wait for a stage is the exact `s_waitcnt vmcnt(n)` that leaves the
// younger stages (and the epilogue's stores) in flight. Lanes outside the image read out of the buffer's range:
// the hardware writes zeros, which IS the zero padding of a raw input; an input read through a BatchNorm
// (+ReLU) is transformed IN PLACE in LDS by the lane that fetched it (own data: its own vmcnt is the only
// ordering needed), one stage ahead of the MFMAs, zeros outside the image written after the transform as the
// reference pads the activated tensor. One barrier per stage.
//
// LDS image of a stage: pixel-major, 4 x 16-byte channel chunks per pixel, chunk c of pixel P stored at chunk
// position c ^ ((P >> 1) & 2): the 16 pixels x 4 chunks one MFMA B-fragment read (ds_read_b128) touches fall on
// 16 different 16-byte bank slots for every lane group and every tap shift. The swizzle is applied on the SOURCE
// address (the LDS destination of a direct load is lane-linear). Weight rows [tap][cout] are stored the same
// way, in MFMA order, so a lane ends up with 4*FC contiguous output channels of one pixel (16-byte NHWC stores).
//
// BatchNorm batch statistics of the output (sum, sum of squares) come from the f32 accumulators, reduced once
// per workgroup and added to sums[8][2][Cout] (float atomics), as in conv_body.h. The input BatchNorm given as
// batch sums is turned into scale/shift with hr_bn_from_sums's arithmetic from a copy of the sums fetched by
// the same direct loads (no register-destination load shares the queue with the ring).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_ring.h"

namespace {

struct RingArgs {
  const char* x;         // [N,H,W,Cin] bf16
  const char* w;         // packed [Cout][9][Cin] bf16 (hrnet_pack_weights mode 0; mode 1 for an input gradient)
  char* y;               // [N,H,W,Cout] bf16
  const float* in_sums;  // [8][2][Cin] batch sums of the input's BatchNorm, or NULL
  const float* in_gb;    // gamma[Cin] | beta[Cin] (contiguous), with in_sums
  const float* in_scale; // or: precomputed scale / shift arrays
  const float* in_shift;
  float* stats;          // forward: [8][2][Cout] batch sums (atomic) or NULL; backward statistics: rows [gx][2][Cout]
  // backward-statistics launches (an input gradient whose output is the gradient of a BatchNorm'ed activation):
  // rows get (sum dz, sum dz*bs_y), dz = v * [m > 0], m = bs_mask (or bs_y) mapped through bs_scale/bs_shift
  const char* bs_y;
  const char* bs_mask;
  const float* bs_scale;
  const float* bs_shift;
  // residual-sum launches (hrnet_conv2d_sum, template X2): the conv's input is a = relu(bn(x) + x2), formed while the
  // stage is transformed in LDS and written to `side` once per pixel (by the first output-channel block)
  const char* x2;        // [N,H,W,Cin] identity term
  char* side;            // [N,H,W,Cin] the sum, for the next residual add and for backward
  float in_inv_count, in_eps;
  int N, H, W, Cin, Cout;
  int tiles_y, tiles_x, total_tiles, tpw, gx, gy;
  int in_relu, accumulate, bs_store_masked;
  unsigned x_bytes, y_bytes, w_bytes;
#ifdef HR_RING_STAMP
  unsigned long long* stamp;   // measurement build only: 32 s_memrealtime stamps per workgroup
#endif
};

// s_waitcnt vmcnt(n) for a wave-uniform run-time n: the largest immediate <= n (waiting for more is safe)
__device__ __forceinline__ void wait_vmcnt(int n) {
#define WV(K) case K: asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory"); break;
  switch (n < 0 ? 0 : (n > 47 ? 47 : n)) {
    WV(0) WV(1) WV(2) WV(3) WV(4) WV(5) WV(6) WV(7) WV(8) WV(9) WV(10) WV(11) WV(12) WV(13) WV(14) WV(15)
    WV(16) WV(17) WV(18) WV(19) WV(20) WV(21) WV(22) WV(23) WV(24) WV(25) WV(26) WV(27) WV(28) WV(29) WV(30) WV(31)
    WV(32) WV(33) WV(34) WV(35) WV(36) WV(37) WV(38) WV(39) WV(40) WV(41) WV(42) WV(43) WV(44) WV(45) WV(46) WV(47)
  }
#undef WV
}

__device__ __forceinline__ void lds_barrier() {
  // the LDS writes of this wave are done, then the workgroup meets; no vmcnt wait: direct loads stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned OOB = 0x80000000u;   // beyond every buffer's num_records: reads return 0, stores are dropped

// A tile = TI images x TH x TW output pixels (TI > 1: whole images, TH x TW = the map), NB output channels per
// workgroup, NW waves as WPX (pixel groups) x NW/WPX (channel groups), R ring slots, WCH > 0: the weights of all
// WCH input-channel chunks stay resident (else one weight chunk travels with every stage).
// Halo rows are stored with a pitch of TW + 2 pixels (TW = 16) or 16 pixels (TW = 8: the two rows a pixel fragment
// spans then sit on complementary bank slots, and a 1 KiB piece is exactly one halo row).
template <int TH, int TW, int TI, int NB, int NW, int WPX, int R, int WCH>
struct RingCfg {
  static constexpr int NT = NW * 64;
  static constexpr int HH = TH + 2, HW = TW + 2, HWP = TW == 8 ? 16 : HW;
  static constexpr int IPX = HH * HWP, HPX = TI * IPX;   // pixel slots per image / per tile
  static constexpr int NPX = (HPX + 15) / 16;            // 1 KiB pieces of a halo chunk
  static constexpr int XSB = HPX * 64;                   // bytes of one ring slot's input image
  static constexpr int NPWC = NB * 9 / 16;               // pieces of one weight chunk
  static constexpr int WCB = NB * 9 * 64;
  static constexpr bool WRES = WCH > 0;
  static constexpr int KPX = (NPX + NW - 1) / NW;        // pieces per wave
  static constexpr int KPW = (NPWC + NW - 1) / NW;
  static constexpr int TPX = TH * TW;                    // output pixels per image of the tile
  static constexpr int NPF = TI * TPX / 16, NCF = NB / 16, WCO = NW / WPX;
  static constexpr int FP = NPF / WPX, FC = NCF / WCO, CN = FC * 16, LANE_C = 4 * FC;
  static constexpr int CMAX = WRES ? WCH * 32 : HR_RING_MAXC;
  static constexpr int XOFF = 0, WOFF = R * XSB, TOFF = WOFF + (WRES ? WCH : R) * WCB;
  static constexpr int LDSB = TOFF + 2 * CMAX * 4;
  static_assert(TW == 16 || TW == 8, "a pixel fragment is one row of 16 or two rows of 8");
  static_assert(NPF % WPX == 0 && NCF % WCO == 0 && WPX * WCO == NW, "wave grid");
  static_assert(NB * 9 % 16 == 0 && XSB % 256 == 0, "whole weight pieces; bank-aligned slots");
  static_assert(R >= 2 && R <= 4, "ring slots");
  static_assert(LDSB <= 160 * 1024, "LDS");
  static_assert(LANE_C == 8 || LANE_C == 16, "a lane stores 16-byte vectors");
};

template <int TH, int TW, int TI, int NB, int NW, int WPX, int R, int WCH, bool BS, bool X2 = false>
__global__ __launch_bounds__(NW * 64, 2) void conv_ring_kernel(RingArgs a) {
  using C = RingCfg<TH, TW, TI, NB, NW, WPX, R, WCH>;
  static_assert(!X2 || (R == 3 && TI == 1 && !BS), "the residual-sum form: three slots, one image per tile, forward");
  __shared__ __attribute__((aligned(1024))) char lds[C::LDSB];
  char* xl = lds + C::XOFF;
  char* wl = lds + C::WOFF;
  float* bntab = (float*)(lds + C::TOFF);     // [scale Cin][shift Cin]
  const unsigned lds_x0 = (unsigned)(uintptr_t)(LDS_AS char*)xl;   // LDS byte address of the ring

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int wpx = wv % WPX, wco = wv / WPX;

  // workgroup -> (pixel walk, output-channel block): ids round-robin over the 8 XCDs, so logical index
  // L = (id % 8) * (grid / 8) + id / 8 gives every XCD one contiguous run of walks - neighbouring tiles (shared
  // halo rows) and the channel blocks of one walk (same input) meet in one L2
  const int G8 = gridDim.x >> 3;
  const int L = (blockIdx.x & 7) * G8 + (blockIdx.x >> 3);
  if (L >= a.gx * a.gy) return;
  const int wg_p = L / a.gy, n0 = (L - wg_p * a.gy) * NB;

#ifdef HR_RING_STAMP
  int stamp_i = 0;
#define RSTAMP() do { if (a.stamp && tid == 0 && stamp_i < 32) a.stamp[(size_t)blockIdx.x * 32 + stamp_i++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RSTAMP() do { } while (0)
#endif
  RSTAMP();
  const int nch = a.Cin >> 5;
  const int tile0 = wg_p * a.tpw;
  const int ntile = min(a.tpw, a.total_tiles - tile0);
  const int S = ntile * nch;
  const bool from_sums = BS ? false : a.in_sums != nullptr;
  const bool has_aff = BS ? false : (from_sums || a.in_scale != nullptr);
  const bool in_relu = BS ? false : a.in_relu != 0;
  const bool xf = has_aff || in_relu;       // the input needs the in-place transform

  const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
  const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.w_bytes, 0x00020000);
  const auto ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (int)a.y_bytes, 0x00020000);

  // ---- per-lane constants of the direct loads: lane j of a piece = 16-byte slot j: pixel slot j>>2 of the piece,
  // chunk position j&3, which holds channel chunk cj (fixed per lane: pieces start at multiples of 16 slots)
  const int pj = lane >> 2, cj = (lane & 3) ^ ((pj >> 1) & 2);
  const int pixB = a.Cin * 2, rowB = a.W * pixB, imgB = a.H * rowB;
  // packed per piece: bit 31 = not a halo pixel of the tile; [29:20] image of the tile, [19:10] halo row, [9:0] column
  int hyx[C::KPX], goff[C::KPX];
#pragma unroll
  for (int k = 0; k < C::KPX; ++k) {
    const int P = (wv + k * NW) * 16 + pj;
    const int img = P / C::IPX, r = P - img * C::IPX;
    const int hy = r / C::HWP, hx = r - hy * C::HWP;
    hyx[k] = (P < C::HPX && hx < C::HW) ? (img << 20) | (hy << 10) | hx : (int)0x80000000;
    goff[k] = img * imgB + hy * rowB + hx * pixB + cj * 16;
  }
  // weight pieces: row (tap, co) of the chunk image; LDS column co holds the channel the MFMA row order needs
  int woff[C::KPW];
#pragma unroll
  for (int k = 0; k < C::KPW; ++k) {
    const int row = (wv + k * NW) * 16 + pj;
    const int tap = row / NB, col = row - tap * NB;
    const int q = col % C::CN;
    const int co = n0 + (col / C::CN) * C::CN + ((q & 15) >> 2) * C::LANE_C + (q >> 4) * 4 + (q & 3);
    woff[k] = co < a.Cout ? ((co * 9 + tap) * a.Cin) * 2 + cj * 16 : (int)OOB;
  }

  int cnt = 0;                 // vector-memory instructions this wave has issued
  int mark[R];                 // cnt right after stage (slot) was issued
#pragma unroll
  for (int r = 0; r < R; ++r) mark[r] = 0;
  auto set_mark = [&](int slot, int v) {
#pragma unroll
    for (int r = 0; r < R; ++r) if (slot == r) mark[r] = v;
  };
  auto get_mark = [&](int slot) {
    int v = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) if (slot == r) v = mark[r];
    return v;
  };

  // tile cursors: issue (loads), transform, compute; each walks (tile, chunk) in the same order. n counts groups
  // of TI images.
  struct Cur { int n, ty, tx, ch; };
  auto cur_init = [&](Cur& c) {
    int bq = tile0;
    c.tx = bq % a.tiles_x; bq /= a.tiles_x;
    c.ty = bq % a.tiles_y; c.n = bq / a.tiles_y; c.ch = 0;
  };
  auto cur_next = [&](Cur& c) {
    if (++c.ch == nch) {
      c.ch = 0;
      if (++c.tx == a.tiles_x) { c.tx = 0; if (++c.ty == a.tiles_y) { c.ty = 0; ++c.n; } }
    }
  };
  Cur ci, ct, cc;
  cur_init(ci); cur_init(ct); cur_init(cc);

  // is halo pixel (packed h) of the tile at (n, iy0, ix0) inside the tensor?
  auto inside = [&](int h, int n, int iy0, int ix0) {
    const int gy = iy0 + ((h >> 10) & 0x3ff), gx = ix0 + (h & 0x3ff);
    return h >= 0 && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W && n * TI + (h >> 20) < a.N;
  };

  auto issue_w = [&](int chunk, int slot_or_chunk) {     // one weight chunk image
#pragma unroll
    for (int k = 0; k < C::KPW; ++k) {
      const int r = wv + k * NW;
      if (r < C::NPWC) {
        const unsigned vo = woff[k] == (int)OOB ? OOB : (unsigned)(woff[k] + chunk * 64);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(wl + slot_or_chunk * C::WCB + r * 1024), 16, vo, 0, 0, 0);
        ++cnt;
      }
    }
  };
  // X2: the identity term's 16 bytes of every piece this lane fetches, in REGISTERS from issue to transform (stage s
  // in set s & 1: two stages are in flight between the two). Loaded by inline-asm global loads the compiler does not
  // see (a load it tracked would get an `s_waitcnt vmcnt(0)` in front of its use and drain the ring); they are counted
  // in `cnt` like every other vector-memory instruction, the stage's counted wait covers them, and an empty asm with
  // the registers as "+v" operands behind that wait keeps their readers behind it.
  V16 x2r[X2 ? 2 : 1][X2 ? C::KPX : 1];
  auto issue_stage = [&](int s, auto par) {
    constexpr int P = decltype(par)::value;
    const int slot = s % R;
    const int n = __builtin_amdgcn_readfirstlane(ci.n), ty = __builtin_amdgcn_readfirstlane(ci.ty),
              tx = __builtin_amdgcn_readfirstlane(ci.tx), ch = __builtin_amdgcn_readfirstlane(ci.ch);
    const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
    const int sbase = ((n * TI * a.H + iy0) * a.W + ix0) * pixB + ch * 64;     // may be negative: only valid lanes use it
#pragma unroll
    for (int k = 0; k < C::KPX; ++k) {
      const int q = wv + k * NW;
      if (q < C::NPX) {
        const bool ok = inside(hyx[k], n, iy0, ix0);
        const unsigned vo = ok ? (unsigned)(sbase + goff[k]) : OOB;
        // (slots beyond the tile's halo stay out of the next ring slot; padding slots of a row get zeros)
        if ((q * 16 + pj) < C::HPX)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(xl + slot * C::XSB + q * 1024), 16, vo, 0, 0, 0);
        ++cnt;
        if constexpr (X2) {
          // (every lane loads: a lane outside the image reads the tensor's first bytes and its value is never used)
          const char* p2 = a.x2 + (ok ? (long long)(sbase + goff[k]) : 0ll);
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x2r[P][k]) : "v"(p2) : "memory");
          ++cnt;
        }
      }
    }
    if constexpr (!C::WRES) issue_w(ch, slot);
    set_mark(slot, cnt);
    cur_next(ci);
  };

  // backward-statistics coefficients of this lane's channels (plain loads, issued before any direct load)
  const int cbase = n0 + wco * C::CN + lg * C::LANE_C;
  const bool cok = cbase < a.Cout;
  float bsc[BS ? C::LANE_C : 1], bsh[BS ? C::LANE_C : 1];
  if constexpr (BS) {
#pragma unroll
    for (int k = 0; k < C::LANE_C; ++k) {
      bsc[k] = (a.bs_scale && cok) ? a.bs_scale[cbase + k] : 1.f;
      bsh[k] = (a.bs_scale && cok) ? a.bs_shift[cbase + k] : 0.f;
    }
  }

  // ---- prologue: the BatchNorm inputs (into the last ring slot), resident weights, the first R-1 stages ----
  char* stg = xl + (R - 1) * C::XSB;                    // staging: [8][2][Cin] sums | gamma | beta  (or scale | shift)
  int tab_mark = 0;
  if (has_aff) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(from_sums ? a.in_sums : a.in_scale), 0,
                                                      (from_sums ? 16 : 1) * a.Cin * 4, 0x00020000);
    const auto rg = __builtin_amdgcn_make_buffer_rsrc((void*)(from_sums ? a.in_gb : a.in_shift), 0,
                                                      (from_sums ? 2 : 1) * a.Cin * 4, 0x00020000);
    if (wv == 0) {
      const int np = from_sums ? (a.Cin * 64 + 1023) / 1024 : (a.Cin * 4 + 1023) / 1024;
      for (int q = 0; q < np; ++q) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(stg + q * 1024), 16, (unsigned)(q * 1024 + lane * 16), 0, 0, 0);
        ++cnt;
      }
      const int go = from_sums ? a.Cin * 64 : a.Cin * 4;
      const int ng = ((from_sums ? 2 : 1) * a.Cin * 4 + 1023) / 1024;
      for (int q = 0; q < ng; ++q) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_ptr_t)(stg + ((go + 1023) & ~1023) + q * 1024), 16,
                                                 (unsigned)(q * 1024 + lane * 16), 0, 0, 0);
        ++cnt;
      }
    }
    tab_mark = cnt;
  }
  if constexpr (C::WRES) {
    for (int ch = 0; ch < nch; ++ch) issue_w(ch, ch);
  }
  const int npro = S < R - 1 ? S : R - 1;
  if (npro > 0) issue_stage(0, std::integral_constant<int, 0>{});
  if (npro > 1) issue_stage(1, std::integral_constant<int, 1>{});
  if constexpr (R > 3) {
    if (npro > 2) issue_stage(2, std::integral_constant<int, 0>{});
  }
  RSTAMP();

  if (has_aff) {
    wait_vmcnt(cnt - tab_mark);
    lds_barrier();
    const int go = ((from_sums ? a.Cin * 64 : a.Cin * 4) + 1023) & ~1023;
    const float* sv = (const float*)stg;
    const float* gv = (const float*)(stg + go);
    for (int c = tid; c < a.Cin; c += C::NT) {
      float sc_, sh_;
      if (from_sums) {
        float m_, r_, v_;
        hr_bn_from_sums(sv, a.Cin, c, a.in_inv_count, a.in_eps, gv[c], gv[a.Cin + c], sc_, sh_, m_, r_, v_);
      } else {
        sc_ = sv[c]; sh_ = gv[c];
      }
      bntab[c] = sc_;
      bntab[a.Cin + c] = sh_;
    }
    lds_barrier();         // the table is readable; the staging slot is free for the ring
  }
  RSTAMP();

  // ---- per-lane MFMA operand offsets: halo slot of this lane's pixel of fragment fp, shifted by tap (dy, dx) ----
  int boff[C::FP][9];
#pragma unroll
  for (int fp = 0; fp < C::FP; ++fp) {
    const int p = (wpx * C::FP + fp) * 16 + li;
    const int img = p / C::TPX, q = p - img * C::TPX;
    const int oy = q / TW, ox = q - oy * TW;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      const int P = img * C::IPX + (oy + tp / 3) * C::HWP + ox + tp % 3;
      boff[fp][tp] = P * 64 + ((lg ^ ((P >> 1) & 2)) << 4);
    }
  }
  int aoff[C::FC];
#pragma unroll
  for (int fc = 0; fc < C::FC; ++fc) {
    const int col = wco * C::CN + fc * 16 + li;
    aoff[fc] = col * 64 + ((lg ^ ((col >> 1) & 2)) << 4);
  }

  float sc[8], sh[8];
  auto load_coef = [&](int ch) {
    if (has_aff) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sc[j] = bntab[ch * 32 + cj * 8 + j];
        sh[j] = bntab[a.Cin + ch * 32 + cj * 8 + j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) { sc[j] = 1.f; sh[j] = 0.f; }
    }
  };
  if (nch == 1) load_coef(0);

  // in-place transform of the stage the transform cursor points at (own pieces only)
  const auto rsd = __builtin_amdgcn_make_buffer_rsrc((void*)(X2 ? a.side : a.y), 0, X2 ? (int)a.x_bytes : 0, 0x00020000);
  const bool side_wg = X2 && n0 == 0;      // the first output-channel block of a walk writes the sum out
  auto transform = [&](int s, auto par) {
    constexpr int P = decltype(par)::value;
    const int slot = s % R;
    const int n = __builtin_amdgcn_readfirstlane(ct.n), ty = __builtin_amdgcn_readfirstlane(ct.ty),
              tx = __builtin_amdgcn_readfirstlane(ct.tx), ch = __builtin_amdgcn_readfirstlane(ct.ch);
    if constexpr (X2) {
      // (behind the stage's counted wait: the identity term's registers become readable here, not earlier)
#pragma unroll
      for (int k = 0; k < C::KPX; ++k) asm volatile("" : "+v"(x2r[P][k]));
    }
    if (xf) {
      if (nch != 1) load_coef(ch);
      const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
      const int sbase = ((n * TI * a.H + iy0) * a.W + ix0) * pixB + ch * 64;
#pragma unroll
      for (int k = 0; k < C::KPX; ++k) {
        const int q = wv + k * NW;
        if (q < C::NPX) {
          const bool ok = inside(hyx[k], n, iy0, ix0);
          if constexpr (X2) {
            // a = relu(scale * x + shift + x2), rounded once: what goes to LDS IS what goes to `side`
            // (conv_body.h CONV_FWDS: the same arithmetic in the same order)
            V16 v = v16_zero();
            if ((q * 16 + pj) < C::HPX) {
              const unsigned la = lds_x0 + (unsigned)(slot * C::XSB + q * 1024 + lane * 16);
              asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(la) : "memory");
              float f[8], g2[8];
              v16_unpack<bf16_t>(v, f);
              v16_unpack<bf16_t>(x2r[P][k], g2);
#pragma unroll
              for (int j = 0; j < 8; ++j) { f[j] = fmaf(f[j], sc[j], sh[j]) + g2[j]; f[j] = f[j] > 0.f ? f[j] : 0.f; }
              v = ok ? v16_pack<bf16_t>(f) : v16_zero();
              asm volatile("ds_write_b128 %0, %1" :: "v"(la), "v"(v) : "memory");
            }
            if (side_wg) {
              // centre (non-halo) pixels of the tile, once per pixel; every lane issues the store (the others beyond
              // the descriptor's range), so that the instruction count below is exact
              const int hy = (hyx[k] >> 10) & 0x3ff, hx = hyx[k] & 0x3ff;
              const bool centre = ok && (q * 16 + pj) < C::HPX && hy >= 1 && hy <= TH && hx >= 1 && hx <= TW;
              __builtin_amdgcn_raw_buffer_store_b128(v, rsd, centre ? (unsigned)(sbase + goff[k]) : OOB, 0, 0);
              ++cnt;
            }
            continue;
          }
          if ((q * 16 + pj) < C::HPX) {
            // (LDS accesses in inline asm: hipcc puts `s_waitcnt vmcnt(0)` in front of a ds_read of the address a
            // direct load wrote - it would drain the younger stages; the data is this lane's own and the counted
            // wait above covers it)
            const unsigned la = lds_x0 + (unsigned)(slot * C::XSB + q * 1024 + lane * 16);
            V16 v;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(la) : "memory");
            float f[8];
            v16_unpack<bf16_t>(v, f);
            if (in_relu) {
#pragma unroll
              for (int j = 0; j < 8; ++j) { f[j] = fmaf(f[j], sc[j], sh[j]); f[j] = f[j] > 0.f ? f[j] : 0.f; }
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) f[j] = fmaf(f[j], sc[j], sh[j]);
            }
            v = ok ? v16_pack<bf16_t>(f) : v16_zero();
            asm volatile("ds_write_b128 %0, %1" :: "v"(la), "v"(v) : "memory");
          }
        }
      }
    }
    cur_next(ct);
  };

  float s1[C::LANE_C], s2[C::LANE_C];
#pragma unroll
  for (int k = 0; k < C::LANE_C; ++k) s1[k] = s2[k] = 0.f;
  constexpr int LV = C::LANE_C / 8;           // 16-byte vectors of a lane's channels

  if (S > 0) {
    wait_vmcnt(cnt - get_mark(0));
    RSTAMP();
    transform(0, std::integral_constant<int, 0>{});
  }
  lds_barrier();
  RSTAMP();

  f32x4 acc[C::FC][C::FP];
  // one iteration; `par` = parity of i (compile time: it names the register set of the identity term, X2)
  auto iteration = [&](int i, auto par) {
    constexpr int P = decltype(par)::value;
    // (R = 3: stage i + 2 has the parity of i; R = 2 / 4 are never instantiated with X2)
    if (i + R - 1 < S) issue_stage(i + R - 1, std::integral_constant<int, (P + R - 1) & 1>{});
    if constexpr (R > 2) {
      // the stage after this one is transformed before this one's matrix work (its loads were issued two iterations ago)
      if (i + 1 < S) {
        wait_vmcnt(cnt - get_mark((i + 1) % R));
        RSTAMP();
        transform(i + 1, std::integral_constant<int, 1 - P>{});
      }
    }
    RSTAMP();
    const int ch = __builtin_amdgcn_readfirstlane(cc.ch);
    const int n = __builtin_amdgcn_readfirstlane(cc.n), ty = __builtin_amdgcn_readfirstlane(cc.ty),
              tx = __builtin_amdgcn_readfirstlane(cc.tx);
    if (ch == 0) {
#pragma unroll
      for (int fc = 0; fc < C::FC; ++fc)
#pragma unroll
        for (int fp = 0; fp < C::FP; ++fp) acc[fc][fp] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // epilogue operands of a backward-statistics / accumulating launch: plain loads, issued with the tile's last
    // stage (the wide layers these launches serve have one tile per workgroup: nothing is left in the ring)
    V16 pre_y[BS ? C::FP : 1][LV], pre_m[BS ? C::FP : 1][LV], pre_o[BS ? C::FP : 1][LV];
    if constexpr (BS) {
      if (ch == nch - 1) {
        const auto rby = __builtin_amdgcn_make_buffer_rsrc((void*)a.bs_y, 0, (int)a.y_bytes, 0x00020000);
        const auto rbm = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bs_mask ? a.bs_mask : a.bs_y), 0, (int)a.y_bytes, 0x00020000);
#pragma unroll
        for (int fp = 0; fp < C::FP; ++fp) {
          const int p = (wpx * C::FP + fp) * 16 + li;
          const int img = p / C::TPX, q = p - img * C::TPX;
          const int oy = ty * TH + q / TW, ox = tx * TW + q % TW, ni = n * TI + img;
          const bool pok = cok && oy < a.H && ox < a.W && ni < a.N;
          const unsigned vo = pok ? (unsigned)((((ni * a.H + oy) * a.W + ox) * a.Cout + cbase) * 2) : OOB;
#pragma unroll
          for (int v = 0; v < LV; ++v) {
            pre_y[fp][v] = a.bs_y ? __builtin_amdgcn_raw_buffer_load_b128(rby, pok ? vo + v * 16 : OOB, 0, 0) : v16_zero();
            pre_m[fp][v] = a.bs_mask ? __builtin_amdgcn_raw_buffer_load_b128(rbm, pok ? vo + v * 16 : OOB, 0, 0) : v16_zero();
            pre_o[fp][v] = a.accumulate ? __builtin_amdgcn_raw_buffer_load_b128(ry, pok ? vo + v * 16 : OOB, 0, 0) : v16_zero();
          }
        }
        cnt += C::FP * LV * ((a.bs_y ? 1 : 0) + (a.bs_mask ? 1 : 0) + (a.accumulate ? 1 : 0));
      }
    }
    const char* xs = xl + (i % R) * C::XSB;
    const char* ws = wl + (C::WRES ? ch : i % R) * C::WCB;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      V16 af[C::FC], bf[C::FP];
#pragma unroll
      for (int fc = 0; fc < C::FC; ++fc) af[fc] = *(const V16*)(ws + aoff[fc] + tp * NB * 64);
#pragma unroll
      for (int fp = 0; fp < C::FP; ++fp) bf[fp] = *(const V16*)(xs + boff[fp][tp]);
#pragma unroll
      for (int fc = 0; fc < C::FC; ++fc)
#pragma unroll
        for (int fp = 0; fp < C::FP; ++fp) acc[fc][fp] = mma16<bf16_t>(af[fc], bf[fp], acc[fc][fp]);
    }
    RSTAMP();
    if constexpr (R == 2) {
      // two slots: the next stage lands while this one's matrix work runs, and is transformed after it
      if (i + 1 < S) {
        wait_vmcnt(cnt - get_mark((i + 1) % R));
        transform(i + 1, std::integral_constant<int, 1 - P>{});
      }
    }
    if (ch == nch - 1) {
      // ---- tile epilogue: 4*FC contiguous output channels per pixel, statistics ----
#pragma unroll
      for (int fp = 0; fp < C::FP; ++fp) {
        const int p = (wpx * C::FP + fp) * 16 + li;
        const int img = p / C::TPX, q = p - img * C::TPX;
        const int oy = ty * TH + q / TW, ox = tx * TW + q % TW, ni = n * TI + img;
        const bool pok = cok && oy < a.H && ox < a.W && ni < a.N;
        float vals[C::LANE_C];
#pragma unroll
        for (int fc = 0; fc < C::FC; ++fc) {
          vals[fc * 4 + 0] = acc[fc][fp].x; vals[fc * 4 + 1] = acc[fc][fp].y;
          vals[fc * 4 + 2] = acc[fc][fp].z; vals[fc * 4 + 3] = acc[fc][fp].w;
        }
        if constexpr (!BS) {
          if (pok) {
#pragma unroll
            for (int k = 0; k < C::LANE_C; ++k) {
              s1[k] += vals[k];
              s2[k] += vals[k] * vals[k];
            }
          }
        } else {
          if (a.accumulate) {
            float old[C::LANE_C];
#pragma unroll
            for (int v = 0; v < LV; ++v) v16_unpack<bf16_t>(pre_o[fp][v], old + v * 8);
#pragma unroll
            for (int k = 0; k < C::LANE_C; ++k) vals[k] += old[k];
          }
          if (a.bs_y && pok) {
            // vals hold the finished gradient of this output element
            float yv[C::LANE_C], mv[C::LANE_C];
#pragma unroll
            for (int v = 0; v < LV; ++v) v16_unpack<bf16_t>(pre_y[fp][v], yv + v * 8);
            if (a.bs_mask) {
#pragma unroll
              for (int v = 0; v < LV; ++v) v16_unpack<bf16_t>(pre_m[fp][v], mv + v * 8);
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
        }
        const unsigned vo = pok ? (unsigned)((((ni * a.H + oy) * a.W + ox) * a.Cout + cbase) * 2) : OOB;
#pragma unroll
        for (int k0 = 0; k0 < C::LANE_C; k0 += 8) {
          __builtin_amdgcn_raw_buffer_store_b128(v16_pack<bf16_t>(vals + k0), ry, pok ? vo + k0 * 2 : OOB, 0, 0);
          ++cnt;
        }
      }
    }
    cur_next(cc);
    lds_barrier();     // slot i % R is free; the transformed image of stage i + 1 is visible
    RSTAMP();
  };
  if constexpr (X2) {
    for (int i = 0; i < S; i += 2) {
      iteration(i, std::integral_constant<int, 0>{});
      if (i + 1 < S) iteration(i + 1, std::integral_constant<int, 1>{});
    }
  } else {
    for (int i = 0; i < S; ++i) iteration(i, std::integral_constant<int, 0>{});
  }

  if (a.stats) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (nothing in flight into LDS: the ring doubles as scratch)
    float* sl = (float*)lds;  // [WPX][2][NB]
#pragma unroll
    for (int k = 0; k < C::LANE_C; ++k) {
      s1[k] = wave_sum16(s1[k]);
      s2[k] = wave_sum16(s2[k]);
    }
    if (li == 0) {
#pragma unroll
      for (int k = 0; k < C::LANE_C; ++k) {
        const int cl = wco * C::CN + lg * C::LANE_C + k;
        sl[(wpx * 2 + 0) * NB + cl] = s1[k];
        sl[(wpx * 2 + 1) * NB + cl] = s2[k];
      }
    }
    lds_barrier();
    if (tid < 2 * NB) {
      const int which = tid / NB, cl = tid % NB;
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < WPX; ++q) s += sl[(q * 2 + which) * NB + cl];
      if (n0 + cl < a.Cout) {
        if constexpr (BS) a.stats[((size_t)wg_p * 2 + which) * a.Cout + n0 + cl] = s;       // one row per pixel walk
        else atomicAdd(a.stats + ((size_t)(wg_p & (HR_BN_COPIES - 1)) * 2 + which) * a.Cout + n0 + cl, s);
      }
    }
  }
}

struct RingPlan {
  int id;        // 0 = not served
  int th, tw, ti, nb, per_cu;   // per_cu: workgroups per CU the grid aims at (0: one tile per workgroup)
};

// which instantiation serves a layer (0: none). Shapes: the branch widths of pose_hrnet (w32: 32 / 64 / 128 / 256).
// The narrow, HBM-bound layers keep their weights resident and walk several tiles; the wide layers (small maps,
// large weight matrices) get FEW fat workgroups - 64 output channels x a whole image (or several 8x8 images), the
// weight slices streamed once per workgroup - because what they move is mostly weights: 256 ch @ 8x8 as 512
// workgroups of (one image, 32 channels) pulls 75 MB of weights through the CUs' load paths for a 6 MB layer.
int g_ring_enabled = -1;
inline RingPlan ring_plan_all(int N, int H, int W, int Cin, int Cout, bool bs);
// bytes of one ring slot's input image per instantiation (RingCfg::XSB), checked against the kernels below
constexpr int ring_xsb(int id) {
  return id == 1 ? RingCfg<16, 16, 1, 32, 4, 4, 3, 1>::XSB : id == 2 ? RingCfg<8, 16, 1, 32, 4, 4, 3, 2>::XSB
       : id == 3 ? RingCfg<16, 16, 1, 64, 8, 4, 2, 0>::XSB : id == 4 ? RingCfg<8, 8, 4, 64, 8, 4, 2, 0>::XSB
       : id == 5 ? RingCfg<8, 8, 2, 64, 8, 4, 2, 0>::XSB : id == 6 ? RingCfg<16, 16, 2, 64, 8, 4, 2, 0>::XSB : 0;
}
// The prologue stages the input BatchNorm's [8][2][Cin] batch sums and gamma | beta in the LAST ring slot (whole
// 1 KiB pieces): a launch whose staging image is larger than a slot would spill into weight slot 0 while other waves
// direct-load it. Such shapes (Cin > 256 on the 16x16 / two-image instantiations: w48's widest branch on 512x512
// inputs) are not served - the tile-walking body takes them.
constexpr int ring_stage_bytes(int Cin) { return ((Cin * 64 + 1023) & ~1023) + ((Cin * 8 + 1023) & ~1023); }
static_assert(ring_stage_bytes(32) <= ring_xsb(1) && ring_stage_bytes(64) <= ring_xsb(2), "resident-weight instantiations");
static_assert(ring_stage_bytes(HR_RING_MAXC) <= ring_xsb(4) && ring_stage_bytes(HR_RING_MAXC) <= ring_xsb(6), "4 x 8x8 / 2 x 16x16 images");
static_assert(ring_stage_bytes(256) <= ring_xsb(3) && ring_stage_bytes(256) <= ring_xsb(5), "w32's widest branch");
inline RingPlan ring_plan(int N, int H, int W, int Cin, int Cout, bool bs) {
  // (measurement: HRNET_RING_IDS = bit mask of the instantiations that may be chosen)
  static const int ids = hr_knob("HRNET_RING_IDS", 0x7e);
  const RingPlan p = ring_plan_all(N, H, W, Cin, Cout, bs);
  if (!bs && p.id && ring_stage_bytes(Cin) > ring_xsb(p.id)) return RingPlan{0, 0, 0, 0, 0, 0};
  return ((ids >> p.id) & 1) ? p : RingPlan{0, 0, 0, 0, 0, 0};
}
inline RingPlan ring_plan_all(int N, int H, int W, int Cin, int Cout, bool bs) {
  (void)N;
  static const int wide8 = hr_knob("HRNET_RING_TI8", 4);   // (measurement: 2 or 4)
  static const int wide16 = hr_knob("HRNET_RING_TI16", 1);   // (measurement: 1 or 2)
  if (Cin % 32 != 0 || Cout % 16 != 0 || Cin > HR_RING_MAXC) return RingPlan{0, 0, 0, 0, 0, 0};
  // (one workgroup per CU for the 32-channel instantiation: 256 workgroups of 80 KB leave room for the other lanes'
  // workgroups on every CU - 15.32-15.35 against 15.36-15.39 ms/step with two per CU, round 4)
  if (!bs && Cin == 32 && Cout <= 32 && H >= 16 && W >= 16) return RingPlan{1, 16, 16, 1, 32, 1};
  if (!bs && Cin == 64 && Cout >= 32 && H >= 16 && W >= 16) return RingPlan{2, 8, 16, 1, 32, 2};
  if (Cin >= 96 && Cout >= 64 && H == 8 && W == 8) return wide8 == 2 ? RingPlan{5, 8, 8, 2, 64, 0} : RingPlan{4, 8, 8, 4, 64, 0};
  if (Cin >= 96 && Cout >= 64 && H == 16 && W == 16 && wide16 == 2) return RingPlan{6, 16, 16, 2, 64, 0};
  // instantiation 3 serves any map of at least 16x16, but inside the training step it only wins on the 16x16 maps
  // (w48: 96 channels at 32x32 lose 0.5 ms/step with it): hrnet_conv_ring_enable(2) (or HRNET_CONV_RING=2) lifts the
  // restriction (tests of the overhanging-tile walk, measurements)
  const bool id3_any = g_ring_enabled == 2;
  if (Cin >= 96 && Cout >= 64 && H >= 16 && W >= 16 && (id3_any || (H <= 16 && W <= 16))) return RingPlan{3, 16, 16, 1, 64, 0};
  return RingPlan{0, 0, 0, 0, 0, 0};
}

#ifdef HR_RING_STAMP
unsigned long long* g_ring_stamp = nullptr;
#endif

inline void ring_grid(const RingPlan& p, int N, int H, int W, int Cout, int& tiles_y, int& tiles_x, int& total, int& tpw,
                      int& gx, int& gy) {
  tiles_y = (H + p.th - 1) / p.th; tiles_x = (W + p.tw - 1) / p.tw;
  total = ((N + p.ti - 1) / p.ti) * tiles_y * tiles_x;
  gy = (Cout + p.nb - 1) / p.nb;
  tpw = 1;
  if (p.per_cu > 0) {
    // (measurement overrides: workgroups of the two narrow instantiations)
    static const int wgs1 = hr_knob("HRNET_RING_WGS1", 0);
    static const int wgs2 = hr_knob("HRNET_RING_WGS2", 0);
    int target = 256 * p.per_cu;
    if (p.id == 1 && wgs1 > 0) target = wgs1;
    if (p.id == 2 && wgs2 > 0) target = wgs2;
    tpw = (total * gy + target - 1) / target;
    if (tpw < 1) tpw = 1;
  }
  gx = (total + tpw - 1) / tpw;
}

}  // namespace

#ifdef HR_RING_STAMP
extern "C" int hrnet_conv_ring_set_stamp(void* p) { g_ring_stamp = (unsigned long long*)p; return 0; }
#endif

extern "C" int hrnet_conv_ring_enable(int on) {
  const int prev = g_ring_enabled;
  g_ring_enabled = on;
  return prev;
}

int hr_conv_ring_enabled() {
  if (g_ring_enabled < 0) {
    g_ring_enabled = hr_knob("HRNET_CONV_RING", 1);     // (the run-time switch is hrnet_conv_ring_enable())
  }
  return g_ring_enabled;
}

extern "C" int hrnet_conv_ring_supported(int dtype, int N, int H, int W, int Cin, int Cout) {
  return hr_conv_ring_enabled() ? hr_conv_ring_supported(dtype, N, H, W, Cin, Cout, 0) : 0;
}

int hr_conv_ring_supported(int dtype, int N, int H, int W, int Cin, int Cout, int bs) {
  if (dtype != HR_BF16) return 0;
  if ((double)N * H * W * (Cin > Cout ? Cin : Cout) * 2.0 >= 2147483648.0) return 0;   // 32-bit buffer offsets
  return ring_plan(N, H, W, Cin, Cout, bs != 0).id;
}

// The residual-sum form is OFF by default: bit-identical to the tile-walking body, but inside the training step it loses (15.50-15.56 against 15.39-15.41 ms/step in three A/B
// pairs: 48 more VGPRs per wave for the identity term and the side stores inside the ring). hrnet_conv_ring_sum_enable(1)
// (or HRNET_RING_SUM=1 with HRNET_MEASURE=1) turns it on: tests/test_conv_ring_gpu.py, measurements.
static int g_ring_sum = -1;
extern "C" int hrnet_conv_ring_sum_enable(int on) {
  const int prev = g_ring_sum < 0 ? hr_knob("HRNET_RING_SUM", 0) : g_ring_sum;
  g_ring_sum = on;
  return prev;
}
int hr_conv_ring_sum_supported(int dtype, int N, int H, int W, int Cin, int Cout) {
  if (g_ring_sum < 0) g_ring_sum = hr_knob("HRNET_RING_SUM", 0);
  const int id = (g_ring_sum && hr_conv_ring_enabled()) ? hr_conv_ring_supported(dtype, N, H, W, Cin, Cout, 0) : 0;
  return (id == 1 || id == 2) ? 1 : 0;
}

int hr_conv_ring_rows(int N, int H, int W, int Cin, int Cout) {
  const RingPlan p = ring_plan(N, H, W, Cin, Cout, true);
  if (!p.id) return 0;
  int ty, tx, total, tpw, gx, gy;
  ring_grid(p, N, H, W, Cout, ty, tx, total, tpw, gx, gy);
  return gx;
}

int hr_conv_ring_name(int id, int bs, char* buf, int buflen) {
  static const char* names[] = {"", "16, 16, 1, 32, 4, 4, 3, 1", "8, 16, 1, 32, 4, 4, 3, 2", "16, 16, 1, 64, 8, 4, 2, 0",
                                "8, 8, 4, 64, 8, 4, 2, 0", "8, 8, 2, 64, 8, 4, 2, 0", "16, 16, 2, 64, 8, 4, 2, 0"};
  if (id < 1 || id > 6) return snprintf(buf, buflen, "%s", "");
  return snprintf(buf, buflen, "conv_ring_kernel<%s, %s>", names[id], bs ? "true" : "false");
}

int hr_conv_ring_launch(const HrRingConv& c, hipStream_t s) {
  const bool bs = c.bs_y != nullptr || c.accumulate;
  const RingPlan p = ring_plan(c.N, c.H, c.W, c.Cin, c.Cout, bs);
  HR_REQUIRE(p.id != 0, "conv_ring: shape not served (Cin %d Cout %d %dx%d)", c.Cin, c.Cout, c.H, c.W);
  HR_REQUIRE(c.x && c.w && c.y, "conv_ring: null pointer");
  HR_REQUIRE(!c.in_sums || c.in_gb, "conv_ring: batch sums need gamma/beta");
  HR_REQUIRE((c.in_scale == nullptr) == (c.in_shift == nullptr), "conv_ring: scale/shift must come together");
  HR_REQUIRE(!(c.in_sums && c.in_scale), "conv_ring: batch sums OR scale/shift");
  HR_REQUIRE(!bs || (!c.in_sums && !c.in_scale && !c.in_relu), "conv_ring: an input-gradient launch reads a raw input");
  HR_REQUIRE(!c.bs_y || c.stats, "conv_ring: backward statistics need a rows buffer");
  HR_REQUIRE((c.bs_scale == nullptr) == (c.bs_shift == nullptr), "conv_ring: mask scale/shift must come together");
  RingArgs a;
  a.x = (const char*)c.x; a.w = (const char*)c.w; a.y = (char*)c.y;
  a.in_sums = c.in_sums; a.in_gb = c.in_gb; a.in_scale = c.in_scale; a.in_shift = c.in_shift;
  a.stats = c.stats; a.in_inv_count = c.in_inv_count; a.in_eps = c.in_eps;
  a.bs_y = (const char*)c.bs_y; a.bs_mask = (const char*)c.bs_mask; a.bs_scale = c.bs_scale; a.bs_shift = c.bs_shift;
  a.x2 = (const char*)c.x2; a.side = (char*)c.side;
  const bool x2 = c.x2 != nullptr;
  HR_REQUIRE(!x2 || (c.side && !bs && (p.id == 1 || p.id == 2) && (c.in_sums || c.in_scale)),
             "conv_ring: the residual-sum form needs side, a BatchNorm term and a resident-weight instantiation (id %d)", p.id);
  a.N = c.N; a.H = c.H; a.W = c.W; a.Cin = c.Cin; a.Cout = c.Cout;
  ring_grid(p, c.N, c.H, c.W, c.Cout, a.tiles_y, a.tiles_x, a.total_tiles, a.tpw, a.gx, a.gy);
  a.in_relu = c.in_relu; a.accumulate = c.accumulate; a.bs_store_masked = c.bs_store_masked;
  a.x_bytes = (unsigned)((size_t)c.N * c.H * c.W * c.Cin * 2);
  a.y_bytes = (unsigned)((size_t)c.N * c.H * c.W * c.Cout * 2);
  a.w_bytes = (unsigned)((size_t)c.Cout * 9 * c.Cin * 2);
#ifdef HR_RING_STAMP
  a.stamp = g_ring_stamp;
#endif
  const unsigned grid = (unsigned)((a.gx * a.gy + 7) / 8 * 8);
#define RING(BS_, NW_, ...) hipLaunchKernelGGL((conv_ring_kernel<__VA_ARGS__, BS_>), dim3(grid), dim3(NW_ * 64), 0, s, a)
#define RINGX(NW_, ...) hipLaunchKernelGGL((conv_ring_kernel<__VA_ARGS__, false, true>), dim3(grid), dim3(NW_ * 64), 0, s, a)
  if (x2) {
    if (p.id == 1) RINGX(4, 16, 16, 1, 32, 4, 4, 3, 1);
    else RINGX(4, 8, 16, 1, 32, 4, 4, 3, 2);
  } else if (!bs) {
    switch (p.id) {
      case 1: RING(false, 4, 16, 16, 1, 32, 4, 4, 3, 1); break;
      case 2: RING(false, 4, 8, 16, 1, 32, 4, 4, 3, 2); break;
      case 3: RING(false, 8, 16, 16, 1, 64, 8, 4, 2, 0); break;
      case 4: RING(false, 8, 8, 8, 4, 64, 8, 4, 2, 0); break;
      case 6: RING(false, 8, 16, 16, 2, 64, 8, 4, 2, 0); break;
      default: RING(false, 8, 8, 8, 2, 64, 8, 4, 2, 0); break;
    }
  } else {
    switch (p.id) {
      case 3: RING(true, 8, 16, 16, 1, 64, 8, 4, 2, 0); break;
      case 4: RING(true, 8, 8, 8, 4, 64, 8, 4, 2, 0); break;
      case 6: RING(true, 8, 16, 16, 2, 64, 8, 4, 2, 0); break;
      default: RING(true, 8, 8, 8, 2, 64, 8, 4, 2, 0); break;
    }
  }
#undef RING
#undef RINGX
  return hr_check_launch("conv_ring");
}
