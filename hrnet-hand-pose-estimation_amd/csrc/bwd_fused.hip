// Fused backward of a 3x3 stride-1 convolution whose output feeds a BatchNorm, for gfx950.
//
// One launch does what the unfused backward does in four (grad_term, wgrad, conv_bs, and the
// gradient-side half of the residual add):
//
//   g[p,co]   = A[co]*dz[p,co] + B[co]*y[p,co] + C[co]           BatchNorm backward applied while the
//                                                                output-side tile is staged into LDS
//                                                                (dz = upstream gradient, already masked
//                                                                by its ReLU; y = raw conv output)
//   dW[co,t,ci] += sum_p g[p,co] * a[p+t,ci]                     weight gradient (pixels = K, transposing
//                                                                LDS reads of the SAME two images)
//   v[p,ci]   = sum_{t,co} Wt[ci,t,co] * g[p+t,co] (+ addend)    input gradient (+ the residual stream)
//   dx[p,ci]  = v * [a[p,ci] > 0]                                masked by the ReLU in front of the conv:
//                                                                what is stored IS the next dz
//   rows      = (sum dx, sum dx*yb)                              statistics of the next BatchNorm backward
//
// a = relu?(scale*x+shift) is the conv's input as the forward pass saw it (staged once, used as the
// weight-gradient operand AND as the mask). Per 16x16 pixel tile the kernel reads three halo images
// (dz, y, x) and writes one tile: ~4.8 tensor passes where the unfused sequence makes ~11, and the
// MFMA work per staged byte doubles.
//
// Geometry: 512 threads (8 waves), one workgroup per CU, 16x16 tiles, all Cout (<= 64) channels of the
// output side staged, one 32-channel block of the input side per workgroup (its slice of dW stays in
// registers across the tiles the workgroup walks -> one f32 slab per workgroup, summed by
// hrnet_wgrad_reduce(_table) like the unfused slabs). The next tile's global loads are issued into
// registers before the MFMAs of the current one.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#ifndef HR_FUSED_XRAW
#define HR_FUSED_XRAW 0     // 1: keep x as stored in a second LDS image and take the statistics operand from it when bs_y == x (measured: -17 MB per launch, +0.05 ms/step - the extra 20 KB of LDS cost more than the bytes)
#endif

namespace {

struct BwdArgs {
  const char* dz;        // [N,H,W,Cout] upstream gradient w.r.t. the BatchNorm output, ReLU mask applied
  const char* y;         // [N,H,W,Cout] raw conv output
  const float* coef;     // [3][Cout] A,B,C of hrnet_bn_bwd_finalize, or NULL: g = dz
  HrBnBwdRef ref;        // ref.rows != NULL: the coefficients are built here from the partial rows (coef unused)
  const char* x;         // [N,H,W,Cin] conv input as stored
  const float* in_scale; // optional per-Cin affine (+ReLU) the forward applied on load
  const float* in_shift;
  const char* wT;        // packed [Cin][9 flipped][Cout] (hrnet_pack_weights mode 1)
  char* dx;              // [N,H,W,Cin] out
  const char* addend;    // optional [N,H,W,Cin] added before the mask (may alias dx)
  const char* bs_y;      // optional [N,H,W,Cin]: rows get sum(dx*bs_y)
  float* rows;           // optional [nsplit][2][Cin]
  float* slabs;          // [nsplit][Cout][9][Cin] f32
  int N, H, W, Cin, Cout;
  unsigned g_bytes, a_bytes;   // bytes of the [N,H,W,Cout] / [N,H,W,Cin] tensors (buffer descriptors: 32-bit offsets)
  int tiles_y, tiles_x, total_tiles;
  int ncb, nsplit;
  int in_relu, mask_out;
  // atomic = 1: `slabs` IS the OIHW f32 gradient [Cout_real][Cin_real][3][3]; the workgroups ADD their tiles into it
  // with float atomics, through LDS so that a wave instruction covers 64 consecutive floats (no slabs, no reduce)
  int atomic, Cout_real, Cin_real;
#ifdef HR_MEASURE
  // measurement builds only (scratch/build_variant.py measure -DHR_MEASURE; never in the shipped library):
  unsigned long long* stamp;   // HRNET_FUSED_STAMP_PTR: 32 s_memtime stamps per workgroup
  int ablate;   // HRNET_FUSED_ABLATE: 1 skip input-gradient MFMAs, 2 skip weight-gradient MFMAs, 4 skip the epilogue's
                // global traffic, 8 skip the tile loads (stage zeros)
#endif
};

#ifdef HR_MEASURE
#define HR_ABLATE(a, bit) (((a).ablate & (bit)) != 0)
#else
#define HR_ABLATE(a, bit) (false)
#endif

template <typename T>
__device__ __forceinline__ V16 trl(const char* base, int r0, int rstep, int choff, int lane);
template <>
__device__ __forceinline__ V16 trl<bf16_t>(const char* base, int r0, int rstep, int choff, int lane) {
  const int q = (lane & 15) >> 2, p4 = lane & 3;
  const int addr = r0 + q * rstep + choff + p4 * 8;
  const LDS_AS char* l = (const LDS_AS char*)base;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(l + addr));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(l + addr + 4 * rstep));
  const bf16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return __builtin_bit_cast(V16, v);
}
template <>
__device__ __forceinline__ V16 trl<float>(const char* base, int r0, int rstep, int choff, int lane) {
  const char* p = base + r0 + choff + (lane & 15) * 4;
  return V16{*(const uint32_t*)p, *(const uint32_t*)(p + rstep), *(const uint32_t*)(p + 2 * rstep),
             *(const uint32_t*)(p + 3 * rstep)};
}

// LDS images are pixel-major with unpadded rows of RB = 64 / 128 / 256 bytes; the 16-byte chunk c of row P is stored at
// chunk position c ^ lds_swz<RB>(P). With it both access patterns of the two matrix phases are free of bank conflicts
// (round 3's padded rows: 45 % of the LDS cycles of a launch were conflict cycles, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE):
//   ds_read_b128 of 16 consecutive rows x one chunk per lane group (input-gradient operands; serviced in the lane
//   groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...): the 16 rows of a group fall on 16 different 16-byte slots;
//   ds_read_b64_tr_b16 of rows P..P+3 (lane group 0) and P+4..P+7 (lane group 1) x two chunks (weight-gradient operands;
//   32-lane halves): the 8 rows x 32 bytes cover the 64 banks once.
// P = the pixel's COLUMN in its halo row (every read above stays inside one row, and a row's start only shifts all of its
// slots alike), so a tap's row shift never changes the swizzle: a lane keeps one LDS offset per column shift and every
// other displacement is an instruction offset.
// (SWZ = false - the f32 instantiation, whose one-float transposing reads would need an offset per pixel: rows padded by
// 16 bytes as in round 3, no swizzle.)
template <int RB, bool SWZ = true>
__device__ __forceinline__ int lds_swz(int P) {
  static_assert(!SWZ || RB == 64 || RB == 128 || RB == 256, "row bytes");
  if constexpr (!SWZ) return 0;
  return RB == 64 ? ((P >> 1) & 2) : RB == 128 ? (P & 6) : ((P & 7) << 1);
}

// COP: staged output-side channels (Cout padded to 32 or 64). WCO x WN = 8 waves over the weight-gradient
// outputs: WCO co-halves x WN groups of (tap, ci-fragment).
// YLDS: the raw conv output y of the NEXT tile is parked in LDS instead of registers while it is in flight
// (direct global->LDS loads, each lane reads back exactly the 16 bytes it requested): 24 fewer live VGPRs.
// TH: tile height (tile width 16). NW: waves per workgroup (NW*64 threads): 8 waves = one workgroup per CU;
// 4 waves = two or three co-resident workgroups per CU, each with its own tile in flight, so that one
// workgroup's load waits overlap another's matrix work.
template <typename T, int COP, int TH, int NW, int WCO, int WN, bool YLDS>
__global__ __launch_bounds__(NW * 64) void bwd_fused_kernel(BwdArgs a) {
  constexpr int VEC = TT<T>::VEC;
  constexpr int KSTEP = TT<T>::KSTEP;
  constexpr int ES = (int)sizeof(T);
  constexpr int NT = NW * 64;
  constexpr int TW = 16, HALO = 18, HALO_H = TH + 2, HPX = HALO_H * HALO, BM = TH * TW;
  constexpr int PW = BM / NW, FP = PW / 16;    // input-gradient pixels / pixel fragments per wave
  constexpr int CB = 32;                       // input-side channels per workgroup
  constexpr bool SWZ = ES == 2;                // bf16: swizzled images; f32: padded rows (lds_swz)
  constexpr int HP = HALO;                     // halo row pitch of the LDS images in pixels
  constexpr int RBG = COP * ES + (SWZ ? 0 : 16), RBA = CB * ES + (SWZ ? 0 : 16);   // row (pixel) bytes of the g and a images
  constexpr int NKK = COP / KSTEP;             // 64-byte k-chunks of the output-side channels
  // weights: [tap][k-chunk][input channel (row)][64 bytes = KSTEP output channels], rows swizzled like a 64-byte image
  constexpr int GBYTES = HALO_H * HP * RBG, ABYTES = HALO_H * HP * RBA, WBYTES = 9 * NKK * CB * 64;
  static_assert(KSTEP * ES == 64, "a k-chunk of the weight image is one 64-byte row");
  constexpr int VPG = COP / VEC, VPA = CB / VEC;     // 16-byte vectors per pixel
  constexpr int GVECS = HPX * VPG, AVECS = HPX * VPA, WVECS = CB * 9 * VPG;
  constexpr int XG = (GVECS + NT - 1) / NT, XA = (AVECS + NT - 1) / NT, XW = (WVECS + NT - 1) / NT;
  constexpr int FCO = COP / 16;                // co fragments
  // weight gradient: the (co-fragment, ci-fragment) blocks of dW[COP][9][CB] - nine 16x16 tiles each, one per tap - are
  // dealt to the waves: BPW whole blocks per wave when there are at least as many blocks as waves, else the pixel
  // (contraction) axis is split over KSPLIT groups of waves, each wave owning one block for every KSPLIT-th k-step; the
  // groups' partial sums meet in LDS after the walk. Every wave issues the same 9 * BPW MFMAs per k-step (round 3's
  // (tap, ci-fragment) lists were uneven - 3 / 2 per wave - and their `if (fr < NFR)` tests cut the phase into 24
  // read -> wait -> 2 MFMA blocks per tile).
  constexpr int NBLK = FCO * (CB / 16);
  // (round 4, measured and rejected: TWO co-fragments per wave - 2 + 9 transposed fragment reads for 18 MFMAs per
  // k-step instead of 1 + 9 for 9, the pixel axis split over twice as many wave groups. In-kernel stamps had the phase
  // at 1955 cycles per 16x16 tile for 1150 cycles of MFMA issue, so it looked bound by the LDS pipe; with the change
  // the 32-channel instantiation went from 43.0 to 41.2 us alone, the 64-channel one from 35.7 to 42.2 (256 VGPRs and
  // 60 bytes of scratch) and the step from 15.45 to 16.15 ms: what bounds the phase is the latency of a dependent
  // transposing read at two waves per SIMD, and registers are what pays for hiding it. TWO = true brings it back.)
  constexpr int KS2 = NBLK >= NW * 2 ? 1 : NW * 2 / NBLK;         // k-groups when a wave takes two blocks
  constexpr bool TWO = false && ES == 2 && FCO % 2 == 0 && NBLK * KS2 >= NW * 2 && (BM / KSTEP) % KS2 == 0;
  constexpr int KSPLIT = TWO ? KS2 : (NBLK >= NW ? 1 : NW / NBLK);
  constexpr int NWB = NW / KSPLIT;             // waves of one k-group
  constexpr int BPW = NBLK / NWB;              // blocks per wave (consecutive co-fragments of one ci-fragment)
  constexpr int NFR = 9 * (CB / 16);           // (tap, ci-fragment) outputs per co-fragment = 18
  static_assert(WCO * WN == NW && PW % 16 == 0, "wave grid");
  static_assert(NBLK % NWB == 0 && NW % KSPLIT == 0 && FCO % BPW == 0, "weight-gradient blocks per wave");
  static_assert((BM / KSTEP) % KSPLIT == 0 && KSTEP % TW == 0, "k-steps per wave group; a k-step is whole tile rows");
  static_assert(NT % VPG == 0 && NT % VPA == 0, "a thread keeps one channel vector");
  constexpr int YBYTES = YLDS ? XG * NT * 16 : 0;   // one 16-byte slot per (thread, k): lane-linear
  constexpr int CBYTES = (3 * COP + 2 * CB) * 4;     // per-channel coefficient tables: A, B, C | scale, shift
  // XRAW (bf16): a second input-side image holding x AS STORED. When the next BatchNorm's raw input IS this conv's
  // input (bs_y == x: the second conv of a block, whose input relu(bn1(y1)) is y1 read through its BatchNorm) the
  // epilogue takes y1 at the pixel from this image instead of fetching the tensor a second time: one tensor pass of
  // five less for half of the fused launches, bit-identical statistics. (f32: the images leave no room.)
  constexpr bool XRAW = ES == 2 && HR_FUSED_XRAW != 0;
  constexpr int XRBYTES = XRAW ? ABYTES : 0;
  static_assert(GBYTES + ABYTES + WBYTES + YBYTES + XRBYTES + CBYTES <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) char lds[GBYTES + ABYTES + WBYTES + YBYTES + XRBYTES + CBYTES];
  char* gl = lds;
  char* al = lds + GBYTES;
  char* wl = lds + GBYTES + ABYTES;
  char* yl = lds + GBYTES + ABYTES + WBYTES;
  char* xrl = lds + GBYTES + ABYTES + WBYTES + YBYTES;
  float* ctab = (float*)(lds + GBYTES + ABYTES + WBYTES + YBYTES + XRBYTES);   // [A COP][B COP][C COP][scale CB][shift CB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  // workgroup -> (split, input-channel block): the ncb blocks of one split get ids 8 apart (same XCD: they
  // stage the same output-side tiles)
  // ... and (round 3) every XCD takes a CONTIGUOUS run of splits: at each step of the walk the splits hold consecutive
  // pixel tiles, whose halos overlap - they meet in one L2 instead of being fetched by two
  const int S8 = (a.nsplit + 7) >> 3;                       // splits per XCD
  const int xcd = (int)blockIdx.x & 7, j8 = (int)blockIdx.x >> 3;
  const int cb = j8 % a.ncb;
  const int split = xcd * S8 + j8 / a.ncb;
  if (split >= a.nsplit || j8 / a.ncb >= S8) return;
  const int c0 = cb * CB;
  int stamp_i = 0;
#ifdef HR_MEASURE
#define FSTAMP() do { if (a.stamp && tid == 0 && stamp_i < 32) a.stamp[(size_t)blockIdx.x * 32 + stamp_i++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FSTAMP() do { (void)stamp_i; } while (0)
#endif
  FSTAMP();

  // ---- weights of this input-channel block: resident for the whole walk ----
  {
    const int v = tid % VPG;
#pragma unroll
    for (int k = 0; k < XW; ++k) {
      const int idx = tid + k * NT;
      if (idx < WVECS) {
        const int rt = idx / VPG;
        const int tp = rt % 9, q = rt / 9;          // LDS row q holds the channel the MFMA row order needs
        const int ci = c0 + ((q & 15) >> 2) * 8 + (q >> 4) * 4 + (q & 3);
        const int co = v * VEC;
        V16 w = v16_zero();
        if (ci < a.Cin && co < a.Cout) w = *(const V16*)(a.wT + ((size_t)(ci * 9 + tp) * a.Cout + co) * ES);
        *(V16*)(wl + ((tp * NKK + (v >> 2)) * CB + q) * 64 + (((v & 3) ^ lds_swz<64>(q)) << 4)) = w;
      }
    }
  }

  FSTAMP();
  // coefficient tables (read back per tile: a dependent global load inside the staging pass costs its full latency)
  const bool from_rows = a.ref.rows != nullptr;
  if (from_rows) {
    static_assert(NT * 2 * 8 <= GBYTES, "row-sum scratch fits the g image");
    hr_bnbwd_coef_from_rows<NT, COP>(a.ref, a.Cout, (double*)gl, ctab, blockIdx.x == 0);
  }
  for (int i = tid; i < 3 * COP + 2 * CB; i += NT) {
    float v = 0.f;
    if (i < 3 * COP) {
      if (from_rows) continue;
      const int w = i / COP, c = i % COP;
      if (a.coef && c < a.Cout) v = a.coef[w * a.Cout + c];
    } else {
      const int j = i - 3 * COP, w = j / CB, c = c0 + j % CB;
      v = w == 0 ? 1.f : 0.f;
      if (a.in_scale && c < a.Cin) v = w == 0 ? a.in_scale[c] : a.in_shift[c];
    }
    ctab[i] = v;
  }
  __syncthreads();   // the first staging pass reads the tables
  const int vg = tid % VPG, va = tid % VPA;
  const int cg = vg * VEC, ca = c0 + va * VEC;
  const bool cg_ok = cg < a.Cout, ca_ok = ca < a.Cin;
  const bool has_coef = a.coef != nullptr || from_rows, has_aff = a.in_scale != nullptr, in_relu = a.in_relu != 0;
  // the statistics operand is the conv's own input tensor: read it from LDS (the raw image when the staged one is
  // transformed, else the staged image itself)
  const bool bs_from_x = XRAW && a.rows != nullptr && a.bs_y != nullptr && a.bs_y == a.x;
  const bool xr_store = bs_from_x && (has_aff || in_relu);
  const char* bsl = xr_store ? xrl : al;

  V16 rz[XG], ry[XG], rx[XA];
  unsigned okg = 0, oka = 0;

  auto tile_of = [&](int t, int& n, int& ty, int& tx) {
    tx = t % a.tiles_x;
    t /= a.tiles_x;
    ty = t % a.tiles_y;
    n = t / a.tiles_y;
  };

  // per-thread staging geometry is the same for every tile: the halo coordinates of each staged vector and its
  // byte offset from the tile's first halo pixel are computed once; per tile two adds and two unsigned compares
  // per vector remain (the address is a wave-uniform base + a 32-bit lane offset)
  const int rowG = a.W * a.Cout * ES, pixG = a.Cout * ES, rowA = a.W * a.Cin * ES, pixA = a.Cin * ES;
  int hyxg[XG], offg[XG], hyxa[XA], offa[XA];
  int lofg[XG], lofa[XA];                    // LDS byte offsets of the staged vectors (swizzled images)
#pragma unroll
  for (int k = 0; k < XG; ++k) {
    const int pix = (tid + k * NT) / VPG;
    const int hy = pix / HALO, hx = pix - hy * HALO;
    hyxg[k] = (tid + k * NT) < GVECS && cg_ok ? (hy << 16) | hx : (0x4000 << 16);    // invalid: row far outside
    offg[k] = hy * rowG + hx * pixG + cg * ES;
    lofg[k] = (hy * HP + hx) * RBG + ((vg ^ lds_swz<RBG, SWZ>(hx)) << 4);
  }
#pragma unroll
  for (int k = 0; k < XA; ++k) {
    const int pix = (tid + k * NT) / VPA;
    const int hy = pix / HALO, hx = pix - hy * HALO;
    hyxa[k] = (tid + k * NT) < AVECS && ca_ok ? (hy << 16) | hx : (0x4000 << 16);
    offa[k] = hy * rowA + hx * pixA + ca * ES;
    lofa[k] = (hy * HP + hx) * RBA + ((va ^ lds_swz<RBA, SWZ>(hx)) << 4);
  }

  // the next tile's loads are issued in NS steps spread over the nine taps of the input-gradient loop instead of
  // in one burst: a burst of ~100 KB per CU stalls the issuing waves on the CU's miss queue (~2.5 us per tile
  // measured) and leaves the memory pipe idle during the matrix phases
  // Buffer loads: a lane outside the image (or a walk with no next tile) gets an offset beyond the descriptor's
  // range - the hardware returns zeros and fetches nothing - so the prefetch is free of branches and sits inside the
  // matrix phases without cutting them into scheduling regions (round 3 branched around every batch of loads).
  constexpr unsigned OOB = 0x80000000u;
  const auto rdz = __builtin_amdgcn_make_buffer_rsrc((void*)a.dz, 0, (int)a.g_bytes, 0x00020000);
  const auto rdy = __builtin_amdgcn_make_buffer_rsrc((void*)(has_coef ? a.y : a.dz), 0, has_coef ? (int)a.g_bytes : 0, 0x00020000);
  const auto rdx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.a_bytes, 0x00020000);
  int tbg = 0, tba = 0;        // byte offset of the loaded tile's first halo pixel (may be negative: valid lanes add >= its magnitude)
  int liy0 = 0, lix0 = 0;
  bool lnext = false;
  auto load_begin = [&](int t, bool valid) {
    int n, ty, tx;
    tile_of(valid ? t : 0, n, ty, tx);
    n = __builtin_amdgcn_readfirstlane(n); ty = __builtin_amdgcn_readfirstlane(ty); tx = __builtin_amdgcn_readfirstlane(tx);
    liy0 = ty * TH - 1; lix0 = tx * TW - 1;
    tbg = (n * a.H + liy0) * rowG + lix0 * pixG;
    tba = (n * a.H + liy0) * rowA + lix0 * pixA;
    lnext = valid;
    okg = oka = 0;
  };
  auto load_g = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const int gy = liy0 + (hyxg[k] >> 16), gx = lix0 + (hyxg[k] & 0xffff);
    const bool ok = lnext && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    const unsigned vo = ok ? (unsigned)(tbg + offg[k]) : OOB;
    rz[k] = __builtin_amdgcn_raw_buffer_load_b128(rdz, vo, 0, 0);
    ry[k] = __builtin_amdgcn_raw_buffer_load_b128(rdy, vo, 0, 0);   // (no coefficients: an empty descriptor, nothing is fetched)
    okg |= (ok ? 1u : 0u) << k;
  };
  auto load_a = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const int gy = liy0 + (hyxa[k] >> 16), gx = lix0 + (hyxa[k] & 0xffff);
    const bool ok = lnext && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    rx[k] = __builtin_amdgcn_raw_buffer_load_b128(rdx, ok ? (unsigned)(tba + offa[k]) : OOB, 0, 0);
    oka |= (ok ? 1u : 0u) << k;
  };
  constexpr int NS = XG + XA;
  constexpr int NKW = (BM / KSTEP) / KSPLIT;                       // weight-gradient k-steps of one wave
#ifndef HR_FUSED_LOAD_SLOTS
#define HR_FUSED_LOAD_SLOTS 0     // measurement builds: n > 0 = the next tile's loads are all issued within the first n taps of the input-gradient phase (round 4: n = 9 / 5 / 3 change nothing, 15.33-15.35 against 15.26-15.31 ms/step - the wait at the top of a tile is not a matter of when its loads were issued)
#endif
  constexpr bool WSLOTS = HR_FUSED_LOAD_SLOTS == 0 && NKW <= 8;     // the weight-gradient k-steps are issue slots too
  constexpr int NSLOT = HR_FUSED_LOAD_SLOTS > 0 ? HR_FUSED_LOAD_SLOTS : 9 + (WSLOTS ? NKW : 0);   // issue slots: the 9 taps (+ the weight-gradient k-steps)
  auto load_step = [&](auto sc) {       // step s < XG: dz + y vector s; else x vector s - XG
    constexpr int s_ = decltype(sc)::value;
    if constexpr (s_ < XG) load_g(std::integral_constant<int, s_>{});
    else if constexpr (s_ < NS) load_a(std::integral_constant<int, s_ - XG>{});
  };
  auto load_steps = [&](auto lo, auto hi) {
    constexpr int L = decltype(lo)::value, Hh = decltype(hi)::value;
    if constexpr (L < Hh) {
      load_step(std::integral_constant<int, L>{});
      if constexpr (L + 1 < Hh) load_step(std::integral_constant<int, L + 1>{});
      if constexpr (L + 2 < Hh) load_step(std::integral_constant<int, L + 2>{});
      if constexpr (L + 3 < Hh) load_step(std::integral_constant<int, L + 3>{});
      static_assert(Hh - L <= 4, "at most four steps per tap");
    }
  };
  auto load_tile = [&](int t) {         // all at once (the first tile)
    load_begin(t, true);
    load_steps(std::integral_constant<int, 0>{}, std::integral_constant<int, (NS > 4 ? 4 : NS)>{});
    if constexpr (NS > 4) load_steps(std::integral_constant<int, 4>{}, std::integral_constant<int, (NS > 8 ? 8 : NS)>{});
    if constexpr (NS > 8) load_steps(std::integral_constant<int, 8>{}, std::integral_constant<int, (NS > 12 ? 12 : NS)>{});
    if constexpr (NS > 12) load_steps(std::integral_constant<int, 12>{}, std::integral_constant<int, NS>{});
    static_assert(NS <= 16, "staging steps");
  };

  auto store_tile = [&]() {
    if constexpr (YLDS) {
      // the direct-to-LDS loads of this wave have landed (each lane reads only its own slots)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < XG; ++k)
        if (has_coef && ((okg >> k) & 1u)) ry[k] = *(const V16*)(yl + (tid + k * NT) * 16);
    }
    if (has_coef) {
      float cA[VEC], cB[VEC], cC[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        cA[j] = ctab[cg + j];
        cB[j] = ctab[COP + cg + j];
        cC[j] = ctab[2 * COP + cg + j];
      }
#pragma unroll
      for (int k = 0; k < XG; ++k) {
        if ((okg >> k) & 1u) {          // outside the image the gradient is zero, not B*0+C
          float fz[VEC], fy[VEC];
          v16_unpack<T>(rz[k], fz);
          v16_unpack<T>(ry[k], fy);
#pragma unroll
          for (int j = 0; j < VEC; ++j) fz[j] = fmaf(cA[j], fz[j], fmaf(cB[j], fy[j], cC[j]));
          rz[k] = v16_pack<T>(fz);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < XG; ++k) {
      const int idx = tid + k * NT;
      if (idx < GVECS) *(V16*)(gl + lofg[k]) = ((okg >> k) & 1u) ? rz[k] : v16_zero();
    }
    if constexpr (XRAW) {
      if (xr_store) {
#pragma unroll
        for (int k = 0; k < XA; ++k) {
          const int idx = tid + k * NT;
          if (idx < AVECS) *(V16*)(xrl + lofa[k]) = rx[k];      // (outside the image: zeros out of the descriptor)
        }
      }
    }
    if (has_aff || in_relu) {
      float sc[VEC], sh[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        sc[j] = ctab[3 * COP + va * VEC + j];
        sh[j] = ctab[3 * COP + CB + va * VEC + j];
      }
#pragma unroll
      for (int k = 0; k < XA; ++k) {
        if ((oka >> k) & 1u) {          // zero padding is applied AFTER the transform, as the forward did
          float f[VEC];
          v16_unpack<T>(rx[k], f);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            f[j] = fmaf(f[j], sc[j], sh[j]);
            if (in_relu) f[j] = f[j] > 0.f ? f[j] : 0.f;
          }
          rx[k] = v16_pack<T>(f);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < XA; ++k) {
      const int idx = tid + k * NT;
      if (idx < AVECS) *(V16*)(al + lofa[k]) = ((oka >> k) & 1u) ? rx[k] : v16_zero();
    }
  };

  // ---- per-lane operand offsets ----
  // input gradient D[ci 32][pixel 256]: a wave owns 32 pixels (2 fragments) x all 32 channels (2 fragments);
  // a lane ends up with 8 contiguous channels of one pixel per pixel fragment
  int aoff[2], boff[FP][3][NKK], moff[FP][8 * ES / 16];
#pragma unroll
  for (int fc = 0; fc < 2; ++fc) aoff[fc] = (fc * 16 + li) * 64 + ((lg ^ lds_swz<64>(fc * 16 + li)) << 4);
#pragma unroll
  for (int fp = 0; fp < FP; ++fp) {
    const int p = wave * PW + fp * 16 + li;
    const int Pb = (p / TW) * HP + (p % TW);          // halo slot of the pixel's tap (0, 0) neighbour
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk)
        boff[fp][dx][kk] = (Pb + dx) * RBG + (((kk * 4 + lg) ^ lds_swz<RBG, SWZ>(p % TW + dx)) << 4);
    // the staged input a at the pixel itself, this lane's 8 channels (the ReLU mask of the epilogue)
#pragma unroll
    for (int q = 0; q < 8 * ES / 16; ++q)
      moff[fp][q] = (Pb + HP + 1) * RBA + (((lg * (8 * ES / 16) + q) ^ lds_swz<RBA, SWZ>(p % TW + 1)) << 4);
  }
  // weight gradient: this wave's k-group and blocks (co-fragments cof0 .. cof0 + BPW - 1 of ci-fragment cif)
  const int kh = wave / NWB, wb = wave % NWB;
  const int cof0 = (wb * BPW) % FCO, cif = (wb * BPW) / FCO;
  constexpr int KROWS = KSTEP / TW;            // tile rows one k-step spans
  // LDS byte offsets of this lane group's first pixel in the wave's first k-step: g at the pixel itself (the centre
  // of the halo image), a at its tap (0, 0) neighbour; later k-steps and taps add compile-time constants
  // bf16: lane group lg reads pixels 4 lg .. 4 lg + 3 of the k-step's first tile row (first transposing read) and of
  // its second row (second read): the pairing of pixels with k indices is free as long as both operands use the same
  // one, and this one keeps the 32-lane halves of a read on 8 consecutive halo slots (lds_swz). f32: 4 lg .. 4 lg + 3.
  const int wq = (lane & 15) >> 2, wp4 = lane & 3;
  const int wPA = (kh * KROWS + 1) * HP + 1 + 4 * lg + (ES == 2 ? wq : 0);   // g: the pixel itself (image centre)
  const int wPB = (kh * KROWS) * HP + 4 * lg + (ES == 2 ? wq : 0);           // a: its tap (0, 0) neighbour
  int wao[BPW], wbo[3];
#pragma unroll
  for (int q = 0; q < BPW; ++q)
    wao[q] = wPA * RBG + ((((cof0 + q) * 2 + (wp4 >> 1)) ^ lds_swz<RBG, SWZ>(1 + 4 * lg + wq)) << 4) + (wp4 & 1) * 8;
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
    wbo[dx] = (wPB + dx) * RBA + (((cif * 2 + (wp4 >> 1)) ^ lds_swz<RBA, SWZ>(4 * lg + wq + dx)) << 4) + (wp4 & 1) * 8;
  // one 16-channel x KSTEP-pixel fragment out of a swizzled image. bf16: `off` = wao / wbo entry + row displacement;
  // f32: P0 = the lane group's first pixel (four one-float reads, the swizzle recomputed per pixel)
  auto tr_frag = [&](const char* img, auto rbc, int off, int P0, int cfrag) -> V16 {
    constexpr int RB = decltype(rbc)::value;
    if constexpr (ES == 2) {
      const LDS_AS char* l = (const LDS_AS char*)img;
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(l + off));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(l + off + HP * RB));
      const bf16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      return __builtin_bit_cast(V16, v);
    } else {
      uint32_t r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        r[j] = *(const uint32_t*)(img + (P0 + j) * RB + (((cfrag * 4 + (li >> 2)) ^ lds_swz<RB, SWZ>(P0 + j)) << 4) + (li & 3) * 4);
      return V16{r[0], r[1], r[2], r[3]};
    }
  };
  f32x4 accw[BPW][9];
#pragma unroll
  for (int q = 0; q < BPW; ++q)
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) accw[q][tp] = f32x4{0.f, 0.f, 0.f, 0.f};
  float s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s1[k] = s2[k] = 0.f;

  const int ci0 = c0 + lg * 8;             // this lane's 8 output channels of the input gradient
  const bool ci_ok = ci0 < a.Cin;
  constexpr int EV = 8 * ES / 16;          // 16-byte vectors of 8 channels: 1 (bf16) or 2 (f32)
  // epilogue operands and the output through buffer descriptors too (an absent operand: an empty descriptor)
  const auto rad = __builtin_amdgcn_make_buffer_rsrc((void*)(a.addend ? a.addend : a.x), 0, a.addend ? (int)a.a_bytes : 0, 0x00020000);
  const auto rbs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bs_y ? a.bs_y : a.x), 0, (a.bs_y && !bs_from_x) ? (int)a.a_bytes : 0, 0x00020000);
  const auto rdo = __builtin_amdgcn_make_buffer_rsrc((void*)a.dx, 0, (int)a.a_bytes, 0x00020000);

  // MFMAs / LDS-read instructions behind one fragment (sched_group_barrier counts)
  constexpr int MF = ES == 2 ? 1 : 4;          // mma16<T>: one 16x16x32 bf16 MFMA or four 16x16x4 f32 ones
  constexpr int RDF = ES == 2 ? 2 : 4;         // trl<T>: two ds_read_b64_tr_b16 or four ds_read_b32

  int t = split;
  if (t < a.total_tiles) load_tile(t);
  if HR_ABLATE(a, 8) { okg = oka = 0; }
  FSTAMP();
  for (; t < a.total_tiles; t += a.nsplit) {
    FSTAMP();
    store_tile();
    FSTAMP();
    __syncthreads();
    FSTAMP();
    const bool has_next = t + a.nsplit < a.total_tiles && !HR_ABLATE(a, 8);
    load_begin(t + a.nsplit, has_next);
    int n, ty, tx;
    tile_of(t, n, ty, tx);
    n = __builtin_amdgcn_readfirstlane(n); ty = __builtin_amdgcn_readfirstlane(ty); tx = __builtin_amdgcn_readfirstlane(tx);
    // epilogue operands of this tile (addend, next BatchNorm's raw input), fetched before the matrix work
    V16 pa[FP][EV], pb[FP][EV];
    bool pok[FP];
    unsigned poff[FP];
#pragma unroll
    for (int fp = 0; fp < FP; ++fp) {
      const int p = wave * PW + fp * 16 + li;
      const int oy = ty * TH + p / TW, ox = tx * TW + p % TW;
      pok[fp] = ci_ok && oy < a.H && ox < a.W && !HR_ABLATE(a, 4);
      poff[fp] = pok[fp] ? (unsigned)(((n * a.H + oy) * a.W + ox) * pixA + ci0 * ES) : OOB;
#pragma unroll
      for (int q = 0; q < EV; ++q) {
        pa[fp][q] = __builtin_amdgcn_raw_buffer_load_b128(rad, pok[fp] ? poff[fp] + q * 16 : OOB, 0, 0);
        pb[fp][q] = __builtin_amdgcn_raw_buffer_load_b128(rbs, pok[fp] ? poff[fp] + q * 16 : OOB, 0, 0);
      }
    }

    FSTAMP();
    // ---- input gradient: conv of g with the transposed, flipped kernel. Steps (tap, 32-channel k-chunk); the
    // operands of step s + 1 are read from LDS before the MFMAs of step s are issued (two register sets) ----
    f32x4 accd[2][FP];
#pragma unroll
    for (int fc = 0; fc < 2; ++fc)
#pragma unroll
      for (int fp = 0; fp < FP; ++fp) accd[fc][fp] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      constexpr int NDS = 9 * NKK;
      // (DPF steps ahead: an LDS read returns after ~150-200 cycles under eight reading waves, a step's MFMAs issue in 64)
      constexpr int DPF = COP <= 32 ? 2 : 1, DNB = DPF + 1;
      V16 daf[DNB][2], dbf[DNB][FP];
      auto rd_d = [&](int s_, int buf) {
        const int tp = s_ / NKK, kk = s_ % NKK;
#pragma unroll
        for (int fc = 0; fc < 2; ++fc) daf[buf][fc] = *(const V16*)(wl + aoff[fc] + (tp * NKK + kk) * (CB * 64));
#pragma unroll
        for (int fp = 0; fp < FP; ++fp) dbf[buf][fp] = *(const V16*)(gl + boff[fp][tp % 3][kk] + (tp / 3) * (HP * RBG));
      };
#pragma unroll
      for (int s_ = 0; s_ < DPF; ++s_) rd_d(s_, s_ % DNB);
      __builtin_amdgcn_sched_group_barrier(0x100, DPF * (2 + FP), 0);      // (the pipeline's first reads are a group of their own)
#pragma unroll
      for (int s_ = 0; s_ < NDS; ++s_) {
        if (s_ + DPF < NDS) rd_d(s_ + DPF, (s_ + DPF) % DNB);
        if (!HR_ABLATE(a, 1)) {
#pragma unroll
          for (int fc = 0; fc < 2; ++fc)
#pragma unroll
            for (int fp = 0; fp < FP; ++fp) accd[fc][fp] = mma16<T>(daf[s_ % DNB][fc], dbf[s_ % DNB][fp], accd[fc][fp]);
        }
        if (s_ + DPF < NDS) __builtin_amdgcn_sched_group_barrier(0x100, 2 + FP, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * FP * MF, 0);
        if ((s_ + 1) % NKK == 0) {
          // this tap's share of the next tile's loads; vector-memory instructions may not move across (everything
          // else may): the requests enter the memory pipe evenly while the matrix pipe works
          __builtin_amdgcn_sched_barrier(0x38F);
          if (s_ / NKK < NSLOT) switch (s_ / NKK) {
#define LS(SL) case SL: load_steps(std::integral_constant<int, (SL * NS) / NSLOT>{}, std::integral_constant<int, ((SL + 1) * NS) / NSLOT>{}); break;
            LS(0) LS(1) LS(2) LS(3) LS(4) LS(5) LS(6) LS(7) LS(8)
#undef LS
          }
          __builtin_amdgcn_sched_barrier(0x38F);
        }
      }
    }

    FSTAMP();
    // ---- weight gradient: pixels are the contraction index; both operands come transposed out of the images
    // staged above (centre pixels of g, tap-shifted pixels of a). Steps (k-step, tap): one B fragment (a, this
    // wave's ci-fragment) per step in a ring RB - 1 steps ahead of its MFMAs, the A fragments (g, the wave's
    // co-fragments) of the next k-step half a k-step ahead ----
    {
      constexpr int NWS = NKW * 9, RB = COP <= 32 ? 8 : 6;
      V16 waf[2][BPW], wbf[RB];
      auto rd_wa = [&](int i, int buf) {
#pragma unroll
        for (int q = 0; q < BPW; ++q)
          waf[buf][q] = tr_frag(gl, std::integral_constant<int, RBG>{}, wao[q] + i * (KSPLIT * KROWS * HP * RBG),
                                wPA + i * (KSPLIT * KROWS * HP), cof0 + q);
      };
      auto rd_wb = [&](int s_) {
        const int i = s_ / 9, tp = s_ % 9;
        wbf[s_ % RB] = tr_frag(al, std::integral_constant<int, RBA>{}, wbo[tp % 3] + (i * KSPLIT * KROWS + tp / 3) * (HP * RBA),
                              wPB + (i * KSPLIT * KROWS + tp / 3) * HP + tp % 3, cif);
      };
      rd_wa(0, 0);
#pragma unroll
      for (int s_ = 0; s_ < RB - 1; ++s_) rd_wb(s_);
      __builtin_amdgcn_sched_group_barrier(0x100, (BPW + RB - 1) * RDF, 0);
#pragma unroll
      for (int s_ = 0; s_ < NWS; ++s_) {
        const int i = s_ / 9, tp = s_ % 9;
        if (s_ + RB - 1 < NWS) rd_wb(s_ + RB - 1);
        if (tp == 4 && i + 1 < NKW) rd_wa(i + 1, (i + 1) & 1);
        if (!HR_ABLATE(a, 2)) {
#pragma unroll
          for (int q = 0; q < BPW; ++q) accw[q][tp] = mma16<T>(waf[i & 1][q], wbf[s_ % RB], accw[q][tp]);
        }
        {
          constexpr int dummy = 0; (void)dummy;
          const int nrd = (s_ + RB - 1 < NWS ? RDF : 0) + ((tp == 4 && i + 1 < NKW) ? BPW * RDF : 0);
          // (the counts are compile-time constants once the loop is unrolled; the builtin needs literals)
          switch (nrd) {
            case 2: __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); break;
            case 10: __builtin_amdgcn_sched_group_barrier(0x100, 10, 0); break;
            case 16: __builtin_amdgcn_sched_group_barrier(0x100, 16, 0); break;
            case 4: __builtin_amdgcn_sched_group_barrier(0x100, 4, 0); break;
            case 6: __builtin_amdgcn_sched_group_barrier(0x100, 6, 0); break;
            case 8: __builtin_amdgcn_sched_group_barrier(0x100, 8, 0); break;
            case 12: __builtin_amdgcn_sched_group_barrier(0x100, 12, 0); break;
            default: break;
          }
          __builtin_amdgcn_sched_group_barrier(0x008, BPW * MF, 0);
        }
        if constexpr (WSLOTS) {
          if (tp == 8) {
            __builtin_amdgcn_sched_barrier(0x38F);
            switch (i) {
#define LS(SL) case SL - 9: load_steps(std::integral_constant<int, (SL * NS) / NSLOT>{}, std::integral_constant<int, ((SL + 1) * NS) / NSLOT>{}); break;
              LS(9) LS(10) LS(11) LS(12) LS(13) LS(14) LS(15) LS(16)
#undef LS
            }
            __builtin_amdgcn_sched_barrier(0x38F);
          }
        }
      }
    }

    FSTAMP();
    // ---- tile epilogue: residual addend, ReLU mask from the staged input image, store, statistics ----
#pragma unroll
    for (int fp = 0; fp < FP; ++fp) {
      const int p = wave * PW + fp * 16 + li;
      float v[8];
      v[0] = accd[0][fp].x; v[1] = accd[0][fp].y; v[2] = accd[0][fp].z; v[3] = accd[0][fp].w;
      v[4] = accd[1][fp].x; v[5] = accd[1][fp].y; v[6] = accd[1][fp].z; v[7] = accd[1][fp].w;
      {
        float ad[8];               // (no addend: zeros out of the empty descriptor)
#pragma unroll
        for (int q = 0; q < EV; ++q) v16_unpack<T>(pa[fp][q], ad + q * VEC);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += ad[k];
      }
      if (a.mask_out) {
        float am[8];
#pragma unroll
        for (int q = 0; q < EV; ++q) v16_unpack<T>(*(const V16*)(al + moff[fp][q]), am + q * VEC);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = am[k] > 0.f ? v[k] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < EV; ++q)
        __builtin_amdgcn_raw_buffer_store_b128(v16_pack<T>(v + q * VEC), rdo, pok[fp] ? poff[fp] + q * 16 : OOB, 0, 0);
      if (a.rows) {
        float yb[8];
        if (bs_from_x) {
#pragma unroll
          for (int q = 0; q < EV; ++q) v16_unpack<T>(*(const V16*)(bsl + moff[fp][q]), yb + q * VEC);
        } else {
#pragma unroll
          for (int q = 0; q < EV; ++q) v16_unpack<T>(pb[fp][q], yb + q * VEC);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float vv = pok[fp] ? v[k] : 0.f;
          s1[k] += vv;
          hr_fma_acc(s2[k], vv, yb[k]);
        }
      }
    }
    FSTAMP();
    __syncthreads();   // the images are free again
  }
  FSTAMP();

  // ---- backward statistics: lanes -> waves -> one row per workgroup (deterministic) ----
  if (a.rows) {
    float* sl = (float*)lds;   // [NW waves][2][32]
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s1[k] = wave_sum16(s1[k]);
      s2[k] = wave_sum16(s2[k]);
    }
    if (li == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        sl[(wave * 2 + 0) * CB + lg * 8 + k] = s1[k];
        sl[(wave * 2 + 1) * CB + lg * 8 + k] = s2[k];
      }
    }
    __syncthreads();
    if (tid < 2 * CB) {
      const int which = tid / CB, cl = tid % CB;
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < NW; ++q) s += sl[(q * 2 + which) * CB + cl];
      if (c0 + cl < a.Cin) a.rows[((size_t)split * 2 + which) * a.Cin + c0 + cl] = s;
    }
  }

  FSTAMP();
  // ---- the k-groups' partial weight gradients meet in LDS, in a fixed order: group 0 keeps the total ----
  if constexpr (KSPLIT > 1) {
    // one group per round (a round's exchange is NWB * BPW * 9 KB: all groups at once would not fit beside nothing)
    static_assert(NWB * BPW * 9 * 1024 <= GBYTES + ABYTES + WBYTES, "one k-group's exchange fits the LDS images");
    f32x4* kl = (f32x4*)lds;  // [NWB][BPW * 9][64 lanes]
#pragma unroll
    for (int g = 1; g < KSPLIT; ++g) {
      __syncthreads();        // (the statistics reduction / the previous round read the same bytes)
      if (kh == g) {
#pragma unroll
        for (int q = 0; q < BPW; ++q)
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) kl[((wb * BPW + q) * 9 + tp) * 64 + lane] = accw[q][tp];
      }
      __syncthreads();
      if (kh == 0) {
#pragma unroll
        for (int q = 0; q < BPW; ++q)
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) {
            const f32x4 o = kl[((wb * BPW + q) * 9 + tp) * 64 + lane];
            accw[q][tp] = f32x4{accw[q][tp].x + o.x, accw[q][tp].y + o.y, accw[q][tp].z + o.z, accw[q][tp].w + o.w};
          }
      }
    }
  }
  const bool wwriter = kh == 0;
  // ---- weight-gradient slab of this workgroup: slab[split][co][tap][ci], D: col = ci, row = co ----
  if (a.atomic) {
    // [co][ci][tap] tiles of 16 output channels through LDS (the walk ended on a barrier: the images are free)
    __syncthreads();          // (the reductions above read their scratch in the same bytes)
    float* tl = (float*)lds;
    static_assert(16 * CB * 9 * 4 <= GBYTES + ABYTES + WBYTES, "weight-gradient tile fits the LDS images");
#pragma unroll
    for (int h = 0; h < FCO; ++h) {
#pragma unroll
      for (int q = 0; q < BPW; ++q) {
        if (wwriter && cof0 + q == h) {
          const int cil = cif * 16 + li;
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) {
            const float v4[4] = {accw[q][tp].x, accw[q][tp].y, accw[q][tp].z, accw[q][tp].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) tl[((lg * 4 + r) * CB + cil) * 9 + tp] = v4[r];
          }
        }
      }
      __syncthreads();
      const int nci = min(CB, a.Cin_real - c0);
      for (int idx = tid; idx < 16 * CB * 9; idx += NT) {
        const int col = idx / (CB * 9), rem = idx - col * (CB * 9);
        const int co = h * 16 + col;
        if (co < a.Cout_real && rem < nci * 9) atomicAdd(a.slabs + ((size_t)co * a.Cin_real + c0) * 9 + rem, tl[idx]);
      }
      if (h + 1 < FCO) __syncthreads();
    }
  } else if (wwriter) {
    float* slab = a.slabs + (size_t)split * a.Cout * 9 * a.Cin;
    const int ci = c0 + cif * 16 + li;
#pragma unroll
    for (int q = 0; q < BPW; ++q) {
      const int co = (cof0 + q) * 16 + lg * 4;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const float v4[4] = {accw[q][tp].x, accw[q][tp].y, accw[q][tp].z, accw[q][tp].w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (co + r < a.Cout && ci < a.Cin) slab[((size_t)(co + r) * 9 + tp) * a.Cin + ci] = v4[r];
      }
    }
  }
  FSTAMP();
}

struct FusedCfg {
  int cop, th, nw, wco, wn, ylds, per_cu;   // per_cu: co-resident workgroups per CU (LDS-limited)
};

// Variant: 0 = 16x16 tiles, 8 waves, one workgroup per CU; 1 = 16x16 tiles, 4 waves, two per CU;
// 2 = 8x16 tiles, 4 waves, two or three per CU. (HRNET_FUSED_VARIANT overrides the default for measurements.)
inline int fused_variant(int cop) {
  static const int v = hr_knob("HRNET_FUSED_VARIANT", -1);
  static const int v32 = hr_knob("HRNET_FUSED_V32", v);
  static const int v64 = hr_knob("HRNET_FUSED_V64", v);
  return cop <= 32 ? v32 : v64;
}

inline FusedCfg fused_cfg(int dtype, int Cout) {
  // 0 in .cop = shape not served by the fused kernel (LDS: 160 KB per CU)
  const int var = fused_variant(Cout <= 32 ? 32 : 64);
  if (dtype == HR_F32) return Cout <= 32 ? FusedCfg{32, 16, 8, 1, 8, 0, 1} : FusedCfg{0, 0, 0, 0, 0, 0, 0};
  if (Cout <= 32) {
    if (var == 1) return FusedCfg{32, 16, 4, 1, 4, 0, 2};
    if (var == 2) return FusedCfg{32, 8, 4, 1, 4, 0, 2};   // (188 VGPRs: two 4-wave workgroups per CU)
    // (round 4: 32x16 pixel tiles for this instantiation - half the barriers and staging waits per pixel - take 256 VGPRs
    // + 112 bytes of scratch and 97 KB of LDS: 16.5 against 15.24 ms/step)
    return FusedCfg{32, 16, 8, 1, 8, 0, 1};
  }
  if (Cout <= 64) {
    if (var == 2) return FusedCfg{64, 8, 4, 2, 2, 0, 2};
    return FusedCfg{64, 16, 8, 2, 4, 0, 1};
  }
  // 128 output-side channels (the third branch: 16x16 maps): 8x16 tiles so that the g image (49 KB), the 32-channel
  // input image (14 KB) and the 32 x 9 x 128 weights (74 KB) share one CU's LDS
  if (Cout <= 128) return FusedCfg{128, 8, 8, 2, 4, 0, 1};
  return FusedCfg{0, 0, 0, 0, 0, 0, 0};
}

}  // namespace

// 1 if hrnet_conv3x3_bwd_fused serves this layer (3x3 stride 1, Cout <= 128 for bf16 / <= 32 for f32)
extern "C" int hrnet_bwd_fused_supported(int dtype, int Cin, int Cout) {
  if (dtype != HR_F32 && dtype != HR_BF16) return 0;
  if (Cin % (dtype == HR_F32 ? 4 : 8) != 0 || Cout % 16 != 0 || Cin <= 0 || Cout <= 0) return 0;
  return fused_cfg(dtype, Cout).cop != 0 ? 1 : 0;
}

// number of slabs / statistics rows: one per walk of pixel tiles (a walk is shared by the ceil(Cin/32)
// workgroups that own its input-channel blocks); the grid fills the CUs once
extern "C" int hrnet_bwd_fused_splits(int dtype, int N, int H, int W, int Cin, int Cout) {
  const FusedCfg c = fused_cfg(dtype, Cout);
  if (!c.cop) return 0;
  const int tiles = N * ((H + c.th - 1) / c.th) * ((W + 15) / 16);
  const int ncb = (Cin + 31) / 32;
  // half the CUs: measured on MI355X inside the training step (4 lanes in flight), 128 workgroups per launch
  // beat 256 by 0.6 ms/step (20.6 vs 21.3 ms; 64: 22.0, 96: 22.0, 160: 20.7) - a grid that takes every CU with a
  // 100 KB / 256-VGPR workgroup locks the other lanes' kernels out - and it halves the slab traffic
  static const int cus = hr_knob("HRNET_FUSED_CUS", 128);
  // ... except where the launch is the whole step for a while: the 64-channel 3x3 conv of a layer1 Bottleneck
  // (64x64 maps: 2048 workgroup-tiles at batch 64) sits in the single-lane tail of the backward pass and takes
  // every CU (165 us on 128 CUs)
  static const int big = hr_knob("HRNET_FUSED_CUS_BIG", 256);
  // the 128-channel instantiation (w48's 96-channel branch: three input-channel blocks per walk): 96 workgroups =
  // 32 walks of 18 tiles; 28.3 against 28.8 ms/step with 128 (36 walks), 29.2 with 80, 29.0 with 192 / 256
  static const int cus128 = hr_knob("HRNET_FUSED_CUS_128", 96);
  // (measurement: separate grids for the 32- and 64-channel instantiations, which run side by side on two lanes)
  static const int cus32 = hr_knob("HRNET_FUSED_CUS32", cus);
  static const int cus64 = hr_knob("HRNET_FUSED_CUS64", cus);
  const int small = c.cop == 32 ? cus32 : cus64;
  int ns = (c.cop == 128 ? cus128 : (long long)tiles * ncb * c.th >= 2048 * 16 ? big : small) * c.per_cu / ncb;
  if (ns < 1) ns = 1;
  if (ns > tiles) ns = tiles;
  // even walks: every split takes the same number of tiles when possible
  int even = ns;
  while (even > 1 && tiles % even != 0) --even;
  if (even * 2 > ns) ns = even;
  return ns;
}

// name of the instantiation hrnet_conv3x3_bwd_fused launches, as rocprofv3 demangles it (returns its length)
extern "C" int hrnet_bwd_fused_kernel_name(int dtype, int Cin, int Cout, char* buf, int buflen) {
  (void)Cin;
  const FusedCfg c = fused_cfg(dtype, Cout);
  return snprintf(buf, buflen, "bwd_fused_kernel<%s, %d, %d, %d, %d, %d, %s>", dtype == HR_F32 ? "float" : "__bf16", c.cop,
                  c.th, c.nw, c.wco, c.wn, c.ylds ? "true" : "false");
}

extern "C" int hrnet_conv3x3_bwd_fused(int dtype, const void* dz, const void* y, const float* coef, const void* x,
                                       const float* in_scale, const float* in_shift, int in_relu, const void* wT,
                                       void* dx, const void* addend, int mask_out, float* rows, const void* bs_y,
                                       float* slabs, int N, int H, int W, int Cin, int Cout, hr_stream_t stream) {
  return hrnet_conv3x3_bwd_fused_bnref(dtype, dz, y, coef, nullptr, x, in_scale, in_shift, in_relu, wT, dx, addend,
                                       mask_out, rows, bs_y, slabs, N, H, W, Cin, Cout, stream);
}

static int bwd_fused_launch(int dtype, const void* dz, const void* y, const float* coef, const HrBnBwdRef* ref,
                            const void* x, const float* in_scale, const float* in_shift, int in_relu, const void* wT,
                            void* dx, const void* addend, int mask_out, float* rows, const void* bs_y, float* slabs,
                            int N, int H, int W, int Cin, int Cout, int atomic, int Cout_real, int Cin_real,
                            hr_stream_t stream);

extern "C" int hrnet_conv3x3_bwd_fused_bnref(int dtype, const void* dz, const void* y, const float* coef,
                                             const HrBnBwdRef* ref, const void* x, const float* in_scale,
                                             const float* in_shift, int in_relu, const void* wT, void* dx,
                                             const void* addend, int mask_out, float* rows, const void* bs_y,
                                             float* slabs, int N, int H, int W, int Cin, int Cout,
                                             hr_stream_t stream) {
  return bwd_fused_launch(dtype, dz, y, coef, ref, x, in_scale, in_shift, in_relu, wT, dx, addend, mask_out, rows, bs_y,
                          slabs, N, H, W, Cin, Cout, 0, 0, 0, stream);
}

static int bwd_fused_launch(int dtype, const void* dz, const void* y, const float* coef, const HrBnBwdRef* ref,
                            const void* x, const float* in_scale, const float* in_shift, int in_relu, const void* wT,
                            void* dx, const void* addend, int mask_out, float* rows, const void* bs_y, float* slabs,
                            int N, int H, int W, int Cin, int Cout, int atomic, int Cout_real, int Cin_real,
                            hr_stream_t stream) {
  HR_REQUIRE(!atomic || (Cout_real >= 1 && Cout_real <= Cout && Cin_real >= 1 && Cin_real <= Cin),
             "bwd_fused: the atomic form needs the gradient's real channel counts (%d, %d)", Cout_real, Cin_real);
  HR_REQUIRE(hrnet_bwd_fused_supported(dtype, Cin, Cout), "bwd_fused: dtype %d Cin %d Cout %d not served", dtype, Cin, Cout);
  HR_REQUIRE(dz && x && wT && dx && slabs, "bwd_fused: null pointer");
  HR_REQUIRE((!coef && !ref) || y, "bwd_fused: coef needs y");
  HR_REQUIRE(!ref || (ref->rows && ref->gamma && ref->save_mean && ref->save_invstd && ref->dgamma && ref->dbeta &&
                      ref->nrows >= 1 && (long long)ref->nrows * Cout <= 16384 && ref->count > 0.f),
             "bwd_fused: incomplete HrBnBwdRef (or nrows * Cout > 16384)");
  HR_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "bwd_fused: scale/shift must come together");
  HR_REQUIRE(!bs_y || rows, "bwd_fused: bs_y needs rows");
  HR_REQUIRE(N > 0 && H > 0 && W > 0, "bwd_fused: empty shape");
  HR_REQUIRE((double)N * H * W * (Cin > Cout ? Cin : Cout) * (dtype == HR_F32 ? 4.0 : 2.0) < 2147483648.0,
             "bwd_fused: a tensor of 2 GiB or more (32-bit buffer offsets)");
  const FusedCfg c = fused_cfg(dtype, Cout);
  BwdArgs a;
  a.dz = (const char*)dz; a.y = (const char*)y; a.coef = ref ? nullptr : coef; a.x = (const char*)x;
  if (ref) a.ref = *ref; else a.ref = HrBnBwdRef{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0, 0, 0};
  a.in_scale = in_scale; a.in_shift = in_shift; a.wT = (const char*)wT; a.dx = (char*)dx;
  a.addend = (const char*)addend; a.bs_y = (const char*)bs_y; a.rows = rows; a.slabs = slabs;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.g_bytes = (unsigned)((size_t)N * H * W * Cout * (dtype == HR_F32 ? 4 : 2));
  a.a_bytes = (unsigned)((size_t)N * H * W * Cin * (dtype == HR_F32 ? 4 : 2));
  a.tiles_y = (H + c.th - 1) / c.th; a.tiles_x = (W + 15) / 16; a.total_tiles = N * a.tiles_y * a.tiles_x;
  a.ncb = (Cin + 31) / 32;
  a.nsplit = hrnet_bwd_fused_splits(dtype, N, H, W, Cin, Cout);
  a.in_relu = in_relu; a.mask_out = mask_out;
  a.atomic = atomic; a.Cout_real = Cout_real; a.Cin_real = Cin_real;
#ifdef HR_MEASURE
  { static const char* sp = getenv("HRNET_FUSED_STAMP_PTR"); a.stamp = sp ? (unsigned long long*)strtoull(sp, nullptr, 16) : nullptr; }
  { static const int abl = hr_knob("HRNET_FUSED_ABLATE", 0); a.ablate = abl; }
#endif
  const unsigned grid = (unsigned)((a.nsplit + 7) / 8 * 8 * a.ncb);
  hipStream_t s = (hipStream_t)stream;
#define FUSED(T_, COP_, TH_, NW_, WCO_, WN_, Y_) \
  hipLaunchKernelGGL((bwd_fused_kernel<T_, COP_, TH_, NW_, WCO_, WN_, Y_>), dim3(grid), dim3(NW_ * 64), 0, s, a)
  if (dtype == HR_F32) FUSED(float, 32, 16, 8, 1, 8, false);
  else if (c.cop == 32 && c.nw == 8) FUSED(bf16_t, 32, 16, 8, 1, 8, false);
  else if (c.cop == 32 && c.th == 16) FUSED(bf16_t, 32, 16, 4, 1, 4, false);
  else if (c.cop == 32) FUSED(bf16_t, 32, 8, 4, 1, 4, false);
  else if (c.cop == 128) FUSED(bf16_t, 128, 8, 8, 2, 4, false);
  else if (c.nw == 8) FUSED(bf16_t, 64, 16, 8, 2, 4, false);
  else FUSED(bf16_t, 64, 8, 4, 2, 2, false);
#undef FUSED
  return hr_check_launch("conv3x3_bwd_fused");
}

int hr_launch_bwd_fused(const HrOp& op, hipStream_t s) {
  // p[12]: HOST pointer to a HrBnBwdRef (kept alive by the plan), or NULL; i[8] = 1: p[11] is the OIHW gradient the
  // weight-gradient tiles are ADDED to (float atomics), i[9], i[10] its real Cout, Cin
  return bwd_fused_launch(op.i[0], op.p[0], op.p[1], (const float*)op.p[2], (const HrBnBwdRef*)op.p[12], op.p[3],
                          (const float*)op.p[4], (const float*)op.p[5], op.i[6], op.p[6], op.p[7], op.p[8], op.i[7],
                          (float*)op.p[9], op.p[10], (float*)op.p[11], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5],
                          op.i[8], op.i[9], op.i[10], (hr_stream_t)s);
}
