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
#include <stdio.h>

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
  int N, H, W, Cin;  // stored input
  int Hz, Wz;        // logical input extent (== H, W unless upz)
  int Ho, Wo, Cout;
  int tiles_y, tiles_x;
  int in_relu, upz, accumulate;
};

template <typename T, int KS, int STRIDE, int TH, int TW, int BN, int WP, int WC>
struct ConvCfg {
  static constexpr int VEC = TT<T>::VEC;
  static constexpr int KSTEP = TT<T>::KSTEP;
  static constexpr int KC = KSTEP;                 // channels staged per chunk
  static constexpr int VPP = KC / VEC;             // 16-byte vectors per pixel per chunk (= 4)
  static constexpr int TAPS = KS * KS;
  static constexpr int HALO_H = (TH - 1) * STRIDE + KS;
  static constexpr int HALO_W = (TW - 1) * STRIDE + KS;
  static constexpr int PIXB = KC * (int)sizeof(T) + 16;          // padded pixel stride (bytes)
  static constexpr int WROWB = TAPS * KC * (int)sizeof(T) + 16;  // padded weight-row stride
  static constexpr int XBYTES = HALO_H * HALO_W * PIXB;
  static constexpr int WBYTES = BN * WROWB;
  static constexpr int BM = TH * TW;
  static constexpr int PM = BM / WP;   // pixels per wave
  static constexpr int FP = PM / 16;   // pixel fragments per wave
  static constexpr int CN = BN / WC;   // couts per wave
  static constexpr int FC = CN / 16;   // cout fragments per wave
  static constexpr int LANE_C = 4 * FC;  // contiguous couts per lane
  static constexpr int STATB = WP * BN * 2 * (int)sizeof(float);
  static constexpr int LDSB = (XBYTES + WBYTES) > STATB ? (XBYTES + WBYTES) : STATB;
  static_assert(WP * WC == 4, "4 waves");
  static_assert(PM % 16 == 0 && CN % 16 == 0, "fragment multiples");
  static_assert(VPP == 4, "staging assumes 4 vectors per pixel");
};

template <typename T, int KS, int STRIDE, int TH, int TW, int BN, int WP, int WC>
__global__ __launch_bounds__(256) void conv_kernel(ConvArgs a) {
  using C = ConvCfg<T, KS, STRIDE, TH, TW, BN, WP, WC>;
  constexpr int VEC = C::VEC;
  __shared__ __attribute__((aligned(16))) char lds[C::LDSB];
  char* xl = lds;
  char* wl = lds + C::XBYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wp = wave % WP;
  const int wc = wave / WP;
  const int li = lane & 15;
  const int lg = lane >> 4;

  int bid = blockIdx.x;
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int n = bid / a.tiles_y;
  const int n0 = blockIdx.y * BN;
  constexpr int PAD = KS / 2;
  const int iy0 = ty * TH * STRIDE - PAD;
  const int ix0 = tx * TW * STRIDE - PAD;

  // per-lane LDS byte offsets of the MFMA operands
  int aoff[C::FC];
#pragma unroll
  for (int fc = 0; fc < C::FC; ++fc) {
    const int row = wc * C::CN + (li >> 2) * C::LANE_C + fc * 4 + (li & 3);
    aoff[fc] = row * C::WROWB + lg * 16;
  }
  int boff[C::FP];
#pragma unroll
  for (int fp = 0; fp < C::FP; ++fp) {
    const int p = wp * C::PM + fp * 16 + li;
    const int py = p / TW, px = p % TW;
    boff[fp] = ((py * STRIDE) * C::HALO_W + px * STRIDE) * C::PIXB + lg * 16;
  }

  f32x4 acc[C::FC][C::FP];
#pragma unroll
  for (int fc = 0; fc < C::FC; ++fc)
#pragma unroll
    for (int fp = 0; fp < C::FP; ++fp) acc[fc][fp] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int v = tid & 3;  // this thread's 16-byte vector within a weight tap (fixed)

  for (int c0 = 0; c0 < a.Cin; c0 += C::KC) {
    const int c = c0 + v * VEC;
    const bool cvalid = c < a.Cin;
    stage_halo<T, C::KC, C::HALO_H, C::HALO_W, C::PIXB>(xl, a.x, n, a.H, a.W, a.Cin, a.Hz, a.Wz, iy0,
                                                        ix0, c0, a.in_scale, a.in_shift, a.in_relu,
                                                        a.upz, tid);
    // ---- stage the weight slice [BN][TAPS][KC] ----
    for (int idx = tid; idx < BN * C::TAPS * 4; idx += 256) {
      const int rt = idx >> 2;
      const int t = rt % C::TAPS, r = rt / C::TAPS;
      const int co = n0 + r;
      V16 val = v16_zero();
      if (cvalid && co < a.Cout)
        val = *(const V16*)(a.w + ((size_t)(co * C::TAPS + t) * a.Cin + c) * sizeof(T));
      *(V16*)(wl + r * C::WROWB + (t * C::KC) * (int)sizeof(T) + v * 16) = val;
    }
    __syncthreads();
    // ---- MFMA over the taps of this chunk ----
#pragma unroll
    for (int t = 0; t < C::TAPS; ++t) {
      const int tapb = ((t / KS) * C::HALO_W + (t % KS)) * C::PIXB;
      V16 af[C::FC], bf[C::FP];
#pragma unroll
      for (int fc = 0; fc < C::FC; ++fc)
        af[fc] = *(const V16*)(wl + aoff[fc] + t * C::KC * (int)sizeof(T));
#pragma unroll
      for (int fp = 0; fp < C::FP; ++fp) bf[fp] = *(const V16*)(xl + boff[fp] + tapb);
#pragma unroll
      for (int fc = 0; fc < C::FC; ++fc)
#pragma unroll
        for (int fp = 0; fp < C::FP; ++fp) acc[fc][fp] = mma16<T>(af[fc], bf[fp], acc[fc][fp]);
    }
    __syncthreads();
  }

  // ---- epilogue: bias, (accumulate), store 4*FC contiguous couts per pixel, BN statistics ----
  const int cbase = n0 + wc * C::CN + lg * C::LANE_C;
  const bool cok = cbase < a.Cout;
  float bias[C::LANE_C];
#pragma unroll
  for (int k = 0; k < C::LANE_C; ++k) bias[k] = (a.bias && cok) ? a.bias[cbase + k] : 0.f;
  float s1[C::LANE_C], s2[C::LANE_C];
#pragma unroll
  for (int k = 0; k < C::LANE_C; ++k) s1[k] = s2[k] = 0.f;

#pragma unroll
  for (int fp = 0; fp < C::FP; ++fp) {
    const int p = wp * C::PM + fp * 16 + li;
    const int oy = ty * TH + p / TW, ox = tx * TW + p % TW;
    const bool pok = cok && oy < a.Ho && ox < a.Wo;
    float vals[C::LANE_C];
#pragma unroll
    for (int fc = 0; fc < C::FC; ++fc)
#pragma unroll
      for (int r = 0; r < 4; ++r) vals[fc * 4 + r] = acc[fc][fp][r] + bias[fc * 4 + r];
    if (pok) {
      char* dst = a.y + ((size_t)((n * a.Ho + oy) * a.Wo + ox) * a.Cout + cbase) * sizeof(T);
#pragma unroll
      for (int k = 0; k < C::LANE_C; ++k) {
        s1[k] += vals[k];
        s2[k] += vals[k] * vals[k];
      }
#pragma unroll
      for (int k0 = 0; k0 < C::LANE_C; k0 += VEC) {
        if constexpr (C::LANE_C >= VEC) {
          if (a.accumulate) {
            float old[VEC];
            v16_unpack<T>(*(const V16*)(dst + k0 * sizeof(T)), old);
#pragma unroll
            for (int j = 0; j < VEC; ++j) vals[k0 + j] += old[j];
          }
          *(V16*)(dst + k0 * sizeof(T)) = v16_pack<T>(vals + k0);
        } else {
          // LANE_C == 4 with bf16: one 8-byte store
          bf16x4 o;
          if (a.accumulate) {
            const bf16x4 old = *(const bf16x4*)dst;
#pragma unroll
            for (int j = 0; j < 4; ++j) vals[j] += (float)old[j];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (bf16_t)vals[j];
          *(bf16x4*)dst = o;
        }
      }
    }
  }

  if (a.stats) {
    __syncthreads();  // LDS reuse
    float* sl = (float*)lds;  // [WP][2][BN]
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
      if (n0 + cl < a.Cout)
        a.stats[((size_t)blockIdx.x * 2 + which) * a.Cout + n0 + cl] = s;
    }
  }
}

// ---- tile configuration choice (host) --------------------------------------------------
struct TileChoice {
  int th, tw, bn, id;
};

TileChoice choose_tile(int Ho, int Wo, int Cout, int ks, int stride) {
  // id: 0 = 16x16/BN32 (4x1 waves), 1 = 8x16/BN64 (2x2), 2 = 8x8/BN64 (2x2), 3 = 8x8/BN32 (2x2)
  if (stride == 2) return Cout >= 64 ? TileChoice{8, 8, 64, 2} : TileChoice{8, 8, 32, 3};
  if (Cout <= 32) return (Ho >= 16 && Wo >= 16) ? TileChoice{16, 16, 32, 0} : TileChoice{8, 8, 32, 3};
  if (Wo >= 16 && Ho >= 16) return TileChoice{8, 16, 64, 1};
  return TileChoice{8, 8, 64, 2};
}

template <typename T, int KS, int STRIDE>
int launch_cfg(const ConvArgs& a, const TileChoice& tc, int N, hipStream_t s) {
  dim3 grid((unsigned)(N * a.tiles_y * a.tiles_x), (unsigned)((a.Cout + tc.bn - 1) / tc.bn));
  switch (tc.id) {
    case 0:
      hipLaunchKernelGGL((conv_kernel<T, KS, STRIDE, 16, 16, 32, 4, 1>), grid, dim3(256), 0, s, a);
      break;
    case 1:
      hipLaunchKernelGGL((conv_kernel<T, KS, STRIDE, 8, 16, 64, 2, 2>), grid, dim3(256), 0, s, a);
      break;
    case 2:
      hipLaunchKernelGGL((conv_kernel<T, KS, STRIDE, 8, 8, 64, 2, 2>), grid, dim3(256), 0, s, a);
      break;
    default:
      hipLaunchKernelGGL((conv_kernel<T, KS, STRIDE, 8, 8, 32, 2, 2>), grid, dim3(256), 0, s, a);
      break;
  }
  return hr_check_launch("conv2d");
}

template <typename T>
int launch_t(const ConvArgs& a, const TileChoice& tc, int N, int ks, int stride, hipStream_t s) {
  if (ks == 1) return launch_cfg<T, 1, 1>(a, tc, N, s);
  if (stride == 1) return launch_cfg<T, 3, 1>(a, tc, N, s);
  return launch_cfg<T, 3, 2>(a, tc, N, s);
}

}  // namespace

extern "C" int hrnet_conv_tiles(int N, int Ho, int Wo, int Cout, int ks, int stride) {
  const TileChoice tc = choose_tile(Ho, Wo, Cout, ks, stride);
  return N * ((Ho + tc.th - 1) / tc.th) * ((Wo + tc.tw - 1) / tc.tw);
}

int hr_launch_conv(const HrOp& op, hipStream_t s) {
  const int dtype = op.i[0], N = op.i[1], H = op.i[2], W = op.i[3], Cin = op.i[4], Ho = op.i[5],
            Wo = op.i[6], Cout = op.i[7], ks = op.i[8], upz = op.i[10];
  int stride = op.i[9];
  HR_REQUIRE(dtype == HR_F32 || dtype == HR_BF16, "conv2d: bad dtype %d", dtype);
  HR_REQUIRE(ks == 1 || ks == 3, "conv2d: kernel size %d not supported (1 or 3)", ks);
  HR_REQUIRE(stride == 1 || stride == 2, "conv2d: stride %d not supported", stride);
  HR_REQUIRE(!(ks == 1 && stride != 1), "conv2d: 1x1 stride-2 not supported");
  HR_REQUIRE(Cin % (dtype == HR_F32 ? 4 : 8) == 0, "conv2d: Cin=%d not a 16-byte multiple", Cin);
  HR_REQUIRE(Cout % 16 == 0, "conv2d: Cout=%d must be a multiple of 16 (pad the tensor)", Cout);
  HR_REQUIRE(N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "conv2d: empty shape");
  HR_REQUIRE(op.p[0] && op.p[1] && op.p[5], "conv2d: null tensor pointer");
  HR_REQUIRE((op.p[2] == nullptr) == (op.p[3] == nullptr), "conv2d: scale/shift must come together");
  ConvArgs a;
  a.x = (const char*)op.p[0];
  a.w = (const char*)op.p[1];
  a.in_scale = (const float*)op.p[2];
  a.in_shift = (const float*)op.p[3];
  a.bias = (const float*)op.p[4];
  a.y = (char*)op.p[5];
  a.stats = (float*)op.p[6];
  a.N = N; a.H = H; a.W = W; a.Cin = Cin;
  a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
  a.in_relu = op.i[11]; a.upz = upz; a.accumulate = op.i[12];
  if (upz) {
    HR_REQUIRE(ks == 3, "conv2d: upz needs a 3x3 kernel");
    HR_REQUIRE((Ho + 1) / 2 == H && (Wo + 1) / 2 == W, "conv2d: upz shape mismatch");
    stride = 1;
    a.Hz = Ho; a.Wz = Wo;
  } else {
    const int pad = ks / 2;
    HR_REQUIRE((H + 2 * pad - ks) / stride + 1 == Ho && (W + 2 * pad - ks) / stride + 1 == Wo,
               "conv2d: output %dx%d does not match input %dx%d ks=%d stride=%d", Ho, Wo, H, W, ks, stride);
    a.Hz = H; a.Wz = W;
  }
  const TileChoice tc = choose_tile(Ho, Wo, Cout, ks, op.i[9]);
  a.tiles_y = (Ho + tc.th - 1) / tc.th;
  a.tiles_x = (Wo + tc.tw - 1) / tc.tw;
  // the tile choice is keyed on the ORIGINAL stride so hrnet_conv_tiles() agrees; a upz conv
  // runs the stride-1 kernel with that tile
  if (dtype == HR_F32) return launch_t<float>(a, tc, N, ks, stride, s);
  return launch_t<bf16_t>(a, tc, N, ks, stride, s);
}

extern "C" int hrnet_conv2d(int dtype, const void* x, const void* w, const float* in_scale,
                            const float* in_shift, const float* bias, void* y, float* stats, int N,
                            int H, int W, int Cin, int Ho, int Wo, int Cout, int ks, int stride,
                            int upz, int in_relu, int accumulate, hr_stream_t stream) {
  HrOp op = {};
  op.kind = HR_OP_CONV;
  const int iv[13] = {dtype, N, H, W, Cin, Ho, Wo, Cout, ks, stride, upz, in_relu, accumulate};
  for (int k = 0; k < 13; ++k) op.i[k] = iv[k];
  op.p[0] = (void*)x; op.p[1] = (void*)w; op.p[2] = (void*)in_scale; op.p[3] = (void*)in_shift;
  op.p[4] = (void*)bias; op.p[5] = y; op.p[6] = stats;
  return hr_launch_conv(op, (hipStream_t)stream);
}

// Demangled-style name of the kernel instantiation hrnet_conv2d launches for this shape (so that
// bench.py's per-kernel timings can be matched against rocprofv3's kernel trace).
extern "C" int hrnet_conv_kernel_name(int dtype, int Ho, int Wo, int Cout, int ks, int stride, int upz,
                                      char* buf, int buflen) {
  const TileChoice tc = choose_tile(Ho, Wo, Cout, ks, stride);
  static const int wp[4] = {4, 2, 2, 2}, wc[4] = {1, 2, 2, 2};
  const int kstride = (ks == 1 || upz) ? 1 : stride;
  return snprintf(buf, buflen, "conv_kernel<%s, %d, %d, %d, %d, %d, %d, %d>", dtype == HR_F32 ? "float" : "__bf16",
                  ks, kstride, tc.th, tc.tw, tc.bn, wp[tc.id], wc[tc.id]);
}
