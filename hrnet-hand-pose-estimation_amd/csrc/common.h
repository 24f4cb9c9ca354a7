// Shared device helpers for the gfx950 (CDNA4) HRNet kernels.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hrnet_hip.h"

typedef __bf16 bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 V16;  // one 16-byte vector: 8 bf16 or 4 f32

#define LDS_AS __attribute__((address_space(3)))

// Per-dtype constants. VEC = elements per 16-byte vector. KSTEP = K extent of one
// "fragment step": lane (i = l&15, g = l>>4) owns k = k0 + g*VEC + j, j < VEC, so one
// 16-byte LDS read feeds one bf16 16x16x32 MFMA, or four f32 16x16x4 MFMAs.
template <typename T>
struct TT;
template <>
struct TT<float> {
  static constexpr int VEC = 4;
  static constexpr int KSTEP = 16;
  static constexpr int ID = HR_F32;
};
template <>
struct TT<bf16_t> {
  static constexpr int VEC = 8;
  static constexpr int KSTEP = 32;
  static constexpr int ID = HR_BF16;
};

__device__ __forceinline__ V16 v16_zero() { return V16{0u, 0u, 0u, 0u}; }

template <typename T>
__device__ __forceinline__ void v16_unpack(const V16& v, float* f);
template <>
__device__ __forceinline__ void v16_unpack<float>(const V16& v, float* f) {
  const f32x4 t = __builtin_bit_cast(f32x4, v);
  f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
}
template <>
__device__ __forceinline__ void v16_unpack<bf16_t>(const V16& v, float* f) {
  const uint32_t w0 = v.x, w1 = v.y, w2 = v.z, w3 = v.w;
  f[0] = __builtin_bit_cast(float, w0 << 16); f[1] = __builtin_bit_cast(float, w0 & 0xffff0000u);
  f[2] = __builtin_bit_cast(float, w1 << 16); f[3] = __builtin_bit_cast(float, w1 & 0xffff0000u);
  f[4] = __builtin_bit_cast(float, w2 << 16); f[5] = __builtin_bit_cast(float, w2 & 0xffff0000u);
  f[6] = __builtin_bit_cast(float, w3 << 16); f[7] = __builtin_bit_cast(float, w3 & 0xffff0000u);
}

template <typename T>
__device__ __forceinline__ V16 v16_pack(const float* f);
template <>
__device__ __forceinline__ V16 v16_pack<float>(const float* f) {
  return __builtin_bit_cast(V16, f32x4{f[0], f[1], f[2], f[3]});
}
template <>
__device__ __forceinline__ V16 v16_pack<bf16_t>(const float* f) {
  // plain casts lower to v_cvt_pk_bf16_f32 (RNE, NaN-safe)
  const bf16x8 b = {(bf16_t)f[0], (bf16_t)f[1], (bf16_t)f[2], (bf16_t)f[3],
                    (bf16_t)f[4], (bf16_t)f[5], (bf16_t)f[6], (bf16_t)f[7]};
  return __builtin_bit_cast(V16, b);
}

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T>
__device__ __forceinline__ T from_f32(float x) {
  return (T)x;
}

// acc(16 couts x 16 pixels) += A(16 x KSTEP) * B(KSTEP x 16); a/b are the lane's 16-byte
// fragments. D layout (all dtypes): col = lane&15, row = 4*(lane>>4) + reg.
template <typename T>
__device__ __forceinline__ f32x4 mma16(const V16& a, const V16& b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 mma16<bf16_t>(const V16& a, const V16& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mma16<float>(const V16& a, const V16& b, f32x4 c) {
  const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bf.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bf.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bf.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bf.w, c, 0, 0, 0);
  return c;
}

// Measurement knobs (grid sizes, kernel variants, instantiation masks ...): honoured only when HRNET_MEASURE=1 is set -
// the scratch/ sweep scripts set it - so that a stray variable cannot change what a recorded program launches. Product
// switches are few and live on the host side (HRNET_DETERMINISTIC, HRNET_WGRAD_ATOMIC, HRNET_DP_PLAN, HRNET_LANES).
inline int hr_knob(const char* name, int dflt) {
  static const bool on = getenv("HRNET_MEASURE") != nullptr && atoi(getenv("HRNET_MEASURE")) == 1;
  if (!on) return dflt;
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

__device__ __forceinline__ float wave_sum16(float v) {
  // sum over the 16 lanes that share (lane >> 4)
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v;
}
// acc += a * b as ONE scalar v_fma_f32 the vectoriser cannot touch: the sum(dz * y) accumulate of the backward-statistics
// epilogues. hipcc 7.2's SLP pass turns two neighbouring accumulates into `v_pk_fma_f32 acc2, dz2, y2, acc2
// op_sel:[0,1,0] op_sel_hi:[1,0,1]` (y2 = a bf16 pair unpacked high half first), and inside conv_ring's input-gradient
// launches that instruction returned a wrong LOW half in lanes 48-63 in about every second launch (run-to-run
// different; DESIGN section 4, trap 4: scratch/trap4_run.py reproduces it, the experiment builds of
// scratch/trap4_build2.sh show that taking THIS statement away from the vectoriser - and nothing else - ends it,
// scratch/pk_fma_t.hip that the instruction sequence alone does not fail). Every epilogue that accumulates
// sum(dz * y) goes through here.
__device__ __forceinline__ void hr_fma_acc(float& acc, float a, float b) {
  asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

__device__ __forceinline__ float wave_sum64(float v) {
  v = wave_sum16(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// source taps of one destination coordinate of F.interpolate(mode='bilinear') (PyTorch's
// area_pixel_compute_source_index): the two source indices and the weight of the second one
__device__ __forceinline__ void bilin_src(int d, int in_size, int out_size, int align, int& i0, int& i1,
                                          float& l1) {
  float src;
  if (align) {
    // PyTorch area_pixel_compute_source_index(align_corners=True): src = d * (in-1)/(out-1)
    src = out_size > 1 ? (float)d * ((float)(in_size - 1) / (float)(out_size - 1)) : 0.f;
  } else {
    // align_corners=False: src = (d+0.5)*scale-0.5, clamp >= 0
    const float scale = (float)in_size / (float)out_size;
    src = ((float)d + 0.5f) * scale - 0.5f;
    if (src < 0.f) src = 0.f;
  }
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

// Batch statistics kept as 8 partial copies sums[8][2][C] (sum y, sum y^2): producers add their per-workgroup sums
// with float atomics into copy (workgroup id & 7); consumers turn them into the BatchNorm affine on the fly, so
// no finalize launch sits between a conv and the ops that read its output. The SAME function serves the
// consumers and the batched finalize that fills the arrays the backward pass reads, so both see identical values.
constexpr int HR_BN_COPIES = 8;
__device__ __forceinline__ void hr_bn_from_sums(const float* sums, int C, int c, float inv_count, float eps,
                                                float gamma, float beta, float& scale, float& shift, float& mean,
                                                float& invstd, float& var_biased) {
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int k = 0; k < HR_BN_COPIES; ++k) {
    s1 += (double)sums[(k * 2 + 0) * C + c];
    s2 += (double)sums[(k * 2 + 1) * C + c];
  }
  const double m = s1 * (double)inv_count;
  double v = s2 * (double)inv_count - m * m;
  if (v < 0.0) v = 0.0;
  mean = (float)m;
  var_biased = (float)v;
  invstd = 1.0f / sqrtf(var_biased + eps);
  scale = gamma * invstd;
  shift = beta - mean * scale;
}

// BatchNorm-backward coefficients built by the consuming workgroup from the partial rows a previous launch left
// (HrBnBwdRef): thread t sums rows t/CP, t/CP + NT/CP, ... of channel t%CP in f64, the row groups are added in a
// fixed order, then A,B,C as hrnet_bn_bwd_finalize computes them. `scratch`: NT*2 doubles of LDS; `tab`: [3][CP]
// floats of LDS. Ends with a barrier: tab is readable, scratch is free. `writer`: this workgroup stores dgamma/dbeta.
template <int NT, int CP>
__device__ __forceinline__ void hr_bnbwd_coef_from_rows(const HrBnBwdRef& r, int C, double* scratch, float* tab,
                                                        bool writer) {
  static_assert(NT % CP == 0, "row groups");
  constexpr int NRG = NT / CP;
  const int tid = threadIdx.x, c = tid % CP, rg = tid / CP;
  double a = 0.0, b = 0.0;
  if (c < C)
    for (int t = rg; t < r.nrows; t += NRG) {
      a += (double)r.rows[((size_t)t * 2 + 0) * C + c];
      b += (double)r.rows[((size_t)t * 2 + 1) * C + c];
    }
  scratch[(rg * 2 + 0) * CP + c] = a;
  scratch[(rg * 2 + 1) * CP + c] = b;
  __syncthreads();
  if (tid < CP) {
    float A = 0.f, B = 0.f, Cc = 0.f;
    if (c < C) {
      double s1 = 0.0, s2 = 0.0;
      for (int q = 0; q < NRG; ++q) {
        s1 += scratch[(q * 2 + 0) * CP + c];
        s2 += scratch[(q * 2 + 1) * CP + c];
      }
      const double mu = r.save_mean[c], iv = r.save_invstd[c], g = r.gamma[c], n = (double)r.count;
      const double dg = iv * (s2 - mu * s1);   // sum dz * xhat
      if (writer) {
        r.dgamma[c] = r.accumulate ? r.dgamma[c] + (float)dg : (float)dg;
        r.dbeta[c] = r.accumulate ? r.dbeta[c] + (float)s1 : (float)s1;
      }
      A = (float)(g * iv);
      B = (float)(-g * iv * iv * dg / n);
      Cc = (float)(-g * iv * s1 / n + g * iv * iv * mu * dg / n);
    }
    tab[c] = A;
    tab[CP + c] = B;
    tab[2 * CP + c] = Cc;
  }
  __syncthreads();
}

// host-side error plumbing (api.hip)
void hr_set_error(const char* fmt, ...);
int hr_check_launch(const char* what);

#define HR_REQUIRE(cond, ...)  \
  do {                         \
    if (!(cond)) {             \
      hr_set_error(__VA_ARGS__); \
      return HR_E_BADARG;      \
    }                          \
  } while (0)

// internal launchers (one per op kind), shared by the C entry points and the program runner
int hr_launch_conv(const HrOp& op, hipStream_t s);
int hr_launch_wgrad(const HrOp& op, hipStream_t s);
int hr_launch_wgrad_reduce(const HrOp& op, hipStream_t s);
int hr_launch_bn_finalize(const HrOp& op, hipStream_t s);
int hr_launch_sum_terms(const HrOp& op, hipStream_t s);
int hr_launch_grad_term(const HrOp& op, hipStream_t s);
int hr_launch_bn_bwd_reduce(const HrOp& op, hipStream_t s);
int hr_launch_bn_bwd_finalize(const HrOp& op, hipStream_t s);
int hr_launch_bilinear_cat(const HrOp& op, hipStream_t s);
int hr_launch_bilinear_cat_bwd(const HrOp& op, hipStream_t s);
int hr_launch_im2col_stem(const HrOp& op, hipStream_t s);
int hr_launch_nhwc_to_nchw(const HrOp& op, hipStream_t s);
int hr_launch_nchw_to_nhwc(const HrOp& op, hipStream_t s);
int hr_launch_pack_weights(const HrOp& op, hipStream_t s);
int hr_launch_bias_grad(const HrOp& op, hipStream_t s);
int hr_launch_fill(const HrOp& op, hipStream_t s);
int hr_launch_pack_table(const HrOp& op, hipStream_t s);
int hr_launch_wgrad_reduce_table(const HrOp& op, hipStream_t s);
int hr_launch_bwd_fused(const HrOp& op, hipStream_t s);
int hr_launch_bwd_pw(const HrOp& op, hipStream_t s);
int hr_launch_conv_sum(const HrOp& op, hipStream_t s);
int hr_gemm_pw_supported(int dtype, int Cin, int Cout);
int hr_gemm_pw(const void* x, const void* w, const float* bias, void* y, float* sums, long long pixels, int Cin,
               int Cout, hipStream_t s);
int hr_launch_bn_finalize_table(const HrOp& op, hipStream_t s);
int hr_launch_ew_table(const HrOp& op, hipStream_t s);
int hr_launch_head_mix(const HrOp& op, hipStream_t s);
int hr_launch_upsample_t(const HrOp& op, hipStream_t s);
int hr_launch_head_bwd(const HrOp& op, hipStream_t s);
int hr_launch_pool_reduce(const HrOp& op, hipStream_t s);
