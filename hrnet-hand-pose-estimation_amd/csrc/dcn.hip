// Deformable convolution v1 (config 5 of BASELINE.json; reference lib/deformable_conv:
// src/cuda/deform_im2col_cuda.cuh:24-53 bilinear rule, :127-189 sampling + offset channel order,
// :192-310 gradients; src/cuda/deform_conv_cuda.cu:19-271 shapes / bias / groups).
//
//   out[b,o,y,x] = bias[o] + sum_{c,i,j} W[o,c,i,j] * sample(in[b,c], y*s-p+i*d+dy, x*s-p+j*d+dx)
//
// HBM / gather-bound: the offsets (2*kh*kw floats per deformable group and pixel) dominate the
// bytes, the samples are L2-resident gathers and the 21x189 contraction is tiny - so the column
// buffer of the reference (198 MB at B=64) is never materialised: one thread owns one output pixel,
// streams its offsets once (coalesced along x), samples, and applies the weights straight from LDS
// (wave-uniform broadcast reads). NCHW f32 like the reference op.
#include "common.h"

#ifndef DCN_ABLATE
#define DCN_ABLATE 0      // measurement builds (scratch/dcn_ablate.sh): 1 no LDS atomics, 2 one weight vector per tap, 4 no offset-gradient stores
#endif

namespace {

struct DcnArgs {
  const float* in;    // [B,C,H,W]
  const float* off;   // [B,DG*2*K,Ho,Wo]
  const float* w;     // [Co,C/G,kh,kw]
  const float* bias;  // [Co] or null
  const float* gout;  // [B,Co,Ho,Wo]
  float* out;         // [B,Co,Ho,Wo]
  float* gin;         // [B,C,H,W]   (atomic accumulation: zeroed by the caller)
  float* goff;        // [B,DG*2*K,Ho,Wo]
  float* partial;     // [blocks][Og][Cg][K]
  const float* gmax;  // dcn_bwd_fused_kernel: [B] max |gout[b]| (the fixed-point scale of the input-gradient plane)
  // modulated form (DCNv2, reference src/cuda/modulated_deform_im2col_cuda.cuh:128-257): every sample is multiplied by
  // mask[b, dg*K + k, y, x]; NULL = plain v1
  const float* mask;  // [B,DG*K,Ho,Wo] or NULL
  float* gmask;       // [B,DG*K,Ho,Wo] gradient of the mask (backward, with mask)
  int B, C, H, W, Co, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, DG;
  int c0, Cg, o0, Og;  // this launch's conv group: input channels [c0,c0+Cg), outputs [o0,o0+Og)
};

// reference bilinear (deform_im2col_cuda.cuh:24-53): corners outside the image contribute 0
__device__ __forceinline__ float dcn_sample(const float* plane, int H, int W, float h, float w) {
  const int hl = (int)floorf(h), wl = (int)floorf(w);
  const int hh_ = hl + 1, wh = wl + 1;
  const float lh = h - hl, lw = w - wl, hh = 1.f - lh, hw = 1.f - lw;
  const float v1 = (hl >= 0 && wl >= 0) ? plane[hl * W + wl] : 0.f;
  const float v2 = (hl >= 0 && wh <= W - 1) ? plane[hl * W + wh] : 0.f;
  const float v3 = (hh_ <= H - 1 && wl >= 0) ? plane[hh_ * W + wl] : 0.f;
  const float v4 = (hh_ <= H - 1 && wh <= W - 1) ? plane[hh_ * W + wh] : 0.f;
  return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

// the same sample out of a plane with a one-pixel border of zeros (row pitch W2 = W + 2), for a position inside
// (-1, H) x (-1, W): no bounds tests, identical arithmetic
__device__ __forceinline__ float dcn_sample_bordered(const float* plane, int W2, float h, float w) {
  const float fh = floorf(h), fw = floorf(w);
  const int idx = ((int)fh + 1) * W2 + (int)fw + 1;
  const float lh = h - fh, lw = w - fw, hh = 1.f - lh, hw = 1.f - lw;
  const float v1 = plane[idx], v2 = plane[idx + 1], v3 = plane[idx + W2], v4 = plane[idx + W2 + 1];
  return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

__device__ __forceinline__ bool dcn_inside(float h, float w, int H, int W) {
  return h > -1.f && w > -1.f && h < (float)H && w < (float)W;
}

// OC: output channels per thread pass (32; 24 when a conv group has at most 24 - PoseAggr's 21: a quarter fewer
// multiply-adds and weight reads per sample)
template <int OC>
__global__ __launch_bounds__(256) void dcn_fwd_kernel(DcnArgs a, int oc0, int ocn) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [Cg*K][OC], zero beyond ocn
  const int K = a.kh * a.kw;
  for (int i = threadIdx.x; i < OC * a.Cg * K; i += 256) {
    const int o = i % OC, ck = i / OC;
    wl[i] = o < ocn ? a.w[(size_t)(oc0 + o) * a.Cg * K + ck] : 0.f;
  }
  __syncthreads();
  const long long npix = (long long)a.B * a.Ho * a.Wo;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= npix) return;
  const int x = (int)(idx % a.Wo);
  const int y = (int)((idx / a.Wo) % a.Ho);
  const int b = (int)(idx / ((long long)a.Wo * a.Ho));
  float acc[OC];
#pragma unroll
  for (int o = 0; o < OC; ++o) acc[o] = (a.bias && o < ocn) ? a.bias[oc0 + o] : 0.f;
  const int cpd = a.C / a.DG;
  const size_t plane_o = (size_t)a.Ho * a.Wo;
  for (int cl = 0; cl < a.Cg; ++cl) {
    const int c = a.c0 + cl;
    const float* plane = a.in + ((size_t)b * a.C + c) * a.H * a.W;
    const float* offp = a.off + ((size_t)b * a.DG + c / cpd) * 2 * K * plane_o + (size_t)y * a.Wo + x;
    // three taps at a time: their 6 offset loads, then their 12 gathers, are in flight together
    for (int k0 = 0; k0 < K; k0 += 3) {
      float hh[3], ww[3], vv[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int k = k0 + u < K ? k0 + u : K - 1;
        const int i = k / a.kw, j = k % a.kw;
        hh[u] = (float)(y * a.sh - a.ph + i * a.dh) + offp[(size_t)(2 * k) * plane_o];
        ww[u] = (float)(x * a.sw - a.pw + j * a.dw) + offp[(size_t)(2 * k + 1) * plane_o];
      }
#pragma unroll
      for (int u = 0; u < 3; ++u)
        vv[u] = (k0 + u < K && dcn_inside(hh[u], ww[u], a.H, a.W)) ? dcn_sample(plane, a.H, a.W, hh[u], ww[u]) : 0.f;
      if (a.mask) {
        const float* mp = a.mask + ((size_t)b * a.DG + c / cpd) * K * plane_o + (size_t)y * a.Wo + x;
#pragma unroll
        for (int u = 0; u < 3; ++u) vv[u] *= mp[(size_t)(k0 + u < K ? k0 + u : K - 1) * plane_o];
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int k = k0 + u < K ? k0 + u : K - 1;
        const float val = vv[u];
        const float4* wk = reinterpret_cast<const float4*>(wl + (cl * K + k) * OC);
#pragma unroll
        for (int o = 0; o < OC / 4; ++o) {
          const float4 w4 = wk[o];
          acc[4 * o] = fmaf(w4.x, val, acc[4 * o]);
          acc[4 * o + 1] = fmaf(w4.y, val, acc[4 * o + 1]);
          acc[4 * o + 2] = fmaf(w4.z, val, acc[4 * o + 2]);
          acc[4 * o + 3] = fmaf(w4.w, val, acc[4 * o + 3]);
        }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < OC; ++o)
    if (o < ocn) a.out[((size_t)b * a.Co + oc0 + o) * plane_o + (size_t)y * a.Wo + x] = acc[o];
}

// Forward with the input planes staged through LDS (round 4): a block owns NT consecutive output pixels of one image
// and walks the conv group's channels; channel c's whole plane sits in LDS while its nine taps are sampled - the four
// corner reads of a sample are LDS reads instead of L2 gathers - and channel c + 1's plane is loaded into registers
// meanwhile (two LDS planes, one barrier per channel). Chosen when a plane fits (H*W <= 8 * NT floats, 2 planes + the
// weights in 64 KB) and an image has at least NT output pixels: PoseAggr's 64x64 planes -> 4 blocks of 1024 threads per
// image, 256 blocks. Same arithmetic as dcn_fwd_kernel in the same order: bit-identical output.
template <int OC, int NT>
__global__ __launch_bounds__(NT) void dcn_fwd_planes_kernel(DcnArgs a, int oc0, int ocn) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = a.kh * a.kw, HW = a.H * a.W;
  // the planes carry a one-pixel border of zeros: the four corners of a sample inside (-1, H) x (-1, W) are then read
  // without bounds tests (two ds_read2_b32 from one address), and a corner outside the image contributes its weight
  // times +0 - the value the gather kernel's `ok ? v : 0` gives
  const int W2 = a.W + 2, HW2 = (a.H + 2) * W2;
  float* wl = sm;                          // [Cg*K][OC], zero beyond ocn
  float* pl = sm + OC * a.Cg * K;          // [2][(H+2)*(W+2)]
  for (int i = threadIdx.x; i < OC * a.Cg * K; i += NT) {
    const int o = i % OC, ck = i / OC;
    wl[i] = o < ocn ? a.w[(size_t)(oc0 + o) * a.Cg * K + ck] : 0.f;
  }
  const int b = blockIdx.y;
  const int plane_o = a.Ho * a.Wo;
  const int p = blockIdx.x * NT + threadIdx.x;
  const bool live = p < plane_o;
  const int y = live ? p / a.Wo : 0, x = live ? p % a.Wo : 0;
  constexpr int PV = 8;                    // plane floats per thread (host: H*W <= PV * NT)
  float nx[PV];
  int dst[PV];                             // bordered position of this thread's q-th plane element
  const float* in_b = a.in + ((size_t)b * a.C + a.c0) * HW;
  for (int i = threadIdx.x; i < 2 * HW2; i += NT) pl[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PV; ++q) {
    const int i = threadIdx.x + q * NT;
    const int r = i / a.W;
    dst[q] = (r + 1) * W2 + (i - r * a.W) + 1;
    if (i < HW) pl[dst[q]] = in_b[i];
  }
  float acc[OC];
#pragma unroll
  for (int o = 0; o < OC; ++o) acc[o] = (a.bias && o < ocn) ? a.bias[oc0 + o] : 0.f;
  const int cpd = a.C / a.DG;
  __syncthreads();
  for (int cl = 0; cl < a.Cg; ++cl) {
    const float* plane = pl + (cl & 1) * HW2;
    if (cl + 1 < a.Cg) {
#pragma unroll
      for (int q = 0; q < PV; ++q) {
        const int i = threadIdx.x + q * NT;
        nx[q] = i < HW ? in_b[(size_t)(cl + 1) * HW + i] : 0.f;
      }
    }
    if (live) {
      const int c = a.c0 + cl;
      const float* offp = a.off + ((size_t)b * a.DG + c / cpd) * 2 * K * plane_o + p;
      for (int k0 = 0; k0 < K; k0 += 3) {
        float hh[3], ww[3], vv[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int k = k0 + u < K ? k0 + u : K - 1;
          const int i = k / a.kw, j = k % a.kw;
          hh[u] = (float)(y * a.sh - a.ph + i * a.dh) + offp[(size_t)(2 * k) * plane_o];
          ww[u] = (float)(x * a.sw - a.pw + j * a.dw) + offp[(size_t)(2 * k + 1) * plane_o];
        }
#pragma unroll
        for (int u = 0; u < 3; ++u)
          vv[u] = (k0 + u < K && dcn_inside(hh[u], ww[u], a.H, a.W)) ? dcn_sample_bordered(plane, W2, hh[u], ww[u]) : 0.f;
        if (a.mask) {
          const float* mp = a.mask + ((size_t)b * a.DG + c / cpd) * K * plane_o + p;
#pragma unroll
          for (int u = 0; u < 3; ++u) vv[u] *= mp[(size_t)(k0 + u < K ? k0 + u : K - 1) * plane_o];
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int k = k0 + u < K ? k0 + u : K - 1;
          const float val = vv[u];
          const float4* wk = reinterpret_cast<const float4*>(wl + (cl * K + k) * OC);
#pragma unroll
          for (int o = 0; o < OC / 4; ++o) {
            const float4 w4 = wk[o];
            acc[4 * o] = fmaf(w4.x, val, acc[4 * o]);
            acc[4 * o + 1] = fmaf(w4.y, val, acc[4 * o + 1]);
            acc[4 * o + 2] = fmaf(w4.z, val, acc[4 * o + 2]);
            acc[4 * o + 3] = fmaf(w4.w, val, acc[4 * o + 3]);
          }
        }
      }
    }
    if (cl + 1 < a.Cg) {
      float* nxt = pl + ((cl + 1) & 1) * HW2;    // read by every thread in iteration cl - 1: the barrier below closed it
#pragma unroll
      for (int q = 0; q < PV; ++q) {
        const int i = threadIdx.x + q * NT;
        if (i < HW) nxt[dst[q]] = nx[q];
      }
    }
    __syncthreads();
  }
  if (live) {
#pragma unroll
    for (int o = 0; o < OC; ++o)
      if (o < ocn) a.out[((size_t)b * a.Co + oc0 + o) * plane_o + p] = acc[o];
  }
}

// grad wrt input (atomics, as the reference's col2im) and wrt offsets (direct store: one thread owns
// every (pixel, offset channel) of its deformable groups)
template <int OCP>
__global__ __launch_bounds__(256) void dcn_bwd_data_kernel(DcnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [Cg*K][OCP], zero beyond Og
  const int K = a.kh * a.kw;
  for (int i = threadIdx.x; i < OCP * a.Cg * K; i += 256) {
    const int o = i % OCP, ck = i / OCP;
    wl[i] = o < a.Og ? a.w[(size_t)(a.o0 + o) * a.Cg * K + ck] : 0.f;
  }
  __syncthreads();
  const long long npix = (long long)a.B * a.Ho * a.Wo;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= npix) return;
  const int x = (int)(idx % a.Wo);
  const int y = (int)((idx / a.Wo) % a.Ho);
  const int b = (int)(idx / ((long long)a.Wo * a.Ho));
  const size_t plane_o = (size_t)a.Ho * a.Wo;
  float g[OCP];
#pragma unroll
  for (int o = 0; o < OCP; ++o)
    g[o] = o < a.Og ? a.gout[((size_t)b * a.Co + a.o0 + o) * plane_o + (size_t)y * a.Wo + x] : 0.f;
  const int cpd = a.C / a.DG;
  for (int cl = 0; cl < a.Cg; ++cl) {
    const int c = a.c0 + cl;
    const int dgi = c / cpd;
    const float* plane = a.in + ((size_t)b * a.C + c) * a.H * a.W;
    float* gplane = a.gin + ((size_t)b * a.C + c) * a.H * a.W;
    const size_t obase = ((size_t)b * a.DG + dgi) * 2 * K * plane_o + (size_t)y * a.Wo + x;
    for (int k = 0; k < K; ++k) {
      const int i = k / a.kw, j = k % a.kw;
      const float h = (float)(y * a.sh - a.ph + i * a.dh) + a.off[obase + (size_t)(2 * k) * plane_o];
      const float w = (float)(x * a.sw - a.pw + j * a.dw) + a.off[obase + (size_t)(2 * k + 1) * plane_o];
      float gc = 0.f;  // d loss / d column(c,k) at this pixel
      const float4* wk = reinterpret_cast<const float4*>(wl + (cl * K + k) * OCP);
#pragma unroll
      for (int o = 0; o < OCP / 4; ++o) {
        const float4 w4 = wk[o];
        gc = fmaf(w4.x, g[4 * o], gc);
        gc = fmaf(w4.y, g[4 * o + 1], gc);
        gc = fmaf(w4.z, g[4 * o + 2], gc);
        gc = fmaf(w4.w, g[4 * o + 3], gc);
      }
      float gh = 0.f, gw = 0.f, gm = 0.f;
      // modulated: column = mask * sample, so d/d sample carries the mask and d/d mask = gc * sample
      // (modulated_deform_im2col_cuda.cuh:196-257, 259-330)
      const size_t mbase = ((size_t)b * a.DG + dgi) * K * plane_o + (size_t)y * a.Wo + x + (size_t)k * plane_o;
      const float gcu = gc;
      if (a.mask) gc *= a.mask[mbase];
      if (dcn_inside(h, w, a.H, a.W)) {
        const int hl = (int)floorf(h), wl_ = (int)floorf(w);
        const int hh_ = hl + 1, wh = wl_ + 1;
        const float lh = h - hl, lw = w - wl_, hh = 1.f - lh, hw = 1.f - lw;
        const bool ok1 = hl >= 0 && wl_ >= 0, ok2 = hl >= 0 && wh <= a.W - 1;
        const bool ok3 = hh_ <= a.H - 1 && wl_ >= 0, ok4 = hh_ <= a.H - 1 && wh <= a.W - 1;
        const float v1 = ok1 ? plane[hl * a.W + wl_] : 0.f, v2 = ok2 ? plane[hl * a.W + wh] : 0.f;
        const float v3 = ok3 ? plane[hh_ * a.W + wl_] : 0.f, v4 = ok4 ? plane[hh_ * a.W + wh] : 0.f;
        // d sample / d h, d w (deform_im2col_cuda.cuh:82-124)
        gh = gc * (-hw * v1 - lw * v2 + hw * v3 + lw * v4);
        gw = gc * (-hh * v1 + hh * v2 - lh * v3 + lh * v4);
        gm = gcu * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4);
        if (ok1) atomicAdd(gplane + hl * a.W + wl_, gc * hh * hw);
        if (ok2) atomicAdd(gplane + hl * a.W + wh, gc * hh * lw);
        if (ok3) atomicAdd(gplane + hh_ * a.W + wl_, gc * lh * hw);
        if (ok4) atomicAdd(gplane + hh_ * a.W + wh, gc * lh * lw);
      }
      // input channels that share a deformable group add up in channel order (the group's first
      // channel stores; a group that straddles conv groups is finished by the later launch)
      float* gp = a.goff + obase + (size_t)(2 * k) * plane_o;
      const bool first = c % cpd == 0;
      gp[0] = first ? gh : gp[0] + gh;
      gp[plane_o] = first ? gw : gp[plane_o] + gw;
      if (a.gmask) a.gmask[mbase] = first ? gm : a.gmask[mbase] + gm;
    }
  }
}

// Fast path of the data gradients when the input planes of one deformable group fit LDS: a block
// owns (image b, deformable group) and accumulates the scattered input gradient of the group's
// channels in LDS (ds_add_f32) instead of global atomics, reads each offset once for all channels of
// the group, and writes the finished planes with plain coalesced stores. Requires every deformable
// group to lie inside one conv group.
template <int OCP>
__global__ __launch_bounds__(256) void dcn_bwd_data_lds_kernel(DcnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = a.kh * a.kw;
  const int cpd = a.C / a.DG;
  const int HW = a.H * a.W;
  float* wl = sm;                       // [cpd*K][OCP], zero beyond Og
  float* gpl = sm + cpd * K * OCP;      // [cpd][H*W]
  const int c_first = a.c0 + blockIdx.x * cpd;          // first channel of this deformable group
  const int dgi = c_first / cpd;
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < OCP * cpd * K; i += 256) {
    const int o = i % OCP, ck = i / OCP;
    wl[i] = o < a.Og ? a.w[((size_t)(a.o0 + o) * a.Cg + (c_first - a.c0)) * K + ck] : 0.f;
  }
  for (int i = threadIdx.x; i < cpd * HW; i += 256) gpl[i] = 0.f;
  __syncthreads();
  const int plane_o = a.Ho * a.Wo;
  const float* in_b = a.in + ((size_t)b * a.C + c_first) * HW;
  for (int p0 = 0; p0 < plane_o; p0 += 256) {
    const int p = p0 + threadIdx.x;
    if (p >= plane_o) break;
    const int y = p / a.Wo, x = p % a.Wo;
    float g[OCP];
#pragma unroll
    for (int o = 0; o < OCP; ++o)
      g[o] = o < a.Og ? a.gout[((size_t)b * a.Co + a.o0 + o) * plane_o + p] : 0.f;
    const size_t obase = ((size_t)b * a.DG + dgi) * 2 * K * plane_o + p;
    for (int k = 0; k < K; ++k) {
      const int i = k / a.kw, j = k % a.kw;
      const float h = (float)(y * a.sh - a.ph + i * a.dh) + a.off[obase + (size_t)(2 * k) * plane_o];
      const float w = (float)(x * a.sw - a.pw + j * a.dw) + a.off[obase + (size_t)(2 * k + 1) * plane_o];
      float gh = 0.f, gw = 0.f;
      if (dcn_inside(h, w, a.H, a.W)) {
        const int hl = (int)floorf(h), wl_ = (int)floorf(w);
        const int hh_ = hl + 1, wh = wl_ + 1;
        const float lh = h - hl, lw = w - wl_, hh = 1.f - lh, hw = 1.f - lw;
        const bool ok1 = hl >= 0 && wl_ >= 0, ok2 = hl >= 0 && wh <= a.W - 1;
        const bool ok3 = hh_ <= a.H - 1 && wl_ >= 0, ok4 = hh_ <= a.H - 1 && wh <= a.W - 1;
        for (int cl = 0; cl < cpd; ++cl) {
          float gc = 0.f;
          const float4* wk = reinterpret_cast<const float4*>(wl + (cl * K + k) * OCP);
#pragma unroll
          for (int o = 0; o < OCP / 4; ++o) {
            const float4 w4 = wk[o];
            gc = fmaf(w4.x, g[4 * o], gc);
            gc = fmaf(w4.y, g[4 * o + 1], gc);
            gc = fmaf(w4.z, g[4 * o + 2], gc);
            gc = fmaf(w4.w, g[4 * o + 3], gc);
          }
          const float* plane = in_b + (size_t)cl * HW;
          float* gp = gpl + cl * HW;
          const float v1 = ok1 ? plane[hl * a.W + wl_] : 0.f, v2 = ok2 ? plane[hl * a.W + wh] : 0.f;
          const float v3 = ok3 ? plane[hh_ * a.W + wl_] : 0.f, v4 = ok4 ? plane[hh_ * a.W + wh] : 0.f;
          gh = fmaf(gc, -hw * v1 - lw * v2 + hw * v3 + lw * v4, gh);
          gw = fmaf(gc, -hh * v1 + hh * v2 - lh * v3 + lh * v4, gw);
          if (ok1) atomicAdd(gp + hl * a.W + wl_, gc * hh * hw);
          if (ok2) atomicAdd(gp + hl * a.W + wh, gc * hh * lw);
          if (ok3) atomicAdd(gp + hh_ * a.W + wl_, gc * lh * hw);
          if (ok4) atomicAdd(gp + hh_ * a.W + wh, gc * lh * lw);
        }
      }
      a.goff[obase + (size_t)(2 * k) * plane_o] = gh;
      a.goff[obase + (size_t)(2 * k + 1) * plane_o] = gw;
    }
  }
  __syncthreads();
  float* out = a.gin + ((size_t)b * a.C + c_first) * HW;
  for (int i = threadIdx.x; i < cpd * HW; i += 256) out[i] = gpl[i];
}

// weight gradient: a block walks WCH chunks of 256 pixels; per chunk and input channel the K samples
// and the Og output gradients of every pixel go through LDS, thread t owns the (o,k) pair t, dots the
// 256 pixels and keeps its running sums in LDS. One partial [Og][Cg][K] per block, summed in fixed
// order by dcn_weight_reduce_kernel (deterministic).
constexpr int WCH = 1;  // more chunks per block leave too few blocks in flight at B*Ho*Wo ~ 2.6e5

__global__ __launch_bounds__(256) void dcn_bwd_weight_kernel(DcnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = a.kh * a.kw;
  constexpr int LD = 257;            // padded row: (row + p) mod 32 banks
  float* gl = sm;                    // [Og][LD]
  float* vl = sm + a.Og * LD;        // [K][LD]
  float* accl = vl + K * LD;         // [Og][Cg][K], element (o,cl,k) owned by thread (o*K+k) % 256
  const long long npix = (long long)a.B * a.Ho * a.Wo;
  const size_t plane_o = (size_t)a.Ho * a.Wo;
  const int cpd = a.C / a.DG;
  const int npair = a.Og * K;
  for (int t = threadIdx.x; t < npair; t += 256)
    for (int cl = 0; cl < a.Cg; ++cl) accl[((t / K) * a.Cg + cl) * K + t % K] = 0.f;
  for (int ch = 0; ch < WCH; ++ch) {
    const long long first = ((long long)blockIdx.x * WCH + ch) * 256;
    if (first >= npix) break;
    const long long idx = first + threadIdx.x;
    const bool live = idx < npix;
    const int x = live ? (int)(idx % a.Wo) : 0;
    const int y = live ? (int)((idx / a.Wo) % a.Ho) : 0;
    const int b = live ? (int)(idx / ((long long)a.Wo * a.Ho)) : 0;
    __syncthreads();
    for (int o = 0; o < a.Og; ++o)
      gl[o * LD + threadIdx.x] =
          live ? a.gout[((size_t)b * a.Co + a.o0 + o) * plane_o + (size_t)y * a.Wo + x] : 0.f;
    for (int cl = 0; cl < a.Cg; ++cl) {
      const int c = a.c0 + cl;
      const float* plane = a.in + ((size_t)b * a.C + c) * a.H * a.W;
      const size_t obase = ((size_t)b * a.DG + c / cpd) * 2 * K * plane_o + (size_t)y * a.Wo + x;
      __syncthreads();
      for (int k = 0; k < K; ++k) {
        float val = 0.f;
        if (live) {
          const int i = k / a.kw, j = k % a.kw;
          const float h = (float)(y * a.sh - a.ph + i * a.dh) + a.off[obase + (size_t)(2 * k) * plane_o];
          const float w = (float)(x * a.sw - a.pw + j * a.dw) + a.off[obase + (size_t)(2 * k + 1) * plane_o];
          if (dcn_inside(h, w, a.H, a.W)) val = dcn_sample(plane, a.H, a.W, h, w);
          if (a.mask) val *= a.mask[((size_t)b * a.DG + c / cpd) * K * plane_o + (size_t)k * plane_o + (size_t)y * a.Wo + x];
        }
        vl[k * LD + threadIdx.x] = val;
      }
      __syncthreads();
      for (int t = threadIdx.x; t < npair; t += 256) {
        const int o = t / K, k = t % K;
        const float* gr = gl + o * LD;
        const float* vr = vl + k * LD;
        float s0 = 0.f, s1 = 0.f;
        for (int p = 0; p < 256; p += 2) {
          s0 = fmaf(gr[p], vr[p], s0);
          s1 = fmaf(gr[p + 1], vr[p + 1], s1);
        }
        accl[(o * a.Cg + cl) * K + k] += s0 + s1;
      }
    }
  }
  float* part = a.partial + (size_t)blockIdx.x * a.Og * a.Cg * K;
  for (int t = threadIdx.x; t < npair; t += 256)
    for (int cl = 0; cl < a.Cg; ++cl) {
      const int e = ((t / K) * a.Cg + cl) * K + t % K;
      part[e] = accl[e];
    }
}

// Single-pass backward for the PoseAggr geometry (one input channel per deformable group, one conv group,
// Og <= 32, planes that fit LDS): a block owns (image b, channel c) and reads that channel's offsets ONCE for
// all three gradients (the two-kernel path above streams the 396 MB of offsets twice and re-samples
// everything for the weight gradient). Per step of 256 pixels:
//   phase 1 (thread = pixel): g[o] of the pixel, then per tap gc = sum_o W[o,c,k] * g[o], the four bilinear
//           corners from the input plane staged in LDS, the offset gradient stored directly, the scattered
//           input gradient added into an LDS plane; the sample and g go to LDS rows;
//   phase 2 (thread = (o,k) pair): dW[o,c,k] += sum over the 256 pixels of g[o,p] * sample[k,p] (16-byte LDS
//           reads), so the weight-gradient sums cost a thread one register instead of Og*K.
// The planes are stored coalesced at the end; one partial [Og][C][K] per image, summed over the images in
// fixed order by dcn_weight_reduce_kernel (deterministic).
//
// The input-gradient plane is a 64-bit FIXED-POINT plane (round 4). LDS float atomics (ds_add_f32) were 0.74 of this
// kernel's 1.06 ms; integer LDS atomics run ~3.5x faster (scratch/lds_atom_t.hip: ds_add_u64 as fast as ds_add_u32).
// Every contribution v (an f32 product) is added as round(v * 2^e): 2^e from a bound on the contributions of this
// plane - max |gout[b]| (a[B] pre-pass, dcn_absmax_kernel) times the largest tap's sum of |W[o,c,k]| - chosen so that
// one contribution stays below 2^50 and all plane_o * K of them in ONE cell below 2^62. An f32 value has 24 significant
// bits, so what is added is the contribution itself unless it is more than 2^26 times smaller than the bound: the plane
// holds the exact sum of the f32 contributions, rounded to f32 once when it is stored - closer to the f64 oracle than
// float atomics, and the same bits on every run (integer addition commutes). The conversion is one f64 FMA: v * 2^e +
// 1.5 * 2^52 leaves the rounded integer in the low mantissa bits.
template <int OGP, int KK>
__global__ __launch_bounds__(256) void dcn_bwd_fused_kernel(DcnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int LD = 260;           // padded row of 256 pixels (16-byte aligned, rows 4 banks apart)
  const int HW = a.H * a.W;
  unsigned long long* gpl = reinterpret_cast<unsigned long long*>(sm);   // [H*W] input-gradient plane, 64-bit fixed point
  float* pl = sm + 2 * HW;          // [H*W] input plane
  float* wl = sm + 3 * HW;          // [KK][OGP] weights of this channel, zero beyond Og
  float* gl = wl + KK * OGP;        // [Og][LD] output gradients of the step's pixels (Og rows: two workgroups per CU fit)
  float* vl = gl + a.Og * LD;       // [KK][LD] samples of the step's pixels
  const int c = blockIdx.x, b = blockIdx.y;
  for (int i = threadIdx.x; i < KK * OGP; i += 256) {
    const int o = i % OGP, k = i / OGP;
    wl[i] = o < a.Og ? a.w[((size_t)o * a.C + c) * KK + k] : 0.f;
  }
  const float* in_p = a.in + ((size_t)b * a.C + c) * HW;
  for (int i = threadIdx.x; i < HW; i += 256) {
    pl[i] = in_p[i];
    gpl[i] = 0ull;
  }
  const int plane_o = a.Ho * a.Wo;
  __syncthreads();          // wl, pl and the zeroed gradient plane are complete
  // fixed-point scale 2^fe of this plane (wave-uniform arithmetic, every thread the same)
  float wsum = 0.f;
  for (int k = 0; k < KK; ++k) {
    float t = 0.f;
    for (int o = 0; o < OGP; ++o) t += fabsf(wl[k * OGP + o]);
    wsum = fmaxf(wsum, t);
  }
  const float bound = a.gmax[b] * wsum;                 // >= |gc| of every (pixel, tap) of this plane
  const bool finite = bound <= 3.0e38f;                 // (false for inf / NaN gradients: the plane is stored as NaN)
  int bx = 0;
  (void)frexpf(finite && bound > 0.f ? bound : 1.f, &bx);          // bound < 2^bx
  int cells_log = 0;
  while ((1ll << cells_log) < (long long)plane_o * KK) ++cells_log;
  const int fe = (50 < 61 - cells_log ? 50 : 61 - cells_log) - bx;
  const double fscale = ldexp(1.0, fe), funscale = ldexp(1.0, -fe);
  constexpr double MAGIC = 6755399441055744.0;          // 1.5 * 2^52
  // a contribution gc * w as round(gs * w), gs = gc * 2^fe in f64 (one conversion of gc per tap; the product is rounded
  // once, in f64: |gs * w| < 2^50)
  auto to_fixed2 = [&](double gs, float w) -> unsigned long long {
    const double d = __builtin_fma(gs, (double)w, MAGIC);
    return (unsigned long long)(__builtin_bit_cast(long long, d) - __builtin_bit_cast(long long, MAGIC));
  };
  const size_t obase0 = ((size_t)b * a.DG + c) * 2 * KK * plane_o;
  // phase 2 on the matrix pipe (round 4): dW[o][k] = sum_p g[o][p] * sample[k][p] is a (32 x 16) x K = 256 product per
  // step. v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulate): lane (i = lane & 15, kk = lane >> 4) of a
  // 16-pixel chunk reads FOUR consecutive pixels 4 kk .. 4 kk + 3 of its row with one ds_read_b128 and feeds them to
  // four MFMAs (the pairing of pixels with k indices is free as long as both operands use the same one). A wave takes
  // the four chunks of its own 64 pixels: 4 chunks x (2 + 1) 16-byte reads per step instead of 128 per (o, k) thread - the LDS pipe was
  // active 87 % of the kernel's time before (SQ_LDS_IDX_ACTIVE 2.0 M cycles per CU of 2.3 M), phase 2 alone issued
  // ~3000 of its ~6300 LDS cycles per step. Rows beyond Og / taps beyond 9 are fed zeros.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mi = lane & 15, mk = lane >> 4;
  f32x4 accw[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};   // o tiles 0-15, 16-31 x taps (cols)
  for (int p0 = 0; p0 < plane_o; p0 += 256) {
    const int p = p0 + threadIdx.x;
    const bool live = p < plane_o;
    // (no workgroup barrier inside the walk: a wave's phase 2 reads only the 64 pixels its own lanes wrote - LDS
    // operations of one wave complete in order - so the four waves drift apart and fill each other's waits)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (live) {
      const int y = p / a.Wo, x = p % a.Wo;
      float g[OGP];
#pragma unroll
      for (int o = 0; o < OGP; ++o) g[o] = o < a.Og ? a.gout[((size_t)b * a.Co + o) * plane_o + p] : 0.f;
      // all offsets of this pixel first (18 independent loads in flight). (Round 4: loading the NEXT step's offsets a
      // step ahead - 18 more registers, 243 in all - made the launch slower, 571 against 540 us: the co-resident
      // workgroup already fills the wait.)
      float oh[KK], ow[KK];
#pragma unroll
      for (int k = 0; k < KK; ++k) {
        oh[k] = a.off[obase0 + (size_t)(2 * k) * plane_o + p];
        ow[k] = a.off[obase0 + (size_t)(2 * k + 1) * plane_o + p];
      }
#pragma unroll
      for (int o = 0; o < OGP; ++o)
        if (o < a.Og) gl[o * LD + threadIdx.x] = g[o];
#pragma unroll
      for (int k = 0; k < KK; ++k) {
        const int i = k / a.kw, j = k % a.kw;
        const float h = (float)(y * a.sh - a.ph + i * a.dh) + oh[k];
        const float w = (float)(x * a.sw - a.pw + j * a.dw) + ow[k];
        float gc = 0.f;
        const float4* wk = reinterpret_cast<const float4*>(wl + k * OGP);
#pragma unroll
        for (int o = 0; o < ((DCN_ABLATE & 2) ? 1 : OGP / 4); ++o) {
          const float4 w4 = wk[o];
          gc = fmaf(w4.x, g[4 * o], gc);
          gc = fmaf(w4.y, g[4 * o + 1], gc);
          gc = fmaf(w4.z, g[4 * o + 2], gc);
          gc = fmaf(w4.w, g[4 * o + 3], gc);
        }
        float gh = 0.f, gw = 0.f, val = 0.f;
        if (dcn_inside(h, w, a.H, a.W)) {
          const int hl = (int)floorf(h), wl_ = (int)floorf(w);
          const int hh_ = hl + 1, wh = wl_ + 1;
          const float lh = h - hl, lw = w - wl_, hh = 1.f - lh, hw = 1.f - lw;
          const bool ok1 = hl >= 0 && wl_ >= 0, ok2 = hl >= 0 && wh <= a.W - 1;
          const bool ok3 = hh_ <= a.H - 1 && wl_ >= 0, ok4 = hh_ <= a.H - 1 && wh <= a.W - 1;
          const float v1 = ok1 ? pl[hl * a.W + wl_] : 0.f, v2 = ok2 ? pl[hl * a.W + wh] : 0.f;
          const float v3 = ok3 ? pl[hh_ * a.W + wl_] : 0.f, v4 = ok4 ? pl[hh_ * a.W + wh] : 0.f;
          val = hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
          gh = gc * (-hw * v1 - lw * v2 + hw * v3 + lw * v4);
          gw = gc * (-hh * v1 + hh * v2 - lh * v3 + lh * v4);
#if !(DCN_ABLATE & 1)
          // the scattered input gradient: integer atomics into the fixed-point LDS plane
          const double gs = (double)gc * fscale;
          if (ok1) __hip_atomic_fetch_add(gpl + hl * a.W + wl_, to_fixed2(gs, hh * hw), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (ok2) __hip_atomic_fetch_add(gpl + hl * a.W + wh, to_fixed2(gs, hh * lw), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (ok3) __hip_atomic_fetch_add(gpl + hh_ * a.W + wl_, to_fixed2(gs, lh * hw), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (ok4) __hip_atomic_fetch_add(gpl + hh_ * a.W + wh, to_fixed2(gs, lh * lw), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
        }
#if !(DCN_ABLATE & 4)
        a.goff[obase0 + (size_t)(2 * k) * plane_o + p] = gh;
        a.goff[obase0 + (size_t)(2 * k + 1) * plane_o + p] = gw;
#else
        if (gh == 12345.f) a.goff[obase0 + p] = gw;
#endif
        vl[k * LD + threadIdx.x] = val;
      }
    } else {
#pragma unroll
      for (int o = 0; o < OGP; ++o)
        if (o < a.Og) gl[o * LD + threadIdx.x] = 0.f;
#pragma unroll
      for (int k = 0; k < KK; ++k) vl[k * LD + threadIdx.x] = 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    {
      const bool r0 = mi < a.Og, r1 = 16 + mi < a.Og, kc = mi < KK;
      const float* g0p = gl + (r0 ? mi : 0) * LD + 4 * mk;
      const float* g1p = gl + (r1 ? 16 + mi : 0) * LD + 4 * mk;
      const float* vp = vl + (kc ? mi : 0) * LD + 4 * mk;
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = (wave * 4 + q) * 16;                  // q-th chunk of 16 of this wave's own 64 pixels
        const float4 a0 = r0 ? *reinterpret_cast<const float4*>(g0p + ch) : z4;
        const float4 a1 = r1 ? *reinterpret_cast<const float4*>(g1p + ch) : z4;
        const float4 bv = kc ? *reinterpret_cast<const float4*>(vp + ch) : z4;
        accw[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, bv.x, accw[0], 0, 0, 0);
        accw[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, bv.y, accw[0], 0, 0, 0);
        accw[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, bv.z, accw[0], 0, 0, 0);
        accw[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, bv.w, accw[0], 0, 0, 0);
        if constexpr (OGP > 16) {
          accw[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, bv.x, accw[1], 0, 0, 0);
          accw[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, bv.y, accw[1], 0, 0, 0);
          accw[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, bv.z, accw[1], 0, 0, 0);
          accw[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, bv.w, accw[1], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();
  float* out = a.gin + ((size_t)b * a.C + c) * HW;
  for (int i = threadIdx.x; i < HW; i += 256)
    out[i] = finite ? (float)((double)(long long)gpl[i] * funscale) : __builtin_nanf("");
  // the four waves' partial products meet in LDS (fixed order: deterministic); D layout: col (lane & 15) = tap,
  // row 4 * (lane >> 4) + r = output channel of the tile
  float* red = gl;                   // [4 waves][32 o][16 taps] (the step buffers are free: 8 KB of gl + vl's >= 10 KB)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float v4[4] = {accw[t].x, accw[t].y, accw[t].z, accw[t].w};
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * 32 + t * 16 + 4 * mk + r) * 16 + mi] = v4[r];
  }
  __syncthreads();
  // partial[b][o][c][k]: one [Og][C][K] block per image
  if ((int)threadIdx.x < a.Og * KK) {
    const int po = threadIdx.x / KK, pk = threadIdx.x % KK;
    const float dw = ((red[(0 * 32 + po) * 16 + pk] + red[(1 * 32 + po) * 16 + pk]) + red[(2 * 32 + po) * 16 + pk]) +
                     red[(3 * 32 + po) * 16 + pk];
    a.partial[(size_t)b * a.Og * a.C * KK + ((size_t)po * a.C + c) * KK + pk] = dw;
  }
}

// out[b] = max |x[b, :]| (NaN / inf propagate as inf: the consumer then stores NaN planes). gridDim.x blocks per image
// meet in out[b] with an unsigned atomic max - non-negative floats order like their bit patterns - so out must be zero
// before the launch. (One block per image took 139 us at B = 64: a fifth of the backward's kernel time.)
__global__ __launch_bounds__(256) void dcn_absmax_kernel(const float* x, float* out, long long n) {
  __shared__ float red[4];
  const float* p = x + (size_t)blockIdx.y * n;
  float m = 0.f;
  const long long n4 = n >> 2;
  if ((n & 3) == 0 && ((size_t)p & 15) == 0) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
      const float4 v = p4[i];
      const float a = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
      const bool bad = v.x != v.x || v.y != v.y || v.z != v.z || v.w != v.w;
      m = bad ? __builtin_inff() : fmaxf(m, a);
    }
  } else {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
      const float v = fabsf(p[i]);
      m = v != v ? __builtin_inff() : fmaxf(m, v);
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    atomicMax(reinterpret_cast<unsigned*>(out) + blockIdx.y, __float_as_uint(t));
  }
}

// one wave per weight element: lanes stride over the per-block partials, fixed-shape tree at the end
__global__ __launch_bounds__(256) void dcn_weight_reduce_kernel(const float* partial, float* gw, int blocks,
                                                                int n, int accumulate) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= n) return;
  float s = 0.f;
  for (int b = threadIdx.x & 63; b < blocks; b += 64) s += partial[(size_t)b * n + e];
  s = wave_sum64(s);
  if ((threadIdx.x & 63) == 0) gw[e] = accumulate ? gw[e] + s : s;
}

__global__ __launch_bounds__(256) void dcn_bias_grad_kernel(const float* gout, float* gb, int B, int Co,
                                                            long long hw, int accumulate) {
  __shared__ float red[4];
  const int o = blockIdx.x;
  float s = 0.f;
  for (int b = 0; b < B; ++b)
    for (long long i = threadIdx.x; i < hw; i += 256) s += gout[((size_t)b * Co + o) * hw + i];
  s = wave_sum64(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = (red[0] + red[1]) + (red[2] + red[3]);
    gb[o] = accumulate ? gb[o] + t : t;
  }
}

// dynamic LDS beyond the default 64 KB window has to be requested per kernel
template <typename F>
void want_lds(F kernel, size_t bytes) {
  if (bytes > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)bytes);
}

int fill_common(DcnArgs& a, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw,
                int dh, int dw, int G, int DG, int* Ho, int* Wo) {
  HR_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Co > 0 && kh > 0 && kw > 0, "deform_conv: empty shape");
  HR_REQUIRE(sh > 0 && sw > 0 && dh > 0 && dw > 0 && ph >= 0 && pw >= 0, "deform_conv: bad stride/pad/dilation");
  HR_REQUIRE(G >= 1 && C % G == 0 && Co % G == 0, "deform_conv: channels not divisible by groups %d", G);
  HR_REQUIRE(DG >= 1 && C % DG == 0, "deform_conv: channels not divisible by deformable_groups %d", DG);
  *Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  *Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  HR_REQUIRE(*Ho > 0 && *Wo > 0, "deform_conv: empty output");
  a.B = B; a.C = C; a.H = H; a.W = W; a.Co = Co; a.Ho = *Ho; a.Wo = *Wo; a.kh = kh; a.kw = kw;
  a.sh = sh; a.sw = sw; a.ph = ph; a.pw = pw; a.dh = dh; a.dw = dw; a.DG = DG;
  return HR_OK;
}

}  // namespace

static int dcn_forward_impl(const float* input, const float* offset, const float* mask, const float* weight,
                            const float* bias, float* output, int B, int C, int H, int W, int Co,
                            int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                            int groups, int deformable_groups, hr_stream_t stream) {
  DcnArgs a = {};
  int Ho, Wo;
  if (int e = fill_common(a, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, deformable_groups, &Ho, &Wo))
    return e;
  HR_REQUIRE(input && offset && weight && output, "deform_conv_forward: null pointer");
  a.in = input; a.off = offset; a.w = weight; a.bias = bias; a.out = output; a.mask = mask;
  const int Cg = C / groups, Og = Co / groups, K = kh * kw;
  const int OC = Og <= 24 ? 24 : 32;
  HR_REQUIRE((size_t)OC * Cg * K * 4 <= 96 * 1024, "deform_conv_forward: C/groups * kh * kw = %d too large", Cg * K);
  const long long npix = (long long)B * Ho * Wo;
  const unsigned blocks = (unsigned)((npix + 255) / 256);
  if (OC == 24) want_lds(dcn_fwd_kernel<24>, (size_t)OC * Cg * K * 4);
  else want_lds(dcn_fwd_kernel<32>, (size_t)OC * Cg * K * 4);
  // planes through LDS (dcn_fwd_planes_kernel) where they fit and an image fills 1024-thread blocks
  constexpr int PNT = 1024;
  const size_t lds_planes = ((size_t)OC * Cg * K + (size_t)2 * (H + 2) * (W + 2)) * 4;
  static const int planes_on = hr_knob("HRNET_DCN_FWD_PLANES", 1);   // (measurement: 0 = the gather kernel)
  if (planes_on && OC == 24 && (long long)H * W <= 8 * PNT && lds_planes <= 64 * 1024 && Ho * Wo >= PNT && B <= 65535) {
    want_lds(dcn_fwd_planes_kernel<24, PNT>, lds_planes);
    for (int g = 0; g < groups; ++g) {
      a.c0 = g * Cg; a.Cg = Cg;
      hipLaunchKernelGGL((dcn_fwd_planes_kernel<24, PNT>), dim3((Ho * Wo + PNT - 1) / PNT, B), dim3(PNT), lds_planes,
                         (hipStream_t)stream, a, g * Og, Og);
    }
    return hr_check_launch("deform_conv_forward");
  }
  for (int g = 0; g < groups; ++g) {
    a.c0 = g * Cg; a.Cg = Cg;
    for (int o = 0; o < Og; o += OC) {
      const int ocn = Og - o < OC ? Og - o : OC;
      if (OC == 24)
        hipLaunchKernelGGL(dcn_fwd_kernel<24>, dim3(blocks), dim3(256), (size_t)OC * Cg * K * 4, (hipStream_t)stream, a,
                           g * Og + o, ocn);
      else
        hipLaunchKernelGGL(dcn_fwd_kernel<32>, dim3(blocks), dim3(256), (size_t)OC * Cg * K * 4, (hipStream_t)stream, a,
                           g * Og + o, ocn);
    }
  }
  return hr_check_launch("deform_conv_forward");
}

extern "C" int hrnet_deform_conv_forward(const float* input, const float* offset, const float* weight,
                                         const float* bias, float* output, int B, int C, int H, int W, int Co,
                                         int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                                         int groups, int deformable_groups, hr_stream_t stream) {
  return dcn_forward_impl(input, offset, nullptr, weight, bias, output, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw,
                          groups, deformable_groups, stream);
}

// modulated form (DCNv2): mask [B, deformable_groups * kh * kw, Ho, Wo] multiplies every sample
// (reference lib/deformable_conv/src/modulated_deform_conv.h:10-44, src/cuda/modulated_deform_conv_cuda.cu:20-118)
extern "C" int hrnet_modulated_deform_conv_forward(const float* input, const float* offset, const float* mask,
                                                   const float* weight, const float* bias, float* output, int B,
                                                   int C, int H, int W, int Co, int kh, int kw, int sh, int sw,
                                                   int ph, int pw, int dh, int dw, int groups,
                                                   int deformable_groups, hr_stream_t stream) {
  HR_REQUIRE(mask, "modulated_deform_conv_forward: null mask");
  return dcn_forward_impl(input, offset, mask, weight, bias, output, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw,
                          groups, deformable_groups, stream);
}

extern "C" int hrnet_deform_conv_wgrad_blocks(int B, int Ho, int Wo) {
  return (int)(((long long)B * Ho * Wo + 256 * WCH - 1) / (256 * WCH));
}

static int dcn_backward_impl(const float* input, const float* offset, const float* mask, const float* weight,
                             const float* grad_output, float* grad_input, float* grad_offset, float* grad_mask,
                             float* grad_weight, float* grad_bias, float* scratch, int B, int C,
                             int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw,
                             int dh, int dw, int groups, int deformable_groups, hr_stream_t stream);

extern "C" int hrnet_deform_conv_backward(const float* input, const float* offset, const float* weight,
                                          const float* grad_output, float* grad_input, float* grad_offset,
                                          float* grad_weight, float* grad_bias, float* scratch, int B, int C,
                                          int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw,
                                          int dh, int dw, int groups, int deformable_groups,
                                          hr_stream_t stream) {
  return dcn_backward_impl(input, offset, nullptr, weight, grad_output, grad_input, grad_offset, nullptr, grad_weight,
                           grad_bias, scratch, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups,
                           deformable_groups, stream);
}

// gradients of the modulated form: + grad_mask [B, deformable_groups * kh * kw, Ho, Wo]
// (reference src/cuda/modulated_deform_conv_cuda.cu:120-285)
extern "C" int hrnet_modulated_deform_conv_backward(const float* input, const float* offset, const float* mask,
                                                    const float* weight, const float* grad_output,
                                                    float* grad_input, float* grad_offset, float* grad_mask,
                                                    float* grad_weight, float* grad_bias, float* scratch, int B,
                                                    int C, int H, int W, int Co, int kh, int kw, int sh, int sw,
                                                    int ph, int pw, int dh, int dw, int groups,
                                                    int deformable_groups, hr_stream_t stream) {
  HR_REQUIRE(mask && grad_mask, "modulated_deform_conv_backward: null mask / grad_mask");
  return dcn_backward_impl(input, offset, mask, weight, grad_output, grad_input, grad_offset, grad_mask, grad_weight,
                           grad_bias, scratch, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups,
                           deformable_groups, stream);
}

static int dcn_backward_impl(const float* input, const float* offset, const float* mask, const float* weight,
                             const float* grad_output, float* grad_input, float* grad_offset, float* grad_mask,
                             float* grad_weight, float* grad_bias, float* scratch, int B, int C,
                             int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw,
                             int dh, int dw, int groups, int deformable_groups, hr_stream_t stream) {
  DcnArgs a = {};
  int Ho, Wo;
  if (int e = fill_common(a, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, deformable_groups, &Ho, &Wo))
    return e;
  HR_REQUIRE(input && offset && weight && grad_output && grad_input && grad_offset && grad_weight && scratch,
             "deform_conv_backward: null pointer");
  hipStream_t s = (hipStream_t)stream;
  a.in = input; a.off = offset; a.w = weight; a.gout = grad_output; a.gin = grad_input; a.goff = grad_offset;
  a.partial = scratch; a.mask = mask; a.gmask = grad_mask;
  const int Cg = C / groups, Og = Co / groups, K = kh * kw;
  HR_REQUIRE(Og <= 64, "deform_conv_backward: out_channels/groups = %d > 64 not supported", Og);
  HR_REQUIRE((size_t)(Og <= 32 ? 32 : 64) * Cg * K * 4 <= 96 * 1024,
             "deform_conv_backward: C/groups * kh * kw = %d too large", Cg * K);
  HR_REQUIRE(((size_t)(Og + K) * 257 + (size_t)Og * Cg * K) * 4 <= 160 * 1024,
             "deform_conv_backward: weight-gradient tile does not fit LDS (Og %d, Cg %d, taps %d)", Og, Cg, K);
  const unsigned wblocks = (unsigned)hrnet_deform_conv_wgrad_blocks(B, Ho, Wo);
  const long long npix = (long long)B * Ho * Wo;
  const unsigned blocks = (unsigned)((npix + 255) / 256);
  const int cpd = C / deformable_groups;
  // single pass over the offsets for all three gradients (PoseAggr: 21 channels = 21 deformable groups, 3x3)
  const int ogp = Og <= 24 ? 24 : 32;
  // 64-bit gradient plane + f32 input plane + weights + Og + 9 rows of 260 floats (PoseAggr: 81 216 bytes - two workgroups per CU)
  const size_t lds_one = ((size_t)3 * H * W + (size_t)9 * ogp + (size_t)(Og + 9) * 260) * 4;
  // (scratch holds hrnet_deform_conv_wgrad_blocks() partials of [Og][C][K]; this path writes B of them, and the B
  // per-image maxima of |grad_output| behind them)
  if (!mask && groups == 1 && cpd == 1 && K == 9 && kh == 3 && Og <= 28 && lds_one <= 128 * 1024 && B <= 65535 &&
      (size_t)(Og + 9) * 260 >= 4 * 32 * 16 &&
      (size_t)hrnet_deform_conv_wgrad_blocks(B, Ho, Wo) * Og * Cg * K >= (size_t)B * Og * Cg * K + B) {
    a.c0 = 0; a.Cg = Cg; a.o0 = 0; a.Og = Og;
    float* gmax = scratch + (size_t)B * Og * Cg * K;
    a.gmax = gmax;
    (void)hipMemsetAsync(gmax, 0, (size_t)B * sizeof(float), s);
    {
      const long long per = (long long)Co * Ho * Wo;
      int gx = (int)((per / 4 + 2047) / 2048);            // ~8 float4 per thread
      gx = gx < 1 ? 1 : gx > 64 ? 64 : gx;
      hipLaunchKernelGGL(dcn_absmax_kernel, dim3(gx, B), dim3(256), 0, s, grad_output, gmax, per);
    }
    if (ogp == 24) {
      want_lds(dcn_bwd_fused_kernel<24, 9>, lds_one);
      hipLaunchKernelGGL((dcn_bwd_fused_kernel<24, 9>), dim3(C, B), dim3(256), lds_one, s, a);
    } else {
      want_lds(dcn_bwd_fused_kernel<32, 9>, lds_one);
      hipLaunchKernelGGL((dcn_bwd_fused_kernel<32, 9>), dim3(C, B), dim3(256), lds_one, s, a);
    }
    const int n = Og * Cg * K;
    hipLaunchKernelGGL(dcn_weight_reduce_kernel, dim3((n + 3) / 4), dim3(256), 0, s, (const float*)scratch, grad_weight,
                       B, n, 0);
    if (grad_bias)
      hipLaunchKernelGGL(dcn_bias_grad_kernel, dim3(Co), dim3(256), 0, s, grad_output, grad_bias, B, Co,
                         (long long)Ho * Wo, 0);
    return hr_check_launch("deform_conv_backward");
  }
  const size_t lds_fast = ((size_t)cpd * K * (Og <= 32 ? 32 : 64) + (size_t)cpd * H * W) * 4;
  const bool fast = !mask && Cg % cpd == 0 && lds_fast <= 64 * 1024;     // (the modulated form runs the generic kernels)
  if (!fast) (void)hipMemsetAsync(grad_input, 0, (size_t)B * C * H * W * sizeof(float), s);
  if (Og <= 32) want_lds(dcn_bwd_data_kernel<32>, (size_t)32 * Cg * K * 4);
  else want_lds(dcn_bwd_data_kernel<64>, (size_t)64 * Cg * K * 4);
  want_lds(dcn_bwd_weight_kernel, ((size_t)(Og + K) * 257 + (size_t)Og * Cg * K) * 4);
  for (int g = 0; g < groups; ++g) {
    a.c0 = g * Cg; a.Cg = Cg; a.o0 = g * Og; a.Og = Og;
    if (fast && Og <= 32)
      hipLaunchKernelGGL(dcn_bwd_data_lds_kernel<32>, dim3(Cg / cpd, B), dim3(256), lds_fast, s, a);
    else if (fast)
      hipLaunchKernelGGL(dcn_bwd_data_lds_kernel<64>, dim3(Cg / cpd, B), dim3(256), lds_fast, s, a);
    else if (Og <= 32)
      hipLaunchKernelGGL(dcn_bwd_data_kernel<32>, dim3(blocks), dim3(256), (size_t)32 * Cg * K * 4, s, a);
    else
      hipLaunchKernelGGL(dcn_bwd_data_kernel<64>, dim3(blocks), dim3(256), (size_t)64 * Cg * K * 4, s, a);
    const int n = Og * Cg * K;
    hipLaunchKernelGGL(dcn_bwd_weight_kernel, dim3(wblocks), dim3(256), ((size_t)(Og + K) * 257 + n) * 4, s, a);
    hipLaunchKernelGGL(dcn_weight_reduce_kernel, dim3((n + 3) / 4), dim3(256), 0, s, (const float*)scratch,
                       grad_weight + (size_t)g * n, (int)wblocks, n, 0);
  }
  if (grad_bias)
    hipLaunchKernelGGL(dcn_bias_grad_kernel, dim3(Co), dim3(256), 0, s, grad_output, grad_bias, B, Co,
                       (long long)Ho * Wo, 0);
  return hr_check_launch("deform_conv_backward");
}
