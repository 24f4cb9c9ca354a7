// Heat-map loss, key-point decode, key-point loss and the Adam step: small HBM/latency-bound
// kernels on the NCHW f32 tensors of the module contract. One wavefront-friendly block per
// (batch, joint) map; reductions by DPP/shuffle then LDS, in a fixed order.
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum64(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
  for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k];
  __syncthreads();
  return s;
}

__device__ __forceinline__ float block_max(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float m = red[0];
  for (int k = 1; k < (int)(blockDim.x >> 6); ++k) m = fmaxf(m, red[k]);
  __syncthreads();
  return m;
}

// spatial softmax with temperature, one block per (batch, joint) map (pose_hrnet_softmax.py:520-524)
__global__ __launch_bounds__(256) void spatial_softmax_fwd_kernel(const float* x, const float* temp, float* out,
                                                                  int HW) {
  __shared__ float red[4];
  const size_t base = (size_t)blockIdx.x * HW;
  const float t = temp[0];
  float m = -INFINITY;
  for (int i = threadIdx.x; i < HW; i += 256) m = fmaxf(m, x[base + i] * t);
  m = block_max(m, red);
  float s = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) {
    const float e = expf(x[base + i] * t - m);
    out[base + i] = e;
    s += e;
  }
  s = block_sum(s, red);
  const float inv = 1.f / s;
  for (int i = threadIdx.x; i < HW; i += 256) out[base + i] *= inv;
}

__global__ __launch_bounds__(256) void spatial_softmax_bwd_kernel(const float* x, const float* out,
                                                                  const float* gout, const float* temp, float* dx,
                                                                  float* dtemp_partial, int HW) {
  __shared__ float red[4];
  const size_t base = (size_t)blockIdx.x * HW;
  const float t = temp[0];
  float dot = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) dot = fmaf(gout[base + i], out[base + i], dot);
  dot = block_sum(dot, red);
  float dt = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) {
    const float dz = out[base + i] * (gout[base + i] - dot);
    dx[base + i] = t * dz;
    dt = fmaf(dz, x[base + i], dt);
  }
  dt = block_sum(dt, red);
  if (threadIdx.x == 0) dtemp_partial[blockIdx.x] = dt;
}

// Gaussian heat-map targets (reference lib/dataset/target_generators/target_generators.py:14-53): peak 1 at
// int(coord), window [x-3s-1, x+3s+2), zero map for an invisible or out-of-range joint. One block per map.
__global__ __launch_bounds__(256) void gaussian_targets_kernel(const float* pose2d, const float* vis, float* out,
                                                               int H, int W, float sigma) {
  const int bk = blockIdx.x;
  const float fx = pose2d[bk * 2 + 0], fy = pose2d[bk * 2 + 1];
  const int x = (int)fx, y = (int)fy;                      // int(): truncation, as the reference
  const bool on = (!vis || vis[bk] > 0.f) && x >= 0 && y >= 0 && x < W && y < H;
  const int ulx = (int)rintf((float)x - 3.f * sigma - 1.f), uly = (int)rintf((float)y - 3.f * sigma - 1.f);
  const int brx = (int)rintf((float)x + 3.f * sigma + 2.f), bry = (int)rintf((float)y + 3.f * sigma + 2.f);
  // the reference's kernel image g has its peak at index 3*sigma+1 of a (6*sigma+3)-wide window placed at ul
  const double c = 3.0 * (double)sigma + 1.0, inv = 1.0 / (2.0 * (double)sigma * (double)sigma);
  float* o = out + (size_t)bk * H * W;
  for (int i = threadIdx.x; i < H * W; i += 256) {
    const int py = i / W, px = i % W;
    float v = 0.f;
    if (on && px >= ulx && px < brx && py >= uly && py < bry) {
      const double dx = (double)(px - ulx) - c, dy = (double)(py - uly) - c;
      v = (float)exp(-(dx * dx + dy * dy) * inv);
    }
    o[i] = v;
  }
}

// ToTensor + Normalize of the reference's input pipeline (lib/dataset/transforms/build.py:84-85,
// transforms.py:38-51): HWC u8 -> CHW f32, (v/255 - mean[c]) / std[c]
__global__ __launch_bounds__(256) void normalize_u8_kernel(const unsigned char* img, float* out, long long npix,
                                                           long long hw, float m0, float m1, float m2, float s0,
                                                           float s1, float s2) {
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix;
       p += (long long)gridDim.x * blockDim.x) {
    const long long n = p / hw, r = p % hw;
    const unsigned char* q = img + p * 3;
    float* o = out + n * 3 * hw + r;
    o[0] = ((float)q[0] / 255.f - m0) / s0;
    o[hw] = ((float)q[1] / 255.f - m1) / s1;
    o[2 * hw] = ((float)q[2] / 255.f - m2) / s2;
  }
}

// per-map sum of (pred-gt)^2 or |pred-gt|
__global__ __launch_bounds__(256) void hm_loss_map_kernel(const float* pred, const float* gt,
                                                          float* partial, int HW, int mode) {
  __shared__ float red[4];
  const size_t base = (size_t)blockIdx.x * HW;
  float s = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) {
    const float d = pred[base + i] - gt[base + i];
    s += mode == 0 ? d * d : fabsf(d);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void mean_kernel(const float* partial, float* out, int n, float denom) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s / denom;
}

__global__ __launch_bounds__(256) void hm_loss_bwd_kernel(const float* pred, const float* gt,
                                                          const float* gout, float* dpred,
                                                          long long n, float inv_bk, int mode) {
  const float g = gout[0] * inv_bk;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float d = pred[i] - gt[i];
    dpred[i] = mode == 0 ? 2.f * d * g : (d > 0.f ? g : (d < 0.f ? -g : 0.f));
  }
}

__global__ __launch_bounds__(256) void decode_expect_kernel(const float* hms, float* preds, int H, int W) {
  __shared__ float red[4];
  const size_t base = (size_t)blockIdx.x * H * W;
  float sx = 0.f, sy = 0.f;
  for (int i = threadIdx.x; i < H * W; i += 256) {
    const float h = hms[base + i];
    sx = fmaf(h, (float)(i % W), sx);
    sy = fmaf(h, (float)(i / W), sy);
  }
  sx = block_sum(sx, red);
  sy = block_sum(sy, red);
  if (threadIdx.x == 0) {
    preds[blockIdx.x * 2 + 0] = sx;
    preds[blockIdx.x * 2 + 1] = sy;
  }
}

__global__ __launch_bounds__(256) void decode_expect_bwd_kernel(const float* gpreds, float* dhms, int H,
                                                                int W, int accumulate) {
  const size_t base = (size_t)blockIdx.x * H * W;
  const float gx = gpreds[blockIdx.x * 2], gy = gpreds[blockIdx.x * 2 + 1];
  for (int i = threadIdx.x; i < H * W; i += 256) {
    const float v = gx * (float)(i % W) + gy * (float)(i / W);
    dhms[base + i] = accumulate ? dhms[base + i] + v : v;
  }
}

// first maximal flat index (torch.argmax / numpy.argmax tie rule), NaN treated as maximal like torch
__global__ __launch_bounds__(256) void decode_argmax_kernel(const float* hms, float* preds, float* maxvals,
                                                            int H, int W, int inference_style) {
  __shared__ float rv[256];
  __shared__ int ri[256];
  const size_t base = (size_t)blockIdx.x * H * W;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < H * W; i += 256) {
    const float h = hms[base + i];
    const bool better = (h > best) || (h != h && !(best != best)) || (bi == 0x7fffffff);
    if (better) {
      best = h;
      bi = i;
    }
  }
  rv[threadIdx.x] = best;
  ri[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const float ov = rv[threadIdx.x + s];
      const int oi = ri[threadIdx.x + s];
      const float mv = rv[threadIdx.x];
      const int mi = ri[threadIdx.x];
      const bool o_nan = ov != ov, m_nan = mv != mv;
      bool take;
      if (oi == 0x7fffffff) take = false;
      else if (mi == 0x7fffffff) take = true;
      else if (o_nan != m_nan) take = o_nan;
      else if (o_nan) take = oi < mi;
      else take = (ov > mv) || (ov == mv && oi < mi);
      if (take) {
        rv[threadIdx.x] = ov;
        ri[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int idx = ri[0];
    const float mx = rv[0];
    float u, v;
    if (inference_style) {  // lib/core/inference.py:38-44: W for % and /, zero where max <= 0
      u = (float)(idx % W);
      v = (float)(idx / W);
      if (!(mx > 0.f)) u = v = 0.f;
    } else {  // lib/utils/heatmap_decoding.py:105: H for both
      u = (float)(idx % H);
      v = (float)(idx / H);
    }
    preds[blockIdx.x * 2] = u;
    preds[blockIdx.x * 2 + 1] = v;
    if (maxvals) maxvals[blockIdx.x] = mx;
  }
}

__global__ __launch_bounds__(256) void joints_loss_fwd_kernel(const float* pred, const float* gt,
                                                              const float* vis, float* loss, int B, int K) {
  __shared__ float red[4];
  float s = 0.f, sv = 0.f;
  for (int i = threadIdx.x; i < B * K; i += 256) {
    const float dx = pred[2 * i] - gt[2 * i], dy = pred[2 * i + 1] - gt[2 * i + 1];
    const float nrm = sqrtf(dx * dx + dy * dy);
    const float w = vis ? vis[i] : 1.f;
    s = fmaf(nrm, w, s);
    sv += w;
  }
  s = block_sum(s, red);
  sv = block_sum(sv, red);
  if (threadIdx.x == 0) loss[0] = vis ? s / fmaxf(1.f, sv) : s / (float)K;
}

__global__ __launch_bounds__(256) void joints_loss_bwd_kernel(const float* pred, const float* gt,
                                                              const float* vis, const float* gout,
                                                              float* dpred, int B, int K) {
  __shared__ float red[4];
  float sv = 0.f;
  if (vis)
    for (int i = threadIdx.x; i < B * K; i += 256) sv += vis[i];
  sv = block_sum(sv, red);
  const float denom = vis ? fmaxf(1.f, sv) : (float)K;
  const float g = gout[0] / denom;
  for (int i = threadIdx.x; i < B * K; i += 256) {
    const float dx = pred[2 * i] - gt[2 * i], dy = pred[2 * i + 1] - gt[2 * i + 1];
    const float nrm = sqrtf(dx * dx + dy * dy);
    const float w = (vis ? vis[i] : 1.f) * g;
    // torch.norm backward: x / ||x||, 0 at the origin
    dpred[2 * i] = nrm > 0.f ? w * dx / nrm : 0.f;
    dpred[2 * i + 1] = nrm > 0.f ? w * dy / nrm : 0.f;
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v,
                                                   long long n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2_sqrt, float gscale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    gi = fmaf(wd, pi, gi);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}

}  // namespace

extern "C" int hrnet_heatmap_loss_fwd(const float* pred, const float* gt, float* partial, float* loss,
                                      int BK, int HW, int mode, hr_stream_t stream) {
  HR_REQUIRE(pred && gt && partial && loss && BK > 0 && HW > 0 && (mode == 0 || mode == 1),
             "heatmap_loss_fwd: args");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(hm_loss_map_kernel, dim3(BK), dim3(256), 0, s, pred, gt, partial, HW, mode);
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, s, (const float*)partial, loss, BK, (float)BK);
  return hr_check_launch("heatmap_loss_fwd");
}

extern "C" int hrnet_heatmap_loss_bwd(const float* pred, const float* gt, const float* gout,
                                      float* dpred, int BK, int HW, int mode, hr_stream_t stream) {
  HR_REQUIRE(pred && gt && gout && dpred && BK > 0 && HW > 0, "heatmap_loss_bwd: args");
  const long long n = (long long)BK * HW;
  long long grid = (n + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(hm_loss_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, pred, gt,
                     gout, dpred, n, 1.0f / (float)BK, mode);
  return hr_check_launch("heatmap_loss_bwd");
}

extern "C" int hrnet_gaussian_targets(const float* pose2d, const float* visibility, float* heatmaps, int BK, int H,
                                      int W, float sigma, hr_stream_t stream) {
  HR_REQUIRE(pose2d && heatmaps && BK > 0 && H > 0 && W > 0 && sigma > 0.f, "gaussian_targets: args");
  hipLaunchKernelGGL(gaussian_targets_kernel, dim3(BK), dim3(256), 0, (hipStream_t)stream, pose2d, visibility,
                     heatmaps, H, W, sigma);
  return hr_check_launch("gaussian_targets");
}

extern "C" int hrnet_normalize_u8(const unsigned char* img_nhwc, float* out_nchw, int N, int H, int W,
                                  const float* mean3, const float* std3, hr_stream_t stream) {
  HR_REQUIRE(img_nhwc && out_nchw && mean3 && std3 && N > 0 && H > 0 && W > 0, "normalize_u8: args");
  HR_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "normalize_u8: zero std");
  const long long hw = (long long)H * W, npix = hw * N;
  long long grid = (npix + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(normalize_u8_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, img_nhwc, out_nchw,
                     npix, hw, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  return hr_check_launch("normalize_u8");
}

extern "C" int hrnet_spatial_softmax_fwd(const float* x, const float* temp, float* out, int BK, int HW,
                                         hr_stream_t stream) {
  HR_REQUIRE(x && temp && out && BK > 0 && HW > 0, "spatial_softmax_fwd: args");
  hipLaunchKernelGGL(spatial_softmax_fwd_kernel, dim3(BK), dim3(256), 0, (hipStream_t)stream, x, temp, out, HW);
  return hr_check_launch("spatial_softmax_fwd");
}

extern "C" int hrnet_spatial_softmax_bwd(const float* x, const float* out, const float* gout, const float* temp,
                                         float* dx, float* dtemp_partial, int BK, int HW, hr_stream_t stream) {
  HR_REQUIRE(x && out && gout && temp && dx && dtemp_partial && BK > 0 && HW > 0, "spatial_softmax_bwd: args");
  hipLaunchKernelGGL(spatial_softmax_bwd_kernel, dim3(BK), dim3(256), 0, (hipStream_t)stream, x, out, gout, temp,
                     dx, dtemp_partial, HW);
  return hr_check_launch("spatial_softmax_bwd");
}

extern "C" int hrnet_decode_expectation(const float* hms, float* preds, int BK, int H, int W,
                                        hr_stream_t stream) {
  HR_REQUIRE(hms && preds && BK > 0 && H > 0 && W > 0, "decode_expectation: args");
  hipLaunchKernelGGL(decode_expect_kernel, dim3(BK), dim3(256), 0, (hipStream_t)stream, hms, preds, H, W);
  return hr_check_launch("decode_expectation");
}

extern "C" int hrnet_decode_expectation_bwd(const float* gpreds, float* dhms, int BK, int H, int W,
                                            int accumulate, hr_stream_t stream) {
  HR_REQUIRE(gpreds && dhms && BK > 0, "decode_expectation_bwd: args");
  hipLaunchKernelGGL(decode_expect_bwd_kernel, dim3(BK), dim3(256), 0, (hipStream_t)stream, gpreds, dhms,
                     H, W, accumulate);
  return hr_check_launch("decode_expectation_bwd");
}

extern "C" int hrnet_decode_argmax(const float* hms, float* preds, float* maxvals, int BK, int H, int W,
                                   int inference_style, hr_stream_t stream) {
  HR_REQUIRE(hms && preds && BK > 0 && H > 0 && W > 0, "decode_argmax: args");
  hipLaunchKernelGGL(decode_argmax_kernel, dim3(BK), dim3(256), 0, (hipStream_t)stream, hms, preds,
                     maxvals, H, W, inference_style);
  return hr_check_launch("decode_argmax");
}

extern "C" int hrnet_joints_loss_fwd(const float* pred, const float* gt, const float* vis, float* loss,
                                     int B, int K, hr_stream_t stream) {
  HR_REQUIRE(pred && gt && loss && B > 0 && K > 0, "joints_loss_fwd: args");
  hipLaunchKernelGGL(joints_loss_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, gt, vis,
                     loss, B, K);
  return hr_check_launch("joints_loss_fwd");
}

extern "C" int hrnet_joints_loss_bwd(const float* pred, const float* gt, const float* vis,
                                     const float* gout, float* dpred, int B, int K, hr_stream_t stream) {
  HR_REQUIRE(pred && gt && gout && dpred && B > 0 && K > 0, "joints_loss_bwd: args");
  hipLaunchKernelGGL(joints_loss_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, gt, vis,
                     gout, dpred, B, K);
  return hr_check_launch("joints_loss_bwd");
}

extern "C" int hrnet_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                               int64_t n, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int step, float grad_scale, hr_stream_t stream) {
  HR_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adam_step: args");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  long long grid = (n + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, param, grad,
                     exp_avg, exp_avg_sq, (long long)n, lr, beta1, beta2, eps, weight_decay, (float)bc1,
                     (float)sqrt(bc2), grad_scale);
  return hr_check_launch("adam_step");
}
