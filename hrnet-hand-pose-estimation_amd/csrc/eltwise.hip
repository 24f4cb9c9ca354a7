// HBM-bound companions of the conv kernels: BatchNorm finalisation, fused multi-term sums
// (residual adds, fuse-layer sums with nearest upsampling), their backward, the bilinear
// head concat, layout conversions and weight packing. All are 16-byte-per-lane streaming
// kernels; reductions are two-stage and deterministic (no float atomics).
#include "common.h"

namespace {

constexpr int EW_BLOCK = 256;
inline unsigned ew_grid(long long n) {
  long long g = (n + EW_BLOCK - 1) / EW_BLOCK;
  if (g > 256LL * 16) g = 256LL * 16;  // cap + grid-stride
  if (g < 1) g = 1;
  return (unsigned)g;
}

// ---------------------------------------------------------------------------------------
// BatchNorm finalize: stats[tiles][2][C] -> scale/shift (+ running stats update)
// ---------------------------------------------------------------------------------------
// Column sums of rows[n][2][C] (per-workgroup partials) for the block's 32 channels: a block of
// 1024 threads = 32 channels x 32 row lanes, 4 independent loads in flight per lane (the row count
// reaches ~4000, and a serial per-channel loop is pure memory latency), f64 accumulation, LDS tree.
constexpr int FIN_LANES = 32;
__device__ __forceinline__ void partial_sums(const float* rows, int n, int C, int c, int tl, int cl,
                                             double (*red)[FIN_LANES][32], double& s1, double& s2) {
  double a0 = 0, a1 = 0, b0 = 0, b1 = 0;
  if (c < C) {
    int t = tl;
    for (; t + FIN_LANES < n; t += 2 * FIN_LANES) {
      const float x0 = rows[((size_t)t * 2 + 0) * C + c], y0 = rows[((size_t)t * 2 + 1) * C + c];
      const float x1 = rows[((size_t)(t + FIN_LANES) * 2 + 0) * C + c];
      const float y1 = rows[((size_t)(t + FIN_LANES) * 2 + 1) * C + c];
      a0 += (double)x0; b0 += (double)y0; a1 += (double)x1; b1 += (double)y1;
    }
    if (t < n) {
      a0 += (double)rows[((size_t)t * 2 + 0) * C + c];
      b0 += (double)rows[((size_t)t * 2 + 1) * C + c];
    }
  }
  red[0][tl][cl] = a0 + a1;
  red[1][tl][cl] = b0 + b1;
  __syncthreads();
  for (int s = FIN_LANES / 2; s > 0; s >>= 1) {
    if (tl < s) {
      red[0][tl][cl] += red[0][tl + s][cl];
      red[1][tl][cl] += red[1][tl + s][cl];
    }
    __syncthreads();
  }
  s1 = red[0][0][cl];
  s2 = red[1][0][cl];
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(
    const float* stats, int tiles, int C, float count, const float* gamma, const float* beta,
    float* running_mean, float* running_var, long long* nbt, float momentum, float eps, int training,
    float* scale, float* shift, float* save_mean, float* save_invstd) {
  __shared__ double red[2][FIN_LANES][32];
  const int cl = threadIdx.x & 31, tl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (training) partial_sums(stats, tiles, C, c, tl, cl, red, s1, s2);
  if (tl == 0 && c < C) {
    float mean, var;
    if (training) {
      const double m = s1 / (double)count;
      double v = s2 / (double)count - m * m;
      if (v < 0.0) v = 0.0;
      mean = (float)m;
      var = (float)v;
      if (running_mean) {
        const float unbiased = count > 1.f ? (float)(v * (double)count / ((double)count - 1.0)) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
      }
    } else {
      mean = running_mean[c];
      var = running_var[c];
    }
    const float invstd = 1.0f / sqrtf(var + eps);
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    if (save_mean) save_mean[c] = mean;
    if (save_invstd) save_invstd[c] = invstd;
  }
  if (training && nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
}

// ---------------------------------------------------------------------------------------
// sum_terms: out = relu_out( sum_t relu_t( src_t[up 2^sh_t] * scale_t + shift_t ) )
// ---------------------------------------------------------------------------------------
struct SumArgs {
  char* out;
  const char* src[4];
  const float* scale[4];
  const float* shift[4];
  int sh[4];
  int relu[4];
  int N, Ho, Wo, C, nterms, relu_out;
  // bit t of sums_mode: term t's BatchNorm is given as batch sums (scale[t] = sums[8][2][C], shift[t] = gamma with
  // beta = gamma + C) and turned into scale/shift per block (hr_bn_from_sums): no finalize launch before this op
  int sums_mode;
  float inv_count[4];
  float eps;
};

constexpr int SUM_MAXC = 384;

// SUMS: some term's BatchNorm comes as batch sums (training forward); the plain instantiation (eval forward,
// deterministic training) carries neither the table nor the per-term test in its inner loop
// (block `block0` of `nblocks`: a launch of its own, or one job of a table-driven launch)
template <typename T, bool SUMS>
__device__ __forceinline__ void sum_terms_block(const SumArgs& a, int block0, int nblocks,
                                                float (*tab)[2][SUMS ? SUM_MAXC : 1]) {
  constexpr int VEC = TT<T>::VEC;
  if (SUMS && a.sums_mode) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < a.nterms && ((a.sums_mode >> t) & 1)) {
        for (int c = threadIdx.x; c < a.C; c += 256) {
          float sc_, sh_, m_, r_, v_;
          hr_bn_from_sums(a.scale[t], a.C, c, a.inv_count[t], a.eps, a.shift[t][c], a.shift[t][a.C + c], sc_, sh_, m_, r_, v_);
          tab[t][0][c] = sc_;
          tab[t][1][c] = sh_;
        }
      }
    }
    __syncthreads();
  }
  const int cv = a.C / VEC;
  const long long total = (long long)a.N * a.Ho * a.Wo * cv;
  for (long long idx = (long long)block0 * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)nblocks * blockDim.x) {
    const int v = (int)(idx % cv);
    long long pix = idx / cv;
    const int ox = (int)(pix % a.Wo);
    pix /= a.Wo;
    const int oy = (int)(pix % a.Ho);
    const int n = (int)(pix / a.Ho);
    const int c = v * VEC;
    float accv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) accv[j] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < a.nterms) {
        const int sh = a.sh[t];
        const int hs = a.Ho >> sh, ws = a.Wo >> sh;
        const size_t off = ((size_t)((n * hs + (oy >> sh)) * ws + (ox >> sh)) * a.C + c) * sizeof(T);
        float f[VEC];
        v16_unpack<T>(*(const V16*)(a.src[t] + off), f);
        if (SUMS && ((a.sums_mode >> t) & 1)) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) f[j] = fmaf(f[j], tab[SUMS ? t : 0][0][SUMS ? c + j : 0], tab[SUMS ? t : 0][1][SUMS ? c + j : 0]);
        } else if (a.scale[t]) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) f[j] = fmaf(f[j], a.scale[t][c + j], a.shift[t][c + j]);
        }
        if (a.relu[t]) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) f[j] = fmaxf(f[j], 0.f);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) accv[j] += f[j];
      }
    }
    if (a.relu_out) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) accv[j] = fmaxf(accv[j], 0.f);
    }
    *(V16*)(a.out + (size_t)idx * 16) = v16_pack<T>(accv);
  }
}

template <typename T, bool SUMS>
__global__ __launch_bounds__(256) void sum_terms_kernel(SumArgs a) {
  __shared__ float tab[SUMS ? 4 : 1][2][SUMS ? SUM_MAXC : 1];
  sum_terms_block<T, SUMS>(a, (int)blockIdx.x, (int)gridDim.x, tab);
}

// slots of a HR_OP_SUM_TERMS op -> SumArgs (host: hr_launch_sum_terms; device: the table-driven launch)
__host__ __device__ inline void sum_args_from_op(const HrOp& op, SumArgs& a) {
  a.N = op.i[1]; a.Ho = op.i[2]; a.Wo = op.i[3]; a.C = op.i[4]; a.nterms = op.i[5]; a.relu_out = op.i[6];
  a.out = (char*)op.p[0];
  a.sums_mode = op.i[15];
  for (int t = 0; t < 4; ++t) a.inv_count[t] = op.f[t];
  a.eps = __builtin_bit_cast(float, op.i[16]);
  for (int t = 0; t < 4; ++t) {
    a.sh[t] = op.i[7 + t];
    a.relu[t] = op.i[11 + t];
    a.src[t] = (const char*)op.p[1 + t];
    a.scale[t] = (const float*)op.p[5 + t];
    a.shift[t] = (const float*)op.p[9 + t];
  }
}

// ---------------------------------------------------------------------------------------
// grad_term / bn_bwd_reduce share the dz computation:
//   dz[q,c] = sum_{p in block(q)} g[p,c] * [mask_out[p,c] > 0] * [scale*y+shift > 0 if inner_relu]
// ---------------------------------------------------------------------------------------
struct GradArgs {
  char* dst;
  const char* g;
  const char* mask;
  const char* y;
  const float* scale;
  const float* shift;
  const float* coef;
  float* partials;
  char* dst2;   // optional second destination: dst2 (+)= dz (the residual / identity term of the same sum)
  int N, H, W, C, sh, inner_relu, accumulate, accumulate2;
};

template <typename T>
__device__ __forceinline__ void compute_dz(const GradArgs& a, int n, int qy, int qx, int c,
                                           float* dz, float* yv, bool need_y) {
  constexpr int VEC = TT<T>::VEC;
  const int f = 1 << a.sh;
  const int Hg = a.H << a.sh, Wg = a.W << a.sh;
#pragma unroll
  for (int j = 0; j < VEC; ++j) dz[j] = 0.f;
  for (int dy = 0; dy < f; ++dy)
    for (int dx = 0; dx < f; ++dx) {
      const size_t off =
          ((size_t)((n * Hg + (qy << a.sh) + dy) * Wg + (qx << a.sh) + dx) * a.C + c) * sizeof(T);
      float gv[VEC];
      v16_unpack<T>(*(const V16*)(a.g + off), gv);
      if (a.mask) {
        float mv[VEC];
        v16_unpack<T>(*(const V16*)(a.mask + off), mv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) gv[j] = mv[j] > 0.f ? gv[j] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) dz[j] += gv[j];
    }
  if (need_y || a.inner_relu) {
    const size_t yoff = ((size_t)((n * a.H + qy) * a.W + qx) * a.C + c) * sizeof(T);
    v16_unpack<T>(*(const V16*)(a.y + yoff), yv);
    if (a.inner_relu) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float z = a.scale ? fmaf(yv[j], a.scale[c + j], a.shift[c + j]) : yv[j];
        dz[j] = z > 0.f ? dz[j] : 0.f;
      }
    }
  }
}

// (bid, nb: this block's index / the number of blocks of the job - a launch of its own, or a slice of a batched one)
template <typename T>
__device__ __forceinline__ void grad_term_block(const GradArgs& a, int bid, int nb) {
  constexpr int VEC = TT<T>::VEC;
  const int cv = a.C / VEC;
  const long long total = (long long)a.N * a.H * a.W * cv;
  for (long long idx = (long long)bid * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)nb * blockDim.x) {
    const int v = (int)(idx % cv);
    long long pix = idx / cv;
    const int qx = (int)(pix % a.W);
    pix /= a.W;
    const int qy = (int)(pix % a.H);
    const int n = (int)(pix / a.H);
    const int c = v * VEC;
    float dz[VEC], yv[VEC];
    compute_dz<T>(a, n, qy, qx, c, dz, yv, a.coef != nullptr);
    float o[VEC];
    if (a.coef) {
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        o[j] = fmaf(a.coef[c + j], dz[j], fmaf(a.coef[a.C + c + j], yv[j], a.coef[2 * a.C + c + j]));
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = dz[j];
    }
    char* d = a.dst + (size_t)idx * 16;
    if (a.accumulate) {
      float old[VEC];
      v16_unpack<T>(*(const V16*)d, old);
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] += old[j];
    }
    *(V16*)d = v16_pack<T>(o);
    if (a.dst2) {
      char* d2 = a.dst2 + (size_t)idx * 16;
      if (a.accumulate2) {
        float old[VEC];
        v16_unpack<T>(*(const V16*)d2, old);
#pragma unroll
        for (int j = 0; j < VEC; ++j) dz[j] += old[j];
      }
      *(V16*)d2 = v16_pack<T>(dz);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void grad_term_kernel(GradArgs a) {
  grad_term_block<T>(a, (int)blockIdx.x, (int)gridDim.x);
}

// The same for large tensors (several grid-stride steps per thread): a thread owns ONE channel vector
// (v = tid % cv) and walks pixels, so the per-channel coefficients are loaded once instead of per element
// (the 480-channel head tensor ran at 1.6 TB/s with 40 coefficient loads per 16 bytes of data). sh == 0 only.
template <typename T>
__global__ __launch_bounds__(256) void grad_term_rows_kernel(GradArgs a) {
  constexpr int VEC = TT<T>::VEC;
  const int cv = a.C / VEC;
  const int rows = 256 / cv;
  const int v = threadIdx.x % cv, row = threadIdx.x / cv;
  if (row >= rows) return;
  const int c = v * VEC;
  const unsigned npix = (unsigned)a.N * a.H * a.W;
  float cA[VEC], cB[VEC], cC[VEC], sc[VEC], sf[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    cA[j] = a.coef ? a.coef[c + j] : 1.f;
    cB[j] = a.coef ? a.coef[a.C + c + j] : 0.f;
    cC[j] = a.coef ? a.coef[2 * a.C + c + j] : 0.f;
    sc[j] = (a.inner_relu && a.scale) ? a.scale[c + j] : 1.f;
    sf[j] = (a.inner_relu && a.scale) ? a.shift[c + j] : 0.f;
  }
  for (unsigned pix = blockIdx.x * rows + row; pix < npix; pix += gridDim.x * rows) {
    const size_t off = ((size_t)pix * a.C + c) * sizeof(T);
    float dz[VEC], yv[VEC];
    v16_unpack<T>(*(const V16*)(a.g + off), dz);
    if (a.mask) {
      float mv[VEC];
      v16_unpack<T>(*(const V16*)(a.mask + off), mv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) dz[j] = mv[j] > 0.f ? dz[j] : 0.f;
    }
    if (a.coef || a.inner_relu) {
      v16_unpack<T>(*(const V16*)(a.y + off), yv);
      if (a.inner_relu) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) dz[j] = fmaf(yv[j], sc[j], sf[j]) > 0.f ? dz[j] : 0.f;
      }
    }
    float o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = a.coef ? fmaf(cA[j], dz[j], fmaf(cB[j], yv[j], cC[j])) : dz[j];
    char* d = a.dst + off;
    if (a.accumulate) {
      float old[VEC];
      v16_unpack<T>(*(const V16*)d, old);
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] += old[j];
    }
    *(V16*)d = v16_pack<T>(o);
    if (a.dst2) {
      char* d2 = a.dst2 + off;
      if (a.accumulate2) {
        float old[VEC];
        v16_unpack<T>(*(const V16*)d2, old);
#pragma unroll
        for (int j = 0; j < VEC; ++j) dz[j] += old[j];
      }
      *(V16*)d2 = v16_pack<T>(dz);
    }
  }
}

// partials[block][2][C]; block = rows x cv threads, each thread owns one channel vector
template <typename T>
__device__ __forceinline__ void bn_bwd_reduce_block(const GradArgs& a, int bid, int nb, float* red) {
  constexpr int VEC = TT<T>::VEC;
  const int cv = a.C / VEC;
  const int rows = 256 / cv;
  const int v = threadIdx.x % cv, row = threadIdx.x / cv;
  const int c = v * VEC;
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s1[j] = s2[j] = 0.f;
  const long long npix = (long long)a.N * a.H * a.W;
  if (row < rows) {
    for (long long pix = (long long)bid * rows + row; pix < npix; pix += (long long)nb * rows) {
      const int qx = (int)(pix % a.W);
      const long long r = pix / a.W;
      const int qy = (int)(r % a.H);
      const int n = (int)(r / a.H);
      float dz[VEC], yv[VEC];
      compute_dz<T>(a, n, qy, qx, c, dz, yv, true);
      // (optional) keep the pooled, masked gradient: the apply pass reads it instead of pooling and masking the
      // full-resolution tensors a second time; the sums are those of the STORED values (rounded to T once)
      if (a.dst) {
        const V16 packed = v16_pack<T>(dz);
        *(V16*)(a.dst + (size_t)(pix * cv + v) * 16) = packed;
        v16_unpack<T>(packed, dz);
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        s1[j] += dz[j];
        s2[j] = fmaf(dz[j], yv[j], s2[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    red[(threadIdx.x * 2 + 0) * VEC + j] = s1[j];
    red[(threadIdx.x * 2 + 1) * VEC + j] = s2[j];
  }
  __syncthreads();
  // thread t < 2*C sums over rows for (which, channel)
  for (int o = threadIdx.x; o < 2 * a.C; o += 256) {
    const int which = o / a.C, ch = o % a.C;
    const int vv = ch / VEC, jj = ch % VEC;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += red[((r * cv + vv) * 2 + which) * VEC + jj];
    a.partials[((size_t)bid * 2 + which) * a.C + ch] = s;
  }
}

// The reduction pass of SEVERAL up-sampled terms of one fuse sum at once (HR_OP_POOL_REDUCE): output i of a
// HighResolutionModule adds nearest-up-sampled terms from the branches below it (pose_hrnet.py:257-264), so the
// BatchNorm backward of the term from branch i + l needs G_i (masked by the sum's ReLU) pooled over 2^l x 2^l blocks -
// the blocks nest, so ONE walk over G_i and the mask forms every level (level l + 1 = the sum of four level-l values,
// in f32), stores each level's dz (rounded to T once; the apply pass reads it) and its (sum dz, sum dz*y) partials.
// Before, every term pooled the full-resolution tensors on its own (three terms on branch 0: 3 x 33.6 MB).
struct PoolArgs {
  const char* g;         // [N][H][W][C] upstream gradient of the sum output
  const char* mask;      // [N][H][W][C] the sum output (ReLU mask [> 0]) or NULL
  const char* y[3];      // raw conv output of the term at level l + 1: [N][H >> (l+1)][W >> (l+1)][C]
  char* dz[3];           // same shape: pooled, masked gradient (out)
  float* partials[3];    // [blocks][2][C]
  int N, H, W, C, nlev;
};

// One thread per level-1 block (2x2 pixels) and channel vector; the 256/cv "rows" of a workgroup are consecutive
// level-1 blocks in nested (Morton) order, so the four children of a level-2 block are rows 4k .. 4k+3 and the sixteen
// grand-children of a level-3 block rows 16k .. 16k+15: the higher levels are added through LDS in a fixed order.
template <typename T, int NLEV>
__device__ __forceinline__ void pool_reduce_block(const PoolArgs& a, int bid, int nb, float* red) {
  constexpr int VEC = TT<T>::VEC;
  const int cv = a.C / VEC;
  constexpr int GROUP = NLEV == 3 ? 16 : (NLEV == 2 ? 4 : 1);      // level-1 blocks per coarsest block
  const int rows = 256 / cv / GROUP * GROUP;    // level-1 blocks per workgroup step: whole coarsest blocks
  const int v = threadIdx.x % cv, row = threadIdx.x / cv;
  const int c = v * VEC;
  float s1[3][VEC], s2[3][VEC];
#pragma unroll
  for (int l = 0; l < 3; ++l)
#pragma unroll
    for (int j = 0; j < VEC; ++j) s1[l][j] = s2[l][j] = 0.f;
  const int Ht = a.H >> NLEV, Wt = a.W >> NLEV;
  const long long ntop = (long long)a.N * Ht * Wt;
  const long long n1 = ntop * GROUP;                                // level-1 blocks, nested order
  float* l1 = red;                      // [rows][cv][VEC] level-1 (then level-2) values of the step
  for (long long base = (long long)bid * rows; base < n1; base += (long long)nb * rows) {
    const long long q = base + row;
    const bool live = row < rows && q < n1;
    float d1[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) d1[j] = 0.f;
    int n = 0, y1 = 0, x1 = 0;
    if (live) {
      const long long top = q / GROUP;
      const int sub = (int)(q % GROUP);
      const int tx = (int)(top % Wt);
      const long long r = top / Wt;
      const int ty = (int)(r % Ht);
      n = (int)(r / Ht);
      // nested order: sub = (child of level 3) * 4 + (child of level 2); a child index is (dy << 1) | dx
      const int c3 = NLEV == 3 ? sub >> 2 : 0, c2 = NLEV >= 2 ? sub & 3 : 0;
      int by = ty, bx = tx;
      if (NLEV == 3) { by = by * 2 + (c3 >> 1); bx = bx * 2 + (c3 & 1); }
      if (NLEV >= 2) { by = by * 2 + (c2 >> 1); bx = bx * 2 + (c2 & 1); }
      y1 = by; x1 = bx;                          // level-1 coordinates
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const size_t off = ((size_t)((n * a.H + y1 * 2 + (p >> 1)) * a.W + x1 * 2 + (p & 1)) * a.C + c) * sizeof(T);
        float gv[VEC];
        v16_unpack<T>(*(const V16*)(a.g + off), gv);
        if (a.mask) {
          float mv[VEC];
          v16_unpack<T>(*(const V16*)(a.mask + off), mv);
#pragma unroll
          for (int j = 0; j < VEC; ++j) gv[j] = mv[j] > 0.f ? gv[j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) d1[j] += gv[j];
      }
      const size_t o1 = ((size_t)((n * (a.H >> 1) + y1) * (a.W >> 1) + x1) * a.C + c) * sizeof(T);
      const V16 packed = v16_pack<T>(d1);
      *(V16*)(a.dz[0] + o1) = packed;
      float dr[VEC], yv[VEC];
      v16_unpack<T>(packed, dr);
      v16_unpack<T>(*(const V16*)(a.y[0] + o1), yv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        s1[0][j] += dr[j];
        s2[0][j] = fmaf(dr[j], yv[j], s2[0][j]);
      }
    }
    if constexpr (NLEV >= 2) {
      __syncthreads();                           // (the previous step's values have been read)
#pragma unroll
      for (int j = 0; j < VEC; ++j) l1[(row * cv + v) * VEC + j] = d1[j];
      __syncthreads();
      float d2[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) d2[j] = 0.f;
      const bool head2 = live && (row & 3) == 0;
      if (head2) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int j = 0; j < VEC; ++j) d2[j] += l1[((row + k) * cv + v) * VEC + j];
        const int y2 = y1 >> 1, x2 = x1 >> 1;
        const size_t o2 = ((size_t)((n * (a.H >> 2) + y2) * (a.W >> 2) + x2) * a.C + c) * sizeof(T);
        const V16 packed = v16_pack<T>(d2);
        *(V16*)(a.dz[1] + o2) = packed;
        float dr[VEC], yv[VEC];
        v16_unpack<T>(packed, dr);
        v16_unpack<T>(*(const V16*)(a.y[1] + o2), yv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          s1[1][j] += dr[j];
          s2[1][j] = fmaf(dr[j], yv[j], s2[1][j]);
        }
      }
      if constexpr (NLEV == 3) {
        __syncthreads();
        if ((row & 3) == 0) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) l1[(row * cv + v) * VEC + j] = d2[j];
        }
        __syncthreads();
        if (live && (row & 15) == 0) {
          float d3[VEC];
#pragma unroll
          for (int j = 0; j < VEC; ++j) d3[j] = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < VEC; ++j) d3[j] += l1[((row + 4 * k) * cv + v) * VEC + j];
          const int y3 = y1 >> 2, x3 = x1 >> 2;
          const size_t o3 = ((size_t)((n * (a.H >> 3) + y3) * (a.W >> 3) + x3) * a.C + c) * sizeof(T);
          const V16 packed = v16_pack<T>(d3);
          *(V16*)(a.dz[2] + o3) = packed;
          float dr[VEC], yv[VEC];
          v16_unpack<T>(packed, dr);
          v16_unpack<T>(*(const V16*)(a.y[2] + o3), yv);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            s1[2][j] += dr[j];
            s2[2][j] = fmaf(dr[j], yv[j], s2[2][j]);
          }
        }
      }
    }
  }
  // per level: the workgroup's partial row, rows added in a fixed order (as bn_bwd_reduce_block)
#pragma unroll
  for (int l = 0; l < NLEV; ++l) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      red[(threadIdx.x * 2 + 0) * VEC + j] = s1[l][j];
      red[(threadIdx.x * 2 + 1) * VEC + j] = s2[l][j];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 2 * a.C; o += 256) {
      const int which = o / a.C, ch = o % a.C;
      const int vv = ch / VEC, jj = ch % VEC;
      float sacc = 0.f;
      for (int r = 0; r < rows; ++r) sacc += red[((r * cv + vv) * 2 + which) * VEC + jj];
      a.partials[l][((size_t)bid * 2 + which) * a.C + ch] = sacc;
    }
  }
}

__host__ __device__ inline void pool_args_from_op(const HrOp& op, PoolArgs& a) {
  // slots: i = {dtype, N, H, W, C, nlev}, p = {g, mask, y1, dz1, partials1, y2, dz2, partials2, y3, dz3, partials3}
  a.N = op.i[1]; a.H = op.i[2]; a.W = op.i[3]; a.C = op.i[4]; a.nlev = op.i[5];
  a.g = (const char*)op.p[0]; a.mask = (const char*)op.p[1];
  for (int l = 0; l < 3; ++l) {
    a.y[l] = (const char*)op.p[2 + 3 * l]; a.dz[l] = (char*)op.p[3 + 3 * l]; a.partials[l] = (float*)op.p[4 + 3 * l];
  }
}

template <typename T>
__device__ __forceinline__ void pool_reduce_dispatch(const PoolArgs& a, int bid, int nb, float* red) {
  if (a.nlev == 1) pool_reduce_block<T, 1>(a, bid, nb, red);
  else if (a.nlev == 2) pool_reduce_block<T, 2>(a, bid, nb, red);
  else pool_reduce_block<T, 3>(a, bid, nb, red);
}

template <typename T>
__global__ __launch_bounds__(256) void pool_reduce_kernel(PoolArgs a) {
  __shared__ float red[256 * 2 * TT<T>::VEC];
  pool_reduce_dispatch<T>(a, (int)blockIdx.x, (int)gridDim.x, red);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(GradArgs a) {
  __shared__ float red[256 * 2 * TT<T>::VEC];
  bn_bwd_reduce_block<T>(a, (int)blockIdx.x, (int)gridDim.x, red);
}

__device__ __forceinline__ void bn_bwd_finalize_block(
    const float* partials, int blocks, int C, float count, const float* gamma, const float* save_mean,
    const float* save_invstd, float* dgamma, float* dbeta, float* coef, int accumulate, int bid,
    double (*red)[FIN_LANES][32]) {
  const int cl = threadIdx.x & 31, tl = threadIdx.x >> 5;
  const int c = bid * 32 + cl;
  double s1 = 0.0, s2 = 0.0;
  partial_sums(partials, blocks, C, c, tl, cl, red, s1, s2);
  if (tl == 0 && c < C) {
    const double mu = save_mean[c], r = save_invstd[c], g = gamma[c];
    const double dg = r * (s2 - mu * s1);  // sum dz * xhat
    const double db = s1;
    if (accumulate) {
      dgamma[c] += (float)dg;
      dbeta[c] += (float)db;
    } else {
      dgamma[c] = (float)dg;
      dbeta[c] = (float)db;
    }
    // dy = A*dz + B*y + Cc
    const double A = g * r;
    const double B = -g * r * r * dg / (double)count;
    const double Cc = -g * r * db / (double)count + g * r * r * mu * dg / (double)count;
    coef[c] = (float)A;
    coef[C + c] = (float)B;
    coef[2 * C + c] = (float)Cc;
  }
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(
    const float* partials, int blocks, int C, float count, const float* gamma, const float* save_mean,
    const float* save_invstd, float* dgamma, float* dbeta, float* coef, int accumulate) {
  __shared__ double red[2][FIN_LANES][32];
  bn_bwd_finalize_block(partials, blocks, C, count, gamma, save_mean, save_invstd, dgamma, dbeta, coef, accumulate,
                        (int)blockIdx.x, red);
}

// ---- batched launches (HR_OP_EW_TABLE): several HR_OP_GRAD_TERM / HR_OP_BN_BWD_REDUCE / HR_OP_BN_BWD_FINALIZE jobs of
// one kind as ONE launch. The table holds the jobs as HrOp records in device memory (slots as for the single ops;
// i[16] = first block of the job, i[17] = its block count); a block finds its job by binary search. The module
// fuse layers leave a dozen such jobs on tensors of a few MB per HighResolutionModule: as launches of their own
// they are latency, not work. Same device code as the single launches: same values bit for bit.
__device__ __forceinline__ const HrOp& ew_table_find(const HrOp* tab, int n, int& local) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].i[16] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  local = (int)blockIdx.x - tab[lo].i[16];
  return tab[lo];
}

__device__ __forceinline__ GradArgs ew_grad_args(const HrOp& op, bool reduce) {
  GradArgs a;
  a.N = op.i[1]; a.H = op.i[2]; a.W = op.i[3]; a.C = op.i[4]; a.sh = op.i[5]; a.inner_relu = op.i[6];
  a.accumulate = reduce ? 0 : op.i[7];
  a.accumulate2 = reduce ? 0 : op.i[8];
  a.dst2 = reduce ? nullptr : (char*)op.p[7];
  a.partials = reduce ? (float*)op.p[0] : nullptr;
  a.dst = reduce ? (char*)op.p[6] : (char*)op.p[0];      // (reduce: optional store of the pooled, masked gradient)
  a.coef = reduce ? nullptr : (const float*)op.p[6];
  a.g = (const char*)op.p[1]; a.mask = (const char*)op.p[2]; a.y = (const char*)op.p[3];
  a.scale = (const float*)op.p[4]; a.shift = (const float*)op.p[5];
  return a;
}

template <typename T>
__global__ __launch_bounds__(256) void grad_term_table_kernel(const HrOp* tab, int n) {
  int local;
  const HrOp& op = ew_table_find(tab, n, local);
  const GradArgs a = ew_grad_args(op, false);
  grad_term_block<T>(a, local, op.i[17]);
}

template <typename T>
__global__ __launch_bounds__(256) void pool_reduce_table_kernel(const HrOp* tab, int n) {
  __shared__ float red[256 * 2 * TT<T>::VEC];
  int local;
  const HrOp& op = ew_table_find(tab, n, local);
  PoolArgs a;
  pool_args_from_op(op, a);
  pool_reduce_dispatch<T>(a, local, op.i[17], red);
}

// the forward sums of a HighResolutionModule's outputs (one per branch) as ONE launch: no fork / join around them
template <typename T, bool SUMS>
__global__ __launch_bounds__(256) void sum_terms_table_kernel(const HrOp* tab, int n) {
  __shared__ float bn[SUMS ? 4 : 1][2][SUMS ? SUM_MAXC : 1];
  int local;
  const HrOp& op = ew_table_find(tab, n, local);
  SumArgs a;
  sum_args_from_op(op, a);
  a.eps = __builtin_bit_cast(float, op.i[18]);      // (a table job: i[16] / i[17] hold its block range, eps moved to i[18])
  sum_terms_block<T, SUMS>(a, local, op.i[17], bn);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_table_kernel(const HrOp* tab, int n) {
  __shared__ float red[256 * 2 * TT<T>::VEC];
  int local;
  const HrOp& op = ew_table_find(tab, n, local);
  const GradArgs a = ew_grad_args(op, true);
  bn_bwd_reduce_block<T>(a, local, op.i[17], red);
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_table_kernel(const HrOp* tab, int n) {
  __shared__ double red[2][FIN_LANES][32];
  int local;
  const HrOp& op = ew_table_find(tab, n, local);
  bn_bwd_finalize_block((const float*)op.p[0], op.i[0], op.i[1], op.f[0], (const float*)op.p[1], (const float*)op.p[2],
                        (const float*)op.p[3], (float*)op.p[4], (float*)op.p[5], (float*)op.p[6], op.i[2], local, red);
}

// ---------------------------------------------------------------------------------------
// bilinear (align_corners=False) upsample + channel concat, and its transpose
// ---------------------------------------------------------------------------------------
struct CatArgs {
  char* cat;
  char* xs[4];
  int hs[4], ws[4], cs[4], coff[4];
  int nbr, N, H, W, Ctot, accumulate;
  int align;   // align_corners=True (pose_hrnet_softmax.py:499-501) instead of False (pose_hrnet.py:560-562)
};

template <typename T>
__global__ __launch_bounds__(256) void bilinear_cat_kernel(CatArgs a) {
  constexpr int VEC = TT<T>::VEC;
  const int cv = a.Ctot / VEC;
  const long long total = (long long)a.N * a.H * a.W * cv;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(idx % cv);
    long long pix = idx / cv;
    const int ox = (int)(pix % a.W);
    pix /= a.W;
    const int oy = (int)(pix % a.H);
    const int n = (int)(pix / a.H);
    const int c = v * VEC;
    int b = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (k < a.nbr && c >= a.coff[k]) b = k;
    const int cl = c - a.coff[b];
    const int hs = a.hs[b], ws = a.ws[b], cs = a.cs[b];
    const char* src = a.xs[b];
    float o[VEC];
    if (hs == a.H && ws == a.W) {
      v16_unpack<T>(*(const V16*)(src + ((size_t)((n * hs + oy) * ws + ox) * cs + cl) * sizeof(T)), o);
    } else {
      int y0, y1, x0, x1;
      float ly, lx;
      bilin_src(oy, hs, a.H, a.align, y0, y1, ly);
      bilin_src(ox, ws, a.W, a.align, x0, x1, lx);
      float f00[VEC], f01[VEC], f10[VEC], f11[VEC];
      const size_t base = (size_t)n * hs * ws;
      v16_unpack<T>(*(const V16*)(src + ((base + (size_t)y0 * ws + x0) * cs + cl) * sizeof(T)), f00);
      v16_unpack<T>(*(const V16*)(src + ((base + (size_t)y0 * ws + x1) * cs + cl) * sizeof(T)), f01);
      v16_unpack<T>(*(const V16*)(src + ((base + (size_t)y1 * ws + x0) * cs + cl) * sizeof(T)), f10);
      v16_unpack<T>(*(const V16*)(src + ((base + (size_t)y1 * ws + x1) * cs + cl) * sizeof(T)), f11);
      const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        o[j] = hy * (hx * f00[j] + lx * f01[j]) + ly * (hx * f10[j] + lx * f11[j]);
    }
    *(V16*)(a.cat + (size_t)idx * 16) = v16_pack<T>(o);
  }
}

// gather form of the transpose: one thread per (branch pixel, channel vector); loops over the
// destination pixels whose bilinear footprint touches it and re-evaluates the forward weights.
// The destination window of a source pixel is at most 2*ceil(out/in) + 1 wide; the x weights of the window are
// evaluated once per thread into registers (CATB_MAXW entries, statically indexed) and reused for every row, so
// the inner loop is one 16-byte load + VEC fmas per touched pixel with the row's loads issued together.
constexpr int CATB_MAXW = 20;

__device__ __forceinline__ void bilin_window(int s, int in_size, int out_size, int align, int& d0, int& d1) {
  // destinations whose source coordinate lies in (s-1, s+1), padded by one on both sides (weights that turn out
  // zero are skipped); the clamped ends of the axis extend to the border
  float lo, hi;
  if (align) {
    const float inv = in_size > 1 ? (float)(out_size - 1) / (float)(in_size - 1) : 0.f;
    lo = (float)(s - 1) * inv;
    hi = (float)(s + 1) * inv;
  } else {
    const float f = (float)out_size / (float)in_size;
    lo = ((float)s - 0.5f) * f - 0.5f;
    hi = ((float)s + 1.5f) * f - 0.5f;
  }
  d0 = (int)floorf(lo) - 1;
  d1 = (int)ceilf(hi) + 2;
  if (d0 < 0 || s == 0) d0 = 0;
  if (d1 > out_size || s == in_size - 1) d1 = out_size;
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_cat_bwd_kernel(CatArgs a, int b) {
  constexpr int VEC = TT<T>::VEC;
  const int hs = a.hs[b], ws = a.ws[b], cs = a.cs[b];
  const int cv = cs / VEC;
  const long long total = (long long)a.N * hs * ws * cv;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(idx % cv);
    long long pix = idx / cv;
    const int sx = (int)(pix % ws);
    pix /= ws;
    const int sy = (int)(pix % hs);
    const int n = (int)(pix / hs);
    const int c = v * VEC;
    float o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = 0.f;
    if (hs == a.H && ws == a.W) {
      v16_unpack<T>(*(const V16*)(a.cat + ((size_t)((n * a.H + sy) * a.W + sx) * a.Ctot + a.coff[b] + c) * sizeof(T)), o);
    } else {
      int dy0, dy1, dx0, dx1;
      bilin_window(sy, hs, a.H, a.align, dy0, dy1);
      bilin_window(sx, ws, a.W, a.align, dx0, dx1);
      for (int dxb = dx0; dxb < dx1; dxb += CATB_MAXW) {      // (one pass unless the factor exceeds 8)
        float wxs[CATB_MAXW];
#pragma unroll
        for (int j = 0; j < CATB_MAXW; ++j) {
          const int dx = dxb + j;
          int x0, x1;
          float lx;
          bilin_src(dx < a.W ? dx : a.W - 1, ws, a.W, a.align, x0, x1, lx);
          const float wx = (x0 == sx ? 1.f - lx : 0.f) + (x1 == sx ? lx : 0.f);
          wxs[j] = dx < dx1 ? wx : 0.f;
        }
        for (int dy = dy0; dy < dy1; ++dy) {
          int y0, y1;
          float ly;
          bilin_src(dy, hs, a.H, a.align, y0, y1, ly);
          const float wy = (y0 == sy ? 1.f - ly : 0.f) + (y1 == sy ? ly : 0.f);
          if (wy == 0.f) continue;
          const char* row = a.cat + ((size_t)((n * a.H + dy) * a.W + dxb) * a.Ctot + a.coff[b] + c) * sizeof(T);
#pragma unroll
          for (int j = 0; j < CATB_MAXW; ++j) {
            if (wxs[j] != 0.f) {
              float gv[VEC];
              v16_unpack<T>(*(const V16*)(row + (size_t)j * a.Ctot * sizeof(T)), gv);
              const float w = wy * wxs[j];
#pragma unroll
              for (int k = 0; k < VEC; ++k) o[k] = fmaf(w, gv[k], o[k]);
            }
          }
        }
      }
    }
    char* d = a.xs[b] + (size_t)idx * 16;
    if (a.accumulate) {
      float old[VEC];
      v16_unpack<T>(*(const V16*)d, old);
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] += old[j];
    }
    *(V16*)d = v16_pack<T>(o);
  }
}

// ---------------------------------------------------------------------------------------
// stem im2col, layout conversions, weight packing, bias grad, wgrad slab reduction
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void im2col_stem_kernel(const float* img, T* cols, int N, int C,
                                                          int H, int W, int Ho, int Wo, int Kpad) {
  const long long total = (long long)N * Ho * Wo * Kpad;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % Kpad);
    long long pix = idx / Kpad;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    float val = 0.f;
    if (k < 9 * C) {
      const int tap = k / C, c = k % C;
      const int iy = oy * 2 - 1 + tap / 3, ix = ox * 2 - 1 + tap % 3;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) val = img[((size_t)(n * C + c) * H + iy) * W + ix];
    }
    cols[idx] = (T)val;
  }
}

// the same with one thread per output PIXEL (C = 3, Kpad = 32: the network's stem): its 27 input values come from
// three image rows per plane that the neighbouring threads of the row share (L1), and the 32-value column is written
// as whole 16-byte vectors - the element-per-thread form above gathers 4 bytes per lane from 27 places and ran at
// 1.1 TB/s (107 us at batch 64)
template <typename T>
__global__ __launch_bounds__(256) void im2col_stem_pixel_kernel(const float* img, char* cols, int N, int H, int W, int Ho,
                                                                int Wo) {
  constexpr int VEC = TT<T>::VEC;
  const long long total = (long long)N * Ho * Wo;
  for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < total;
       pix += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(pix % Wo);
    const int oy = (int)((pix / Wo) % Ho);
    const int n = (int)(pix / ((long long)Wo * Ho));
    float v[32];
#pragma unroll
    for (int k = 27; k < 32; ++k) v[k] = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int iy = oy * 2 - 1 + r;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int ix = ox * 2 - 1 + s;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          v[(r * 3 + s) * 3 + c] = ok ? img[((size_t)(n * 3 + c) * H + iy) * W + ix] : 0.f;
      }
    }
    char* dst = cols + (size_t)pix * 32 * sizeof(T);
#pragma unroll
    for (int q = 0; q < 32 / VEC; ++q) *(V16*)(dst + q * 16) = v16_pack<T>(v + q * VEC);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* src, float* dst, int N, int HW,
                                                           int Cp, int C) {
  // 32 pixels x 32 channels tiles through LDS so both sides stay coalesced
  __shared__ float tile[32][33];
  const int pt = blockIdx.x * 32, ct = blockIdx.y * 32, n = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 8 rows per pass
  for (int r = ty; r < 32; r += 8) {
    const int p = pt + r, c = ct + tx;
    tile[r][tx] = (p < HW && c < Cp) ? to_f32(src[((size_t)n * HW + p) * Cp + c]) : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = ct + r, p = pt + tx;
    if (c < C && p < HW) dst[((size_t)n * C + c) * HW + p] = tile[tx][r];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* src, T* dst, int N, int HW,
                                                           int Cp, int C) {
  __shared__ float tile[32][33];
  const int pt = blockIdx.x * 32, ct = blockIdx.y * 32, n = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int c = ct + r, p = pt + tx;
    tile[r][tx] = (c < C && p < HW) ? src[((size_t)n * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int p = pt + r, c = ct + tx;
    if (p < HW && c < Cp) dst[((size_t)n * HW + p) * Cp + c] = (T)tile[tx][r];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* w, T* out, int Cout, int Cin,
                                                           int ks, int Cout_pad, int Cin_pad, int mode) {
  const int taps = ks * ks;
  const long long total = mode == 1 ? (long long)Cin_pad * taps * Cout_pad
                                    : (mode == 2 ? (long long)Cout_pad * Cin_pad
                                                 : (long long)Cout_pad * taps * Cin_pad);
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    float val = 0.f;
    if (mode == 0) {
      const int ci = (int)(idx % Cin_pad);
      const int t = (int)((idx / Cin_pad) % taps);
      const int co = (int)(idx / ((long long)Cin_pad * taps));
      if (co < Cout && ci < Cin) val = w[((size_t)co * Cin + ci) * taps + t];
    } else if (mode == 1) {
      // transposed conv: out[ci][taps-1-t][co] = w[co][ci][t]
      const int co = (int)(idx % Cout_pad);
      const int tf = (int)((idx / Cout_pad) % taps);
      const int ci = (int)(idx / ((long long)Cout_pad * taps));
      if (co < Cout && ci < Cin) val = w[((size_t)co * Cin + ci) * taps + (taps - 1 - tf)];
    } else {
      const int k = (int)(idx % Cin_pad);
      const int co = (int)(idx / Cin_pad);
      if (co < Cout && k < taps * Cin) {
        const int t = k / Cin, ci = k % Cin;
        val = w[((size_t)co * Cin + ci) * taps + t];
      }
    }
    out[idx] = (T)val;
  }
}

// all convolutions of the network packed by ONE launch: block b finds its entry by binary search.
// The master weights are OIHW f32; both packed layouts permute (ci, tap) or transpose (co <-> ci), so a direct
// gather reads 4 bytes per 36-byte (or row-sized) stride - 5.7x read amplification measured (653 MB fetched for
// 114 MB of weights, 130 us per launch, twice per step). Here a block stages a contiguous piece of the master
// through LDS: mode 0 one output-channel row (Cin*taps floats, contiguous), mode 1 four input channels of every
// output channel (4*taps contiguous floats per row), and writes its packed rows with unit stride.
// 19 KB of staging (eight workgroups per CU): mode 1 takes 16 ... 1 input channels per block, whichever fits
constexpr int PACK_LDS_FLOATS = 4864;
__host__ __device__ inline int pack_nci(int Cout, int taps) {
  for (int p = 16; p > 1; p >>= 1)
    if (Cout * (p * taps + 1) <= PACK_LDS_FLOATS) return p;
  return 1;
}
// mode 0: output-channel rows per block (narrow layers: a 32-channel 3x3 row is 1.1 KB - several per block)
__host__ __device__ inline int pack_rows(int Cin_pad, int taps) {
  const int r = 2048 / (Cin_pad * taps);
  return r < 1 ? 1 : (r > 8 ? 8 : r);
}

template <typename T>
__global__ __launch_bounds__(256) void pack_table_kernel(const HrPackEnt* tab, int n) {
  __shared__ float buf[PACK_LDS_FLOATS];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const HrPackEnt e = tab[lo];
  const int taps = e.ks * e.ks;
  const float* w = (const float*)e.w;
  T* out = (T*)e.out;
  const int u = (int)blockIdx.x - e.block0;
  if (e.mode == 0) {
    // out[co][t][ci_pad] <- w[co][ci][t]
    const int nr = pack_rows(e.Cin_pad, taps), co0 = u * nr, rowlen = e.Cin * taps, olen = taps * e.Cin_pad;
    const int rows = e.Cout_pad - co0 < nr ? e.Cout_pad - co0 : nr;
    for (int i = threadIdx.x; i < rows * rowlen; i += 256) {      // consecutive master rows are contiguous
      const int r = i / rowlen;
      buf[i] = co0 + r < e.Cout ? w[e.ld ? (size_t)(co0 + r) * e.ld + (i - r * rowlen) : (size_t)co0 * rowlen + i] : 0.f;
    }
    __syncthreads();
    T* o = out + (size_t)co0 * olen;
    for (int idx = threadIdx.x; idx < rows * olen; idx += 256) {
      const int r = idx / olen, q = idx - r * olen;
      const int t = q / e.Cin_pad, ci = q - t * e.Cin_pad;
      o[idx] = (T)((co0 + r < e.Cout && ci < e.Cin) ? buf[r * rowlen + ci * taps + t] : 0.f);
    }
  } else if (e.mode == 1) {
    // out[ci][taps flipped][co_pad] <- w[co][ci][t]
    const int per = pack_nci(e.Cout_pad, taps);     // (same rule as hrnet_pack_blocks)
    const int ci0 = u * per, span = per * taps, pitch = span + 1;
    const int nci = e.Cin_pad - ci0 < per ? e.Cin_pad - ci0 : per;
    for (int i = threadIdx.x; i < e.Cout * span; i += 256) {
      const int co = i / span, r = i - co * span;
      const int ci = ci0 + r / taps;
      buf[co * pitch + r] = ci < e.Cin ? w[((size_t)co * e.Cin + ci0) * taps + r] : 0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < nci * taps * e.Cout_pad; idx += 256) {
      const int co = idx % e.Cout_pad;
      const int tf = (idx / e.Cout_pad) % taps;
      const int cl = idx / (e.Cout_pad * taps);
      out[((size_t)(ci0 + cl) * taps + tf) * e.Cout_pad + co] =
          (T)(co < e.Cout ? buf[co * pitch + cl * taps + (taps - 1 - tf)] : 0.f);
    }
  } else {
    // stem: out[co][k_pad], k = tap * Cin + ci (a few KB)
    const long long total = (long long)e.Cout_pad * e.Cin_pad;
    const long long base = (long long)u * 1024;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long idx = base + r * 256 + threadIdx.x;
      if (idx >= total) break;
      const int k = (int)(idx % e.Cin_pad);
      const int co = (int)(idx / e.Cin_pad);
      float val = 0.f;
      if (co < e.Cout && k < taps * e.Cin) val = w[((size_t)co * e.Cin + (k % e.Cin)) * taps + k / e.Cin];
      out[idx] = (T)val;
    }
  }
}

// column sums of dy[pixels][Cp] (conv bias gradient), two-stage and deterministic
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const char* dy, float* partial, long long pixels,
                                                     int Cp) {
  constexpr int VEC = TT<T>::VEC;
  __shared__ float red[256 * VEC];
  const int cv = Cp / VEC;
  const int rows = 256 / cv;
  const int v = threadIdx.x % cv, row = threadIdx.x / cv;
  float s1[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s1[j] = 0.f;
  if (row < rows)
    for (long long p = (long long)blockIdx.x * rows + row; p < pixels; p += (long long)gridDim.x * rows) {
      float f[VEC];
      v16_unpack<T>(*(const V16*)(dy + ((size_t)p * Cp + v * VEC) * sizeof(T)), f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s1[j] += f[j];
    }
#pragma unroll
  for (int j = 0; j < VEC; ++j) red[threadIdx.x * VEC + j] = s1[j];
  __syncthreads();
  for (int ch = threadIdx.x; ch < Cp; ch += 256) {
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += red[(r * cv + ch / VEC) * VEC + ch % VEC];
    partial[(size_t)blockIdx.x * Cp + ch] = s;
  }
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* partial, float* dbias,
                                                              int blocks, int Cp, int C, int accumulate) {
  // block = 32 channels x 8 row lanes
  __shared__ double red[8][32];
  const int cl = threadIdx.x & 31, tl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s0 = 0.0, s1 = 0.0;
  if (c < C) {
    int b = tl;
    for (; b + 8 < blocks; b += 16) {
      const float x0 = partial[(size_t)b * Cp + c], x1 = partial[(size_t)(b + 8) * Cp + c];
      s0 += (double)x0; s1 += (double)x1;
    }
    if (b < blocks) s0 += (double)partial[(size_t)b * Cp + c];
  }
  red[tl][cl] = s0 + s1;
  __syncthreads();
  if (tl == 0 && c < C) {
    double s = 0.0;
    for (int k = 0; k < 8; ++k) s += red[k][cl];
    dbias[c] = accumulate ? dbias[c] + (float)s : (float)s;
  }
}

// one 64-element chunk of a weight gradient. idx enumerates the slab order (co, tap, ci): coalesced reads.
// Vector form (channel counts that are multiples of 4, every layer but the stem): 16 element lanes x float4 by 16
// slab lanes, each thread keeps up to 8 independent 16-byte loads in flight (the slabs are a pure HBM stream: 128
// slabs of a fused 64-channel layer are 18.9 MB); scalar form: 64 element lanes x 4 slab lanes. LDS adds the slab
// lanes in a fixed order (deterministic).
__device__ __forceinline__ void wgrad_reduce_chunk(const float* slabs, float* grad, int nsplit, int Cout, int Cin,
                                                   int ks, int Cout_real, int Cin_real, int kflat, int accumulate,
                                                   long long base, float (*red)[64], int ld = 0) {
  const int taps = ks * ks;
  // ld: floats between consecutive output-channel rows of `grad` (a column slice of a wider 1x1 weight); 0 = dense
  const size_t gpitch = ld ? (size_t)ld : (size_t)Cin_real * taps;
  const long long total = (long long)Cout_real * Cin_real * taps;
  const size_t slab_sz = (size_t)Cout * (kflat ? 1 : taps) * Cin;
  const bool vec = ((Cin_real | Cin) & 3) == 0;
  if (vec) {
    const int el = (threadIdx.x & 15) * 4, sl = threadIdx.x >> 4;
    const long long idx = base + el;
    const bool ok = idx < total;          // total is a multiple of 4 here: a float4 is inside or outside as a whole
    const int ci = (int)(idx % Cin_real);
    const int t = (int)((idx / Cin_real) % taps);
    const int co = (int)(idx / ((long long)Cin_real * taps));
    const size_t soff = kflat ? ((size_t)co * Cin + (size_t)t * Cin_real + ci) : (((size_t)co * taps + t) * Cin + ci);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
      const float* p = slabs + soff;
      int k = sl;
      for (; k + 112 < nsplit; k += 128) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *(const float4*)(p + (size_t)(k + 16 * j) * slab_sz);
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
      }
      for (; k < nsplit; k += 16) {
        const float4 v = *(const float4*)(p + (size_t)k * slab_sz);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    float (*red16)[64] = red;             // [16][64]
    *(float4*)&red16[sl][el] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
      const long long id1 = base + threadIdx.x;
      if (id1 < total) {
        float sacc = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sacc += red16[k][threadIdx.x];
        const int ci1 = (int)(id1 % Cin_real);
        const int t1 = (int)((id1 / Cin_real) % taps);
        const int co1 = (int)(id1 / ((long long)Cin_real * taps));
        float* g = grad + (size_t)co1 * gpitch + (size_t)ci1 * taps + t1;
        *g = accumulate ? *g + sacc : sacc;
      }
    }
    __syncthreads();
    return;
  }
  const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long long idx = base + el;
  const bool ok = idx < total;
  const int ci = (int)(idx % Cin_real);
  const int t = (int)((idx / Cin_real) % taps);
  const int co = (int)(idx / ((long long)Cin_real * taps));
  const size_t soff = kflat ? ((size_t)co * Cin + (size_t)t * Cin_real + ci) : (((size_t)co * taps + t) * Cin + ci);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (ok) {
    int k = sl;
    for (; k + 12 < nsplit; k += 16) {
      const float v0 = slabs[(size_t)k * slab_sz + soff], v1 = slabs[(size_t)(k + 4) * slab_sz + soff];
      const float v2 = slabs[(size_t)(k + 8) * slab_sz + soff], v3 = slabs[(size_t)(k + 12) * slab_sz + soff];
      s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; k < nsplit; k += 4) s0 += slabs[(size_t)k * slab_sz + soff];
  }
  red[sl][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && ok) {
    const float s = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
    float* g = grad + (size_t)co * gpitch + (size_t)ci * taps + t;
    *g = accumulate ? *g + s : s;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* slabs, float* grad, int nsplit,
                                                           int Cout, int Cin, int ks, int Cout_real,
                                                           int Cin_real, int kflat, int accumulate, int ld) {
  __shared__ __attribute__((aligned(16))) float red[16][64];
  const long long total = (long long)Cout_real * Cin_real * ks * ks;
  for (long long base = (long long)blockIdx.x * 64; base < total; base += (long long)gridDim.x * 64)
    wgrad_reduce_chunk(slabs, grad, nsplit, Cout, Cin, ks, Cout_real, Cin_real, kflat, accumulate, base, red, ld);
}

// every weight gradient of a backward segment in ONE launch: block b finds its layer by binary search
__global__ __launch_bounds__(256) void wgrad_reduce_table_kernel(const HrWredEnt* tab, int n) {
  __shared__ __attribute__((aligned(16))) float red[16][64];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const HrWredEnt e = tab[lo];
  wgrad_reduce_chunk(e.slabs, e.grad, e.nsplit, e.Cout_pad, e.Cin_pad, e.ks, e.Cout, e.Cin, e.kflat, e.accumulate,
                     (long long)((int)blockIdx.x - e.block0) * 64, red, e.ld);
}

// every BatchNorm of a forward pass in ONE launch: batch sums -> the arrays the backward pass reads (scale, shift,
// mean, invstd) and the running statistics (momentum update with the unbiased variance, num_batches_tracked).
// Block b finds its entry by binary search (block0 = first block of an entry; 256 channels per block).
__global__ __launch_bounds__(256) void bn_finalize_table_kernel(const HrBnEnt* tab, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const HrBnEnt e = tab[lo];
  const int c = ((int)blockIdx.x - e.block0) * 256 + threadIdx.x;
  if (c < e.C && !e.sums) {
    // eval mode: the affine of the running statistics (same arithmetic as bn_finalize_kernel, training = 0)
    const float invstd = 1.0f / sqrtf(e.running_var[c] + e.eps);
    const float sc = e.gamma[c] * invstd;
    e.scale[c] = sc;
    e.shift[c] = e.beta[c] - e.running_mean[c] * sc;
    return;
  }
  if (c < e.C) {
    float sc, sh, mean, invstd, var;
    hr_bn_from_sums(e.sums, e.C, c, 1.0f / e.count, e.eps, e.gamma[c], e.beta[c], sc, sh, mean, invstd, var);
    e.scale[c] = sc;
    e.shift[c] = sh;
    e.mean[c] = mean;
    e.invstd[c] = invstd;
    if (e.running_mean) {
      const float unbiased = e.count > 1.f ? var * (e.count / (e.count - 1.f)) : var;
      e.running_mean[c] = (1.f - e.momentum) * e.running_mean[c] + e.momentum * mean;
      e.running_var[c] = (1.f - e.momentum) * e.running_var[c] + e.momentum * unbiased;
    }
  }
  if (e.sums && e.num_batches_tracked && (int)blockIdx.x == e.block0 && threadIdx.x == 0) *e.num_batches_tracked += 1;
}

__global__ __launch_bounds__(256) void fill_zero_kernel(V16* p, long long n16, char* tail, int ntail) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
       i += (long long)gridDim.x * blockDim.x)
    p[i] = v16_zero();
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

struct LinArgs {
  const float* src[8];
  float coef[8];
  float* out;
  long long n;
  int k;
};
// out[i] = sum_j coef[j] * src[j][i] over f32 NCHW heat maps: the frame differences and the temporal aggregation
// of pose_hrnet_PoseAggr (reference lib/models/pose_hrnet_PoseAggr.py:612-640)
__global__ __launch_bounds__(256) void lincomb_f32_kernel(LinArgs a) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < a.k) s = fmaf(a.coef[j], a.src[j][i], s);
    a.out[i] = s;
  }
}

}  // namespace

extern "C" int hrnet_lincomb_f32(float* out, long long n, int k, const float* const* srcs, const float* coefs,
                                 hr_stream_t stream) {
  HR_REQUIRE(out && srcs && coefs && n > 0 && k >= 1 && k <= 8, "lincomb_f32: arguments");
  LinArgs a;
  for (int j = 0; j < 8; ++j) {
    a.src[j] = j < k ? srcs[j] : nullptr;
    a.coef[j] = j < k ? coefs[j] : 0.f;
    HR_REQUIRE(j >= k || a.src[j], "lincomb_f32: null source %d", j);
  }
  a.out = out; a.n = n; a.k = k;
  hipLaunchKernelGGL(lincomb_f32_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a);
  return hr_check_launch("lincomb_f32");
}

// =========================================================================================
// launchers
// =========================================================================================
int hr_launch_bn_finalize(const HrOp& op, hipStream_t s) {
  const int tiles = op.i[0], C = op.i[1], training = op.i[2];
  HR_REQUIRE(C > 0, "bn_finalize: C=%d", C);
  HR_REQUIRE(op.p[1] && op.p[2] && op.p[6] && op.p[7], "bn_finalize: null pointer");
  HR_REQUIRE(training ? (op.p[0] != nullptr && tiles > 0) : (op.p[3] && op.p[4]),
             "bn_finalize: missing statistics");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, s, (const float*)op.p[0],
                     tiles, C, op.f[0], (const float*)op.p[1], (const float*)op.p[2], (float*)op.p[3],
                     (float*)op.p[4], (long long*)op.p[5], op.f[1], op.f[2], training, (float*)op.p[6],
                     (float*)op.p[7], (float*)op.p[8], (float*)op.p[9]);
  return hr_check_launch("bn_finalize");
}

int hr_launch_sum_terms(const HrOp& op, hipStream_t s) {
  SumArgs a;
  const int dtype = op.i[0];
  sum_args_from_op(op, a);
  HR_REQUIRE(a.nterms >= 1 && a.nterms <= 4, "sum_terms: nterms=%d", a.nterms);
  HR_REQUIRE(a.C % (dtype == HR_F32 ? 4 : 8) == 0, "sum_terms: C=%d", a.C);
  HR_REQUIRE(a.out, "sum_terms: null out");
  HR_REQUIRE(!a.sums_mode || a.C <= SUM_MAXC, "sum_terms: C=%d too wide for batch-sum terms", a.C);
  for (int t = 0; t < 4; ++t) {
    if (t < a.nterms) {
      HR_REQUIRE(a.src[t], "sum_terms: null src %d", t);
      HR_REQUIRE((a.Ho % (1 << a.sh[t])) == 0 && (a.Wo % (1 << a.sh[t])) == 0, "sum_terms: upsample shift");
      HR_REQUIRE((a.scale[t] == nullptr) == (a.shift[t] == nullptr), "sum_terms: scale/shift pair");
    }
  }
  const long long total = (long long)a.N * a.Ho * a.Wo * (a.C / (dtype == HR_F32 ? 4 : 8));
  if (dtype == HR_F32)
    if (a.sums_mode) hipLaunchKernelGGL((sum_terms_kernel<float, true>), dim3(ew_grid(total)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((sum_terms_kernel<float, false>), dim3(ew_grid(total)), dim3(256), 0, s, a);
  else
    if (a.sums_mode) hipLaunchKernelGGL((sum_terms_kernel<bf16_t, true>), dim3(ew_grid(total)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((sum_terms_kernel<bf16_t, false>), dim3(ew_grid(total)), dim3(256), 0, s, a);
  return hr_check_launch("sum_terms");
}

static int fill_grad_args(const HrOp& op, GradArgs& a, bool reduce) {
  a.N = op.i[1]; a.H = op.i[2]; a.W = op.i[3]; a.C = op.i[4]; a.sh = op.i[5]; a.inner_relu = op.i[6];
  a.accumulate = reduce ? 0 : op.i[7];
  a.accumulate2 = reduce ? 0 : op.i[8];
  a.dst2 = reduce ? nullptr : (char*)op.p[7];
  if (reduce) {
    a.partials = (float*)op.p[0]; a.dst = (char*)op.p[6]; a.coef = nullptr;
  } else {
    a.dst = (char*)op.p[0]; a.partials = nullptr; a.coef = (const float*)op.p[6];
    HR_REQUIRE(!a.dst2 || (a.sh == 0 && !a.inner_relu), "grad_term: dst2 needs sh=0 and no inner ReLU");
  }
  a.g = (const char*)op.p[1]; a.mask = (const char*)op.p[2]; a.y = (const char*)op.p[3];
  a.scale = (const float*)op.p[4]; a.shift = (const float*)op.p[5];
  HR_REQUIRE(op.p[0] && a.g, "grad_term: null pointer");
  HR_REQUIRE(a.sh >= 0 && a.sh <= 4, "grad_term: shift %d", a.sh);
  HR_REQUIRE(!(a.inner_relu || a.coef || reduce) || a.y, "grad_term: y required");
  HR_REQUIRE(a.C % (op.i[0] == HR_F32 ? 4 : 8) == 0, "grad_term: C=%d", a.C);
  return 0;
}

int hr_launch_grad_term(const HrOp& op, hipStream_t s) {
  GradArgs a;
  if (int e = fill_grad_args(op, a, false)) return e;
  const long long total = (long long)a.N * a.H * a.W * (a.C / (op.i[0] == HR_F32 ? 4 : 8));
  const int cvs = a.C / (op.i[0] == HR_F32 ? 4 : 8);
  // large tensors (>= 4 grid-stride steps per thread): the variant that keeps the per-channel coefficients in registers
  if (a.sh == 0 && cvs <= 256 && total >= 4LL * 4096 * 256 && (long long)a.N * a.H * a.W < (1LL << 31)) {
    const int rows = 256 / cvs;
    const long long tot2 = ((long long)a.N * a.H * a.W + rows - 1) / rows * 256;
    if (op.i[0] == HR_F32)
      hipLaunchKernelGGL(grad_term_rows_kernel<float>, dim3(ew_grid(tot2)), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL(grad_term_rows_kernel<bf16_t>, dim3(ew_grid(tot2)), dim3(256), 0, s, a);
    return hr_check_launch("grad_term");
  }
  if (op.i[0] == HR_F32)
    hipLaunchKernelGGL(grad_term_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(grad_term_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, s, a);
  return hr_check_launch("grad_term");
}

extern "C" int hrnet_pack_blocks(int Cout_pad, int Cin_pad, int ks, int mode) {
  if (mode == 0) {
    const int nr = pack_rows(Cin_pad, ks * ks);
    return (Cout_pad + nr - 1) / nr;
  }
  if (mode == 1) {
    const int per = pack_nci(Cout_pad, ks * ks);
    return (Cin_pad + per - 1) / per;
  }
  return (int)(((long long)Cout_pad * Cin_pad + 1023) / 1024);
}

extern "C" int hrnet_reduce_blocks(int N, int H, int W, int C) {
  const long long npix = (long long)N * H * W;
  // 8 blocks per CU keep ~100 KB of loads in flight per CU (these passes are latency-bound otherwise)
  long long b = npix / 64;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  (void)C;
  return (int)b;
}

int hr_launch_bn_bwd_reduce(const HrOp& op, hipStream_t s) {
  GradArgs a;
  if (int e = fill_grad_args(op, a, true)) return e;
  const int vec = op.i[0] == HR_F32 ? 4 : 8;
  HR_REQUIRE(a.C / vec <= 256, "bn_bwd_reduce: C=%d too wide", a.C);
  const int blocks = hrnet_reduce_blocks(a.N, a.H, a.W, a.C);
  if (op.i[0] == HR_F32)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(blocks), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, a);
  return hr_check_launch("bn_bwd_reduce");
}

// HR_OP_EW_TABLE: p[0] = device table of HrOp jobs, i[0] = jobs, i[1] = total blocks, i[2] = kind of the jobs
// (HR_OP_GRAD_TERM / HR_OP_BN_BWD_REDUCE / HR_OP_BN_BWD_FINALIZE), i[3] = dtype
extern "C" int hrnet_ew_table_blocks(int kind, int dtype, int N, int H, int W, int C) {
  const int vec = dtype == HR_F32 ? 4 : 8;
  if (kind == HR_OP_GRAD_TERM) return (int)ew_grid((long long)N * H * W * (C / vec));
  if (kind == HR_OP_BN_BWD_REDUCE) return hrnet_reduce_blocks(N, H, W, C);
  if (kind == HR_OP_BN_BWD_FINALIZE) return (C + 31) / 32;
  if (kind == HR_OP_SUM_TERMS) return (int)ew_grid((long long)N * H * W * (C / vec));      // (output size)
  if (kind == HR_OP_POOL_REDUCE) return hrnet_reduce_blocks(N, H, W, C);      // (N, H, W: the FIRST level's size, H/2 x W/2)
  return 0;
}

int hr_launch_ew_table(const HrOp& op, hipStream_t s) {
  const int n = op.i[0], blocks = op.i[1], kind = op.i[2], dtype = op.i[3];
  HR_REQUIRE(op.p[0] && n >= 1 && blocks >= 1, "ew_table: args");
  HR_REQUIRE(dtype == HR_F32 || dtype == HR_BF16, "ew_table: dtype %d", dtype);
  const HrOp* tab = (const HrOp*)op.p[0];
  if (kind == HR_OP_GRAD_TERM) {
    if (dtype == HR_F32) hipLaunchKernelGGL(grad_term_table_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, tab, n);
    else hipLaunchKernelGGL(grad_term_table_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, tab, n);
  } else if (kind == HR_OP_BN_BWD_REDUCE) {
    if (dtype == HR_F32) hipLaunchKernelGGL(bn_bwd_reduce_table_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, tab, n);
    else hipLaunchKernelGGL(bn_bwd_reduce_table_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, tab, n);
  } else if (kind == HR_OP_BN_BWD_FINALIZE) {
    hipLaunchKernelGGL(bn_bwd_finalize_table_kernel, dim3((unsigned)blocks), dim3(1024), 0, s, tab, n);
  } else if (kind == HR_OP_POOL_REDUCE) {
    if (dtype == HR_F32) hipLaunchKernelGGL(pool_reduce_table_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, tab, n);
    else hipLaunchKernelGGL(pool_reduce_table_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, tab, n);
  } else if (kind == HR_OP_SUM_TERMS) {
    // op.i[4] != 0: some job's BatchNorm comes as batch sums (the instantiation with the coefficient table)
    const bool sums = op.i[4] != 0;
    if (dtype == HR_F32) {
      if (sums) hipLaunchKernelGGL((sum_terms_table_kernel<float, true>), dim3((unsigned)blocks), dim3(256), 0, s, tab, n);
      else hipLaunchKernelGGL((sum_terms_table_kernel<float, false>), dim3((unsigned)blocks), dim3(256), 0, s, tab, n);
    } else {
      if (sums) hipLaunchKernelGGL((sum_terms_table_kernel<bf16_t, true>), dim3((unsigned)blocks), dim3(256), 0, s, tab, n);
      else hipLaunchKernelGGL((sum_terms_table_kernel<bf16_t, false>), dim3((unsigned)blocks), dim3(256), 0, s, tab, n);
    }
  } else {
    HR_REQUIRE(false, "ew_table: kind %d cannot be batched", kind);
  }
  return hr_check_launch("ew_table");
}

// HR_OP_POOL_REDUCE as a launch of its own (slots: pool_args_from_op)
int hr_launch_pool_reduce(const HrOp& op, hipStream_t s) {
  PoolArgs a;
  pool_args_from_op(op, a);
  const int dtype = op.i[0], vec = dtype == HR_F32 ? 4 : 8;
  HR_REQUIRE(dtype == HR_F32 || dtype == HR_BF16, "pool_reduce: dtype");
  HR_REQUIRE(a.g && a.nlev >= 1 && a.nlev <= 3 && a.N > 0 && a.C > 0 && a.C % vec == 0 && a.C / vec <= 256, "pool_reduce: args");
  HR_REQUIRE(a.H % (1 << a.nlev) == 0 && a.W % (1 << a.nlev) == 0, "pool_reduce: %dx%d is not a multiple of 2^%d", a.H, a.W, a.nlev);
  for (int l = 0; l < a.nlev; ++l) HR_REQUIRE(a.y[l] && a.dz[l] && a.partials[l], "pool_reduce: level %d pointers", l + 1);
  HR_REQUIRE(256 / (a.C / vec) >= (a.nlev == 3 ? 16 : a.nlev == 2 ? 4 : 1), "pool_reduce: %d channels are too many for %d levels", a.C, a.nlev);
  const int blocks = hrnet_reduce_blocks(a.N, a.H >> 1, a.W >> 1, a.C);
  if (dtype == HR_F32) hipLaunchKernelGGL(pool_reduce_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(pool_reduce_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  return hr_check_launch("pool_reduce");
}

int hr_launch_bn_bwd_finalize(const HrOp& op, hipStream_t s) {
  const int blocks = op.i[0], C = op.i[1];
  HR_REQUIRE(op.p[0] && op.p[1] && op.p[2] && op.p[3] && op.p[4] && op.p[5] && op.p[6],
             "bn_bwd_finalize: null pointer");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, s, (const float*)op.p[0],
                     blocks, C, op.f[0], (const float*)op.p[1], (const float*)op.p[2],
                     (const float*)op.p[3], (float*)op.p[4], (float*)op.p[5], (float*)op.p[6], op.i[2]);
  return hr_check_launch("bn_bwd_finalize");
}


// Transpose of a bilinear upsampling over ALL channels (the head without its concat, gemm_pw.hip): g[n][h][w][c] =
// sum over the H x W pixels (Y, X) whose bilinear footprint touches (h, w) of weight * G[n][Y][X][c]. Separable and
// streamed: a thread owns one low-resolution column w and one 16-byte channel vector, walks the full-resolution
// rows Y of its workgroup's band in order, forms the row's x-contraction (the <= 2*scale+1 pixels whose x taps hit w;
// weights evaluated once per thread) and adds it into the two low-resolution rows Y touches, which live in
// registers and are stored as the walk leaves them. Every element of G is read once per band it belongs to (bands
// overlap by the footprint of their border rows), with unit-stride 16-byte loads across the channel vectors.
struct UpTArgs {
  const char* g;   // [N][H][W][C]
  char* out;       // [N][hs][ws][C]
  int N, H, W, C, hs, ws, align;
  int cvw;         // channel vectors per workgroup
  int chunks;      // channel chunks
  int rband;       // low-resolution rows per workgroup
  int bands;
};

template <typename T>
__global__ __launch_bounds__(256) void upsample_t_kernel(UpTArgs a) {
  constexpr int VEC = TT<T>::VEC;
  const int cvtot = a.C / VEC;
  int b = blockIdx.x;
  const int chunk = b % a.chunks; b /= a.chunks;
  const int band = b % a.bands;
  const int n = b / a.bands;
  const int cv = chunk * a.cvw + (int)threadIdx.x % a.cvw;
  const int w = (int)threadIdx.x / a.cvw;
  if (w >= a.ws || cv >= cvtot) return;
  const int h_lo = band * a.rband, h_hi = min(a.hs, h_lo + a.rband);
  // x window of column w and its weights (zero entries are skipped in the walk)
  int dx0, dx1;
  bilin_window(w, a.ws, a.W, a.align, dx0, dx1);
  float wxs[CATB_MAXW];
#pragma unroll
  for (int j = 0; j < CATB_MAXW; ++j) {
    const int dx = dx0 + j;
    int x0, x1;
    float lx;
    bilin_src(dx < a.W ? dx : a.W - 1, a.ws, a.W, a.align, x0, x1, lx);
    const float wx = (x0 == w ? 1.f - lx : 0.f) + (x1 == w ? lx : 0.f);
    wxs[j] = (dx < dx1 && dx < a.W) ? wx : 0.f;
  }
  // full-resolution rows whose taps reach [h_lo, h_hi)
  int ya, yb, t0, t1;
  bilin_window(h_lo, a.hs, a.H, a.align, ya, t1);
  bilin_window(h_hi - 1, a.hs, a.H, a.align, t0, yb);
  float acc0[VEC], acc1[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc0[k] = acc1[k] = 0.f;
  int cur = h_lo;      // acc0 belongs to low-resolution row cur, acc1 to cur + 1
  const char* gbase = a.g + ((size_t)n * a.H * a.W * a.C + (size_t)cv * VEC) * sizeof(T);
  char* obase = a.out + ((size_t)n * a.hs * a.ws * a.C + (size_t)w * a.C + (size_t)cv * VEC) * sizeof(T);
  auto flush = [&]() {
    if (cur >= h_lo && cur < h_hi) *(V16*)(obase + (size_t)cur * a.ws * a.C * sizeof(T)) = v16_pack<T>(acc0);
#pragma unroll
    for (int k = 0; k < VEC; ++k) { acc0[k] = acc1[k]; acc1[k] = 0.f; }
    ++cur;
  };
  for (int Y = ya; Y < yb; ++Y) {
    int y0, y1;
    float ly;
    bilin_src(Y, a.hs, a.H, a.align, y0, y1, ly);
    if (y1 < h_lo || y0 >= h_hi) continue;
    while (cur < y0) flush();            // (y0 never decreases along Y)
    float r[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) r[k] = 0.f;
    const char* row = gbase + ((size_t)Y * a.W + dx0) * a.C * sizeof(T);
#pragma unroll
    for (int j = 0; j < CATB_MAXW; ++j) {
      if (wxs[j] != 0.f) {
        float gv[VEC];
        v16_unpack<T>(*(const V16*)(row + (size_t)j * a.C * sizeof(T)), gv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) r[k] = fmaf(wxs[j], gv[k], r[k]);
      }
    }
    const float wa = y1 != y0 ? 1.f - ly : 1.f, wb = y1 != y0 ? ly : 0.f;
    if (y0 == cur) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) { acc0[k] = fmaf(wa, r[k], acc0[k]); acc1[k] = fmaf(wb, r[k], acc1[k]); }
    } else {             // y0 == cur - 1 cannot happen (flushed up to y0); y0 < h_lo: only the upper tap is ours
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc0[k] = fmaf(wb, r[k], acc0[k]);
    }
  }
  while (cur < h_hi) flush();
}

static int fill_cat_args(const HrOp& op, CatArgs& a, bool bwd) {
  a.nbr = op.i[1]; a.N = op.i[2]; a.H = op.i[3]; a.W = op.i[4];
  HR_REQUIRE(a.nbr >= 1 && a.nbr <= 4, "bilinear_cat: nbr=%d", a.nbr);
  a.cat = (char*)op.p[0];
  HR_REQUIRE(a.cat, "bilinear_cat: null cat");
  int off = 0;
  const int vec = op.i[0] == HR_F32 ? 4 : 8;
  for (int k = 0; k < 4; ++k) {
    a.hs[k] = op.i[5 + k]; a.ws[k] = op.i[9 + k]; a.cs[k] = op.i[13 + k];
    a.xs[k] = (char*)op.p[1 + k];
    a.coff[k] = off;
    if (k < a.nbr) {
      HR_REQUIRE(a.xs[k] && a.cs[k] % vec == 0 && a.hs[k] > 0 && a.ws[k] > 0, "bilinear_cat: branch %d", k);
      off += a.cs[k];
    }
  }
  a.Ctot = off;
  a.accumulate = bwd ? op.i[17] : 0;
  a.align = op.f[0] != 0.f;   // (all integer slots are taken)
  return 0;
}

// slots: i = {dtype, N, H, W, C, nout, align, h1, w1, h2, w2, h3, w3}, p = {G [N][H][W][C], out_k [N][h_k][w_k][C]}
int hr_upsample_t_tile(int dtype, const void* g, void* const* outs, const int* hs, const int* ws, int nout, int N, int H,
                       int W, int C, int align, hipStream_t s);

int hr_launch_upsample_t(const HrOp& op, hipStream_t s) {
  const int dtype = op.i[0], N = op.i[1], H = op.i[2], W = op.i[3], Cc = op.i[4], nout = op.i[5], align = op.i[6];
  const int vec = dtype == HR_F32 ? 4 : 8;
  HR_REQUIRE(dtype == HR_F32 || dtype == HR_BF16, "upsample_t: dtype");
  HR_REQUIRE(op.p[0] && N > 0 && H > 0 && W > 0 && Cc > 0 && Cc % vec == 0 && nout >= 1 && nout <= 3, "upsample_t: args");
  void* outs[3];
  int hs[3], ws[3];
  for (int k = 0; k < nout; ++k) {
    outs[k] = op.p[1 + k]; hs[k] = op.i[7 + 2 * k]; ws[k] = op.i[8 + 2 * k];
    HR_REQUIRE(outs[k] && hs[k] >= 1 && ws[k] >= 1 && hs[k] <= H && ws[k] <= W && ws[k] <= 256,
               "upsample_t: output %d: %dx%d from %dx%d", k, hs[k], ws[k], H, W);
  }
  // integer scales 2 / 4 / 8 (the head): all outputs from ONE pass over G (head_mix.hip)
  if (op.i[13] == 0 && hr_upsample_t_tile(dtype, op.p[0], outs, hs, ws, nout, N, H, W, Cc, align, s) == 0)
    return hr_check_launch("upsample_t");
  for (int k = 0; k < nout; ++k) {
    UpTArgs a;
    a.g = (const char*)op.p[0]; a.out = (char*)outs[k];
    a.N = N; a.H = H; a.W = W; a.C = Cc; a.hs = hs[k]; a.ws = ws[k]; a.align = align;
    HR_REQUIRE(2 * ((a.W + a.ws - 1) / a.ws) + 4 <= CATB_MAXW, "upsample_t: scale %d x %d too large", a.W, a.ws);
    const int cvtot = a.C / vec;
    a.cvw = 256 / a.ws;
    if (a.cvw > cvtot) a.cvw = cvtot;
    if (a.cvw < 1) a.cvw = 1;
    a.chunks = (cvtot + a.cvw - 1) / a.cvw;
    // ~16 full-resolution rows per band
    a.rband = (16 * a.hs + a.H - 1) / a.H;
    if (a.rband < 1) a.rband = 1;
    a.bands = (a.hs + a.rband - 1) / a.rband;
    const long long blocks = (long long)a.N * a.bands * a.chunks;
    HR_REQUIRE(blocks < (1ll << 31), "upsample_t: grid");
    if (dtype == HR_F32)
      hipLaunchKernelGGL(upsample_t_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL(upsample_t_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  }
  return hr_check_launch("upsample_t");
}

int hr_launch_bilinear_cat(const HrOp& op, hipStream_t s) {
  CatArgs a;
  if (int e = fill_cat_args(op, a, false)) return e;
  const long long total = (long long)a.N * a.H * a.W * (a.Ctot / (op.i[0] == HR_F32 ? 4 : 8));
  if (op.i[0] == HR_F32)
    hipLaunchKernelGGL(bilinear_cat_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(bilinear_cat_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, s, a);
  return hr_check_launch("bilinear_cat");
}

int hr_launch_bilinear_cat_bwd(const HrOp& op, hipStream_t s) {
  CatArgs a;
  if (int e = fill_cat_args(op, a, true)) return e;
  for (int b = 0; b < a.nbr; ++b) {
    const long long total = (long long)a.N * a.hs[b] * a.ws[b] * (a.cs[b] / (op.i[0] == HR_F32 ? 4 : 8));
    if (op.i[0] == HR_F32)
      hipLaunchKernelGGL(bilinear_cat_bwd_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, s, a, b);
    else
      hipLaunchKernelGGL(bilinear_cat_bwd_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, s, a, b);
  }
  return hr_check_launch("bilinear_cat_bwd");
}

int hr_launch_im2col_stem(const HrOp& op, hipStream_t s) {
  const int N = op.i[1], C = op.i[2], H = op.i[3], W = op.i[4], Ho = op.i[5], Wo = op.i[6], Kpad = op.i[7];
  HR_REQUIRE(op.p[0] && op.p[1], "im2col_stem: null pointer");
  HR_REQUIRE(Kpad >= 9 * C && Ho == (H + 1) / 2 && Wo == (W + 1) / 2, "im2col_stem: shape");
  const long long total = (long long)N * Ho * Wo * Kpad;
  if (C == 3 && Kpad == 32) {
    const long long px = (long long)N * Ho * Wo;
    if (op.i[0] == HR_F32)
      hipLaunchKernelGGL(im2col_stem_pixel_kernel<float>, dim3(ew_grid(px)), dim3(256), 0, s, (const float*)op.p[0],
                         (char*)op.p[1], N, H, W, Ho, Wo);
    else
      hipLaunchKernelGGL(im2col_stem_pixel_kernel<bf16_t>, dim3(ew_grid(px)), dim3(256), 0, s, (const float*)op.p[0],
                         (char*)op.p[1], N, H, W, Ho, Wo);
    return hr_check_launch("im2col_stem");
  }
  if (op.i[0] == HR_F32)
    hipLaunchKernelGGL(im2col_stem_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, s,
                       (const float*)op.p[0], (float*)op.p[1], N, C, H, W, Ho, Wo, Kpad);
  else
    hipLaunchKernelGGL(im2col_stem_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, s,
                       (const float*)op.p[0], (bf16_t*)op.p[1], N, C, H, W, Ho, Wo, Kpad);
  return hr_check_launch("im2col_stem");
}

int hr_launch_nhwc_to_nchw(const HrOp& op, hipStream_t s) {
  const int N = op.i[1], HW = op.i[2] * op.i[3], Cp = op.i[4], C = op.i[5];
  HR_REQUIRE(op.p[0] && op.p[1] && C <= Cp, "nhwc_to_nchw: args");
  dim3 grid((HW + 31) / 32, (C + 31) / 32, N);
  if (op.i[0] == HR_F32)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, s, (const float*)op.p[0],
                       (float*)op.p[1], N, HW, Cp, C);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)op.p[0],
                       (float*)op.p[1], N, HW, Cp, C);
  return hr_check_launch("nhwc_to_nchw");
}

int hr_launch_nchw_to_nhwc(const HrOp& op, hipStream_t s) {
  const int N = op.i[1], HW = op.i[2] * op.i[3], Cp = op.i[4], C = op.i[5];
  HR_REQUIRE(op.p[0] && op.p[1] && C <= Cp, "nchw_to_nhwc: args");
  dim3 grid((HW + 31) / 32, (Cp + 31) / 32, N);
  if (op.i[0] == HR_F32)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, s, (const float*)op.p[0],
                       (float*)op.p[1], N, HW, Cp, C);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid, dim3(256), 0, s, (const float*)op.p[0],
                       (bf16_t*)op.p[1], N, HW, Cp, C);
  return hr_check_launch("nchw_to_nhwc");
}

int hr_launch_pack_weights(const HrOp& op, hipStream_t s) {
  const int Cout = op.i[1], Cin = op.i[2], ks = op.i[3], Cout_pad = op.i[4], Cin_pad = op.i[5], mode = op.i[6];
  HR_REQUIRE(op.p[0] && op.p[1] && mode >= 0 && mode <= 2, "pack_weights: args");
  HR_REQUIRE(Cout_pad >= Cout && (mode == 2 ? Cin_pad >= Cin * ks * ks : Cin_pad >= Cin), "pack_weights: pads");
  const long long total = mode == 2 ? (long long)Cout_pad * Cin_pad : (long long)Cout_pad * ks * ks * Cin_pad;
  if (op.i[0] == HR_F32)
    hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, s,
                       (const float*)op.p[0], (float*)op.p[1], Cout, Cin, ks, Cout_pad, Cin_pad, mode);
  else
    hipLaunchKernelGGL(pack_weights_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, s,
                       (const float*)op.p[0], (bf16_t*)op.p[1], Cout, Cin, ks, Cout_pad, Cin_pad, mode);
  return hr_check_launch("pack_weights");
}

int hr_launch_pack_table(const HrOp& op, hipStream_t s) {
  const int n = op.i[1], blocks = op.i[2];
  HR_REQUIRE(op.p[0] && n >= 1 && blocks >= 1, "pack_weights_table: args");
  if (op.i[0] == HR_F32)
    hipLaunchKernelGGL(pack_table_kernel<float>, dim3(blocks), dim3(256), 0, s, (const HrPackEnt*)op.p[0], n);
  else
    hipLaunchKernelGGL(pack_table_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const HrPackEnt*)op.p[0], n);
  return hr_check_launch("pack_weights_table");
}

int hr_launch_bias_grad(const HrOp& op, hipStream_t s) {
  const int pixels = op.i[1], Cp = op.i[2], C = op.i[3];
  const int vec = op.i[0] == HR_F32 ? 4 : 8;
  HR_REQUIRE(op.p[0] && op.p[1] && op.p[2] && C <= Cp && Cp % vec == 0 && Cp / vec <= 256, "bias_grad: args");
  const int blocks = hrnet_reduce_blocks(1, 1, pixels, Cp);
  if (op.i[0] == HR_F32)
    hipLaunchKernelGGL(colsum_kernel<float>, dim3(blocks), dim3(256), 0, s, (const char*)op.p[0],
                       (float*)op.p[2], (long long)pixels, Cp);
  else
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const char*)op.p[0],
                       (float*)op.p[2], (long long)pixels, Cp);
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, s, (const float*)op.p[2],
                     (float*)op.p[1], blocks, Cp, C, op.i[4]);
  return hr_check_launch("bias_grad");
}

int hr_launch_wgrad_reduce(const HrOp& op, hipStream_t s) {
  const int nsplit = op.i[0], Cout = op.i[1], Cin = op.i[2], ks = op.i[3], Cout_real = op.i[4],
            Cin_real = op.i[5], kflat = op.i[6];
  HR_REQUIRE(op.p[0] && op.p[1] && nsplit >= 1, "wgrad_reduce: args");
  HR_REQUIRE(Cout_real <= Cout && (kflat ? Cin_real * ks * ks <= Cin : Cin_real <= Cin), "wgrad_reduce: extents");
  const long long total = (long long)Cout_real * Cin_real * ks * ks;
  long long rgrid = (total + 63) / 64;
  if (rgrid > 4096) rgrid = 4096;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)rgrid), dim3(256), 0, s, (const float*)op.p[0],
                     (float*)op.p[1], nsplit, Cout, Cin, ks, Cout_real, Cin_real, kflat, op.i[7], op.i[8]);
  return hr_check_launch("wgrad_reduce");
}

int hr_launch_wgrad_reduce_table(const HrOp& op, hipStream_t s) {
  const int n = op.i[0], blocks = op.i[1];
  HR_REQUIRE(op.p[0] && n >= 1 && blocks >= 1, "wgrad_reduce_table: args");
  hipLaunchKernelGGL(wgrad_reduce_table_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const HrWredEnt*)op.p[0], n);
  return hr_check_launch("wgrad_reduce_table");
}

int hr_launch_bn_finalize_table(const HrOp& op, hipStream_t s) {
  const int n = op.i[0], blocks = op.i[1];
  HR_REQUIRE(op.p[0] && n >= 1 && blocks >= 1, "bn_finalize_table: args");
  hipLaunchKernelGGL(bn_finalize_table_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const HrBnEnt*)op.p[0], n);
  return hr_check_launch("bn_finalize_table");
}

int hr_launch_fill(const HrOp& op, hipStream_t s) {
  const long long bytes = ((long long)(uint32_t)op.i[1] << 32) | (uint32_t)op.i[0];
  HR_REQUIRE(op.p[0] && bytes >= 0 && ((uintptr_t)op.p[0] % 16) == 0, "fill_zero: args");
  const long long n16 = bytes / 16;
  hipLaunchKernelGGL(fill_zero_kernel, dim3(ew_grid(n16 > 0 ? n16 : 1)), dim3(256), 0, s, (V16*)op.p[0],
                     n16, (char*)op.p[0] + n16 * 16, (int)(bytes % 16));
  return hr_check_launch("fill_zero");
}
