// C-ABI plumbing: per-thread error text, launch checks, the program runner, and the thin
// extern "C" entry points that fill an HrOp and call the launcher of the same op kind.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "common.h"

namespace {
thread_local char g_err[512] = "";
}

void hr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hr_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    hr_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return HR_E_LAUNCH;
  }
  return HR_OK;
}

extern "C" const char* hrnet_last_error_string(void) { return g_err; }
// 2: HR_OP_CONV i[17] (route), HR_OP_WGRAD i[12..15], HrPackEnt.ld, the head / table op kinds of round 3
extern "C" int hrnet_abi_version(void) { return 2; }

static int run_one(const HrOp& op, hipStream_t s, int k) {
  int e;
  switch (op.kind) {
    case HR_OP_CONV: e = hr_launch_conv(op, s); break;
    case HR_OP_WGRAD: e = hr_launch_wgrad(op, s); break;
    case HR_OP_WGRAD_REDUCE: e = hr_launch_wgrad_reduce(op, s); break;
    case HR_OP_BN_FINALIZE: e = hr_launch_bn_finalize(op, s); break;
    case HR_OP_SUM_TERMS: e = hr_launch_sum_terms(op, s); break;
    case HR_OP_GRAD_TERM: e = hr_launch_grad_term(op, s); break;
    case HR_OP_BN_BWD_REDUCE: e = hr_launch_bn_bwd_reduce(op, s); break;
    case HR_OP_BN_BWD_FINALIZE: e = hr_launch_bn_bwd_finalize(op, s); break;
    case HR_OP_BILINEAR_CAT: e = hr_launch_bilinear_cat(op, s); break;
    case HR_OP_BILINEAR_CAT_BWD: e = hr_launch_bilinear_cat_bwd(op, s); break;
    case HR_OP_IM2COL_STEM: e = hr_launch_im2col_stem(op, s); break;
    case HR_OP_NHWC_TO_NCHW: e = hr_launch_nhwc_to_nchw(op, s); break;
    case HR_OP_NCHW_TO_NHWC: e = hr_launch_nchw_to_nhwc(op, s); break;
    case HR_OP_PACK_WEIGHTS: e = hr_launch_pack_weights(op, s); break;
    case HR_OP_BIAS_GRAD: e = hr_launch_bias_grad(op, s); break;
    case HR_OP_FILL: e = hr_launch_fill(op, s); break;
    case HR_OP_PACK_TABLE: e = hr_launch_pack_table(op, s); break;
    case HR_OP_WGRAD_REDUCE_TABLE: e = hr_launch_wgrad_reduce_table(op, s); break;
    case HR_OP_BWD_FUSED: e = hr_launch_bwd_fused(op, s); break;
    case HR_OP_BWD_PW: e = hr_launch_bwd_pw(op, s); break;
    case HR_OP_CONV_SUM: e = hr_launch_conv_sum(op, s); break;
    case HR_OP_BN_FINALIZE_TABLE: e = hr_launch_bn_finalize_table(op, s); break;
    case HR_OP_EW_TABLE: e = hr_launch_ew_table(op, s); break;
    case HR_OP_HEAD_MIX: e = hr_launch_head_mix(op, s); break;
    case HR_OP_UPSAMPLE_T: e = hr_launch_upsample_t(op, s); break;
    case HR_OP_HEAD_BWD: e = hr_launch_head_bwd(op, s); break;
    case HR_OP_POOL_REDUCE: e = hr_launch_pool_reduce(op, s); break;
    case HR_OP_EVENT_RECORD:
      e = hipEventRecord((hipEvent_t)op.p[0], s) == hipSuccess ? HR_OK : HR_E_LAUNCH;
      if (e) hr_set_error("event record failed");
      break;
    case HR_OP_STREAM_WAIT:
      e = hipStreamWaitEvent(s, (hipEvent_t)op.p[0], 0) == hipSuccess ? HR_OK : HR_E_LAUNCH;
      if (e) hr_set_error("stream wait failed");
      break;
    default:
      hr_set_error("program_run: unknown op kind %d at index %d", op.kind, k);
      return HR_E_BADOP;
  }
  if (e != HR_OK) {
    char tmp[400];
    strncpy(tmp, g_err, sizeof(tmp) - 1);
    tmp[sizeof(tmp) - 1] = 0;
    hr_set_error("op %d (kind %d): %s", k, op.kind, tmp);
  }
  return e;
}

extern "C" int hrnet_program_run(const HrOp* ops, int n, hr_stream_t stream) {
  for (int k = 0; k < n; ++k) {
    // single-stream execution is already ordered: lane markers are no-ops
    if (ops[k].kind == HR_OP_EVENT_RECORD || ops[k].kind == HR_OP_STREAM_WAIT) continue;
    const int e = run_one(ops[k], (hipStream_t)stream, k);
    if (e != HR_OK) return e;
  }
  return HR_OK;
}

extern "C" int hrnet_program_run_streams(const HrOp* ops, int n, const hr_stream_t* streams, int nstreams) {
  HR_REQUIRE(streams && nstreams >= 1, "program_run_streams: no streams");
  for (int k = 0; k < n; ++k) {
    const int lane = ops[k].i[HR_LANE_SLOT];
    HR_REQUIRE(lane >= 0 && lane < nstreams, "program_run_streams: op %d lane %d out of range", k, lane);
    const int e = run_one(ops[k], (hipStream_t)streams[lane], k);
    if (e != HR_OK) return e;
  }
  return HR_OK;
}

// The same with a timing event recorded behind every op on its lane: end_ms[k] = when op k was done, in ms since
// the first op's lane reached the start of the call (event ops: when the lane passed them). Synchronises the
// streams before it returns - a measurement entry point (bench.py / scratch timelines), not for the training loop.
extern "C" int hrnet_program_run_streams_timed(const HrOp* ops, int n, const hr_stream_t* streams, int nstreams,
                                               float* end_ms) {
  HR_REQUIRE(streams && nstreams >= 1 && end_ms && n >= 1, "program_run_streams_timed: arguments");
  std::vector<hipEvent_t> ev((size_t)n + 1, nullptr);
  for (auto& e : ev)
    if (hipEventCreate(&e) != hipSuccess) { hr_set_error("program_run_streams_timed: event create"); return HR_E_LAUNCH; }
  int rc = HR_OK;
  const int lane0 = ops[0].i[HR_LANE_SLOT];
  HR_REQUIRE(lane0 >= 0 && lane0 < nstreams, "program_run_streams_timed: lane");
  hipEventRecord(ev[n], (hipStream_t)streams[lane0]);
  int done = 0;
  for (int k = 0; k < n && rc == HR_OK; ++k, ++done) {
    const int lane = ops[k].i[HR_LANE_SLOT];
    if (lane < 0 || lane >= nstreams) { hr_set_error("program_run_streams_timed: op %d lane %d", k, lane); rc = HR_E_BADARG; break; }
    rc = run_one(ops[k], (hipStream_t)streams[lane], k);
    hipEventRecord(ev[k], (hipStream_t)streams[lane]);
  }
  for (int l = 0; l < nstreams; ++l) hipStreamSynchronize((hipStream_t)streams[l]);
  for (int k = 0; k < done; ++k) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, ev[n], ev[k]);
    end_ms[k] = ms;
  }
  for (auto& e : ev) hipEventDestroy(e);
  return rc;
}

extern "C" void* hrnet_event_create(void) {
  hipEvent_t ev = nullptr;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    hr_set_error("hipEventCreate failed");
    return nullptr;
  }
  return (void*)ev;
}

extern "C" int hrnet_event_destroy(void* event) {
  return hipEventDestroy((hipEvent_t)event) == hipSuccess ? HR_OK : HR_E_LAUNCH;
}

#define OP_BEGIN(KIND) \
  HrOp op;             \
  memset(&op, 0, sizeof(op)); \
  op.kind = KIND

extern "C" int hrnet_wgrad_reduce(const float* slabs, float* grad_oihw, int nsplit, int Cout, int Cin,
                                  int ks, int Cout_real, int Cin_real, int kflat, int accumulate,
                                  hr_stream_t stream) {
  OP_BEGIN(HR_OP_WGRAD_REDUCE);
  const int iv[8] = {nsplit, Cout, Cin, ks, Cout_real, Cin_real, kflat, accumulate};
  memcpy(op.i, iv, sizeof(iv));
  op.p[0] = (void*)slabs; op.p[1] = grad_oihw;
  return hr_launch_wgrad_reduce(op, (hipStream_t)stream);
}

extern "C" int hrnet_pack_weights(int dtype, const float* w_oihw, void* packed, int Cout, int Cin, int ks,
                                  int Cout_pad, int Cin_pad, int mode, hr_stream_t stream) {
  OP_BEGIN(HR_OP_PACK_WEIGHTS);
  const int iv[7] = {dtype, Cout, Cin, ks, Cout_pad, Cin_pad, mode};
  memcpy(op.i, iv, sizeof(iv));
  op.p[0] = (void*)w_oihw; op.p[1] = packed;
  return hr_launch_pack_weights(op, (hipStream_t)stream);
}

extern "C" int hrnet_pack_weights_table(int dtype, const HrPackEnt* table, int n, int total_blocks,
                                        hr_stream_t stream) {
  OP_BEGIN(HR_OP_PACK_TABLE);
  op.i[0] = dtype; op.i[1] = n; op.i[2] = total_blocks;
  op.p[0] = (void*)table;
  return hr_launch_pack_table(op, (hipStream_t)stream);
}

extern "C" int hrnet_wgrad_reduce_table(const HrWredEnt* table, int n, int total_blocks, hr_stream_t stream) {
  OP_BEGIN(HR_OP_WGRAD_REDUCE_TABLE);
  op.i[0] = n; op.i[1] = total_blocks;
  op.p[0] = (void*)table;
  return hr_launch_wgrad_reduce_table(op, (hipStream_t)stream);
}

extern "C" int hrnet_bn_finalize(const float* stats, int tiles, int C, float count, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var,
                                 int64_t* num_batches_tracked, float momentum, float eps, int training,
                                 float* scale, float* shift, float* save_mean, float* save_invstd,
                                 hr_stream_t stream) {
  OP_BEGIN(HR_OP_BN_FINALIZE);
  op.i[0] = tiles; op.i[1] = C; op.i[2] = training;
  op.f[0] = count; op.f[1] = momentum; op.f[2] = eps;
  op.p[0] = (void*)stats; op.p[1] = (void*)gamma; op.p[2] = (void*)beta; op.p[3] = running_mean;
  op.p[4] = running_var; op.p[5] = num_batches_tracked; op.p[6] = scale; op.p[7] = shift;
  op.p[8] = save_mean; op.p[9] = save_invstd;
  return hr_launch_bn_finalize(op, (hipStream_t)stream);
}

extern "C" int hrnet_sum_terms(int dtype, void* out, int N, int Ho, int Wo, int C, int nterms,
                               const void* const* src, const float* const* scale,
                               const float* const* shift, const int* shifts, const int* relus,
                               int relu_out, hr_stream_t stream) {
  OP_BEGIN(HR_OP_SUM_TERMS);
  HR_REQUIRE(nterms >= 1 && nterms <= 4 && src && shifts && relus, "sum_terms: args");
  const int iv[7] = {dtype, N, Ho, Wo, C, nterms, relu_out};
  memcpy(op.i, iv, sizeof(iv));
  op.p[0] = out;
  for (int t = 0; t < nterms; ++t) {
    op.i[7 + t] = shifts[t];
    op.i[11 + t] = relus[t];
    op.p[1 + t] = (void*)src[t];
    op.p[5 + t] = scale ? (void*)scale[t] : nullptr;
    op.p[9 + t] = shift ? (void*)shift[t] : nullptr;
  }
  return hr_launch_sum_terms(op, (hipStream_t)stream);
}

extern "C" int hrnet_sum_terms_bnref(int dtype, void* out, int N, int Ho, int Wo, int C, int nterms,
                                     const void* const* src, const float* const* scale, const float* const* shift,
                                     const int* shifts, const int* relus, int relu_out, int sums_mode,
                                     const float* inv_counts, float eps, hr_stream_t stream) {
  OP_BEGIN(HR_OP_SUM_TERMS);
  HR_REQUIRE(nterms >= 1 && nterms <= 4 && src && shifts && relus && scale && shift && inv_counts, "sum_terms: args");
  const int iv[7] = {dtype, N, Ho, Wo, C, nterms, relu_out};
  memcpy(op.i, iv, sizeof(iv));
  op.p[0] = out;
  for (int t = 0; t < nterms; ++t) {
    op.i[7 + t] = shifts[t];
    op.i[11 + t] = relus[t];
    op.p[1 + t] = (void*)src[t];
    op.p[5 + t] = (void*)scale[t];
    op.p[9 + t] = (void*)shift[t];
    op.f[t] = inv_counts[t];
  }
  op.i[15] = sums_mode;
  memcpy(&op.i[16], &eps, sizeof(float));
  return hr_launch_sum_terms(op, (hipStream_t)stream);
}

extern "C" int hrnet_bn_finalize_table(const HrBnEnt* table, int n, int total_blocks, hr_stream_t stream) {
  OP_BEGIN(HR_OP_BN_FINALIZE_TABLE);
  op.i[0] = n; op.i[1] = total_blocks;
  op.p[0] = (void*)table;
  return hr_launch_bn_finalize_table(op, (hipStream_t)stream);
}

extern "C" int hrnet_grad_term(int dtype, void* dst, const void* g, const void* mask_out, const void* y,
                               const float* scale, const float* shift, const float* coef, int N, int H,
                               int W, int C, int sh, int inner_relu, int accumulate, hr_stream_t stream) {
  OP_BEGIN(HR_OP_GRAD_TERM);
  const int iv[8] = {dtype, N, H, W, C, sh, inner_relu, accumulate};
  memcpy(op.i, iv, sizeof(iv));
  op.p[0] = dst; op.p[1] = (void*)g; op.p[2] = (void*)mask_out; op.p[3] = (void*)y;
  op.p[4] = (void*)scale; op.p[5] = (void*)shift; op.p[6] = (void*)coef;
  return hr_launch_grad_term(op, (hipStream_t)stream);
}

extern "C" int hrnet_grad_term2(int dtype, void* dst, void* dst2, const void* g, const void* mask_out,
                                const void* y, const float* scale, const float* shift, const float* coef, int N,
                                int H, int W, int C, int accumulate, int accumulate2, hr_stream_t stream) {
  OP_BEGIN(HR_OP_GRAD_TERM);
  const int iv[9] = {dtype, N, H, W, C, 0, 0, accumulate, accumulate2};
  memcpy(op.i, iv, sizeof(iv));
  op.p[0] = dst; op.p[1] = (void*)g; op.p[2] = (void*)mask_out; op.p[3] = (void*)y;
  op.p[4] = (void*)scale; op.p[5] = (void*)shift; op.p[6] = (void*)coef; op.p[7] = dst2;
  return hr_launch_grad_term(op, (hipStream_t)stream);
}

extern "C" int hrnet_bn_bwd_reduce(int dtype, float* partials, const void* g, const void* mask_out,
                                   const void* y, const float* scale, const float* shift, int N, int H,
                                   int W, int C, int sh, int inner_relu, hr_stream_t stream) {
  OP_BEGIN(HR_OP_BN_BWD_REDUCE);
  const int iv[7] = {dtype, N, H, W, C, sh, inner_relu};
  memcpy(op.i, iv, sizeof(iv));
  op.p[0] = partials; op.p[1] = (void*)g; op.p[2] = (void*)mask_out; op.p[3] = (void*)y;
  op.p[4] = (void*)scale; op.p[5] = (void*)shift;
  return hr_launch_bn_bwd_reduce(op, (hipStream_t)stream);
}

extern "C" int hrnet_bn_bwd_finalize(const float* partials, int blocks, int C, float count,
                                     const float* gamma, const float* save_mean, const float* save_invstd,
                                     float* dgamma, float* dbeta, float* coef, int accumulate,
                                     hr_stream_t stream) {
  OP_BEGIN(HR_OP_BN_BWD_FINALIZE);
  op.i[0] = blocks; op.i[1] = C; op.i[2] = accumulate;
  op.f[0] = count;
  op.p[0] = (void*)partials; op.p[1] = (void*)gamma; op.p[2] = (void*)save_mean;
  op.p[3] = (void*)save_invstd; op.p[4] = dgamma; op.p[5] = dbeta; op.p[6] = coef;
  return hr_launch_bn_bwd_finalize(op, (hipStream_t)stream);
}

static void fill_cat(HrOp& op, int dtype, const int* hs, const int* ws, const int* cs, int nbr, int N,
                     int H, int W) {
  op.i[0] = dtype; op.i[1] = nbr; op.i[2] = N; op.i[3] = H; op.i[4] = W;
  for (int k = 0; k < nbr && k < 4; ++k) {
    op.i[5 + k] = hs[k]; op.i[9 + k] = ws[k]; op.i[13 + k] = cs[k];
  }
}

extern "C" int hrnet_bilinear_cat(int dtype, void* cat, const void* const* xs, const int* hs,
                                  const int* ws, const int* cs, int nbr, int N, int H, int W,
                                  int align_corners, hr_stream_t stream) {
  OP_BEGIN(HR_OP_BILINEAR_CAT);
  HR_REQUIRE(nbr >= 1 && nbr <= 4 && xs && hs && ws && cs, "bilinear_cat: args");
  fill_cat(op, dtype, hs, ws, cs, nbr, N, H, W);
  op.f[0] = align_corners ? 1.f : 0.f;
  op.p[0] = cat;
  for (int k = 0; k < nbr; ++k) op.p[1 + k] = (void*)xs[k];
  return hr_launch_bilinear_cat(op, (hipStream_t)stream);
}

extern "C" int hrnet_bilinear_cat_bwd(int dtype, const void* dcat, void* const* dxs, const int* hs,
                                      const int* ws, const int* cs, int nbr, int N, int H, int W,
                                      int align_corners, int accumulate, hr_stream_t stream) {
  OP_BEGIN(HR_OP_BILINEAR_CAT_BWD);
  HR_REQUIRE(nbr >= 1 && nbr <= 4 && dxs && hs && ws && cs, "bilinear_cat_bwd: args");
  fill_cat(op, dtype, hs, ws, cs, nbr, N, H, W);
  op.f[0] = align_corners ? 1.f : 0.f;
  op.i[17] = accumulate;
  op.p[0] = (void*)dcat;
  for (int k = 0; k < nbr; ++k) op.p[1 + k] = dxs[k];
  return hr_launch_bilinear_cat_bwd(op, (hipStream_t)stream);
}

extern "C" int hrnet_head_mix(int dtype, const void* x0, const void* w0, const float* bias, void* y, float* stats,
                              int rows_mode, const void* const* ts, const int* hs, const int* ws, int nup, int N, int H,
                              int W, int C0, int Cout, int align_corners, hr_stream_t stream) {
  OP_BEGIN(HR_OP_HEAD_MIX);
  HR_REQUIRE(nup >= 0 && nup <= 3 && (nup == 0 || (ts && hs && ws)), "head_mix: args");
  const int iv[8] = {dtype, N, H, W, C0, Cout, nup, align_corners};
  memcpy(op.i, iv, sizeof(iv));
  op.i[14] = rows_mode;
  op.p[0] = (void*)x0; op.p[1] = (void*)w0; op.p[2] = (void*)bias; op.p[3] = y; op.p[4] = stats;
  for (int k = 0; k < nup; ++k) {
    op.i[8 + 2 * k] = hs[k]; op.i[9 + 2 * k] = ws[k];
    op.p[5 + k] = (void*)ts[k];
  }
  return hr_launch_head_mix(op, (hipStream_t)stream);
}

extern "C" int hrnet_upsample_bilinear_t(int dtype, const void* g, void* const* outs, const int* hs, const int* ws,
                                         int nout, int N, int H, int W, int C, int align_corners, int streamed,
                                         hr_stream_t stream) {
  OP_BEGIN(HR_OP_UPSAMPLE_T);
  HR_REQUIRE(nout >= 1 && nout <= 3 && outs && hs && ws, "upsample_t: args");
  const int iv[7] = {dtype, N, H, W, C, nout, align_corners};
  memcpy(op.i, iv, sizeof(iv));
  op.i[13] = streamed;
  op.p[0] = (void*)g;
  for (int k = 0; k < nout; ++k) {
    op.p[1 + k] = outs[k];
    op.i[7 + 2 * k] = hs[k]; op.i[8 + 2 * k] = ws[k];
  }
  return hr_launch_upsample_t(op, (hipStream_t)stream);
}

extern "C" int hrnet_head_bwd(int dtype, int mode, const void* dy, const void* wT, const void* y, void* out,
                              const float* bn_scale, const float* bn_shift, const float* coef, int inner_relu, int N,
                              int H, int W, int K, int Cout, hr_stream_t stream) {
  OP_BEGIN(HR_OP_HEAD_BWD);
  const int iv[8] = {dtype, N, H, W, K, Cout, mode, inner_relu};
  memcpy(op.i, iv, sizeof(iv));
  op.p[0] = (void*)dy; op.p[1] = (void*)wT; op.p[2] = (void*)y; op.p[3] = out;
  op.p[4] = (void*)bn_scale; op.p[5] = (void*)bn_shift; op.p[6] = (void*)coef;
  return hr_launch_head_bwd(op, (hipStream_t)stream);
}

extern "C" int hrnet_im2col_stem(int dtype, const float* img_nchw, void* cols, int N, int C, int H, int W,
                                 int Ho, int Wo, int Kpad, hr_stream_t stream) {
  OP_BEGIN(HR_OP_IM2COL_STEM);
  const int iv[8] = {dtype, N, C, H, W, Ho, Wo, Kpad};
  memcpy(op.i, iv, sizeof(iv));
  op.p[0] = (void*)img_nchw; op.p[1] = cols;
  return hr_launch_im2col_stem(op, (hipStream_t)stream);
}

extern "C" int hrnet_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int H, int W, int Cp,
                                  int C, hr_stream_t stream) {
  OP_BEGIN(HR_OP_NHWC_TO_NCHW);
  const int iv[6] = {dtype, N, H, W, Cp, C};
  memcpy(op.i, iv, sizeof(iv));
  op.p[0] = (void*)src; op.p[1] = dst;
  return hr_launch_nhwc_to_nchw(op, (hipStream_t)stream);
}

extern "C" int hrnet_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int H, int W, int Cp,
                                  int C, hr_stream_t stream) {
  OP_BEGIN(HR_OP_NCHW_TO_NHWC);
  const int iv[6] = {dtype, N, H, W, Cp, C};
  memcpy(op.i, iv, sizeof(iv));
  op.p[0] = (void*)src; op.p[1] = dst;
  return hr_launch_nchw_to_nhwc(op, (hipStream_t)stream);
}

extern "C" int hrnet_bias_grad(int dtype, const void* dy, float* dbias, float* scratch, int pixels, int Cp,
                               int C, int accumulate, hr_stream_t stream) {
  OP_BEGIN(HR_OP_BIAS_GRAD);
  const int iv[5] = {dtype, pixels, Cp, C, accumulate};
  memcpy(op.i, iv, sizeof(iv));
  op.p[0] = (void*)dy; op.p[1] = dbias; op.p[2] = scratch;
  return hr_launch_bias_grad(op, (hipStream_t)stream);
}

extern "C" int hrnet_fill_zero(void* p, int64_t bytes, hr_stream_t stream) {
  OP_BEGIN(HR_OP_FILL);
  op.i[0] = (int32_t)(uint32_t)(bytes & 0xffffffffLL);
  op.i[1] = (int32_t)(uint32_t)((uint64_t)bytes >> 32);
  op.p[0] = p;
  return hr_launch_fill(op, (hipStream_t)stream);
}
