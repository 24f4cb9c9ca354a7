// Convolution entry points: argument checks, tile choice, mode choice; the generic instantiation.
// The device code lives in conv_body.h.
#include <string.h>

#include "conv_body.h"
#include "conv_ring.h"

HR_DEFINE_CONV_LAUNCH(hr_conv_launch_generic, CONV_GENERIC)

namespace {
// workgroup -> pixel-walk mapping of the tile-walking bodies (conv_body.h): contiguous runs per XCD (HRNET_CONV_XCD=0:
// consecutive walks on consecutive XCDs, as rounds 1-2)
inline int conv_xcd_runs() {
  static const int v = hr_knob("HRNET_CONV_XCD", 1);
  return v;
}
// which specialised body serves a launch (see conv_body.h)
inline int conv_mode(const ConvArgs& a, bool in_relu) {
  if (a.bs_y) return CONV_BS;
  if (!a.bias && !a.upz && !a.accumulate && a.stats) return CONV_FWD;
  if (a.bias && !a.upz && !a.accumulate) return CONV_FWDB;
  if (!a.bias && !a.in_scale && !a.in_sums && !in_relu && !a.stats) return CONV_DG;
  return CONV_GENERIC;
}

// 3x3 stride-1 launches conv_ring.hip serves. Forward: no bias / accumulate / zero-stuffing, output statistics by
// atomics (or none), the input's BatchNorm as sums or arrays (or a raw input). Input gradient (returns 2): raw
// input, optional accumulation into y, optional backward statistics (rows: hrnet_conv_rows_bwdstats()).
inline int ring_serves(const ConvArgs& a, int dtype, int ks, int stride, int mode, bool in_relu) {
  if (ks != 3 || stride != 1 || a.upz || a.bias || a.in_dy || a.in_dx || !hr_conv_ring_enabled()) return 0;
  const bool raw = !a.in_scale && !a.in_sums && !in_relu;
  if (mode == CONV_BS || (raw && a.accumulate && !a.stats))
    return hr_conv_ring_supported(dtype, a.N, a.H, a.W, a.Cin, a.Cout, 1) ? 2 : 0;
  if (a.accumulate || a.bs_y) return 0;
  if (mode != CONV_FWD && mode != CONV_GENERIC && mode != CONV_DG) return 0;
  if (a.stats && !a.stats_atomic) return 0;
  if (a.in_sums && a.in_beta != a.in_gamma + a.Cin) return 0;
  return hr_conv_ring_supported(dtype, a.N, a.H, a.W, a.Cin, a.Cout, 0) ? 1 : 0;
}

inline int launch_ring(const ConvArgs& a, int in_relu, hipStream_t s) {
  HrRingConv c;
  c.x = a.x; c.w = a.w; c.y = a.y;
  c.in_sums = a.in_sums; c.in_gb = a.in_gamma; c.in_scale = a.in_scale; c.in_shift = a.in_shift;
  c.stats = a.stats; c.in_inv_count = a.in_inv_count; c.in_eps = a.in_eps;
  c.bs_y = a.bs_y; c.bs_mask = a.bs_mask; c.bs_scale = a.bs_scale; c.bs_shift = a.bs_shift;
  c.N = a.N; c.H = a.H; c.W = a.W; c.Cin = a.Cin; c.Cout = a.Cout; c.in_relu = in_relu;
  c.accumulate = a.accumulate; c.bs_store_masked = a.bs_store_masked;
  c.x2 = a.x2; c.side = a.side;
  return hr_conv_ring_launch(c, s);
}
}  // namespace

extern "C" int hrnet_conv_tiles(int N, int Ho, int Wo, int Cout, int ks, int stride) {
  return choose_tile(N, Ho, Wo, Cout, ks, stride).gx;   // one statistics row per workgroup
}

extern "C" int hrnet_conv_tiles_bwdstats(int N, int Ho, int Wo, int Cout, int ks, int stride) {
  return choose_tile(N, Ho, Wo, Cout, ks, stride, true, ks == 3 && stride == 2).gx;
}

// statistics rows hrnet_conv2d_bwdstats writes for this launch (the kernel family is chosen per dtype and shape:
// the LDS-ring pipeline leaves one row per pixel walk of ITS grid)
extern "C" int hrnet_conv_rows_bwdstats(int dtype, int N, int Ho, int Wo, int Cin, int Cout, int ks, int stride) {
  if (ks == 3 && stride == 1 && hr_conv_ring_enabled() && hr_conv_ring_supported(dtype, N, Ho, Wo, Cin, Cout, 1))
    return hr_conv_ring_rows(N, Ho, Wo, Cin, Cout);
  return hrnet_conv_tiles_bwdstats(N, Ho, Wo, Cout, ks, stride);
}

// Which kernel family a recorded backward-statistics launch is bound to (HrOp.i[17] of HR_OP_CONV): 2 = the LDS-ring
// pipeline, 1 = the tile-walking body. A plan sizes the launch's rows buffer (and the finalize launch's row count) from
// hrnet_conv_rows_bwdstats when it is recorded; hrnet_conv_ring_enable() is a process-wide run-time switch, so the
// op carries the decision and hr_launch_conv() follows it - or fails - instead of deciding again (0: decide at launch,
// the direct C-ABI entry points).
extern "C" int hrnet_conv_route(int dtype, int N, int Ho, int Wo, int Cin, int Cout, int ks, int stride) {
  return (ks == 3 && stride == 1 && hr_conv_ring_enabled() && hr_conv_ring_supported(dtype, N, Ho, Wo, Cin, Cout, 1)) ? 2 : 1;
}

// the tile walk a conv launch of this shape takes: out5 = {tile height, tile width, output-channel block,
// pixel tiles per workgroup, pixel walks}; returns the number of pixel tiles (tests assert that the
// multi-tile walk, tiles-per-workgroup >= 2, is what they exercise)
extern "C" int hrnet_conv_tile_walk(int N, int Ho, int Wo, int Cout, int ks, int stride, int bwdstats, int s2d,
                                    int* out5) {
  const TileChoice tc = choose_tile(N, Ho, Wo, Cout, ks, stride, bwdstats != 0, s2d != 0);
  if (out5) { out5[0] = tc.th; out5[1] = tc.tw; out5[2] = tc.bn; out5[3] = tc.tpw; out5[4] = tc.gx; }
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
  ConvArgs a = {};
  a.bs_store_masked = op.i[14];
  a.in_dy = op.i[15]; a.in_dx = op.i[16];
  HR_REQUIRE((a.in_dy == 0 && a.in_dx == 0) || (ks == 1 && stride == 1 && !upz && !op.p[7]),
             "conv2d: a displaced input window belongs to a 1x1 stride-1 launch");
  a.x = (const char*)op.p[0];
  a.w = (const char*)op.p[1];
  a.in_scale = (const float*)op.p[2];
  a.in_shift = (const float*)op.p[3];
  a.bias = (const float*)op.p[4];
  a.y = (char*)op.p[5];
  a.stats = (float*)op.p[6];
  a.bs_y = (const char*)op.p[7];
  a.bs_mask = (const char*)op.p[8];
  a.bs_scale = (const float*)op.p[9];
  a.bs_shift = (const float*)op.p[10];
  a.in_sums = (const float*)op.p[11];
  a.in_gamma = (const float*)op.p[12];
  a.in_beta = (const float*)op.p[13];
  a.in_inv_count = op.f[0];
  a.in_eps = op.f[1];
  a.stats_atomic = op.i[13];
  HR_REQUIRE(!a.in_sums || (a.in_gamma && a.in_beta && !a.in_scale && Cin <= HR_CONV_MAXC && a.in_inv_count > 0.f),
             "conv2d: input batch sums need gamma/beta, no scale/shift arrays, Cin <= %d", HR_CONV_MAXC);
  HR_REQUIRE(!a.in_sums || !a.bs_y, "conv2d: batch-sum input is for forward launches");
  HR_REQUIRE(!a.bs_y || a.stats, "conv2d: backward statistics need a rows buffer");
  HR_REQUIRE(!a.bs_store_masked || a.bs_y, "conv2d: the masked store belongs to a backward-statistics launch");
  HR_REQUIRE(!a.bs_y || (!a.in_scale && !a.bias && !op.i[11]),
             "conv2d: backward statistics are for input-gradient launches (no input affine / ReLU / bias)");
  HR_REQUIRE((a.bs_scale == nullptr) == (a.bs_shift == nullptr), "conv2d: mask scale/shift must come together");
  a.N = N; a.H = H; a.W = W; a.Cin = Cin;
  a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
  a.in_relu = op.i[11]; a.upz = upz; a.accumulate = op.i[12];
  const int mode = conv_mode(a, op.i[11] != 0);
  HR_REQUIRE(!a.in_sums || mode == CONV_FWD || mode == CONV_FWDB,
             "conv2d: batch-sum input needs a forward launch that writes statistics or adds a bias");
  // the branch 3x3 convolutions: LDS-ring pipeline (conv_ring.hip)
  const int route = op.i[17];
  HR_REQUIRE(route >= 0 && route <= 2, "conv2d: bad route %d", route);
  const int rs = route == 1 ? 0 : ring_serves(a, dtype, ks, op.i[9], mode, op.i[11] != 0);
  HR_REQUIRE(route != 2 || rs, "conv2d: the op was recorded for the LDS-ring kernel, which does not serve it now "
             "(hrnet_conv_ring_enable() changed after the plan was recorded: record the plan again)");
  if (rs) return launch_ring(a, op.i[11], s);
  // the GEMM-shaped head layer (and its input gradient): every output channel of a pixel block in one workgroup
  if (ks == 1 && stride == 1 && !upz && !a.accumulate && !a.bs_y && !a.in_scale && !a.in_sums && !op.i[11] &&
      (!a.stats || a.stats_atomic) && !a.in_dy && !a.in_dx && hr_gemm_pw_supported(dtype, Cin, Cout))
    return hr_gemm_pw(a.x, a.w, a.bias, a.y, a.stats, (long long)N * H * W, Cin, Cout, s);
  // input gradient of a 3x3 stride-2 conv: the input-gradient bodies evaluate the four output parities
  // from the real dY tile (S2D, template stride 4); other modes read it as a zero-stuffed grid
  const bool s2d = upz && (mode == CONV_BS || mode == CONV_DG);
  if (upz) {
    HR_REQUIRE(ks == 3, "conv2d: upz needs a 3x3 kernel");
    HR_REQUIRE((Ho + 1) / 2 == H && (Wo + 1) / 2 == W, "conv2d: upz shape mismatch");
    stride = s2d ? 4 : 1;
    if (s2d) { a.upz = 0; a.Hz = H; a.Wz = W; }
    else { a.Hz = Ho; a.Wz = Wo; }
  } else {
    const int pad = ks / 2;
    HR_REQUIRE((H + 2 * pad - ks) / stride + 1 == Ho && (W + 2 * pad - ks) / stride + 1 == Wo,
               "conv2d: output %dx%d does not match input %dx%d ks=%d stride=%d", Ho, Wo, H, W, ks, stride);
    a.Hz = H; a.Wz = W;
  }
  const TileChoice tc = choose_tile(N, Ho, Wo, Cout, ks, op.i[9], op.p[7] != nullptr, s2d);
  a.tiles_y = (Ho + tc.th - 1) / tc.th;
  a.tiles_x = (Wo + tc.tw - 1) / tc.tw;
  a.total_tiles = N * a.tiles_y * a.tiles_x;
  a.tpw = tc.tpw;
  a.gx = tc.gx;
  a.gy = (Cout + tc.bn - 1) / tc.bn;
  a.xcd_runs = conv_xcd_runs();
  // the tile choice is keyed on the ORIGINAL stride so hrnet_conv_tiles() agrees; a upz conv
  // runs the stride-1 kernel with that tile
  ConvLaunch l;
  l.a = a; l.tc = tc; l.dtype = dtype; l.N = N; l.ks = ks; l.stride = stride;
  switch (mode) {
    case CONV_BS: return hr_conv_launch_bs(l, s);
    case CONV_FWD: return hr_conv_launch_fwd(l, s);
    case CONV_DG: return hr_conv_launch_dg(l, s);
    case CONV_FWDB: return hr_conv_launch_fwdb(l, s);
    default: return hr_conv_launch_generic(l, s);
  }
}

// HR_OP_CONV_SUM: i = dtype,N,H,W,Cin,Cout,ks,stats_atomic; f = 1/count, eps;
// p = x, w, in_scale, in_shift, in_sums, in_gamma, in_beta, y, stats, x2, side
int hr_launch_conv_sum(const HrOp& op, hipStream_t s) {
  const int dtype = op.i[0], N = op.i[1], H = op.i[2], W = op.i[3], Cin = op.i[4], Cout = op.i[5], ks = op.i[6];
  HR_REQUIRE(dtype == HR_F32 || dtype == HR_BF16, "conv2d_sum: bad dtype %d", dtype);
  HR_REQUIRE(ks == 1 || ks == 3, "conv2d_sum: kernel size %d", ks);
  HR_REQUIRE(N > 0 && H > 0 && W > 0, "conv2d_sum: empty shape");
  HR_REQUIRE(Cin % (dtype == HR_F32 ? 4 : 8) == 0 && Cout % 16 == 0, "conv2d_sum: channel counts (Cin=%d Cout=%d)", Cin, Cout);
  HR_REQUIRE((double)H * W * (Cin > Cout ? Cin : Cout) * 4.0 < 2147483648.0, "conv2d_sum: one image exceeds 2 GiB");
  ConvArgs a = {};
  a.x = (const char*)op.p[0];
  a.w = (const char*)op.p[1];
  a.in_scale = (const float*)op.p[2];
  a.in_shift = (const float*)op.p[3];
  a.in_sums = (const float*)op.p[4];
  a.in_gamma = (const float*)op.p[5];
  a.in_beta = (const float*)op.p[6];
  a.y = (char*)op.p[7];
  a.stats = (float*)op.p[8];
  a.x2 = (const char*)op.p[9];
  a.side = (char*)op.p[10];
  a.in_inv_count = op.f[0];
  a.in_eps = op.f[1];
  a.stats_atomic = op.i[7];
  HR_REQUIRE(a.x && a.w && a.y && a.x2 && a.side, "conv2d_sum: null pointer");
  HR_REQUIRE((a.in_scale != nullptr) == (a.in_shift != nullptr), "conv2d_sum: scale/shift must come together");
  HR_REQUIRE((a.in_scale != nullptr) != (a.in_sums != nullptr), "conv2d_sum: the BatchNorm term needs scale/shift OR batch sums");
  HR_REQUIRE(!a.in_sums || (a.in_gamma && a.in_beta && Cin <= HR_CONV_MAXC && a.in_inv_count > 0.f),
             "conv2d_sum: batch sums need gamma/beta, Cin <= %d", HR_CONV_MAXC);
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Ho = H; a.Wo = W; a.Cout = Cout;
  a.in_relu = 1; a.upz = 0; a.accumulate = 0;
  a.Hz = H; a.Wz = W;
  // the narrow branch convolutions: the residual sum inside the LDS-ring pipeline (round 4; conv_ring.hip, X2)
  if (ks == 3 && (!a.stats || a.stats_atomic) && (!a.in_sums || a.in_beta == a.in_gamma + Cin) &&
      hr_conv_ring_sum_supported(dtype, N, H, W, Cin, Cout))
    return launch_ring(a, 1, s);
  const TileChoice tc = choose_tile(N, H, W, Cout, ks, 1, false, false);
  a.tiles_y = (H + tc.th - 1) / tc.th;
  a.tiles_x = (W + tc.tw - 1) / tc.tw;
  a.total_tiles = N * a.tiles_y * a.tiles_x;
  a.tpw = tc.tpw;
  a.gx = tc.gx;
  a.gy = (Cout + tc.bn - 1) / tc.bn;
  a.xcd_runs = conv_xcd_runs();
  ConvLaunch l;
  l.a = a; l.tc = tc; l.dtype = dtype; l.N = N; l.ks = ks; l.stride = 1;
  return hr_conv_launch_fwds(l, s);
}

extern "C" int hrnet_conv2d_sum(int dtype, const void* x, const void* x2, const void* w, const float* in_scale,
                                const float* in_shift, const float* in_sums, const float* in_gamma,
                                const float* in_beta, float in_inv_count, float in_eps, void* side, void* y,
                                float* stats, int stats_atomic, int N, int H, int W, int Cin, int Cout, int ks,
                                hr_stream_t stream) {
  HrOp op = {};
  op.kind = HR_OP_CONV_SUM;
  const int iv[8] = {dtype, N, H, W, Cin, Cout, ks, stats_atomic};
  for (int k = 0; k < 8; ++k) op.i[k] = iv[k];
  op.f[0] = in_inv_count; op.f[1] = in_eps;
  op.p[0] = (void*)x; op.p[1] = (void*)w; op.p[2] = (void*)in_scale; op.p[3] = (void*)in_shift;
  op.p[4] = (void*)in_sums; op.p[5] = (void*)in_gamma; op.p[6] = (void*)in_beta; op.p[7] = y; op.p[8] = stats;
  op.p[9] = (void*)x2; op.p[10] = side;
  return hr_launch_conv_sum(op, (hipStream_t)stream);
}

// y = conv3x3(x, dilation d, padding d): nine 1x1 launches over input windows displaced by ((r-1)d, (s-1)d), the
// first overwriting y and the others accumulating (f32 accumulation inside a launch, y rounded between taps)
extern "C" int hrnet_conv2d_dilated3x3(int dtype, const void* x, const void* w_taps, long long tap_stride_bytes,
                                       void* y, int N, int H, int W, int Cin, int Cout, int dilation,
                                       hr_stream_t stream) {
  HR_REQUIRE(dilation >= 1 && w_taps && tap_stride_bytes > 0, "conv2d_dilated3x3: arguments");
  for (int t = 0; t < 9; ++t) {
    HrOp op = {};
    op.kind = HR_OP_CONV;
    const int iv[13] = {dtype, N, H, W, Cin, H, W, Cout, 1, 1, 0, 0, t > 0 ? 1 : 0};
    for (int k = 0; k < 13; ++k) op.i[k] = iv[k];
    op.i[15] = (t / 3 - 1) * dilation;
    op.i[16] = (t % 3 - 1) * dilation;
    op.p[0] = (void*)x;
    op.p[1] = (void*)((const char*)w_taps + (size_t)t * tap_stride_bytes);
    op.p[5] = y;
    if (int e = hr_launch_conv(op, (hipStream_t)stream)) return e;
  }
  return 0;
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

extern "C" int hrnet_conv2d_bnref(int dtype, const void* x, const void* w, const float* in_sums, const float* in_gamma,
                                  const float* in_beta, float in_inv_count, float in_eps, const float* bias, void* y,
                                  float* out_sums, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int ks,
                                  int stride, int in_relu, hr_stream_t stream) {
  HrOp op = {};
  op.kind = HR_OP_CONV;
  const int iv[14] = {dtype, N, H, W, Cin, Ho, Wo, Cout, ks, stride, 0, in_relu, 0, 1};
  for (int k = 0; k < 14; ++k) op.i[k] = iv[k];
  op.p[0] = (void*)x; op.p[1] = (void*)w; op.p[4] = (void*)bias; op.p[5] = y; op.p[6] = out_sums;
  op.p[11] = (void*)in_sums; op.p[12] = (void*)in_gamma; op.p[13] = (void*)in_beta;
  op.f[0] = in_inv_count; op.f[1] = in_eps;
  return hr_launch_conv(op, (hipStream_t)stream);
}

extern "C" int hrnet_conv2d_bwdstats(int dtype, const void* x, const void* w, void* y, float* stats,
                                     const void* bs_y, const void* bs_mask, const float* bs_scale,
                                     const float* bs_shift, int N, int H, int W, int Cin, int Ho, int Wo,
                                     int Cout, int ks, int stride, int upz, int accumulate, hr_stream_t stream) {
  HrOp op = {};
  op.kind = HR_OP_CONV;
  const int iv[13] = {dtype, N, H, W, Cin, Ho, Wo, Cout, ks, stride, upz, 0, accumulate};
  for (int k = 0; k < 13; ++k) op.i[k] = iv[k];
  op.p[0] = (void*)x; op.p[1] = (void*)w; op.p[5] = y; op.p[6] = stats;
  op.p[7] = (void*)bs_y; op.p[8] = (void*)bs_mask; op.p[9] = (void*)bs_scale; op.p[10] = (void*)bs_shift;
  HR_REQUIRE(bs_y, "conv2d_bwdstats: null bs_y");
  return hr_launch_conv(op, (hipStream_t)stream);
}

// which specialised kernel family hrnet_conv2d / hrnet_conv2d_bwdstats pick for these operands
// (0 conv_kernel, 1 conv_bs_kernel, 2 conv_fwd_kernel, 3 conv_dg_kernel)
extern "C" int hrnet_conv_mode(int bwdstats, int has_bias, int upz, int accumulate, int has_stats, int has_affine,
                               int in_relu) {
  ConvArgs a = {};
  a.bs_y = bwdstats ? (const char*)1 : nullptr;
  a.bias = has_bias ? (const float*)1 : nullptr;
  a.upz = upz; a.accumulate = accumulate;
  a.stats = has_stats ? (float*)1 : nullptr;
  a.in_scale = has_affine ? (const float*)1 : nullptr;
  return conv_mode(a, in_relu != 0);
}

// Demangled-style name of the kernel instantiation hrnet_conv2d launches for this shape (so that
// bench.py's per-kernel timings can be matched against rocprofv3's kernel trace).
extern "C" int hrnet_conv_kernel_name(int dtype, int N, int Ho, int Wo, int Cin, int Cout, int ks, int stride,
                                      int upz, int mode, char* buf, int buflen) {
  // (the head's 480 -> 480 layer and its input gradient run the GEMM kernel; in HRNET_DETERMINISTIC=1 mode the
  // forward launch stays on the tile-walking body, which this name query cannot see)
  if (ks == 1 && stride == 1 && !upz && hr_gemm_pw_supported(dtype, Cin, Cout) &&
      (mode == CONV_FWD || mode == CONV_FWDB || mode == CONV_DG))
    return snprintf(buf, buflen, "gemm_pw_kernel");
  if (ks == 3 && stride == 1 && !upz && mode == CONV_FWDS && hr_conv_ring_sum_supported(dtype, N, Ho, Wo, Cin, Cout)) {
    // (the residual-sum form of the narrow instantiations: "..., false, true"; the batch-statistics mode decides at
    // launch - HRNET_DETERMINISTIC=1 keeps the tile-walking body, which this name query cannot see)
    char tmp[96];
    hr_conv_ring_name(hr_conv_ring_supported(dtype, N, Ho, Wo, Cin, Cout, 0), 0, tmp, sizeof(tmp));
    const int n = (int)strlen(tmp);
    if (n > 1) tmp[n - 1] = 0;          // drop the closing '>'
    return snprintf(buf, buflen, "%s, true>", tmp);
  }
  if (ks == 3 && stride == 1 && !upz && hr_conv_ring_enabled()) {
    const int bsm = mode == CONV_BS ? 1 : 0;
    if ((bsm || mode == CONV_FWD || mode == CONV_GENERIC || mode == CONV_DG) &&
        hr_conv_ring_supported(dtype, N, Ho, Wo, Cin, Cout, bsm))
      return hr_conv_ring_name(hr_conv_ring_supported(dtype, N, Ho, Wo, Cin, Cout, bsm), bsm, buf, buflen);
  }
  const bool s2d = upz && (mode == CONV_BS || mode == CONV_DG);
  const TileChoice tc = choose_tile(N, Ho, Wo, Cout, ks, stride, mode == CONV_BS, s2d);
  static const int wp[8] = {4, 2, 2, 2, 2, 0, 4, 2}, wc[8] = {1, 2, 2, 2, 2, 0, 1, 2};
  const int kstride = s2d ? 4 : (ks == 1 || upz) ? 1 : stride;
  const int km = conv_km(dtype, ks, Cin, tc.id);
  static const char* names[6] = {"conv_kernel", "conv_bs_kernel", "conv_fwd_kernel", "conv_dg_kernel", "conv_fwdb_kernel",
                                 "conv_fwds_kernel"};
  return snprintf(buf, buflen, "%s<%s, %d, %d, %d, %d, %d, %d, %d, %d>", names[mode >= 0 && mode < 6 ? mode : 0],
                  dtype == HR_F32 ? "float" : "__bf16", ks, kstride, tc.th, tc.tw, tc.bn, wp[tc.id], wc[tc.id], km);
}
