/*
 * hrnet_hip.h — C ABI of libhrnet_hip.so: the MI355X (gfx950) device path of the HRNet
 * hand-pose hot path.
 *
 * The reference has no FFI on this path: every op below replaces a stock PyTorch op
 * reached from /root/reference/lib/models/pose_hrnet.py, lib/core/loss.py and
 * lib/utils/heatmap_decoding.py (file:line cited per entry point).  The host side
 * (hrnet-hand-pose-estimation_amd/lib/hipnet/) binds these with ctypes; INTEGRATION.md
 * shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (PyTorch's caching allocator); the library allocates nothing.
 *   - activations are NHWC; `dtype` selects the storage/arithmetic type of activations and
 *     packed weights: HR_F32 (exact f32 MFMA, f32 accumulate) or HR_BF16 (bf16 MFMA, f32
 *     accumulate). Statistics, BN coefficients, partial sums, losses and master weights
 *     are always f32.
 *   - stream-ordered and re-entrant: kernels are enqueued on `stream`, nothing synchronises.
 *   - return 0 on success, a negative HR_E_* code otherwise; hrnet_last_error_string()
 *     describes the last failure of the calling thread. Nothing throws across the ABI.
 */
#ifndef HRNET_HIP_H
#define HRNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* hr_stream_t; /* hipStream_t */

enum { HR_F32 = 0, HR_BF16 = 1 };

enum {
  HR_OK = 0,
  HR_E_BADARG = -1,   /* shape / alignment / dtype the kernels do not support */
  HR_E_LAUNCH = -2,   /* HIP reported a launch error */
  HR_E_BADOP = -3     /* unknown op kind in a program */
};

/* ---- op kinds of a recorded program (hrnet_program_run) -------------------------------- */
enum {
  HR_OP_CONV = 1,          /* conv / dgrad (implicit GEMM, MFMA); a backward-statistics op may set i[14] = 1: the
                              stored gradient is dz = v * [mask > 0] (what the fused backward launches expect) */
  HR_OP_WGRAD = 2,         /* weight gradient partial slabs; i[12] = 1: p[4] is the OIHW f32 gradient itself
                              ([i[13] = real Cout][i[14] = real Cin][ks][ks]) and every workgroup ADDS its tile into it with
                              float atomics - no slabs, no reduce launch, not bit-reproducible. HR_OP_BWD_FUSED / HR_OP_BWD_PW
                              take the same switch in i[8] (gradient in p[11], real Cout / Cin in i[9] / i[10]) */
  HR_OP_WGRAD_REDUCE = 3,  /* slabs -> OIHW f32 gradient */
  HR_OP_BN_FINALIZE = 4,   /* stat partials -> scale/shift (+ running stats) */
  HR_OP_SUM_TERMS = 5,     /* out = relu(sum_t relu_t(affine_t(up_t(src_t)))) */
  HR_OP_GRAD_TERM = 6,     /* dst (+)= A*pool(g*mask) + B*y + C */
  HR_OP_BN_BWD_REDUCE = 7, /* per-channel sum(dz), sum(dz*y) partials; p[6] (optional): dz itself (pooled, masked,
                              [N,H,W,C] in the compute dtype) stored for the apply pass, which then runs
                              HR_OP_GRAD_TERM with g = that tensor, sh = 0 and no masks (in place) */
  HR_OP_BN_BWD_FINALIZE = 8,
  HR_OP_BILINEAR_CAT = 9,
  HR_OP_BILINEAR_CAT_BWD = 10,
  HR_OP_IM2COL_STEM = 11,
  HR_OP_NHWC_TO_NCHW = 12,
  HR_OP_NCHW_TO_NHWC = 13,
  HR_OP_PACK_WEIGHTS = 14,
  HR_OP_BIAS_GRAD = 15,
  HR_OP_FILL = 16,
  HR_OP_PACK_TABLE = 17,
  HR_OP_EVENT_RECORD = 18, /* p[0] = event, recorded on the op's lane */
  HR_OP_STREAM_WAIT = 19,  /* p[0] = event, the op's lane waits for it */
  HR_OP_WGRAD_REDUCE_TABLE = 20, /* p[0] = device HrWredEnt table, i[0] = n, i[1] = total blocks */
  HR_OP_BWD_FUSED = 21,    /* hrnet_conv3x3_bwd_fused */
  HR_OP_BN_FINALIZE_TABLE = 22, /* p[0] = device HrBnEnt table, i[0] = n, i[1] = total blocks */
  HR_OP_BWD_PW = 23,       /* hrnet_conv1x1_bwd_fused (slots as HR_OP_BWD_FUSED) */
  HR_OP_CONV_SUM = 24,     /* hrnet_conv2d_sum */
  HR_OP_EW_TABLE = 25,     /* several HR_OP_GRAD_TERM / HR_OP_BN_BWD_REDUCE / HR_OP_BN_BWD_FINALIZE / HR_OP_SUM_TERMS jobs as
                              ONE launch (HR_OP_SUM_TERMS: i[4] = 1 if a job's BatchNorm is given as batch sums, and a
                              job carries its eps bits in i[18] instead of i[16]):
                              p[0] = device array of HrOp jobs (slots as for the single op; i[16] = first block of the
                              job, i[17] = its blocks: hrnet_ew_table_blocks()), i[0] = jobs, i[1] = total blocks,
                              i[2] = kind of the jobs, i[3] = dtype */
  HR_OP_HEAD_MIX = 26,     /* hrnet_head_mix: i = {dtype, N, H, W, C0, Cout, nup, align, h1, w1, h2, w2, h3, w3, rows
                              mode}, p = {x0, w0 packed, bias, y, statistics, t1, t2, t3} */
  HR_OP_HEAD_BWD = 28,     /* hrnet_head_bwd: i = {dtype, N, H, W, K, Cout, mode, inner_relu}, p = {dY, wT, y, out, bn scale,
                              bn shift, coef} */
  HR_OP_POOL_REDUCE = 29,  /* the BatchNorm-backward reduction of up to three nearest-up-sampled terms of one fuse sum
                              (pose_hrnet.py:257-264) in ONE walk over the sum's gradient: i = {dtype, N, H, W, C,
                              nlev}, p = {g, mask, y1, dz1, partials1, y2, dz2, partials2, y3, dz3, partials3}; level
                              l pools 2^l x 2^l blocks; dz_l is stored for the apply pass (HR_OP_GRAD_TERM with g =
                              dz_l, sh = 0); partials_l[hrnet_reduce_blocks(N, H / 2, W / 2, C)][2][C] */
  HR_OP_UPSAMPLE_T = 27    /* hrnet_upsample_bilinear_t: i = {dtype, N, H, W, C, nout, align, h1, w1, h2, w2, h3, w3,
                              streamed}, p = {G, out1, out2, out3} */
};

/* One recorded op: integer / float / pointer slots, meaning per kind (see the
 * hrnet_* function of the same name; slots are filled in argument order). */
typedef struct HrOp {
  int32_t kind;
  int32_t i[19];
  float f[4];
  void* p[14];
} HrOp;

const char* hrnet_last_error_string(void);
int hrnet_abi_version(void);

/* Run `n` recorded ops in order on `stream` (one host call per forward / backward pass). */
int hrnet_program_run(const HrOp* ops, int n, hr_stream_t stream);
/* blocks of one job of a HR_OP_EW_TABLE launch (kind = the single op's kind; finalize: N = H = W = 1) */
int hrnet_ew_table_blocks(int kind, int dtype, int N, int H, int W, int C);
/* The same over several streams: op.i[HR_LANE_SLOT] selects streams[lane]; HR_OP_EVENT_RECORD /
 * HR_OP_STREAM_WAIT ops express the dependencies between lanes (independent branches of a
 * HighResolutionModule, weight-gradient work off the critical path). Events come from
 * hrnet_event_create (host-side handles; no device memory). */
#define HR_LANE_SLOT 18
int hrnet_program_run_streams(const HrOp* ops, int n, const hr_stream_t* streams, int nstreams);
/* Measurement form of the above: a timing event behind every op on its lane; end_ms[k] = completion of op k in
 * ms since the call began on op 0's lane. Synchronises every stream before returning (not for the training loop). */
int hrnet_program_run_streams_timed(const HrOp* ops, int n, const hr_stream_t* streams, int nstreams, float* end_ms);
void* hrnet_event_create(void);
int hrnet_event_destroy(void* event);

/*
 * Convolution as implicit GEMM on MFMA. Replaces nn.Conv2d forward (pose_hrnet.py:22-25,
 * :65-71, :200-204, :218-222, :283-287, :334-347) and, with packed transposed weights, its
 * input gradient.
 *   x        [N,H,W,Cin]   (Cin % 8 == 0; HR_F32: % 4)
 *   w        packed [Cout][ks*ks][Cin] in `dtype` (hrnet_pack_weights)
 *   in_scale/in_shift  optional per-Cin affine applied to x on load (the producer's
 *            BatchNorm, pose_hrnet.py:45,49 ...), then ReLU if in_relu; zero padding is
 *            applied AFTER the transform, as the reference pads the activated tensor.
 *   bias     optional [Cout] f32
 *   y        [N,Ho,Wo,Cout] (Cout % 16 == 0), overwritten or accumulated into
 *   stats    optional [tiles][2][Cout] f32 per-tile sum / sum of squares of y (BatchNorm
 *            batch statistics, finished by hrnet_bn_finalize); tiles = hrnet_conv_tiles()
 *   ks 1|3, pad = ks/2, stride 1|2.  upz=1: x is read as if zero-stuffed x2 (input gradient
 *   of a stride-2 conv: logical input [N,Ho,Wo,Cin], stride forced to 1).
 */
int hrnet_conv2d(int dtype, const void* x, const void* w, const float* in_scale,
                 const float* in_shift, const float* bias, void* y, float* stats, int N, int H,
                 int W, int Cin, int Ho, int Wo, int Cout, int ks, int stride, int upz,
                 int in_relu, int accumulate, hr_stream_t stream);

/* out[i] = sum_{j<k} coefs[j] * srcs[j][i], f32, k <= 8 (host arrays of k device pointers / k coefficients): the
 * frame differences and the weighted temporal aggregation of pose_hrnet_PoseAggr
 * (lib/models/pose_hrnet_PoseAggr.py:612-640) */
int hrnet_lincomb_f32(float* out, long long n, int k, const float* const* srcs, const float* coefs,
                      hr_stream_t stream);

/*
 * 3x3 convolution with dilation d and padding d (the offset-generating convs of pose_hrnet_PoseAggr,
 * lib/models/pose_hrnet_PoseAggr.py:497-506: dilations 3, 6, 12, 18, 24; a halo of d pixels does not fit the tiled
 * conv body): nine 1x1 launches over input windows displaced by ((r-1)d, (s-1)d), accumulating into y.
 *   w_taps: nine packed 1x1 weight matrices [Cout][Cin] (hrnet_pack_weights mode 0 of w[:, :, r, s]), tap t = 3r+s
 *   at w_taps + t * tap_stride_bytes. No bias / BatchNorm prologue; y is rounded to `dtype` between taps.
 */
int hrnet_conv2d_dilated3x3(int dtype, const void* x, const void* w_taps, long long tap_stride_bytes, void* y,
                            int N, int H, int W, int Cin, int Cout, int dilation, hr_stream_t stream);

/*
 * Input-gradient convolution that also gathers the statistics of the BatchNorm backward pass of
 * its output (autograd of pose_hrnet.py:43-57 - the reduction half of native_batch_norm_backward):
 * y (overwritten or accumulated) is the gradient v of an activation; stats rows receive
 * (sum dz, sum dz*yraw) per channel with dz = v * [m > 0], m = bs_mask (or bs_y when bs_mask is
 * NULL) optionally mapped through bs_scale/bs_shift; no mask at all when bs_mask and bs_scale are
 * both NULL. bs_y / bs_mask are laid out like y. Rows are finished by hrnet_bn_bwd_finalize with
 * blocks = hrnet_conv_tiles_bwdstats(N,Ho,Wo,Cout,ks,stride).
 */
int hrnet_conv2d_bwdstats(int dtype, const void* x, const void* w, void* y, float* stats,
                          const void* bs_y, const void* bs_mask, const float* bs_scale,
                          const float* bs_shift, int N, int H, int W, int Cin, int Ho, int Wo,
                          int Cout, int ks, int stride, int upz, int accumulate, hr_stream_t stream);
/*
 * The 3x3 stride-1 branch convolutions (BasicBlock.conv1 / conv2, pose_hrnet.py:41-57) run as an LDS-ring
 * pipeline (csrc/conv_ring.hip) behind hrnet_conv2d / hrnet_conv2d_bnref when the shape is served (bf16, Cin a
 * multiple of 32, no bias / accumulate / zero-stuffing, output statistics by atomics or none). This switch turns
 * that routing on (1, the default; environment HRNET_CONV_RING) or off (0: the tile-walking body of conv_body.h)
 * for A/B measurements and parity tests; 2: on, and the >= 96-channel 16x16-tile instantiation also takes maps larger
 * than 16x16 (it serves them correctly but loses inside the training step, so 1 keeps it to 16x16 maps). Returns the
 * previous setting (-1: never decided).
 */
int hrnet_conv_ring_enable(int on);
int hrnet_conv_ring_supported(int dtype, int N, int H, int W, int Cin, int Cout);
/* statistics rows hrnet_conv2d_bwdstats leaves for a launch of this shape (hrnet_conv_tiles_bwdstats() for the
 * tile-walking body; the pixel walks of the LDS-ring grid where that serves the launch) */
int hrnet_conv_rows_bwdstats(int dtype, int N, int Ho, int Wo, int Cin, int Cout, int ks, int stride);
/* kernel family a RECORDED backward-statistics launch of this shape is bound to (HR_OP_CONV i[17]): 2 = LDS ring,
 * 1 = tile-walking body; the rows buffer above is sized for that family, so the op keeps the decision and a later
 * hrnet_conv_ring_enable() cannot change how many rows the launch writes (it fails instead). 0 = decide at launch. */
int hrnet_conv_route(int dtype, int N, int Ho, int Wo, int Cin, int Cout, int ks, int stride);
/* hrnet_conv2d_sum on the LDS-ring pipeline for the narrow branch layers (bf16, 3x3, 32 / 64 input channels): off by
 * default (measured slower inside the training step), on = 1; returns the previous setting */
int hrnet_conv_ring_sum_enable(int on);

/* name of the kernel instantiation chosen for a shape, as rocprofv3 demangles it (returns length) */
int hrnet_conv_kernel_name(int dtype, int N, int Ho, int Wo, int Cin, int Cout, int ks, int stride, int upz,
                           int mode, char* buf, int buflen);
/* the specialised kernel family a conv launch uses: 0 conv_kernel (everything by run-time flag), 1
 * conv_bs_kernel (backward statistics), 2 conv_fwd_kernel (forward conv feeding a BatchNorm), 3
 * conv_dg_kernel (plain input gradient), 4 conv_fwdb_kernel (forward conv with bias) */
int hrnet_conv_mode(int bwdstats, int has_bias, int upz, int accumulate, int has_stats, int has_affine,
                    int in_relu);
int hrnet_wgrad_kernel_name(int dtype, int Ho, int Wo, int Cout, int Cin, int ks, int stride, char* buf,
                            int buflen);
/* number of per-tile stat rows hrnet_conv2d writes for this shape */
int hrnet_conv_tiles(int N, int Ho, int Wo, int Cout, int ks, int stride);
/* the tile walk of a conv launch: out5 = {tile h, tile w, output-channel block, pixel tiles per workgroup,
 * pixel walks (= statistics rows)}; returns the number of pixel tiles. s2d = the four-parity input gradient
 * of a 3x3 stride-2 conv (upz with hrnet_conv2d_bwdstats or a plain input-gradient launch). */
int hrnet_conv_tile_walk(int N, int Ho, int Wo, int Cout, int ks, int stride, int bwdstats, int s2d, int* out5);
/* rows of a hrnet_conv2d_bwdstats launch (its tile choice differs for wide 1x1 outputs) */
int hrnet_conv_tiles_bwdstats(int N, int Ho, int Wo, int Cout, int ks, int stride);

/*
 * Weight gradient: slabs[s][Cout][ks*ks][Cin] f32 partial sums over disjoint pixel ranges
 * (deterministic, no atomics); x is transformed on load exactly as in hrnet_conv2d.
 *   dy [N,Ho,Wo,Cout], x [N,H,W,Cin]; nsplit slabs = hrnet_wgrad_splits().
 */
int hrnet_conv2d_wgrad(int dtype, const void* x, const void* dy, const float* in_scale,
                       const float* in_shift, float* slabs, int N, int H, int W, int Cin, int Ho,
                       int Wo, int Cout, int ks, int stride, int in_relu, int nsplit,
                       hr_stream_t stream);
int hrnet_wgrad_splits(int dtype, int N, int Ho, int Wo, int Cout, int Cin, int ks, int stride);
/* pixel tiles the launch walks in all (a split takes tiles / nsplit of them, grid-strided) */
int hrnet_wgrad_tiles(int dtype, int N, int Ho, int Wo, int Cout, int Cin, int ks, int stride);
/* workgroups per split (a launch runs splits x this many) */
int hrnet_wgrad_blocks_per_split(int dtype, int Ho, int Wo, int Cout, int Cin, int ks, int stride);
/* slabs -> grad_oihw[Cout_real][Cin_real][ks][ks] f32 (+= if accumulate). Cout/Cin are the
 * padded slab extents; stem: kflat=1 means slab K index is the flattened (tap,ci) of the
 * im2col'ed stem (Cin_real*ks*ks real entries). */
int hrnet_wgrad_reduce(const float* slabs, float* grad_oihw, int nsplit, int Cout, int Cin,
                       int ks, int Cout_real, int Cin_real, int kflat, int accumulate,
                       hr_stream_t stream);

/*
 * Fused backward of a 3x3 stride-1 convolution y = conv(a), a = relu?(in_scale*x + in_shift), whose output
 * feeds a BatchNorm (autograd of the BasicBlock body, pose_hrnet.py:41-57): ONE launch applies the BatchNorm
 * backward to the upstream gradient while staging it (g = A*dz + B*y + C with coef = [3][Cout] of
 * hrnet_bn_bwd_finalize; coef NULL: g = dz), forms the weight-gradient slabs (layout and reduction as
 * hrnet_conv2d_wgrad; nsplit = hrnet_bwd_fused_splits()), the input gradient (+ `addend`, the residual
 * stream), masks it with [a > 0] when mask_out (so what is stored is the dz of the NEXT BatchNorm backward)
 * and, when `rows` is given, leaves (sum dx, sum dx*bs_y) per channel in rows[nsplit][2][Cin] for
 * hrnet_bn_bwd_finalize (bs_y NULL: the second sum is 0).
 *   dz, y [N,H,W,Cout]; x, dx, addend, bs_y [N,H,W,Cin]; wT = hrnet_pack_weights(mode 1) of the conv.
 * dz must already carry the ReLU mask of the BatchNorm output it belongs to (an in-place hrnet_grad_term with
 * coef NULL does that where the producer did not). Served shapes: hrnet_bwd_fused_supported().
 */
int hrnet_conv3x3_bwd_fused(int dtype, const void* dz, const void* y, const float* coef, const void* x,
                            const float* in_scale, const float* in_shift, int in_relu, const void* wT, void* dx,
                            const void* addend, int mask_out, float* rows, const void* bs_y, float* slabs, int N,
                            int H, int W, int Cin, int Cout, hr_stream_t stream);
int hrnet_bwd_fused_supported(int dtype, int Cin, int Cout);
/*
 * The fused launches can finish the BatchNorm backward of their own output themselves: instead of `coef` (written by
 * a hrnet_bn_bwd_finalize launch in between) they take the partial rows [nrows][2][Cout] the previous launch left
 * - every workgroup sums them in a fixed order (f64) and builds A,B,C itself; the first workgroup also adds
 * dgamma / dbeta. Same arithmetic as hrnet_bn_bwd_finalize, deterministic, one launch and one dependency less per
 * BatchNorm. `ref` is a HOST struct copied into the launch; NULL = use `coef`. Keep nrows * Cout <= 16384
 * (3x3) / 8192 (1x1): every workgroup reads all rows.
 */
typedef struct HrBnBwdRef {
  const float* rows;    /* [nrows][2][Cout]: (sum dz, sum dz*y) partials */
  const float* gamma;
  const float* save_mean;
  const float* save_invstd;
  float* dgamma;        /* += (accumulate) or = */
  float* dbeta;
  float count;          /* elements per channel */
  int32_t nrows, accumulate, reserved;
} HrBnBwdRef;
int hrnet_conv3x3_bwd_fused_bnref(int dtype, const void* dz, const void* y, const float* coef, const HrBnBwdRef* ref,
                                  const void* x, const float* in_scale, const float* in_shift, int in_relu,
                                  const void* wT, void* dx, const void* addend, int mask_out, float* rows,
                                  const void* bs_y, float* slabs, int N, int H, int W, int Cin, int Cout,
                                  hr_stream_t stream);
int hrnet_bwd_fused_splits(int dtype, int N, int H, int W, int Cin, int Cout);
int hrnet_bwd_fused_kernel_name(int dtype, int Cin, int Cout, char* buf, int buflen);

/*
 * The same fused backward for a 1x1 (pointwise) convolution: the conv1 / conv3 / downsample layers of a
 * Bottleneck (autograd of pose_hrnet.py:60-105). Pixels are a flat index: dz, y [pixels,Cout]; x, dx, addend,
 * bs_y [pixels,Cin]; wT = hrnet_pack_weights(mode 1) of the 1x1 kernel ([Cin][Cout]); slabs
 * [hrnet_bwd_pw_splits()][Cout][Cin] f32 (sum with hrnet_wgrad_reduce, ks = 1). bf16 only; served shapes
 * (Cin,Cout) = (64,256), (256,64), (64,64): hrnet_bwd_pw_supported(). `rows` (the next BatchNorm's backward sums,
 * [splits][2][Cin]) where hrnet_bwd_pw_rows_supported() (every served shape).
 */
int hrnet_conv1x1_bwd_fused(int dtype, const void* dz, const void* y, const float* coef, const void* x,
                            const float* in_scale, const float* in_shift, int in_relu, const void* wT, void* dx,
                            const void* addend, int mask_out, float* rows, const void* bs_y, float* slabs,
                            long long pixels, int Cin, int Cout, hr_stream_t stream);
int hrnet_conv1x1_bwd_fused_bnref(int dtype, const void* dz, const void* y, const float* coef, const HrBnBwdRef* ref,
                                  const void* x, const float* in_scale, const float* in_shift, int in_relu,
                                  const void* wT, void* dx, const void* addend, int mask_out, float* rows,
                                  const void* bs_y, float* slabs, long long pixels, int Cin, int Cout,
                                  hr_stream_t stream);
int hrnet_bwd_pw_supported(int dtype, int Cin, int Cout);
int hrnet_bwd_pw_rows_supported(int dtype, int Cin, int Cout);
int hrnet_bwd_pw_splits(int dtype, long long pixels, int Cin, int Cout);
int hrnet_bwd_pw_kernel_name(int dtype, int Cin, int Cout, char* buf, int buflen);

/*
 * Pack f32 OIHW master weights into the kernels' layout.
 *   mode 0: forward  [Cout_pad][ks*ks][Cin_pad]
 *   mode 1: dgrad    [Cin_pad][ks*ks flipped][Cout_pad]  (transposed conv)
 *   mode 2: stem     [Cout_pad][Cin_pad] with k = (r*ks+s)*Cin_real + ci  (im2col order)
 */
int hrnet_pack_weights(int dtype, const float* w_oihw, void* packed, int Cout, int Cin, int ks,
                       int Cout_pad, int Cin_pad, int mode, hr_stream_t stream);
/* The same for every convolution of a network in ONE launch: `table` is a DEVICE array of n
 * entries; entry e covers blocks [block0, block0 + hrnet_pack_blocks(Cout_pad, Cin_pad, ks, mode));
 * total_blocks = their sum. A block stages a master row (mode 0: Cin*ks*ks floats) or 4/2/1 input channels of
 * every output channel (mode 1: Cout_pad * (n*ks*ks + 1) floats) in 19 KB of LDS: Cin*ks*ks and
 * Cout_pad*(ks*ks+1) must not exceed 4864. */
typedef struct HrPackEnt {
  const void* w;   /* f32 OIHW master weights */
  void* out;       /* packed weights */
  int32_t Cout, Cin, ks, Cout_pad, Cin_pad, mode, block0;
  int32_t ld;      /* mode 0: floats between consecutive output-channel rows of w (a column slice of a wider 1x1
                      weight); 0 = dense OIHW */
} HrPackEnt;
int hrnet_pack_weights_table(int dtype, const HrPackEnt* table, int n, int total_blocks,
                             hr_stream_t stream);
int hrnet_pack_blocks(int Cout_pad, int Cin_pad, int ks, int mode);

/* hrnet_wgrad_reduce for many convolutions in ONE launch (each layer keeps its own slab region):
 * `table` is a DEVICE array of n entries; entry e covers blocks [block0, block0 +
 * ceil(Cout*Cin*ks*ks / 64)); total_blocks = their sum. Same element order and summation tree as
 * hrnet_wgrad_reduce, so the result is bit-identical to the per-layer call. */
typedef struct HrWredEnt {
  const float* slabs; /* [nsplit][Cout_pad][taps][Cin_pad] (kflat: [nsplit][Cout_pad][Cin_pad]) */
  float* grad;        /* OIHW f32 [Cout][Cin][ks][ks] */
  int32_t nsplit, Cout_pad, Cin_pad, ks, Cout, Cin, kflat, accumulate, block0;
  int32_t ld;         /* floats between consecutive output-channel rows of grad (column slice); 0 = dense */
} HrWredEnt;
int hrnet_wgrad_reduce_table(const HrWredEnt* table, int n, int total_blocks, hr_stream_t stream);

/*
 * BatchNorm2d statistics -> per-channel affine (nn.BatchNorm2d in pose_hrnet.py:34,37,66-73,
 * :205,:223,:285-288,:340; momentum 0.1, eps 1e-5).
 *   training=1: mean/var from `stats` ([tiles][2][C], `count` elements per channel);
 *               running_mean/var updated in place (unbiased var), num_batches_tracked += 1
 *   training=0: scale/shift from the running statistics.
 *   scale = gamma*invstd, shift = beta - mean*scale; save_mean/save_invstd kept for backward.
 */
int hrnet_bn_finalize(const float* stats, int tiles, int C, float count, const float* gamma,
                      const float* beta, float* running_mean, float* running_var,
                      int64_t* num_batches_tracked, float momentum, float eps, int training,
                      float* scale, float* shift, float* save_mean, float* save_invstd,
                      hr_stream_t stream);

/*
 * Consumer-side BatchNorm: instead of a finalize launch after every conv, a producer accumulates its batch sums
 * into sums[8][2][C] with float atomics (hrnet_conv2d_bnref: out_sums, zeroed by the caller before the pass) and
 * every forward consumer turns sums + gamma/beta into scale/shift itself (one table per workgroup). ONE
 * hrnet_bn_finalize_table launch at the end of the pass fills, for every BatchNorm of the table, the arrays the
 * backward pass reads (scale, shift, save_mean, save_invstd) and updates the running statistics - with the same
 * arithmetic, so both passes see identical coefficients. Atomic accumulation makes the last bits of the batch
 * statistics run-to-run dependent (like cuDNN's); hrnet_conv2d + hrnet_bn_finalize remain the deterministic form.
 *   hrnet_conv2d_bnref: hrnet_conv2d with the input affine given as (in_sums, in_gamma, in_beta, 1/count, eps)
 *   (NULL in_sums: no input BatchNorm) and the output statistics added into out_sums (NULL: none).
 */
int hrnet_conv2d_bnref(int dtype, const void* x, const void* w, const float* in_sums, const float* in_gamma,
                       const float* in_beta, float in_inv_count, float in_eps, const float* bias, void* y,
                       float* out_sums, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int ks, int stride,
                       int in_relu, hr_stream_t stream);
/*
 * Forward conv (3x3 or 1x1, stride 1) whose input is the residual sum that closes the previous block, formed in
 * the conv's own prologue: a = relu(bn(x) + x2) with bn given as scale/shift arrays OR as batch sums + gamma/beta
 * (as hrnet_conv2d_bnref), y = conv(a) (+ statistics of y: rows, or sums[8][2][Cout] when stats_atomic), and `side`
 * = a, written once per pixel - the tensor the next residual add and the backward pass read. Replaces
 * hrnet_sum_terms(relu(bn(x) + x2)) followed by hrnet_conv2d: one launch and one tensor read less
 * (pose_hrnet.py:54-55 + :44 of the next BasicBlock; :95-96 + :79 for Bottlenecks).
 *   x, x2, side [N,H,W,Cin]; y [N,H,W,Cout]; stats may be NULL (eval mode).
 */
int hrnet_conv2d_sum(int dtype, const void* x, const void* x2, const void* w, const float* in_scale,
                     const float* in_shift, const float* in_sums, const float* in_gamma, const float* in_beta,
                     float in_inv_count, float in_eps, void* side, void* y, float* stats, int stats_atomic, int N,
                     int H, int W, int Cin, int Cout, int ks, hr_stream_t stream);

typedef struct HrBnEnt {
  const float* sums;   /* [8][2][C]; NULL: eval mode - scale/shift from running_mean/var, nothing else written */
  const float* gamma;
  const float* beta;
  float* running_mean; /* may be NULL */
  float* running_var;
  int64_t* num_batches_tracked; /* may be NULL */
  float* scale;
  float* shift;
  float* mean;
  float* invstd;
  float count, momentum, eps;
  int32_t C, block0, reserved;
} HrBnEnt;
/* `table`: DEVICE array of n entries; entry e covers blocks [block0, block0 + ceil(C/256)); total_blocks = sum */
int hrnet_bn_finalize_table(const HrBnEnt* table, int n, int total_blocks, hr_stream_t stream);
/* hrnet_sum_terms with BatchNorm terms given as batch sums: bit t of sums_mode set -> scale[t] = sums[8][2][C],
 * shift[t] = gamma (beta = gamma + C), inv_counts[t] = 1 / (elements per channel of term t) */
int hrnet_sum_terms_bnref(int dtype, void* out, int N, int Ho, int Wo, int C, int nterms, const void* const* src,
                          const float* const* scale, const float* const* shift, const int* shifts, const int* relus,
                          int relu_out, int sums_mode, const float* inv_counts, float eps, hr_stream_t stream);

/*
 * out[N,Ho,Wo,C] = relu_out( sum_{t<nterms} relu_t( src_t[nearest-up by 2^sh_t] * scale_t + shift_t ) )
 * Residual adds (pose_hrnet.py:54-55, :95-96) and fuse-layer sums with nearest upsampling
 * (pose_hrnet.py:257-264, :206). scale_t may be NULL (identity term).
 */
int hrnet_sum_terms(int dtype, void* out, int N, int Ho, int Wo, int C, int nterms,
                    const void* const* src, const float* const* scale, const float* const* shift,
                    const int* shifts, const int* relus, int relu_out, hr_stream_t stream);

/*
 * Backward of one term of hrnet_sum_terms / of a BN(+ReLU) that feeds a conv:
 *   dz[q,c]  = sum_{p in 2^sh x 2^sh block of q} g[p,c] * [mask_out[p,c] > 0] * [scale*y+shift > 0 if inner_relu]
 *   dst[q,c] (+)= A[c]*dz + B[c]*y[q,c] + C[c]          (coef = [3][C] from hrnet_bn_bwd_finalize;
 *                                                        coef NULL: dst (+)= dz)
 * g, mask_out: [N,H<<sh,W<<sh,C]; y, dst: [N,H,W,C].
 */
int hrnet_grad_term(int dtype, void* dst, const void* g, const void* mask_out, const void* y,
                    const float* scale, const float* shift, const float* coef, int N, int H, int W,
                    int C, int sh, int inner_relu, int accumulate, hr_stream_t stream);
/* same with sh = 0 and no inner ReLU, plus a second destination dst2 (+)= dz: the BatchNorm term and
 * the residual/identity term of one sum (pose_hrnet.py:54-55) share dz and are written in one pass */
int hrnet_grad_term2(int dtype, void* dst, void* dst2, const void* g, const void* mask_out, const void* y,
                     const float* scale, const float* shift, const float* coef, int N, int H, int W, int C,
                     int accumulate, int accumulate2, hr_stream_t stream);
/* partials[blocks][2][C]: sum(dz), sum(dz*y) with dz as above; blocks = hrnet_reduce_blocks() */
int hrnet_bn_bwd_reduce(int dtype, float* partials, const void* g, const void* mask_out,
                        const void* y, const float* scale, const float* shift, int N, int H, int W,
                        int C, int sh, int inner_relu, hr_stream_t stream);
int hrnet_reduce_blocks(int N, int H, int W, int C);
/* dgamma/dbeta (+=) and coef[3][C]: A = gamma*invstd, B = -gamma*invstd^2*mean(dz*xhat)/... */
int hrnet_bn_bwd_finalize(const float* partials, int blocks, int C, float count,
                          const float* gamma, const float* save_mean, const float* save_invstd,
                          float* dgamma, float* dbeta, float* coef, int accumulate,
                          hr_stream_t stream);

/*
 * Head input: cat[N,H,W,sum(C_j)] = [x0, bilinear(x1), bilinear(x2), bilinear(x3)]
 * (F.upsample(mode='bilinear'), align_corners=False, then torch.cat: pose_hrnet.py:560-565;
 * align_corners=1: F.interpolate(..., align_corners=True) of pose_hrnet_softmax.py:499-503).
 * Branch j has spatial size (H>>j, W>>j) ... given explicitly. nbr <= 4.
 */
int hrnet_bilinear_cat(int dtype, void* cat, const void* const* xs, const int* hs, const int* ws,
                       const int* cs, int nbr, int N, int H, int W, int align_corners,
                       hr_stream_t stream);
/* dxs[j] (+)= bilinear^T(dcat[..., slice_j]) (gather form, deterministic) */
int hrnet_bilinear_cat_bwd(int dtype, const void* dcat, void* const* dxs, const int* hs,
                           const int* ws, const int* cs, int nbr, int N, int H, int W,
                           int align_corners, int accumulate, hr_stream_t stream);

/*
 * The head without its concat (pose_hrnet.py:560-566: F.upsample x3, torch.cat, last_layer[0] = Conv2d(480, 480, 1)).
 * A 1x1 convolution commutes with bilinear upsampling, so
 *     last_layer[0](cat(x0, up(x1), up(x2), up(x3))) = W0 x0 + up(W1 x1) + up(W2 x2) + up(W3 x3) + bias,
 * W_j = the columns of the weight that multiply branch j. The products t_j = W_j x_j are plain hrnet_conv2d launches
 * at branch j's resolution; hrnet_head_mix forms W0 x0 on the full-resolution grid, adds the bias and the bilinear
 * upsampling of t_1..t_nup (align_corners as hrnet_bilinear_cat), stores y once and gathers its batch statistics:
 *   rows_mode 0: stats = sums[8][2][Cout] (float atomics, as hrnet_conv2d with stats_atomic)
 *   rows_mode 1: stats = rows[hrnet_head_mix_rows(N,H,W)][2][Cout], one row per workgroup (deterministic)
 * bf16 (MFMA; C0 a multiple of 16, <= 128) or f32 (plain FMAs: the validation path), Cout <= 512 (hrnet_head_mix_supported).
 * w0: hrnet_pack_weights layout [Cout][C0].
 */
int hrnet_head_mix(int dtype, const void* x0, const void* w0, const float* bias, void* y, float* stats, int rows_mode,
                   const void* const* ts, const int* hs, const int* ws, int nup, int N, int H, int W, int C0, int Cout,
                   int align_corners, hr_stream_t stream);
int hrnet_head_mix_rows(int N, int H, int W);
int hrnet_head_mix_supported(int dtype, int C0, int Cout);
/* Backward of the 1x1 layer BEHIND the head's BatchNorm + ReLU (last_layer[3] of pose_hrnet.py:341-346) fused with that
 * BatchNorm's backward (autograd of nn.Conv2d / nn.ReLU / nn.BatchNorm2d): dz = wT dY (K = the layer's padded output
 * channels) is formed per pixel tile and masked by [bn_scale*y + bn_shift > 0] (inner_relu), never stored.
 *   mode 1: out = rows[hrnet_head_mix_rows(N,H,W)][2][Cout] f32, (sum dz, sum dz*y) per pixel tile (deterministic)
 *   mode 2: out = G[N,H,W,Cout] = A*dz + B*y + C with coef = [3][Cout] from hrnet_bn_bwd_finalize on those rows
 * wT: hrnet_pack_weights mode 1 of the layer's weight ([Cout][K]). Shapes as hrnet_head_mix_supported(dtype, K, Cout). */
int hrnet_head_bwd(int dtype, int mode, const void* dy, const void* wT, const void* y, void* out, const float* bn_scale,
                   const float* bn_shift, const float* coef, int inner_relu, int N, int H, int W, int K, int Cout,
                   hr_stream_t stream);
/* outs[k][N,hs[k],ws[k],C] = bilinear^T(G[N,H,W,C]) over ALL channels, k < nout <= 3: the gradients of t_j above
 * (autograd of F.upsample, pose_hrnet.py:561-563). Deterministic. Integer scales 2 / 4 / 8 with align_corners=0 take
 * ONE pass over G for all outputs (LDS-staged tiles); anything else (or streamed=1) a separable streamed walk per output. */
int hrnet_upsample_bilinear_t(int dtype, const void* g, void* const* outs, const int* hs, const int* ws, int nout,
                              int N, int H, int W, int C, int align_corners, int streamed, hr_stream_t stream);

/* stem: NCHW f32 image -> im2col rows [N,Ho,Wo,Kpad] (k = (r*3+s)*C + c), 3x3 stride 2 pad 1
 * (conv1, pose_hrnet.py:283-284,512). */
int hrnet_im2col_stem(int dtype, const float* img_nchw, void* cols, int N, int C, int H, int W,
                      int Ho, int Wo, int Kpad, hr_stream_t stream);
/* [N,H,W,Cp] dtype -> [N,C,H,W] f32 (first C channels), and back (pad channels zeroed). */
int hrnet_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int H, int W, int Cp, int C,
                       hr_stream_t stream);
int hrnet_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int H, int W, int Cp, int C,
                       hr_stream_t stream);
/* dbias[c] (+)= sum over pixels of dy[pixels, Cp] for c < C (conv bias, pose_hrnet.py:334-347);
 * scratch: hrnet_reduce_blocks(1,1,pixels,Cp) * Cp floats */
int hrnet_bias_grad(int dtype, const void* dy, float* dbias, float* scratch, int pixels, int Cp,
                    int C, int accumulate, hr_stream_t stream);
int hrnet_fill_zero(void* p, int64_t bytes, hr_stream_t stream);

/*
 * HeatmapLoss (lib/core/loss.py:19-28): loss = mean_{b,k} sum_{h,w} (pred-gt)^2 (mode 0)
 * or |pred-gt| (mode 1). pred/gt are NCHW f32 [B,K,H,W] (the module contract).
 * partial: [B*K] f32 scratch; loss: [1] f32.
 */
int hrnet_heatmap_loss_fwd(const float* pred, const float* gt, float* partial, float* loss, int BK,
                           int HW, int mode, hr_stream_t stream);
/* dpred = gout * d loss / d pred */
int hrnet_heatmap_loss_bwd(const float* pred, const float* gt, const float* gout, float* dpred,
                           int BK, int HW, int mode, hr_stream_t stream);

/*
 * get_final_preds (lib/utils/heatmap_decoding.py:87-107), hms NCHW f32 [B,K,H,W] -> preds [B,K,2].
 *   expectation (use_softmax=True): (sum x*h, sum y*h), pixel coordinates, no normalisation
 *   argmax (use_softmax=False): first maximal flat index; u = idx % H, v = idx / H (H as the
 *   reference uses shape[2] for both)
 * maxvals (optional, [B,K]) receives the maximum (get_max_preds, lib/core/inference.py:18-46,
 * which uses W for % and / and zeroes preds whose max <= 0: flag `inference_style`).
 */
int hrnet_decode_expectation(const float* hms, float* preds, int BK, int H, int W,
                             hr_stream_t stream);
int hrnet_decode_expectation_bwd(const float* gpreds, float* dhms, int BK, int H, int W,
                                 int accumulate, hr_stream_t stream);
int hrnet_decode_argmax(const float* hms, float* preds, float* maxvals, int BK, int H, int W,
                        int inference_style, hr_stream_t stream);

/*
 * The data step either side of the path, on the device instead of the host loader:
 *   hrnet_gaussian_targets: lib/dataset/target_generators/target_generators.py:14-53 - un-normalised
 *     Gaussians, peak 1 at int(coord), window 6*sigma+3, zero map if invisible (visibility may be
 *     NULL = all visible) or out of range. pose2d [BK,2] heat-map pixels (x,y), heatmaps [BK,H,W].
 *   hrnet_normalize_u8: ToTensor + Normalize (lib/dataset/transforms/build.py:84-85): HWC u8 image
 *     -> CHW f32 (v/255 - mean[c]) / std[c]; mean3/std3 are HOST arrays of 3 floats.
 */
int hrnet_gaussian_targets(const float* pose2d, const float* visibility, float* heatmaps, int BK, int H,
                           int W, float sigma, hr_stream_t stream);
int hrnet_normalize_u8(const unsigned char* img_nhwc, float* out_nchw, int N, int H, int W,
                       const float* mean3, const float* std3, hr_stream_t stream);

/*
 * Spatial softmax head of pose_hrnet_softmax (lib/models/pose_hrnet_softmax.py:520-524):
 * out[bk, :] = softmax(x[bk, :] * *temp) over the HW positions of each map, NCHW f32.
 * backward: dx = temp * out * (gout - sum(gout*out)); dtemp_partial[bk] = sum_i dz_i * x_i with
 * dz = out * (gout - sum(gout*out)) (the caller sums the BK partials: d loss / d temperature).
 */
int hrnet_spatial_softmax_fwd(const float* x, const float* temp, float* out, int BK, int HW,
                              hr_stream_t stream);
int hrnet_spatial_softmax_bwd(const float* x, const float* out, const float* gout, const float* temp,
                              float* dx, float* dtemp_partial, int BK, int HW, hr_stream_t stream);

/*
 * JointsMSELoss (lib/core/loss.py:37-50): sum_bk ||pred-gt||_2 * vis / max(1, sum vis), or
 * sum/K without visibility (vis NULL). pred/gt [B,K,2] f32, vis [B,K] f32.
 */
int hrnet_joints_loss_fwd(const float* pred, const float* gt, const float* vis, float* loss, int B,
                          int K, hr_stream_t stream);
int hrnet_joints_loss_bwd(const float* pred, const float* gt, const float* vis, const float* gout,
                          float* dpred, int B, int K, hr_stream_t stream);

/* Adam step over a flat f32 parameter buffer (torch.optim.Adam semantics incl. L2 weight
 * decay added to the gradient; lib/utils/utils.py:81-85, lib/core/function.py:101-106). */
int hrnet_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                    float grad_scale, hr_stream_t stream);

/*
 * Deformable convolution v1 (lib/deformable_conv/functions/deform_conv_func.py:18-66, which binds
 * DCN.deform_conv_forward / deform_conv_backward of src/cuda/deform_conv_cuda.cu:19,139; sampling
 * rule src/cuda/deform_im2col_cuda.cuh:24-189). All tensors NCHW f32:
 *   input [B,C,H,W], offset [B, deformable_groups*2*kh*kw, Ho, Wo] (per tap: dy plane, dx plane),
 *   weight [Co, C/groups, kh, kw], bias [Co] or NULL, output [B,Co,Ho,Wo].
 * No column buffer is materialised, so the reference's im2col_step has no counterpart here (any
 * value gives the same result - the reference's own invariant, test.py:218-248).
 * backward: grad_input is accumulated with float atomics like the reference's col2im (zeroed
 * inside) - except on the PoseAggr geometry (one input channel per deformable group, groups = 1, 3x3,
 * Co <= 28, planes that fit LDS), whose single-pass kernel sums it in a 64-bit fixed-point plane: the
 * exact sum of the f32 contributions, the same bits on every run; grad_bias may be NULL; scratch holds hrnet_deform_conv_wgrad_blocks() *
 * (Co/groups)*(C/groups)*kh*kw floats. Limits: Co/groups <= 64 for backward, weight slice of one
 * group <= 96 KB.
 */
int hrnet_deform_conv_forward(const float* input, const float* offset, const float* weight,
                              const float* bias, float* output, int B, int C, int H, int W, int Co,
                              int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                              int groups, int deformable_groups, hr_stream_t stream);
int hrnet_deform_conv_wgrad_blocks(int B, int Ho, int Wo);
int hrnet_deform_conv_backward(const float* input, const float* offset, const float* weight,
                               const float* grad_output, float* grad_input, float* grad_offset,
                               float* grad_weight, float* grad_bias, float* scratch, int B, int C,
                               int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw,
                               int dh, int dw, int groups, int deformable_groups,
                               hr_stream_t stream);

/* Modulated deformable convolution (DCNv2): the same op with every sample multiplied by
 * mask [B, deformable_groups*kh*kw, Ho, Wo]. Replaces DCN.modulated_deform_conv_forward / _backward bound by
 * lib/deformable_conv/functions/modulated_deform_conv_func.py:25-33,44-56 (src/modulated_deform_conv.h:10-86,
 * src/cuda/modulated_deform_conv_cuda.cu:20-285, kernels src/cuda/modulated_deform_im2col_cuda.cuh:128-330).
 * backward: grad_input is zeroed inside; grad_mask like mask; scratch as for hrnet_deform_conv_backward. */
int hrnet_modulated_deform_conv_forward(const float* input, const float* offset, const float* mask,
                                        const float* weight, const float* bias, float* output, int B, int C,
                                        int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw,
                                        int dh, int dw, int groups, int deformable_groups, hr_stream_t stream);
int hrnet_modulated_deform_conv_backward(const float* input, const float* offset, const float* mask,
                                         const float* weight, const float* grad_output, float* grad_input,
                                         float* grad_offset, float* grad_mask, float* grad_weight,
                                         float* grad_bias, float* scratch, int B, int C, int H, int W, int Co,
                                         int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                                         int groups, int deformable_groups, hr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HRNET_HIP_H */
