// LDS-ring convolution kernels (conv_ring.hip): what the dispatchers in conv.hip need
#pragma once
#include <hip/hip_runtime.h>

constexpr int HR_RING_MAXC = 384;   // input channels of the streamed-weight instantiations (w48: 384)

struct HrRingConv {
  const void* x;
  const void* w;
  void* y;
  const float* in_sums;   // [8][2][Cin] batch sums of the input's BatchNorm (then in_gb = gamma | beta), or NULL
  const float* in_gb;
  const float* in_scale;  // or scale / shift arrays, or neither (raw input)
  const float* in_shift;
  float* stats;           // forward: [8][2][Cout] batch sums of the output (float atomics) or NULL;
                          // backward statistics: rows [hr_conv_ring_rows()][2][Cout]
  const void* bs_y;       // backward-statistics operands (hrnet_conv2d_bwdstats), or NULL
  const void* bs_mask;
  const float* bs_scale;
  const float* bs_shift;
  // residual-sum launches (hrnet_conv2d_sum on the narrow instantiations): input a = relu(bn(x) + x2), written to side
  const void* x2;
  void* side;
  float in_inv_count, in_eps;
  int N, H, W, Cin, Cout, in_relu, accumulate, bs_store_masked;
};

int hr_conv_ring_enabled();
// instantiation id (> 0) if conv_ring serves a 3x3 stride-1 pad-1 launch of this shape, else 0.
// bs: an input-gradient launch (raw input) with backward statistics and / or accumulation into y
int hr_conv_ring_supported(int dtype, int N, int H, int W, int Cin, int Cout, int bs);
// 1 if the residual-sum form (HrRingConv.x2 / side) is served for this shape: the resident-weight instantiations (1, 2)
int hr_conv_ring_sum_supported(int dtype, int N, int H, int W, int Cin, int Cout);
// statistics rows a backward-statistics launch of this shape writes (= its pixel walks)
int hr_conv_ring_rows(int N, int H, int W, int Cin, int Cout);
int hr_conv_ring_launch(const HrRingConv& c, hipStream_t s);
// name of instantiation `id` as rocprofv3 prints it (returns its length)
int hr_conv_ring_name(int id, int bs, char* buf, int buflen);
