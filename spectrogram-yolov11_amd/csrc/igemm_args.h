// igemm_args.h — argument block shared by the implicit-GEMM convolution kernels (igemm.hip, conv3x3.hip).
#pragma once
#include "common.h"
#include "bn_tail.h"

struct IgemmArgs {
  const void* x;
  const void* w;
  void* y;
  const float* bias;
  float* stat_sum;
  float* stat_sq;
  int M, N, K, C, T;
  int wK;                 // elements between consecutive filter rows (full T*C of the stored filter)
  int x_ld, y_ld;
  int IH, IW, OH, OW;
  int sy, sx;
  int OHF, OWF, oy_mul, oy_add, ox_mul, ox_add;
  int dense_out;          // 1: output offset is m*y_ld (no decode)
  int tiles_n;
  int chan_major;     // K loop order: 1 = taps innermost (needs C % stage == 0), 0 = channels innermost
  int stat_slots;
  int stat_stride;        // floats between consecutive stat slots (= channel count of the WHOLE stat row)
  unsigned x_bytes, w_bytes;   // extents of the x / w views in bytes (buffer descriptors range-check against them)
  int bpol;               // cache policy of the filter-row copies: 0 default, 1 sc1 (no L1 allocation), 2 nt
  int debug;              // diagnostic builds only: 1 = skip the LDS-DMA issue after the prologue, 2 = skip the MFMAs
  int vec_out;            // 1: output rows are 16-byte addressable -> LDS-transposed wide stores
  unsigned flags;
  BnTailDev tail;         // BN statistics finalised by the last workgroup (ticket == nullptr: off)
  signed char tap_dy[64];
  signed char tap_dx[64];
  signed char tap_w[64];
};


// halo-tiled 3x3 kernel (conv3x3.hip): bn = channels per workgroup (128 / 64 / 32), + 1000 for the 256-pixel tiles; returns SY11_OK or a negative status
bool sy11_halo3x3_legal(const IgemmArgs& a, int bn);
int sy11_halo3x3_launch(const IgemmArgs& a, int bn, hipStream_t st);
// stride-2 3x3 input gradient, all four parity classes in one pass (conv3x3.hip); see the argument convention there
bool sy11_halo_dgrad_s2_legal(const IgemmArgs& a);
int sy11_halo_dgrad_s2_launch(const IgemmArgs& a, hipStream_t st);
// few-channel 3x3 stride-1 kernel (conv3x3s.hip): C = 16 / 32, N <= 32, filters resident in registers; dtype = SY11_F16 / SY11_BF16
bool sy11_smallc3x3_legal(const IgemmArgs& a);
int sy11_smallc3x3_launch(const IgemmArgs& a, int dtype, hipStream_t st);
