// Argument blocks shared by the convolution translation units (conv_igemm.hip: dispatch, generic implicit-GEMM
// kernel, split-K reduction; conv_mfma.hip: the lean-loop kernel generation; conv_cout1.hip: single-channel heads).
#pragma once
#include "common.h"

namespace deva {

// conv_cout1.hip
struct Cout1Args {
  const float* in0;
  const float* in1;
  int64_t bs0, bs1;
  int c0, ctot;
  int H, W, OH, OW, OHW;
  int64_t HW;
  const float* w;
  const float* bias;
  int cout_pad, k_layout;
  int KH, KW, stride, pad;
  int n_total;
  int relu_in;
  const float* res;
  int64_t res_bs;
  int act;
  float* out;
};
int launch_conv_cout1(const Cout1Args& a, hipStream_t st);
int launch_conv3x3_cout1_rows(const Cout1Args& a, hipStream_t st);

struct ConvArgs {
  const float* in0;
  const float* in1;
  int64_t bs0, bs1;  // batch strides (elements)
  int c0, c1, ctot;
  int H, W, OH, OW, OHW;
  int64_t HW;
  const float* w;
  const float* bias;
  int cout, cout_pad;
  int k_layout;
  int KH, KW, stride, pad;
  int K;        // KH*KW*ctot
  int n_total;  // batch*OH*OW
  int relu_in;
  const float* res;
  int64_t res_bs;
  int act;
  float* out;
  int vec_ok;        // inputs are guard-banded + 'same' stride-1 geometry: 4-pixel vector gathers allowed
  int vec_out;       // output and residual rows are 16-byte aligned, OHW % 4 == 0: output stage through LDS
  int tiles_n, tiles_m;
  int group_m;       // cout tiles per tile-order group (conv_epilogue.h: conv_tile_coords); <= 0: all of them
  int64_t ws_elems;
  int splits;        // split-K factor (gridDim.y); > 1 writes raw partial sums to ws
  int per_split;     // K steps per split
  float* ws;         // [splits][cout][n_total]
  int64_t in0_span, in1_span;  // elements from the first to one past the last element of each input
  const void* w16;   // conv_f16.hip: fp16 weights (DEVA_KLAYOUT_H8; hi / lo planes for the split kernels) or null
  int prec;          // conv_f16.hip: 1 = fp16 operands (amp), 2 = hi/lo split of both operands (fp32-accurate)
  float out_scale;   // split kernels: 2^-e of the weight scale, applied (exactly) to the accumulators
  int* flag;         // split kernels: set to 1 when an accumulator came out non-finite (an input beyond the fp16 range)
  const int* gate;   // non-null: the launch does its work only when *gate != 0 (the fp32 re-run behind a split launch)
  int ablate;        // `make PROBES=1` builds only (DEVA_SPLIT_ABLATE): timing runs with parts of the K loop switched off
};

// cout tiles per tile-order group (conv_epilogue.h: conv_tile_coords).  With C workgroups of an XCD (resident at a time,
// ~48, or all the XCD ever gets on a small grid) covering g cout tiles x C/g pixel tiles, that XCD's L2 pulls g weight
// tiles + C/g activation tiles; a weight tile is taps * BM / (BN * stride^2) times the bytes of an activation tile, so
// g ~ sqrt(C / that ratio).
inline int conv_group_m(int taps, int stride, int bm, int bn, int64_t tiles) {
  const float ratio = (float)taps * bm / ((float)bn * stride * stride);
  const float c = (float)(tiles >= 8 * 48 ? 48 : (tiles + 7) / 8);
  int g = 1;
  while ((g + 1) * (g + 1) * ratio <= c * 1.5f) ++g;  // largest g with g^2 <= 1.5 C / ratio
  return g;
}

// conv_igemm.hip: out = act(sum_s ws[s] + bias + residual) for a split-K launch (p.splits > 1)
int launch_splitk_reduce(const ConvArgs& p, hipStream_t st);
// conv_mfma.hip: lean-loop kernels for weights in the k-quad layout (DEVA_KLAYOUT_Q4)
int launch_conv_q4(const ConvArgs& a, hipStream_t st);
// conv_mfma.hip: the gated fp32 re-run behind a split launch as a persistent kernel (<= 1 024 workgroups whatever the
// layer's size); -1 = shape not covered, launch the regular kernels with the gate
int launch_conv_q4_gated(const ConvArgs& a, hipStream_t st);
// conv_wino.hip: Winograd F(2x2, 3x3) on the fp32 matrix pipes for 3x3 / stride 1 / pad 1 layers whose transformed weights the
// caller supplies (deva_conv_desc.weight_wino); -1 = shape not eligible, run the direct kernels
int launch_conv_wino(const ConvArgs& a, const float* u, hipStream_t st);
// conv_f16.hip: fp16-operand kernels (opt-in amp path, a.prec == 1) and the hi/lo split kernels (a.prec == 2: fp32-accurate
// on the f16 matrix pipes); -1 = shape not eligible, run the fp32 kernels
int launch_conv_f16(const ConvArgs& a, hipStream_t st);

}  // namespace deva
