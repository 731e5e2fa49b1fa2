// Single-output-channel convolutions (mask logit head 256->1, shrinkage head 512->1, CBAM 7x7 gate
// 2->1): on the matrix cores 31 of 32 output rows of the smallest MFMA tile would be padding, so these
// run as a coalesced VALU dot product instead -- HBM/L2-bound, a few tens of microseconds.
//
// block = 64 consecutive output pixels x 4 channel groups; a wave reads 64 consecutive pixels of one
// channel (256 B, coalesced) per FMA step, the four partial sums meet in LDS.
#include "common.h"

namespace deva {

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

namespace {

__global__ __launch_bounds__(256) void conv_cout1_kernel(const Cout1Args p) {
  __shared__ float red[4][64];
  const int px = threadIdx.x & 63;
  const int cg = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + px;
  const bool n_ok = n < p.n_total;
  const int nn = n_ok ? n : 0;
  const int b = nn / p.OHW;
  const int pix = nn - b * p.OHW;
  const int oh = pix / p.OW, ow = pix - oh * p.OW;
  const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
  const float* src0 = p.in0 + (int64_t)b * p.bs0;
  const float* src1 = p.in1 ? p.in1 + (int64_t)b * p.bs1 : p.in0;
  const int per = (p.ctot + 3) / 4;
  const int c_lo = cg * per, c_hi = min(p.ctot, c_lo + per);
  const int taps = p.KH * p.KW;
  float acc = 0.0f;
  for (int tap = 0; tap < taps; ++tap) {
    const int dy = tap / p.KW;
    const int ih = ih0 + dy, iw = iw0 + (tap - dy * p.KW);
    const bool ok = n_ok && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
    const int off = ok ? (ih * p.W + iw) : 0;
    for (int c = c_lo; c < c_hi; ++c) {
      const int k = (p.k_layout == DEVA_KLAYOUT_CHUNK32) ? (((c >> 5) * taps + tap) * 32 + (c & 31))
                                                        : (tap * p.ctot + c);
      const float wv = p.w[(int64_t)k * p.cout_pad];
      const float* s = (c < p.c0) ? (src0 + (int64_t)c * p.HW) : (src1 + (int64_t)(c - p.c0) * p.HW);
      float v = s[off];
      if (p.relu_in) v = fmaxf(v, 0.0f);
      acc += ok ? wv * v : 0.0f;
    }
  }
  red[cg][px] = acc;
  __syncthreads();
  if (cg == 0 && n_ok) {
    float v = ((red[0][px] + red[1][px]) + red[2][px]) + red[3][px];
    if (p.bias) v += p.bias[0];
    if (p.res) v += p.res[(int64_t)b * p.res_bs + pix];
    if (p.act == DEVA_ACT_RELU) {
      v = fmaxf(v, 0.0f);
    } else if (p.act == DEVA_ACT_SIGMOID) {
      v = sigmoidf_(v);
    } else if (p.act == DEVA_ACT_SQUARE_PLUS_ONE) {
      v = v * v + 1.0f;
    }
    p.out[(int64_t)b * p.OHW + pix] = v;
  }
}

}  // namespace

int launch_conv_cout1(const Cout1Args& a, hipStream_t st) {
  hipLaunchKernelGGL(conv_cout1_kernel, dim3((unsigned)ceil_div(a.n_total, 64)), dim3(256), 0, st, a);
  return check_launch("deva_conv2d(cout=1)");
}

}  // namespace deva
