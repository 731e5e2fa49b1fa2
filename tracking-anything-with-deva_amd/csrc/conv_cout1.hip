// Single-output-channel convolutions (mask logit head 256->1, shrinkage head 512->1, CBAM 7x7 gate
// 2->1): on the matrix cores 31 of 32 output rows of the smallest MFMA tile would be padding, so these
// run as a coalesced VALU dot product instead -- HBM/L2-bound, a few tens of microseconds.
//
// The weights of the one output channel are cached in LDS; each thread keeps 8 independent input
// loads in flight (a serial load->FMA chain is L2-latency-bound); partial sums meet in LDS.
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

// PX pixels x CG channel groups per 256-thread block: wide frames use 64 x 4, small frames 16 x 16 so
// that a 30x54 feature map still spreads over ~100 workgroups
template <int PX, int CG>
__global__ __launch_bounds__(256) void conv_cout1_kernel(const Cout1Args p) {
  static_assert(PX * CG == 256, "256 threads");
  extern __shared__ float smem[];  // [K] weights of the single output channel, then [CG][PX] partials
  const int K = p.KH * p.KW * p.ctot;
  float* wsm = smem;
  float* red = smem + ((K + 63) & ~63);
  for (int k = threadIdx.x; k < K; k += 256) wsm[k] = p.w[(int64_t)k * p.cout_pad];
  __syncthreads();

  const int px = threadIdx.x % PX;
  const int cg = threadIdx.x / PX;
  const int n = blockIdx.x * PX + px;
  const bool n_ok = n < p.n_total;
  const int nn = n_ok ? n : 0;
  const int b = nn / p.OHW;
  const int pix = nn - b * p.OHW;
  const int oh = pix / p.OW, ow = pix - oh * p.OW;
  const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
  const float* src0 = p.in0 + (int64_t)b * p.bs0;
  const float* src1 = p.in1 ? p.in1 + (int64_t)b * p.bs1 : p.in0;
  const int taps = p.KH * p.KW;
  constexpr int U = 8;  // channels in flight per thread
  float acc = 0.0f;
  for (int tap = 0; tap < taps; ++tap) {
    const int dy = tap / p.KW;
    const int ih = ih0 + dy, iw = iw0 + (tap - dy * p.KW);
    const bool ok = n_ok && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
    const int off = ok ? (ih * p.W + iw) : 0;
    // channels cg, cg+CG, cg+2CG, ...: U independent loads are issued before the FMAs consume them
    for (int c0 = cg; c0 < p.ctot; c0 += CG * U) {
      float v[U], wv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * CG;
        const bool in = c < p.ctot;
        const int cc = in ? c : 0;
        const float* sp = (cc < p.c0) ? (src0 + (int64_t)cc * p.HW) : (src1 + (int64_t)(cc - p.c0) * p.HW);
        v[u] = sp[off];
        const int k = (p.k_layout == DEVA_KLAYOUT_CHUNK32) ? (((cc >> 5) * taps + tap) * 32 + (cc & 31))
                                                          : (tap * p.ctot + cc);
        wv[u] = (in && ok) ? wsm[k] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float x = p.relu_in ? fmaxf(v[u], 0.0f) : v[u];
        acc += wv[u] * x;
      }
    }
  }
  red[cg * PX + px] = acc;
  __syncthreads();
  if (cg == 0 && n_ok) {
    float v = 0.0f;
    for (int g = 0; g < CG; ++g) v += red[g * PX + px];
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
  const int K = a.KH * a.KW * a.ctot;
  const size_t smem = sizeof(float) * (((size_t)K + 63) / 64 * 64 + 256);
  hipLaunchKernelGGL((conv_cout1_kernel<16, 16>), dim3((unsigned)ceil_div(a.n_total, 16)), dim3(256), smem, st, a);
  return check_launch("deva_conv2d(cout=1)");
}

}  // namespace deva
