// Tile -> output stage shared by the MFMA convolution kernels (conv_mfma.hip: fp32 operands, conv_f16.hip: fp16
// operands): the accumulators of v_mfma_f32_32x32x* (lane l, register r: row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31)
// go to the split-K workspace as raw partial sums, or through bias + residual + activation to the NCHW output
// (32 consecutive pixels per half-wave and output channel).
#pragma once
#include "conv_args.h"

namespace deva {

typedef float conv_f32x16 __attribute__((ext_vector_type(16)));

// logical tile of workgroup b: cout tiles fastest, XCD x gets a contiguous range of logical tiles (workgroup b runs on
// XCD b % 8), so the workgroups sharing one pixel tile and neighbouring pixel tiles meet in one L2
__device__ __forceinline__ int conv_logical_tile() {
  const int nb = gridDim.x, b = blockIdx.x;
  const int q = nb >> 3, r = nb & 7, xcd = b & 7;
  return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

template <int TM, int TN>
__device__ __forceinline__ void conv_store_tile(const ConvArgs& p, conv_f32x16 (&acc)[TM][TN], int m0, int wm0, int n0, int wn0,
                                                int l31, int half) {
  if (p.splits > 1) {
    // ---- split-K: raw partial sums, reduced (+ bias / residual / activation) by splitk_reduce_kernel
    float* ws = p.ws + (int64_t)blockIdx.y * p.cout * p.n_total;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn0 + j * 32 + l31;
      if (n >= p.n_total) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (m < p.cout) ws[(int64_t)m * p.n_total + n] = acc[i][j][r];
        }
    }
    return;
  }

  // ---- epilogue: bias + residual + activation, NCHW store (32 consecutive pixels per half-wave)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn0 + j * 32 + l31;
    if (n >= p.n_total) continue;
    const int b = n / p.OHW;
    const int pix = n - b * p.OHW;
    const int64_t obase = (int64_t)b * p.cout * p.OHW + pix;
    const int64_t rbase = (int64_t)b * p.res_bs + pix;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float bv[16], rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int mm = (m < p.cout) ? m : 0;
        bv[r] = p.bias ? p.bias[mm] : 0.0f;
        rv[r] = p.res ? p.res[rbase + (int64_t)mm * p.OHW] : 0.0f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[i][j][r];
        if (p.bias) v += bv[r];
        if (p.res) v += rv[r];
        if (p.act == DEVA_ACT_RELU) {
          v = fmaxf(v, 0.0f);
        } else if (p.act == DEVA_ACT_SIGMOID) {
          v = sigmoidf_(v);
        } else if (p.act == DEVA_ACT_SQUARE_PLUS_ONE) {
          v = v * v + 1.0f;
        }
        if (m < p.cout) p.out[obase + (int64_t)m * p.OHW] = v;
      }
    }
  }
}

}  // namespace deva
