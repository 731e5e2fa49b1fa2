// Tile -> output stage shared by the MFMA convolution kernels (conv_mfma.hip: fp32 operands, conv_f16.hip: fp16
// operands): the accumulators of v_mfma_f32_32x32x* (lane l, register r: row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31)
// go to the split-K workspace as raw partial sums, or through bias + residual + activation to the NCHW output
// (32 consecutive pixels per half-wave and output channel).
#pragma once
#include "conv_args.h"

namespace deva {

typedef float conv_f32x16 __attribute__((ext_vector_type(16)));

// Workgroup b -> (cout tile, pixel tile).  Workgroup b runs on XCD b % 8; XCD x gets a contiguous range of LOGICAL tiles,
// and logical tiles are ordered in groups of `group_m` cout tiles, cout fastest inside a group, pixel tiles next, groups
// last.  The ~32-64 workgroups resident on one XCD then cover group_m cout tiles x ~(32..64)/group_m neighbouring pixel
// tiles, and that rectangle is what streams through the XCD's 4 MB L2 per K step: group_m weight tiles + the pixel tiles'
// activations.  A 3x3 weight tile is ~9x the bytes of a pixel tile's activations, so few cout tiles per group there
// (measured on the 1024->1536 GRU convolution: L2 fills 1.36 GB -> see profiles/r04b), all cout tiles for 1x1.
__device__ __forceinline__ void conv_tile_coords(const ConvArgs& p, int b, int nb, int& tile_m, int& tile_n) {
  const int q = nb >> 3, r = nb & 7, xcd = b & 7;
  const int logical = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  const int g = (p.group_m > 0 && p.group_m < p.tiles_m) ? p.group_m : p.tiles_m;
  const int per_group = g * p.tiles_n;
  const int grp = logical / per_group;
  const int in_grp = logical - grp * per_group;
  const int m_first = grp * g;
  const int gsz = min(g, p.tiles_m - m_first);
  tile_n = in_grp / gsz;
  tile_m = m_first + (in_grp - tile_n * gsz);
}

__device__ __forceinline__ void conv_tile_coords(const ConvArgs& p, int& tile_m, int& tile_n) {
  conv_tile_coords(p, (int)blockIdx.x, (int)gridDim.x, tile_m, tile_n);
}

typedef float conv_f32x4 __attribute__((ext_vector_type(4)));

// Output stage through LDS (p.vec_out: 16-byte aligned rows, OHW % 4 == 0; not for split-K partial sums): a 32x32
// accumulator block goes to the wave's own 4 KB of LDS as [row][pixel] and comes back as 4 consecutive pixels per lane (8 lanes = one 128-byte row segment),
// so the residual is read and the result written with 16-byte accesses -- a quarter of the memory instructions of the
// one-pixel-per-lane form below, same arithmetic in the same order.  `scr`: 1024 floats of LDS nobody else touches
// (the caller has put a barrier between the last tile reads and this call).
template <int TM, int TN>
__device__ __forceinline__ void conv_store_tile_vec(const ConvArgs& p, conv_f32x16 (&acc)[TM][TN], int m0w, int n0w, int lane,
                                                    float* scr) {
  const int l31 = lane & 31, half = lane >> 5;
  const int rrow = lane >> 3, rcol = (lane & 7) * 4;
  // every residual vector and bias value of the wave's tile first (independent of the accumulators: all in flight at once)
  conv_f32x4 rv[TM][TN][4];
  float bv[TM][4];
  int64_t obase[TN];
  bool n_ok[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0w + j * 32 + rcol;
    n_ok[j] = n < p.n_total;
    const int nn = n_ok[j] ? n : 0;
    const int b = nn / p.OHW;
    const int pix = nn - b * p.OHW;
    obase[j] = (int64_t)b * p.cout * p.OHW + pix;
    const int64_t rbase = (int64_t)b * p.res_bs + pix;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0w + i * 32 + q * 8 + rrow;
        rv[i][j][q] = (p.res && n_ok[j] && m < p.cout) ? *reinterpret_cast<const conv_f32x4*>(p.res + rbase + (int64_t)m * p.OHW)
                                                                 : conv_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = m0w + i * 32 + q * 8 + rrow;
      bv[i][q] = (p.bias && m < p.cout) ? p.bias[m] : 0.0f;
    }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[i][j][r];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = q * 8 + rrow;
        const int m = m0w + i * 32 + row;
        conv_f32x4 v = *reinterpret_cast<const conv_f32x4*>(scr + row * 32 + rcol);
        if (!n_ok[j] || m >= p.cout) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float x = v[c];
          if (p.bias) x += bv[i][q];
          if (p.res) x += rv[i][j][q][c];
          if (p.act == DEVA_ACT_RELU) {
            x = fmaxf(x, 0.0f);
          } else if (p.act == DEVA_ACT_SIGMOID) {
            x = sigmoidf_(x);
          } else if (p.act == DEVA_ACT_SQUARE_PLUS_ONE) {
            x = x * x + 1.0f;
          }
          v[c] = x;
        }
        *reinterpret_cast<conv_f32x4*>(p.out + obase[j] + (int64_t)m * p.OHW) = v;
      }
    }
  }
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
