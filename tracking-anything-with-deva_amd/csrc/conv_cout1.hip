// Single-output-channel convolutions (mask logit head 256->1, shrinkage head 512->1, CBAM 7x7 gate
// 2->1): on the matrix cores 31 of 32 output rows of the smallest MFMA tile would be padding, so these
// run as a coalesced VALU dot product instead -- HBM/L2-bound, a few tens of microseconds.
//
// The weights of the one output channel and the decoded (channel, tap) of every reduction index are cached in LDS;
// each thread keeps 8-16 independent input loads in flight (a serial load->FMA chain is L2-latency-bound); partial
// sums meet in LDS.
#include "conv_args.h"

namespace deva {


namespace {

// PX pixels x CG k-groups per 256-thread block.  The reduction index k = (tap, channel) is FLAT: a table in LDS
// holds, per k, the decoded (channel, dy, dx) and the weight, thread (px, g) walks k = g, g + CG, ... with U loads in
// flight.  (The first version looped over taps and split only the channels: the 7x7 gate over 2 channels ran 49
// dependent load rounds on 2 of its 16 channel groups -- 46 us for 0.8 MFLOP.)
template <int PX, int CG, int U>
__global__ __launch_bounds__(256) void conv_cout1_kernel(const Cout1Args p) {
  static_assert(PX * CG == 256, "256 threads");
  extern __shared__ uint2 tab[];  // [K] {channel | dy << 16 | dx << 24, weight bits}, then [CG][PX] partials
  const int K = p.KH * p.KW * p.ctot;
  float* red = reinterpret_cast<float*>(tab + ((K + 63) & ~63));
  const int taps = p.KH * p.KW;
  for (int k = threadIdx.x; k < K; k += 256) {
    int tap, c;
    if (p.k_layout == DEVA_KLAYOUT_CHUNK32) {  // k = ((c/32)*taps + tap)*32 + c%32
      const int slab = k >> 5;
      const int chunk = slab / taps;
      tap = slab - chunk * taps;
      c = chunk * 32 + (k & 31);
    } else {  // k = tap*ctot + c
      tap = k / p.ctot;
      c = k - tap * p.ctot;
    }
    const int dy = tap / p.KW;
    tab[k] = make_uint2((unsigned)c | ((unsigned)dy << 16) | ((unsigned)(tap - dy * p.KW) << 24),
                        __float_as_uint(p.w[(int64_t)k * p.cout_pad]));
  }
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
  float acc = 0.0f;
  for (int k0 = cg; k0 < K; k0 += CG * U) {
    float v[U], wv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u * CG;
      const bool in = k < K;
      const uint2 e = tab[in ? k : 0];
      const int c = (int)(e.x & 0xffffu);
      const int ih = ih0 + (int)((e.x >> 16) & 0xffu), iw = iw0 + (int)(e.x >> 24);
      const bool ok = in && n_ok && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
      const float* sp = (c < p.c0) ? (src0 + (int64_t)c * p.HW) : (src1 + (int64_t)(c - p.c0) * p.HW);
      v[u] = sp[ok ? (ih * p.W + iw) : 0];
      wv[u] = ok ? __uint_as_float(e.y) : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float x = p.relu_in ? fmaxf(v[u], 0.0f) : v[u];
      acc += wv[u] * x;
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

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4_u __attribute__((aligned(4)));  // 4 consecutive pixels, dword-aligned only

// 3x3 / stride 1 / pad 1 on a large map (the 256->1 mask-logit head at 1/4 resolution: 0.6 GFLOP over
// 130 MB of input -- HBM-bound, 31/32 of an MFMA tile would be padding).  A thread owns 4 consecutive
// output pixels of a row; per channel and input row it reads the 6 pixels ow-1 .. ow+4 with two
// dword-aligned 16-B loads (guard-banded tensors, like the vector gather of conv_igemm) and applies the
// three taps of that row from registers: 6 loads per 36 MACs, 24 loads in flight.  The four waves of a
// block take every fourth channel (weights from LDS, broadcast reads: the channel is wave-uniform);
// partial sums meet in LDS.  Needs OW % 4 == 0 and readable guard bands of W + 8 floats.
__global__ __launch_bounds__(256) void conv3x3_cout1_rows_kernel(const Cout1Args p) {
  __shared__ f32x4 red[4][64];
  const int lane = threadIdx.x & 63;
  const int cg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // channel group = wave
  const int strips = p.n_total >> 2;
  // workgroup b runs on XCD b % 8: give every XCD a contiguous range of rows, so that the three workgroups that read
  // an input row (as their row above / own row / row below) meet in one L2 (measured: L2 fills 366 -> ~1.3x input MB)
  int blk;
  {
    const int nb = gridDim.x, bi = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bi & 7;
    blk = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bi >> 3);
  }
  const int sidx = blk * 64 + lane;
  const bool s_ok = sidx < strips;
  const int n = (s_ok ? sidx : 0) * 4;
  const int b = n / p.OHW;
  const int pix = n - b * p.OHW;
  const int oh = pix / p.OW, ow = pix - oh * p.OW;
  const float* src0 = p.in0 + (int64_t)b * p.bs0 + pix;
  const float* src1 = (p.in1 ? p.in1 + (int64_t)b * p.bs1 : p.in0) + pix;
  const bool left = ow > 0, right = ow + 4 < p.W;
  const bool up = oh > 0, down = oh + 1 < p.H;
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};

  // weights of the one output channel, [channel][tap] in LDS (uniform addresses: broadcast reads)
  extern __shared__ float wsm[];
  for (int i = threadIdx.x; i < p.ctot * 9; i += 256) {
    const int c = i / 9, t = i - c * 9;
    const int k = (p.k_layout == DEVA_KLAYOUT_CHUNK32) ? (((c >> 5) * 9 + t) * 32 + (c & 31)) : (t * p.ctot + c);
    wsm[i] = p.w[(int64_t)k * p.cout_pad];
  }
  __syncthreads();
  // out-of-image rows contribute nothing: their taps are zeroed, their (clamped) reads are harmless
  const float u = up ? 1.0f : 0.0f, d = down ? 1.0f : 0.0f;
  const int o_up = up ? -p.W : 0, o_dn = down ? p.W : 0;

  constexpr int U = 4;  // channels in flight per thread: 24 independent 16-B loads
  for (int c0 = cg; c0 < p.ctot; c0 += 4 * U) {  // wave-uniform channels cg, cg+4, ...
    f32x4 a[U][3], e[U][3];
    bool live[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int c = c0 + 4 * q;
      live[q] = c < p.ctot;
      const int cc = live[q] ? c : cg;
      const float* sp = (cc < p.c0) ? (src0 + (int64_t)cc * p.HW) : (src1 + (int64_t)(cc - p.c0) * p.HW);
      a[q][0] = *reinterpret_cast<const f32x4_u*>(sp + o_up - 1);  // ow-1 .. ow+2
      e[q][0] = *reinterpret_cast<const f32x4_u*>(sp + o_up + 3);  // ow+3 .. ow+6 (the last two are not used)
      a[q][1] = *reinterpret_cast<const f32x4_u*>(sp - 1);
      e[q][1] = *reinterpret_cast<const f32x4_u*>(sp + 3);
      a[q][2] = *reinterpret_cast<const f32x4_u*>(sp + o_dn - 1);
      e[q][2] = *reinterpret_cast<const f32x4_u*>(sp + o_dn + 3);
    }
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int c = live[q] ? c0 + 4 * q : cg;
      const float lv = live[q] ? 1.0f : 0.0f;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float rs = (r == 0 ? u : (r == 2 ? d : 1.0f)) * lv;
        const float w0 = wsm[c * 9 + 3 * r] * rs, w1 = wsm[c * 9 + 3 * r + 1] * rs, w2 = wsm[c * 9 + 3 * r + 2] * rs;
        f32x4 x = a[q][r], y = e[q][r];
        if (p.relu_in) {
#pragma unroll
          for (int i = 0; i < 4; ++i) x[i] = fmaxf(x[i], 0.0f);
          y[0] = fmaxf(y[0], 0.0f);
          y[1] = fmaxf(y[1], 0.0f);
        }
        const float x0 = left ? x[0] : 0.0f, x5 = right ? y[1] : 0.0f;
        acc[0] += w0 * x0 + w1 * x[1] + w2 * x[2];
        acc[1] += w0 * x[1] + w1 * x[2] + w2 * x[3];
        acc[2] += w0 * x[2] + w1 * x[3] + w2 * y[0];
        acc[3] += w0 * x[3] + w1 * y[0] + w2 * x5;
      }
    }
  }
  red[cg][lane] = acc;
  __syncthreads();
  if (cg == 0 && s_ok) {
    f32x4 v = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    const f32x4 r = p.res ? *reinterpret_cast<const f32x4*>(p.res + (int64_t)b * p.res_bs + pix) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float x = v[i];
      if (p.bias) x += p.bias[0];
      if (p.res) x += r[i];
      if (p.act == DEVA_ACT_RELU) {
        x = fmaxf(x, 0.0f);
      } else if (p.act == DEVA_ACT_SIGMOID) {
        x = sigmoidf_(x);
      } else if (p.act == DEVA_ACT_SQUARE_PLUS_ONE) {
        x = x * x + 1.0f;
      }
      v[i] = x;
    }
    *reinterpret_cast<f32x4*>(p.out + (int64_t)b * p.OHW + pix) = v;
  }
}

// The same on TWO output rows per thread (OH even): rows oh-1 .. oh+2 are read once for the outputs of rows oh and oh+1
// -- 8 loads per channel for two rows instead of 12, and every input row passes the L2 twice instead of three times (the
// kernel moves 3 TB/s of algorithmic bytes; what binds it is the L2 traffic behind them).  Per output the channel order and
// the tap order are those of the one-row kernel: bit-identical results.
__global__ __launch_bounds__(256) void conv3x3_cout1_rows2_kernel(const Cout1Args p) {
  __shared__ f32x4 red[4][2][64];
  const int lane = threadIdx.x & 63;
  const int cg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int owq = p.OW >> 2, oh2n = p.OH >> 1;
  const int strips = (p.n_total / p.OHW) * oh2n * owq;  // (batch, row pair, pixel quad)
  int blk;
  {
    const int nb = gridDim.x, bi = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bi & 7;
    blk = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bi >> 3);
  }
  const int sidx = blk * 64 + lane;
  const bool s_ok = sidx < strips;
  const int si = s_ok ? sidx : 0;
  const int b = si / (oh2n * owq);
  const int rem = si - b * (oh2n * owq);
  const int oh = 2 * (rem / owq), ow = 4 * (rem % owq);
  const int pix = oh * p.OW + ow;
  const float* src0 = p.in0 + (int64_t)b * p.bs0 + pix;
  const float* src1 = (p.in1 ? p.in1 + (int64_t)b * p.bs1 : p.in0) + pix;
  const bool left = ow > 0, right = ow + 4 < p.W;
  const bool up = oh > 0, down = oh + 2 < p.H;
  f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};

  extern __shared__ float wsm[];
  for (int i = threadIdx.x; i < p.ctot * 9; i += 256) {
    const int c = i / 9, t = i - c * 9;
    const int k = (p.k_layout == DEVA_KLAYOUT_CHUNK32) ? (((c >> 5) * 9 + t) * 32 + (c & 31)) : (t * p.ctot + c);
    wsm[i] = p.w[(int64_t)k * p.cout_pad];
  }
  __syncthreads();
  const float u = up ? 1.0f : 0.0f, d = down ? 1.0f : 0.0f;
  const int o_up = up ? -p.W : 0, o_dn = down ? 2 * p.W : p.W;  // (clamped reads of rows that do not exist: their taps are zeroed)

  constexpr int U = 3;  // channels in flight per thread: 24 independent 16-B loads
  for (int c0 = cg; c0 < p.ctot; c0 += 4 * U) {
    f32x4 a[U][4], e[U][4];
    bool live[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int c = c0 + 4 * q;
      live[q] = c < p.ctot;
      const int cc = live[q] ? c : cg;
      const float* sp = (cc < p.c0) ? (src0 + (int64_t)cc * p.HW) : (src1 + (int64_t)(cc - p.c0) * p.HW);
      const int offs[4] = {o_up, 0, p.W, o_dn};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a[q][r] = *reinterpret_cast<const f32x4_u*>(sp + offs[r] - 1);  // ow-1 .. ow+2
        e[q][r] = *reinterpret_cast<const f32x4_u*>(sp + offs[r] + 3);  // ow+3 .. ow+6 (the last two are not used)
      }
    }
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int c = live[q] ? c0 + 4 * q : cg;
      const float lv = live[q] ? 1.0f : 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f32x4 x = a[q][r], y = e[q][r];
        if (p.relu_in) {
#pragma unroll
          for (int i = 0; i < 4; ++i) x[i] = fmaxf(x[i], 0.0f);
          y[0] = fmaxf(y[0], 0.0f);
          y[1] = fmaxf(y[1], 0.0f);
        }
        const float x0 = left ? x[0] : 0.0f, x5 = right ? y[1] : 0.0f;
        if (r < 3) {  // tap row r of output row oh (its rows: oh-1, oh, oh+1)
          const float rs = (r == 0 ? u : 1.0f) * lv;
          const float w0 = wsm[c * 9 + 3 * r] * rs, w1 = wsm[c * 9 + 3 * r + 1] * rs, w2 = wsm[c * 9 + 3 * r + 2] * rs;
          acc0[0] += w0 * x0 + w1 * x[1] + w2 * x[2];
          acc0[1] += w0 * x[1] + w1 * x[2] + w2 * x[3];
          acc0[2] += w0 * x[2] + w1 * x[3] + w2 * y[0];
          acc0[3] += w0 * x[3] + w1 * y[0] + w2 * x5;
        }
        if (r > 0) {  // tap row r-1 of output row oh+1 (its rows: oh, oh+1, oh+2)
          const int t = r - 1;
          const float rs = (t == 2 ? d : 1.0f) * lv;
          const float w0 = wsm[c * 9 + 3 * t] * rs, w1 = wsm[c * 9 + 3 * t + 1] * rs, w2 = wsm[c * 9 + 3 * t + 2] * rs;
          acc1[0] += w0 * x0 + w1 * x[1] + w2 * x[2];
          acc1[1] += w0 * x[1] + w1 * x[2] + w2 * x[3];
          acc1[2] += w0 * x[2] + w1 * x[3] + w2 * y[0];
          acc1[3] += w0 * x[3] + w1 * y[0] + w2 * x5;
        }
      }
    }
  }
  red[cg][0][lane] = acc0;
  red[cg][1][lane] = acc1;
  __syncthreads();
  if (cg == 0 && s_ok) {
#pragma unroll
    for (int row = 0; row < 2; ++row) {
      f32x4 v = ((red[0][row][lane] + red[1][row][lane]) + red[2][row][lane]) + red[3][row][lane];
      const int px = pix + row * p.OW;
      const f32x4 r = p.res ? *reinterpret_cast<const f32x4*>(p.res + (int64_t)b * p.res_bs + px) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x = v[i];
        if (p.bias) x += p.bias[0];
        if (p.res) x += r[i];
        if (p.act == DEVA_ACT_RELU) {
          x = fmaxf(x, 0.0f);
        } else if (p.act == DEVA_ACT_SIGMOID) {
          x = sigmoidf_(x);
        } else if (p.act == DEVA_ACT_SQUARE_PLUS_ONE) {
          x = x * x + 1.0f;
        }
        v[i] = x;
      }
      *reinterpret_cast<f32x4*>(p.out + (int64_t)b * p.OHW + px) = v;
    }
  }
}

}  // namespace

int launch_conv3x3_cout1_rows(const Cout1Args& a, hipStream_t st) {
  if (a.OH % 2 == 0) {
    const int strips = (a.n_total / a.OHW) * (a.OH / 2) * (a.OW / 4);
    hipLaunchKernelGGL(conv3x3_cout1_rows2_kernel, dim3((unsigned)ceil_div(strips, 64)), dim3(256),
                       sizeof(float) * 9 * (size_t)a.ctot, st, a);
    return check_launch("deva_conv2d(cout=1, 3x3 rows)");
  }
  const int strips = a.n_total / 4;
  hipLaunchKernelGGL(conv3x3_cout1_rows_kernel, dim3((unsigned)ceil_div(strips, 64)), dim3(256),
                     sizeof(float) * 9 * (size_t)a.ctot, st, a);
  return check_launch("deva_conv2d(cout=1, 3x3 rows)");
}

int launch_conv_cout1(const Cout1Args& a, hipStream_t st) {
  const int K = a.KH * a.KW * a.ctot;
  if (a.ctot >= 65536 || a.KH >= 256 || a.KW >= 256) {
    set_error("deva_conv2d(cout=1): channel / kernel size beyond the 16 / 8-bit table fields");
    return 2;
  }
  const size_t smem = sizeof(uint2) * (((size_t)K + 63) / 64 * 64) + sizeof(float) * 256;
  // short reductions: a wave of pixels x 4 k-groups; long ones on small maps: 8 pixels x 32 k-groups (the map is
  // L2-resident, 32-byte runs are fine) so that ~100 k values per thread remain
  if (K <= 512) {
    hipLaunchKernelGGL((conv_cout1_kernel<64, 4, 8>), dim3((unsigned)ceil_div(a.n_total, 64)), dim3(256), smem, st, a);
  } else if (a.n_total >= 8192) {
    hipLaunchKernelGGL((conv_cout1_kernel<16, 16, 16>), dim3((unsigned)ceil_div(a.n_total, 16)), dim3(256), smem, st, a);
  } else {
    hipLaunchKernelGGL((conv_cout1_kernel<8, 32, 16>), dim3((unsigned)ceil_div(a.n_total, 8)), dim3(256), smem, st, a);
  }
  return check_launch("deva_conv2d(cout=1)");
}

}  // namespace deva
