// Pooling, resampling and pointwise blocks of the DEVA decoder / encoders (all HBM-bound).
// One thread per output element, consecutive threads on consecutive addresses (coalesced).
#include <math.h>

#include "common.h"

namespace deva {
namespace {

constexpr int TPB = 256;

inline dim3 grid_for(int64_t n) {
  int64_t b = ceil_div(n, TPB);
  if (b > 65535 * 16) b = 65535 * 16;  // grid-stride beyond this
  return dim3((unsigned)b);
}

// ------------------------------------------------------------------ zero padding of the last two dimensions
// (tensor_utils.py:7-22 pad_divide_by -> F.pad: a fill and a copy launch per frame in ATen).  T = the element as an integer
// of its size: the kernel moves bits.
template <typename T>
__global__ void pad2d_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t total, int H, int W, int top, int left,
                             int OH, int OW) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ow = (int)(i % OW);
    const int64_t t = i / OW;
    const int oh = (int)(t % OH);
    const int64_t plane = t / OH;
    const int ih = oh - top, iw = ow - left;
    const bool inside = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    out[i] = inside ? in[(plane * H + ih) * W + iw] : (T)0;
  }
}

// ------------------------------------------------------------------ taps of a stride-2 convolution as channels
// out[b][t*C + c][oh][ow] = in[b][c][2 oh + dy - pad][2 ow + dx - pad] (zero outside), t = dy*K + dx: a K x K stride-2
// convolution (resnet.py:46-114: conv1 / conv2 / downsample.0 of the first block of layer2, layer3) becomes a 1x1 stride-1
// convolution over K*K*C channels, which the vector-gather kernels of conv_mfma.hip / conv_f16.hip take (the scalar-gather
// kind they would otherwise run reaches 50 TFLOP/s).  One thread per (b, c, oh, ow): K*K loads, K*K coalesced stores.
template <int K>
__global__ void gather_s2_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t total, int C, int H, int W,
                                 int OH, int OW) {
  constexpr int PAD = K / 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ow = (int)(i % OW);
    int64_t t = i / OW;
    const int oh = (int)(t % OH);
    t /= OH;
    const int c = (int)(t % C);
    const int64_t b = t / C;
    const float* src = in + ((b * C + c) * (int64_t)H) * W;
    float v[K * K];
#pragma unroll
    for (int dy = 0; dy < K; ++dy)
#pragma unroll
      for (int dx = 0; dx < K; ++dx) {
        const int ih = 2 * oh + dy - PAD, iw = 2 * ow + dx - PAD;
        v[dy * K + dx] = ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) ? src[(int64_t)ih * W + iw] : 0.0f;
      }
    float* dst = out + ((b * K * K * C + c) * (int64_t)OH + oh) * OW + ow;
#pragma unroll
    for (int tap = 0; tap < K * K; ++tap) dst[(int64_t)tap * C * OH * OW] = v[tap];
  }
}

// ------------------------------------------------------------------ max pool 3x3 / s2 / p1
__global__ void maxpool3x3s2_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t total,
                                    int H, int W, int OH, int OW, int relu_after) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ow = (int)(i % OW);
    const int64_t t = i / OW;
    const int oh = (int)(t % OH);
    const int64_t plane = t / OH;
    const float* src = in + plane * (int64_t)H * W;
    float m = -INFINITY;
    const int h0 = oh * 2 - 1, w0 = ow * 2 - 1;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int h = h0 + dy;
      if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int w = w0 + dx;
        if ((unsigned)w >= (unsigned)W) continue;
        m = fmaxf(m, src[h * W + w]);
      }
    }
    if (relu_after) m = fmaxf(m, 0.0f);
    out[i] = m;
  }
}

// Four output pixels of one row per thread (even width, rows of whole quads, 16-byte aligned planes): the 3 x 9 input window is
// two aligned 16-byte loads + one scalar per row instead of 36 scalar loads -- a vector-memory instruction costs the same
// ~16 cycles of the CU's address path whatever its width (tools/probe/vmem_width_probe.hip), and the scalar kernel above
// spent 9 of them per output.  Same comparisons in the same order as the scalar kernel: bit-identical results.
__global__ __launch_bounds__(256) void maxpool3x3s2_quad_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t quads,
                                                                int H, int W, int OH, int OW, int relu_after) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int QW = OW / 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < quads; i += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(i % QW);
    const int64_t t = i / QW;
    const int oh = (int)(t % OH);
    const int64_t plane = t / OH;
    const float* src = in + plane * (int64_t)H * W;
    const int w0 = 8 * q;  // input columns w0 - 1 .. w0 + 7 feed outputs 4 q .. 4 q + 3
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int h = oh * 2 - 1 + dy;
      if ((unsigned)h >= (unsigned)H) continue;
      const float* row = src + (int64_t)h * W + w0;
      const f32x4 a = *reinterpret_cast<const f32x4*>(row), b = *reinterpret_cast<const f32x4*>(row + 4);
      const float left = w0 > 0 ? row[-1] : -INFINITY;
      // output j: columns 2j - 1, 2j, 2j + 1 of the window (dx = 0, 1, 2 in that order)
      m[0] = fmaxf(fmaxf(fmaxf(m[0], left), a[0]), a[1]);
      m[1] = fmaxf(fmaxf(fmaxf(m[1], a[1]), a[2]), a[3]);
      m[2] = fmaxf(fmaxf(fmaxf(m[2], a[3]), b[0]), b[1]);
      m[3] = fmaxf(fmaxf(fmaxf(m[3], b[1]), b[2]), b[3]);
    }
    if (relu_after) {
#pragma unroll
      for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], 0.0f);
    }
    *reinterpret_cast<f32x4*>(out + (plane * OH + oh) * (int64_t)OW + 4 * q) = m;
  }
}

// ------------------------------------------------------------------ bilinear helpers
// PyTorch area_pixel_compute_source_index(align_corners=False): src = scale*(dst+0.5)-0.5, clamped at 0
struct Lerp {
  int i0, i1;
  float w0, w1;
};
__device__ __forceinline__ Lerp lerp_index(int dst, float scale, int in_size) {
  float src = scale * (dst + 0.5f) - 0.5f;
  if (src < 0.0f) src = 0.0f;
  Lerp l;
  l.i0 = (int)src;  // src >= 0 -> floor
  l.i1 = l.i0 + ((l.i0 < in_size - 1) ? 1 : 0);
  l.w1 = src - (float)l.i0;
  l.w0 = 1.0f - l.w1;
  return l;
}
__device__ __forceinline__ float bilerp(const float* __restrict__ src, int W, const Lerp& ly, const Lerp& lx) {
  const float top = lx.w0 * src[ly.i0 * W + lx.i0] + lx.w1 * src[ly.i0 * W + lx.i1];
  const float bot = lx.w0 * src[ly.i1 * W + lx.i0] + lx.w1 * src[ly.i1 * W + lx.i1];
  return ly.w0 * top + ly.w1 * bot;
}

__global__ void upsample2x_add_kernel(const float* __restrict__ in, const float* __restrict__ skip,
                                      float* __restrict__ out, int64_t total, int C, int h, int w) {
  const int OH = 2 * h, OW = 2 * w;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const int64_t t = i / OW;
    const int oy = (int)(t % OH);
    const int64_t plane = t / OH;  // b*C + c
    const int c = (int)(plane % C);
    const Lerp ly = lerp_index(oy, 0.5f, h), lx = lerp_index(ox, 0.5f, w);
    float v = bilerp(in + plane * (int64_t)h * w, w, ly, lx);
    if (skip) v = skip[((int64_t)c * OH + oy) * OW + ox] + v;
    out[i] = v;
  }
}

// Two output rows x four consecutive output pixels per thread (w even, w >= 4): output rows 2P - 1 and 2P interpolate
// between the SAME two input rows (P - 1, P), and their eight inputs (columns 2j - 1 .. 2j + 2 of both rows) are two
// 16-byte loads -- 6 vector-memory instructions per 2 x 16 bytes of output (two loads, two skip loads, two stores) where the
// one-row form of round 4 issued 10 per 16 bytes (eight scalar loads): that kernel ran at 2.7 TB/s at 1080p / 11 objects,
// bound by the CU's address path (a vector-memory instruction costs the same whatever its width).  The column window is
// shifted inside the row at both ends (no read outside the row) and the clamped neighbours are picked from it.  Same
// per-pixel arithmetic as upsample2x_add_kernel (lerp_index weights, top / bottom order): bit-identical results.
// ds2 (optional, h and w even): the 2x2 box means of `in` -- area_downsample(in, 2), the decoder's p8 -> 1/16 for the sensory
// update (modules.py:121-151) -- written by the threads whose two input rows and inner two columns ARE such a box (odd P):
// no extra loads, the same sum in the same order as area_downsample_kernel, and one pass over `in` less.
typedef float up_f32x4 __attribute__((ext_vector_type(4)));
typedef up_f32x4 up_f32x4_u __attribute__((aligned(4)));
__global__ __launch_bounds__(256) void upsample2x_add_quad_kernel(const float* __restrict__ in, const float* __restrict__ skip,
                                                                  float* __restrict__ out, float* __restrict__ ds2, int C,
                                                                  int h, int w) {
  const int OH = 2 * h, OW = 2 * w, QW = OW >> 2;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= (h + 1) * QW) return;
  const int plane = blockIdx.y;  // b*C + c
  const int P = q / QW, j = q - P * QW;
  const float* src = in + (int64_t)plane * h * w;
  // the rows lerp_index names for output rows 2P - 1 and 2P (identical for both; row 0 pairs with row 1 for output row 0)
  const int rowA = P == 0 ? 0 : P - 1, rowB = P == 0 ? min(1, h - 1) : min(P, h - 1);
  const int base = min(max(2 * j - 1, 0), w - 4);
  const up_f32x4 va = *reinterpret_cast<const up_f32x4_u*>(src + rowA * w + base);
  const up_f32x4 vb = *reinterpret_cast<const up_f32x4_u*>(src + rowB * w + base);
  const int o = (2 * j - 1) - base;  // -1 at the left edge (column -1 clamps to 0), +1 at the right edge (column w clamps to w - 1)
  const float a0 = o > 0 ? va[1] : va[0], b0 = o < 0 ? va[0] : (o > 0 ? va[2] : va[1]);
  const float c0 = o < 0 ? va[1] : (o > 0 ? va[3] : va[2]), d0 = o < 0 ? va[2] : va[3];
  const float a1 = o > 0 ? vb[1] : vb[0], b1 = o < 0 ? vb[0] : (o > 0 ? vb[2] : vb[1]);
  const float c1 = o < 0 ? vb[1] : (o > 0 ? vb[3] : vb[2]), d1 = o < 0 ? vb[2] : vb[3];
  if (ds2 && (P & 1)) {  // rows 2R, 2R + 1 with R = P / 2; columns 2j, 2j + 1
    float sum = 0.0f;
    sum += b0;
    sum += c0;
    sum += b1;
    sum += c1;
    ds2[((int64_t)plane * (h >> 1) + (P >> 1)) * (w >> 1) + j] = sum / 4.0f;
  }
#pragma unroll
  for (int row = 0; row < 2; ++row) {
    const int oy = 2 * P - 1 + row;
    if (oy < 0 || oy >= OH) continue;
    const Lerp ly = lerp_index(oy, 0.5f, h);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const Lerp lx = lerp_index(4 * j + e, 0.5f, w);
      // columns of the four outputs: (2j-1, 2j), (2j, 2j+1), (2j, 2j+1), (2j+1, 2j+2) -- clamped like lerp_index clamps
      const float t0 = e == 0 ? a0 : (e == 3 ? c0 : b0), t1 = e == 0 ? b0 : (e == 3 ? d0 : c0);
      const float u0 = e == 0 ? a1 : (e == 3 ? c1 : b1), u1 = e == 0 ? b1 : (e == 3 ? d1 : c1);
      const float top = lx.w0 * t0 + lx.w1 * t1;
      const float bot = lx.w0 * u0 + lx.w1 * u1;
      v[e] = ly.w0 * top + ly.w1 * bot;
    }
    const int64_t ofs = ((int64_t)plane * OH + oy) * OW + 4 * j;
    up_f32x4 r = {v[0], v[1], v[2], v[3]};
    if (skip) {
      const up_f32x4 sk = *reinterpret_cast<const up_f32x4*>(skip + ((int64_t)(plane % C) * OH + oy) * OW + 4 * j);
      r = up_f32x4{sk[0] + v[0], sk[1] + v[1], sk[2] + v[2], sk[3] + v[3]};
    }
    *reinterpret_cast<up_f32x4*>(out + ofs) = r;
  }
}

// ------------------------------------------------------------------ area downsample (integer factor)
__global__ void area_downsample_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t total,
                                       int H, int W, int f) {
  const int OH = H / f, OW = W / f;
  const float denom = (float)(f * f);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const int64_t t = i / OW;
    const int oy = (int)(t % OH);
    const int64_t plane = t / OH;
    const float* src = in + plane * (int64_t)H * W + (int64_t)oy * f * W + ox * f;
    float s = 0.0f;
    for (int dy = 0; dy < f; ++dy)
      for (int dx = 0; dx < f; ++dx) s += src[dy * W + dx];
    out[i] = s / denom;
  }
}

// factors 2, 4 and 16 on 16-byte aligned rows: the same sums in the same order from 16-byte loads (a vector-memory
// instruction costs the same whatever its width: the scalar form issues f*f of them per output, this one f per OPT outputs)
typedef float pw_f32x4 __attribute__((ext_vector_type(4)));
typedef float pw_f32x2 __attribute__((ext_vector_type(2)));
template <int F, int OPT>  // OPT outputs per thread (1, 2 or 4) from OPT * F / 4 16-byte loads per input row
__global__ void area_downsample_vec_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t total_groups,
                                           int H, int W) {
  constexpr int LOADS = OPT * F / 4;
  static_assert(LOADS >= 1 && LOADS * 4 == OPT * F, "whole 16-byte loads");
  const int OH = H / F, OW = W / F, GW = OW / OPT;
  const float denom = (float)(F * F);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total_groups; i += (int64_t)gridDim.x * blockDim.x) {
    const int gx = (int)(i % GW);
    const int64_t t = i / GW;
    const int oy = (int)(t % OH);
    const int64_t plane = t / OH;
    const float* src = in + plane * (int64_t)H * W + (int64_t)oy * F * W + gx * OPT * F;
    pw_f32x4 v[F][LOADS];
#pragma unroll
    for (int dy = 0; dy < F; ++dy)
#pragma unroll
      for (int l = 0; l < LOADS; ++l) v[dy][l] = *reinterpret_cast<const pw_f32x4*>(src + dy * W + 4 * l);
    float r[OPT];
#pragma unroll
    for (int o = 0; o < OPT; ++o) {
      float sum = 0.0f;
#pragma unroll
      for (int dy = 0; dy < F; ++dy)
#pragma unroll
        for (int dx = 0; dx < F; ++dx) sum += v[dy][(o * F + dx) / 4][(o * F + dx) % 4];
      r[o] = sum / denom;
    }
    float* dst = out + (plane * OH + oy) * (int64_t)OW + gx * OPT;
    if constexpr (OPT == 4) {
      *reinterpret_cast<pw_f32x4*>(dst) = pw_f32x4{r[0], r[1], r[2], r[3]};
    } else if constexpr (OPT == 2) {
      *reinterpret_cast<pw_f32x2*>(dst) = pw_f32x2{r[0], r[1]};
    } else {
      dst[0] = r[0];
    }
  }
}

// ------------------------------------------------------------------ aggregate (network.py:33-40)
__device__ __forceinline__ float logit_clamped(float p) {
  p = fminf(fmaxf(p, 1e-7f), 1.0f - 1e-7f);
  return logf(p / (1.0f - p));
}

template <bool U8>
__global__ void aggregate_kernel(const void* __restrict__ in_, int apply_sigmoid, float* __restrict__ out,
                                 int num, int64_t pixels) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < pixels; i += (int64_t)gridDim.x * blockDim.x) {
    float bg = 1.0f;
    for (int o = 0; o < num; ++o) {
      float p;
      if (U8) {
        p = (float)reinterpret_cast<const unsigned char*>(in_)[(int64_t)o * pixels + i];
      } else {
        p = reinterpret_cast<const float*>(in_)[(int64_t)o * pixels + i];
      }
      if (apply_sigmoid) p = sigmoidf_(p);
      bg *= (1.0f - p);
      out[(int64_t)(o + 1) * pixels + i] = logit_clamped(p);
    }
    out[i] = logit_clamped(bg);
  }
}

// ------------------------------------------------------------------ channel softmax
__global__ void softmax_channels_kernel(const float* __restrict__ in, float* __restrict__ out, int C,
                                        int64_t pixels) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < pixels; i += (int64_t)gridDim.x * blockDim.x) {
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, in[(int64_t)c * pixels + i]);
    float s = 0.0f;
    for (int c = 0; c < C; ++c) s += expf(in[(int64_t)c * pixels + i] - m);
    for (int c = 0; c < C; ++c) out[(int64_t)c * pixels + i] = expf(in[(int64_t)c * pixels + i] - m) / s;
  }
}

// ------------------------------------------------------------------ x4 bilinear + channel softmax
__global__ void upsample4x_softmax_kernel(const float* __restrict__ in, float* __restrict__ logits_up,
                                          float* __restrict__ prob, int C, int h, int w) {
  const int OH = 4 * h, OW = 4 * w;
  const int64_t opix = (int64_t)OH * OW;
  const int64_t ipix = (int64_t)h * w;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < opix; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const int oy = (int)(i / OW);
    const Lerp ly = lerp_index(oy, 0.25f, h), lx = lerp_index(ox, 0.25f, w);
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) {
      const float v = bilerp(in + c * ipix, w, ly, lx);
      if (logits_up) logits_up[c * opix + i] = v;
      m = fmaxf(m, v);
    }
    float s = 0.0f;
    for (int c = 0; c < C; ++c) s += expf(bilerp(in + c * ipix, w, ly, lx) - m);
    for (int c = 0; c < C; ++c) prob[c * opix + i] = expf(bilerp(in + c * ipix, w, ly, lx) - m) / s;
  }
}

// ------------------------------------------------------------------ CBAM pieces
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* smem) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(v, off);
    v = is_max ? fmaxf(v, o) : (v + o);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  const int nw = blockDim.x >> 6;
  float r = smem[0];
  for (int i = 1; i < nw; ++i) r = is_max ? fmaxf(r, smem[i]) : (r + smem[i]);
  return r;
}

__global__ void global_avgmax_kernel(const float* __restrict__ x, float* __restrict__ avg,
                                     float* __restrict__ mx, int hw) {
  __shared__ float smem[8];
  const float* src = x + (int64_t)blockIdx.x * hw;
  float s = 0.0f, m = -INFINITY;
  for (int i = threadIdx.x; i < hw; i += blockDim.x) {
    const float v = src[i];
    s += v;
    m = fmaxf(m, v);
  }
  s = block_reduce(s, false, smem);
  m = block_reduce(m, true, smem);
  if (threadIdx.x == 0) {
    avg[blockIdx.x] = s / (float)hw;
    mx[blockIdx.x] = m;
  }
}

// one block per batch item; scale = sigmoid(mlp(avg) + mlp(max)).  The block is alone on its CU and the whole kernel is a
// chain of dependent latencies, so both layers put all their loads in flight at once: layer 1 runs four threads per hidden
// unit and input vector (each streams a quarter of the weight row with 16-byte loads, the partial sums meet in two
// shuffles), layer 2 reads its `hidden` weights per output channel with 16-byte loads.  (C % 16 == 0, hidden % 4 == 0 and
// 2 * hidden * 4 <= blockDim; other shapes take the plain loops.)
__global__ void cbam_mlp_kernel(const float* __restrict__ avg, const float* __restrict__ mx,
                                const float* __restrict__ w1, const float* __restrict__ b1,
                                const float* __restrict__ w2, const float* __restrict__ b2,
                                float* __restrict__ scale, int C, int hidden) {
  extern __shared__ float sm[];  // [2][C] inputs, then [2][hidden]
  float* vin = sm;
  float* hid = sm + 2 * C;
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < C; i += blockDim.x) {
    vin[i] = avg[(int64_t)b * C + i];
    vin[C + i] = mx[(int64_t)b * C + i];
  }
  __syncthreads();
  const bool fast = (C % 16 == 0) && (hidden % 4 == 0) && (2 * hidden * 4 <= (int)blockDim.x) &&
                    ((((uintptr_t)w1 | (uintptr_t)w2) & 15) == 0);
  if (fast) {
    const int t = threadIdx.x >> 2, part = threadIdx.x & 3;  // unit-and-vector t, quarter `part` of the channels
    float s = 0.0f;
    if (t < 2 * hidden) {
      const int which = t / hidden, j = t % hidden;
      const int quarter = C / 4;
      const pw_f32x4* wr = reinterpret_cast<const pw_f32x4*>(w1 + (int64_t)j * C + part * quarter);
      const float* xv = vin + which * C + part * quarter;
      for (int i = 0; i < quarter / 4; ++i) {
        const pw_f32x4 w = wr[i];
        s += w[0] * xv[4 * i] + w[1] * xv[4 * i + 1] + w[2] * xv[4 * i + 2] + w[3] * xv[4 * i + 3];
      }
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (t < 2 * hidden && part == 0) hid[t] = fmaxf(s + b1[t % hidden], 0.0f);
  } else {
    for (int t = threadIdx.x; t < 2 * hidden; t += blockDim.x) {
      const int which = t / hidden, j = t % hidden;
      float s = 0.0f;
      for (int c = 0; c < C; ++c) s += w1[(int64_t)j * C + c] * vin[which * C + c];
      hid[t] = fmaxf(s + b1[j], 0.0f);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float sa = 0.0f, sm_ = 0.0f;
    if (fast) {
      const pw_f32x4* wr = reinterpret_cast<const pw_f32x4*>(w2 + (int64_t)c * hidden);
      for (int j4 = 0; j4 < hidden / 4; ++j4) {
        const pw_f32x4 w = wr[j4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          sa += w[u] * hid[4 * j4 + u];
          sm_ += w[u] * hid[hidden + 4 * j4 + u];
        }
      }
    } else {
      for (int j = 0; j < hidden; ++j) {
        const float w = w2[(int64_t)c * hidden + j];
        sa += w * hid[j];
        sm_ += w * hid[hidden + j];
      }
    }
    const float att = (sa + b2[c]) + (sm_ + b2[c]);
    scale[(int64_t)b * C + c] = sigmoidf_(att);
  }
}

// block = 64 pixels x POOL_CG channel groups: each wave reduces its share of the channels for 64 consecutive pixels
// (coalesced 256-B reads, POOL_U loads in flight), the partial max / sums meet in LDS in group order.  (Round 5 ran 4 groups
// with one load in flight: 36 us for 16.6 MB at 480p / 5 objects.)
constexpr int POOL_CG = 16, POOL_U = 8;
__global__ __launch_bounds__(64 * POOL_CG) void cbam_channel_pool_kernel(const float* __restrict__ x,
                                                                         const float* __restrict__ scale,
                                                                         float* __restrict__ pooled, int64_t total, int C,
                                                                         int hw) {
  __shared__ float s_max[POOL_CG][64], s_sum[POOL_CG][64];
  const int px = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int64_t i = blockIdx.x * 64ll + px;
  const bool ok = i < total;
  const int64_t ii = ok ? i : 0;
  const int p = (int)(ii % hw);
  const int64_t b = ii / hw;
  const float* src = x + b * C * hw + p;
  const float* sc = scale + b * C;
  const int per = (C + POOL_CG - 1) / POOL_CG;
  const int c_lo = cg * per, c_hi = min(C, c_lo + per);
  float m = -INFINITY, s = 0.0f;
  int c = c_lo;
  for (; c + POOL_U <= c_hi; c += POOL_U) {
    float v[POOL_U];
#pragma unroll
    for (int u = 0; u < POOL_U; ++u) v[u] = src[(int64_t)(c + u) * hw];
#pragma unroll
    for (int u = 0; u < POOL_U; ++u) {
      const float w = v[u] * sc[c + u];
      m = fmaxf(m, w);
      s += w;
    }
  }
  for (; c < c_hi; ++c) {
    const float w = src[(int64_t)c * hw] * sc[c];
    m = fmaxf(m, w);
    s += w;
  }
  s_max[cg][px] = m;
  s_sum[cg][px] = s;
  __syncthreads();
  if (cg == 0 && ok) {
    float mm = s_max[0][px], ss = s_sum[0][px];
#pragma unroll
    for (int g = 1; g < POOL_CG; ++g) {
      mm = fmaxf(mm, s_max[g][px]);
      ss += s_sum[g][px];
    }
    pooled[(b * 2 + 0) * hw + p] = mm;
    pooled[(b * 2 + 1) * hw + p] = ss / (float)C;
  }
}

__global__ void cbam_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                  const float* __restrict__ gate, float* __restrict__ out, int64_t total,
                                  int C, int hw) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % hw);
    const int64_t bc = i / hw;
    const int64_t b = bc / C;
    const float v = x[i];
    const float r = (v * scale[bc]) * sigmoidf_(gate[b * hw + p]);
    out[i] = v + r;
  }
}

// hw % 4 == 0, 16-byte aligned tensors: four pixels per thread (the same arithmetic per element)
__global__ void cbam_apply_vec_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                      const float* __restrict__ gate, float* __restrict__ out, int64_t total4, int C, int hw) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i * 4;
    const int p = (int)(e % hw);
    const int64_t bc = e / hw;
    const int64_t b = bc / C;
    const pw_f32x4 v = *reinterpret_cast<const pw_f32x4*>(x + e);
    const pw_f32x4 g = *reinterpret_cast<const pw_f32x4*>(gate + b * hw + p);
    const float sc = scale[bc];
    pw_f32x4 r;
#pragma unroll
    for (int u = 0; u < 4; ++u) r[u] = v[u] + (v[u] * sc) * sigmoidf_(g[u]);
    *reinterpret_cast<pw_f32x4*>(out + e) = r;
  }
}

// ------------------------------------------------------------------ GRU-style sensory update
__global__ void gru_update_kernel(const float* __restrict__ values, const float* __restrict__ h,
                                  float* __restrict__ new_h, int64_t total, int C, int hw) {
  const int64_t chw = (int64_t)C * hw;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / chw;
    const int64_t r = i - b * chw;
    const float* v = values + b * 3 * chw + r;
    const float f = sigmoidf_(v[0]);
    const float u = sigmoidf_(v[chw]);
    const float n = tanhf(v[2 * chw]);
    const float hv = h[i];
    new_h[i] = f * hv * (1.0f - u) + u * n;
  }
}

// hw % 4 == 0 and 16-byte aligned tensors: four elements per thread (the same arithmetic per element)
__global__ void gru_update_vec_kernel(const float* __restrict__ values, const float* __restrict__ h,
                                      float* __restrict__ new_h, int64_t total4, int C, int hw) {
  const int64_t chw = (int64_t)C * hw;
  for (int64_t i4 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i4 < total4; i4 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = i4 * 4;
    const int64_t b = i / chw;
    const int64_t r = i - b * chw;
    const float* v = values + b * 3 * chw + r;
    const pw_f32x4 vf = *reinterpret_cast<const pw_f32x4*>(v), vu = *reinterpret_cast<const pw_f32x4*>(v + chw);
    const pw_f32x4 vn = *reinterpret_cast<const pw_f32x4*>(v + 2 * chw), hv = *reinterpret_cast<const pw_f32x4*>(h + i);
    pw_f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float f = sigmoidf_(vf[e]);
      const float u = sigmoidf_(vu[e]);
      const float n = tanhf(vn[e]);
      o[e] = f * hv[e] * (1.0f - u) + u * n;
    }
    *reinterpret_cast<pw_f32x4*>(new_h + i) = o;
  }
}

// ------------------------------------------------------------------ input head
// uint8 HWC frame -> normalised fp32 CHW at the network's input size, in one pass
// (deva/inference/data/video_reader.py:139-144 = ToTensor + Normalize + Resize(antialias=True);
// deva/inference/demo_utils.py:10-19 = the same with plain bilinear).  The normalisation goes through
// a 3 x 256 table in LDS with torchvision's arithmetic ((u/255 - mean)/std in fp32); the resize follows
// ATen: antialias -> separable triangle filter of _upsample_bilinear2d_aa (support = scale when
// shrinking, weights normalised per output index, columns before rows, each 1-D sum accumulated in
// tap order); otherwise the 2-tap bilinear of upsample_bilinear2d (align_corners=False).
struct HeadArgs {
  const unsigned char* img;
  int h, w, oh, ow, antialias;
  int pad_left, pad_top, ph, pw;  // the resized frame sits at (pad_top, pad_left) of a zeroed ph x pw plane
  float mean[3], stdv[3];
  float* out;
};

__device__ __forceinline__ float tri(float x) { return fmaxf(0.0f, 1.0f - fabsf(x)); }

// [first, first+size) source indices and their weights for output index i (ATen
// _compute_indices_min_size_weights_aa); at most MAXT taps
constexpr int MAXT = 24;
__device__ __forceinline__ void aa_taps(int i, int in, float scale, int& first, int& size, float (&wgt)[MAXT]) {
  const float support = scale >= 1.0f ? scale : 1.0f;
  const float invscale = scale >= 1.0f ? 1.0f / scale : 1.0f;
  const float center = scale * ((float)i + 0.5f);
  first = max((int)(center - support + 0.5f), 0);
  size = min(min((int)(center + support + 0.5f), in) - first, MAXT);
  float total = 0.0f;
  for (int j = 0; j < size; ++j) {
    wgt[j] = tri(((float)(j + first) - center + 0.5f) * invscale);
    total += wgt[j];
  }
  if (total != 0.0f)
    for (int j = 0; j < size; ++j) wgt[j] /= total;
}

__global__ void input_head_kernel(const HeadArgs p) {
  __shared__ float lut[3][256];
  for (int i = threadIdx.x; i < 768; i += blockDim.x) {
    const int c = i >> 8, u = i & 255;
    lut[c][u] = ((float)u / 255.0f - p.mean[c]) / p.stdv[c];
  }
  __syncthreads();
  const float sy = (float)p.h / (float)p.oh, sx = (float)p.w / (float)p.ow;
  const int64_t total = (int64_t)p.ph * p.pw;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int py = (int)(i / p.pw), px_ = (int)(i - (int64_t)py * p.pw);
    const int y = py - p.pad_top, x = px_ - p.pad_left;
    float r[3];
    if (y < 0 || y >= p.oh || x < 0 || x >= p.ow) {
      r[0] = r[1] = r[2] = 0.0f;  // pad_divide_by's zero border (tensor_utils.py:7-22), fused
    } else if (p.oh == p.h && p.ow == p.w) {
      const unsigned char* px = p.img + ((int64_t)y * p.w + x) * 3;
      for (int c = 0; c < 3; ++c) r[c] = lut[c][px[c]];
    } else if (p.antialias) {
      int y0, ny, x0, nx;
      float wy[MAXT], wx[MAXT];
      aa_taps(y, p.h, sy, y0, ny, wy);
      aa_taps(x, p.w, sx, x0, nx, wx);
      r[0] = r[1] = r[2] = 0.0f;
      for (int j = 0; j < ny; ++j) {
        const unsigned char* row = p.img + ((int64_t)(y0 + j) * p.w + x0) * 3;
        float hsum[3] = {lut[0][row[0]] * wx[0], lut[1][row[1]] * wx[0], lut[2][row[2]] * wx[0]};
        for (int t = 1; t < nx; ++t)
          for (int c = 0; c < 3; ++c) hsum[c] += lut[c][row[3 * t + c]] * wx[t];
        for (int c = 0; c < 3; ++c) r[c] = (j == 0) ? hsum[c] * wy[0] : r[c] + hsum[c] * wy[j];
      }
    } else {
      const float fy = fmaxf(sy * ((float)y + 0.5f) - 0.5f, 0.0f), fx = fmaxf(sx * ((float)x + 0.5f) - 0.5f, 0.0f);
      const int y0 = min((int)fy, p.h - 1), x0 = min((int)fx, p.w - 1);
      const int y1 = y0 + (y0 < p.h - 1 ? 1 : 0), x1 = x0 + (x0 < p.w - 1 ? 1 : 0);
      const float ly1 = fy - (float)y0, lx1 = fx - (float)x0, ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
      for (int c = 0; c < 3; ++c) {
        const float a = lut[c][p.img[((int64_t)y0 * p.w + x0) * 3 + c]], b = lut[c][p.img[((int64_t)y0 * p.w + x1) * 3 + c]];
        const float d = lut[c][p.img[((int64_t)y1 * p.w + x0) * 3 + c]], e = lut[c][p.img[((int64_t)y1 * p.w + x1) * 3 + c]];
        r[c] = ly0 * (lx0 * a + lx1 * b) + ly1 * (lx0 * d + lx1 * e);
      }
    }
    for (int c = 0; c < 3; ++c) p.out[(int64_t)c * total + i] = r[c];
  }
}

}  // namespace
}  // namespace deva

using namespace deva;

extern "C" int deva_pad2d(const void* in, void* out, int elem_bytes, int64_t planes, int height, int width, int top, int left,
                          int out_height, int out_width, void* stream) {
  DEVA_REQUIRE(in && out && planes > 0 && height > 0 && width > 0 && top >= 0 && left >= 0 && out_height >= height + top &&
                   out_width >= width + left, "deva_pad2d: bad args");
  const int64_t total = planes * out_height * out_width;
  switch (elem_bytes) {
    case 1:
      hipLaunchKernelGGL(pad2d_kernel<uint8_t>, grid_for(total), dim3(TPB), 0, (hipStream_t)stream, (const uint8_t*)in,
                         (uint8_t*)out, total, height, width, top, left, out_height, out_width);
      break;
    case 4:
      hipLaunchKernelGGL(pad2d_kernel<uint32_t>, grid_for(total), dim3(TPB), 0, (hipStream_t)stream, (const uint32_t*)in,
                         (uint32_t*)out, total, height, width, top, left, out_height, out_width);
      break;
    case 8:
      hipLaunchKernelGGL(pad2d_kernel<uint64_t>, grid_for(total), dim3(TPB), 0, (hipStream_t)stream, (const uint64_t*)in,
                         (uint64_t*)out, total, height, width, top, left, out_height, out_width);
      break;
    default:
      DEVA_REQUIRE(false, "deva_pad2d: elements of 1, 4 or 8 bytes");
  }
  return check_launch("deva_pad2d");
}

extern "C" int deva_gather_s2(const float* in, float* out, int batch, int channels, int height, int width, int kernel,
                              void* stream) {
  DEVA_REQUIRE(in && out && batch > 0 && channels > 0 && height > 0 && width > 0, "deva_gather_s2: bad args");
  DEVA_REQUIRE(kernel == 1 || kernel == 3, "deva_gather_s2: kernel 1 (pad 0) or 3 (pad 1)");
  const int pad = kernel / 2;
  const int OH = (height + 2 * pad - kernel) / 2 + 1, OW = (width + 2 * pad - kernel) / 2 + 1;
  const int64_t total = (int64_t)batch * channels * OH * OW;
  if (kernel == 1) {
    hipLaunchKernelGGL(gather_s2_kernel<1>, grid_for(total), dim3(TPB), 0, (hipStream_t)stream, in, out, total, channels, height,
                       width, OH, OW);
  } else {
    hipLaunchKernelGGL(gather_s2_kernel<3>, grid_for(total), dim3(TPB), 0, (hipStream_t)stream, in, out, total, channels, height,
                       width, OH, OW);
  }
  return check_launch("deva_gather_s2");
}

extern "C" int deva_maxpool3x3s2(const float* in, float* out, int64_t planes, int height, int width,
                                 int relu_after, void* stream) {
  DEVA_REQUIRE(in && out && planes > 0 && height > 0 && width > 0, "deva_maxpool3x3s2: bad args");
  const int OH = (height + 2 - 3) / 2 + 1, OW = (width + 2 - 3) / 2 + 1;
  const int64_t total = planes * OH * OW;
  if (width % 8 == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {  // OW = width / 2 is a multiple of 4: whole quads, aligned rows
    hipLaunchKernelGGL(maxpool3x3s2_quad_kernel, grid_for(total / 4), dim3(TPB), 0, (hipStream_t)stream, in, out, total / 4,
                       height, width, OH, OW, relu_after);
    return check_launch("deva_maxpool3x3s2");
  }
  hipLaunchKernelGGL(maxpool3x3s2_kernel, grid_for(total), dim3(TPB), 0, (hipStream_t)stream, in, out, total,
                     height, width, OH, OW, relu_after);
  return check_launch("deva_maxpool3x3s2");
}

static int upsample2x_add_impl(const float* in, const float* skip, float* out, float* ds2, int batch, int channels, int height,
                               int width, void* stream, const char* what) {
  const int64_t total = (int64_t)batch * channels * height * 2 * width * 2;
  const int64_t planes = (int64_t)batch * channels;
  const bool aligned = (((uintptr_t)out | (uintptr_t)skip) & 15) == 0;  // 16-byte rows: OW % 4 == 0 and aligned bases
  if (width % 2 == 0 && width >= 4 && planes <= 65535 && aligned && (!ds2 || height % 2 == 0)) {
    const int quads = (height + 1) * (width * 2 / 4);  // (row pair, pixel quad): output rows 2P - 1 and 2P, P = 0 .. height
    hipLaunchKernelGGL(upsample2x_add_quad_kernel, dim3((unsigned)ceil_div(quads, 256), (unsigned)planes), dim3(256), 0,
                       (hipStream_t)stream, in, skip, out, ds2, channels, height, width);
    return check_launch(what);
  }
  hipLaunchKernelGGL(upsample2x_add_kernel, grid_for(total), dim3(TPB), 0, (hipStream_t)stream, in, skip, out,
                     total, channels, height, width);
  if (check_launch(what)) return 1;
  if (ds2) return deva_area_downsample(in, ds2, planes, height, width, 2, stream);  // (shapes the quad kernel does not take)
  return 0;
}

extern "C" int deva_upsample2x_add(const float* in, const float* skip, float* out, int batch, int channels,
                                   int height, int width, void* stream) {
  DEVA_REQUIRE(in && out && batch > 0 && channels > 0 && height > 0 && width > 0, "deva_upsample2x_add: bad args");
  return upsample2x_add_impl(in, skip, out, nullptr, batch, channels, height, width, stream, "deva_upsample2x_add");
}

extern "C" int deva_upsample2x_add_ds2(const float* in, const float* skip, float* out, float* ds2, int batch, int channels,
                                       int height, int width, void* stream) {
  DEVA_REQUIRE(in && out && ds2 && batch > 0 && channels > 0 && height > 0 && width > 0, "deva_upsample2x_add_ds2: bad args");
  DEVA_REQUIRE(height % 2 == 0 && width % 2 == 0, "deva_upsample2x_add_ds2: even input size expected");
  return upsample2x_add_impl(in, skip, out, ds2, batch, channels, height, width, stream, "deva_upsample2x_add_ds2");
}

extern "C" int deva_area_downsample(const float* in, float* out, int64_t planes, int height, int width,
                                    int factor, void* stream) {
  DEVA_REQUIRE(in && out && planes > 0 && factor >= 1, "deva_area_downsample: bad args");
  DEVA_REQUIRE(height % factor == 0 && width % factor == 0, "deva_area_downsample: size %dx%d not divisible by %d",
               height, width, factor);
  const int64_t total = planes * (height / factor) * (width / factor);
  const bool aligned = (((uintptr_t)in | (uintptr_t)out) & 15) == 0 && width % 4 == 0;
  const int ow = width / factor;
#define DEVA_AREA_VEC(F, OPT)                                                                                                \
  do {                                                                                                                        \
    hipLaunchKernelGGL((area_downsample_vec_kernel<F, OPT>), grid_for(total / OPT), dim3(TPB), 0, (hipStream_t)stream, in, out, \
                       total / OPT, height, width);                                                                           \
    return check_launch("deva_area_downsample");                                                                              \
  } while (0)
  if (aligned && factor == 16) DEVA_AREA_VEC(16, 1);  // (the last mask at 1/16: 64 16-byte loads per output instead of 256 scalar ones)
  if (aligned && (factor == 2 || factor == 4)) {
    if (factor == 4 && ow % 4 == 0) DEVA_AREA_VEC(4, 4);
    if (factor == 4 && ow % 2 == 0) DEVA_AREA_VEC(4, 2);
    if (factor == 2 && ow % 4 == 0) DEVA_AREA_VEC(2, 4);
    if (factor == 2 && ow % 2 == 0) DEVA_AREA_VEC(2, 2);
  }
#undef DEVA_AREA_VEC
  hipLaunchKernelGGL(area_downsample_kernel, grid_for(total), dim3(TPB), 0, (hipStream_t)stream, in, out, total,
                     height, width, factor);
  return check_launch("deva_area_downsample");
}

extern "C" int deva_aggregate(const void* in, int in_is_u8, int apply_sigmoid, float* out, int num,
                              int64_t pixels, void* stream) {
  // num == 0 (no object yet, inference_core.py:67-70 / :196) yields the background-only map; `in` may be NULL then
  DEVA_REQUIRE((in || num == 0) && out && num >= 0 && pixels > 0, "deva_aggregate: bad args");
  if (in_is_u8) {
    hipLaunchKernelGGL(aggregate_kernel<true>, grid_for(pixels), dim3(TPB), 0, (hipStream_t)stream, in,
                       apply_sigmoid, out, num, pixels);
  } else {
    hipLaunchKernelGGL(aggregate_kernel<false>, grid_for(pixels), dim3(TPB), 0, (hipStream_t)stream, in,
                       apply_sigmoid, out, num, pixels);
  }
  return check_launch("deva_aggregate");
}

extern "C" int deva_softmax_channels(const float* in, float* out, int channels, int64_t pixels, void* stream) {
  DEVA_REQUIRE(in && out && channels > 0 && pixels > 0, "deva_softmax_channels: bad args");
  hipLaunchKernelGGL(softmax_channels_kernel, grid_for(pixels), dim3(TPB), 0, (hipStream_t)stream, in, out,
                     channels, pixels);
  return check_launch("deva_softmax_channels");
}

extern "C" int deva_upsample4x_softmax(const float* in, float* logits_up, float* prob, int channels, int height,
                                       int width, void* stream) {
  DEVA_REQUIRE(in && prob && channels > 0 && height > 0 && width > 0, "deva_upsample4x_softmax: bad args");
  const int64_t opix = (int64_t)height * 4 * width * 4;
  hipLaunchKernelGGL(upsample4x_softmax_kernel, grid_for(opix), dim3(TPB), 0, (hipStream_t)stream, in, logits_up,
                     prob, channels, height, width);
  return check_launch("deva_upsample4x_softmax");
}

extern "C" int deva_global_avgmax(const float* x, float* avg, float* mx, int64_t planes, int hw, void* stream) {
  DEVA_REQUIRE(x && avg && mx && planes > 0 && hw > 0, "deva_global_avgmax: bad args");
  DEVA_REQUIRE(planes < (1ll << 31), "deva_global_avgmax: too many planes");
  hipLaunchKernelGGL(global_avgmax_kernel, dim3((unsigned)planes), dim3(TPB), 0, (hipStream_t)stream, x, avg, mx,
                     hw);
  return check_launch("deva_global_avgmax");
}

extern "C" int deva_cbam_mlp(const float* avg, const float* mx, const float* w1, const float* b1, const float* w2,
                             const float* b2, float* scale, int batch, int channels, int hidden, void* stream) {
  DEVA_REQUIRE(avg && mx && w1 && b1 && w2 && b2 && scale && batch > 0 && channels > 0 && hidden > 0,
               "deva_cbam_mlp: bad args");
  const size_t smem = sizeof(float) * (2 * (size_t)channels + 2 * (size_t)hidden);
  DEVA_REQUIRE(smem <= 64 * 1024, "deva_cbam_mlp: channels too large");
  hipLaunchKernelGGL(cbam_mlp_kernel, dim3((unsigned)batch), dim3(TPB), smem, (hipStream_t)stream, avg, mx, w1, b1,
                     w2, b2, scale, channels, hidden);
  return check_launch("deva_cbam_mlp");
}

extern "C" int deva_cbam_channel_pool(const float* x, const float* scale, float* pooled, int batch, int channels,
                                      int hw, void* stream) {
  DEVA_REQUIRE(x && scale && pooled && batch > 0 && channels > 0 && hw > 0, "deva_cbam_channel_pool: bad args");
  const int64_t total = (int64_t)batch * hw;
  hipLaunchKernelGGL(cbam_channel_pool_kernel, dim3((unsigned)ceil_div(total, 64)), dim3(64 * POOL_CG), 0, (hipStream_t)stream,
                     x, scale, pooled, total, channels, hw);
  return check_launch("deva_cbam_channel_pool");
}

extern "C" int deva_cbam_apply(const float* x, const float* scale, const float* gate, float* out, int batch,
                               int channels, int hw, void* stream) {
  DEVA_REQUIRE(x && scale && gate && out && batch > 0 && channels > 0 && hw > 0, "deva_cbam_apply: bad args");
  const int64_t total = (int64_t)batch * channels * hw;
  if (hw % 4 == 0 && (((uintptr_t)x | (uintptr_t)gate | (uintptr_t)out) & 15) == 0) {
    hipLaunchKernelGGL(cbam_apply_vec_kernel, grid_for(total / 4), dim3(TPB), 0, (hipStream_t)stream, x, scale, gate, out,
                       total / 4, channels, hw);
    return check_launch("deva_cbam_apply");
  }
  hipLaunchKernelGGL(cbam_apply_kernel, grid_for(total), dim3(TPB), 0, (hipStream_t)stream, x, scale, gate, out,
                     total, channels, hw);
  return check_launch("deva_cbam_apply");
}

extern "C" int deva_gru_update(const float* values, const float* h, float* new_h, int batch, int channels, int hw,
                               void* stream) {
  DEVA_REQUIRE(values && h && new_h && batch > 0 && channels > 0 && hw > 0, "deva_gru_update: bad args");
  const int64_t total = (int64_t)batch * channels * hw;
  if (hw % 4 == 0 && (((uintptr_t)values | (uintptr_t)h | (uintptr_t)new_h) & 15) == 0) {
    hipLaunchKernelGGL(gru_update_vec_kernel, grid_for(total / 4), dim3(TPB), 0, (hipStream_t)stream, values, h, new_h,
                       total / 4, channels, hw);
    return check_launch("deva_gru_update");
  }
  hipLaunchKernelGGL(gru_update_kernel, grid_for(total), dim3(TPB), 0, (hipStream_t)stream, values, h, new_h, total,
                     channels, hw);
  return check_launch("deva_gru_update");
}

extern "C" int deva_input_head(const unsigned char* image_hwc, int height, int width, const float* mean3,
                               const float* std3, int antialias, float* out, int out_height, int out_width,
                               int pad_left, int pad_right, int pad_top, int pad_bottom, void* stream) {
  using namespace deva;
  DEVA_REQUIRE(image_hwc && mean3 && std3 && out && height > 0 && width > 0 && out_height > 0 && out_width > 0,
               "deva_input_head: bad args");
  DEVA_REQUIRE(pad_left >= 0 && pad_right >= 0 && pad_top >= 0 && pad_bottom >= 0, "deva_input_head: negative pad");
  // the antialias filter holds at most 24 taps per axis: shrink factors up to 11
  DEVA_REQUIRE(!antialias || ((float)height / out_height <= 11.0f && (float)width / out_width <= 11.0f),
               "deva_input_head: shrink factor above 11 is not supported with antialias");
  HeadArgs a;
  a.img = image_hwc;
  a.h = height;
  a.w = width;
  a.oh = out_height;
  a.ow = out_width;
  a.antialias = antialias;
  a.pad_left = pad_left;
  a.pad_top = pad_top;
  a.ph = out_height + pad_top + pad_bottom;
  a.pw = out_width + pad_left + pad_right;
  for (int c = 0; c < 3; ++c) {
    a.mean[c] = mean3[c];
    a.stdv[c] = std3[c];
  }
  a.out = out;
  hipLaunchKernelGGL(input_head_kernel, grid_for((int64_t)a.ph * a.pw), dim3(TPB), 0, (hipStream_t)stream, a);
  return check_launch("deva_input_head");
}
