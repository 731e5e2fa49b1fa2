// The 7x7 stride-2 stems (resnet.py:117-122 `conv1` of the key encoder, big_modules.py:58-61,103-107 `conv1` of the
// value encoder over cat(image, mask)) as a DIRECT convolution on the f16 matrix pipes, fp32-accurate through the hi/lo
// operand split of conv_f16.hip (PREC 2) -- the kernel behind --f16_split / --f16_split_key_encoder for these two layers.
//
// Why a kernel of its own: with 3 / 4 input channels the implicit-GEMM machinery has nothing to tile (K = 147 / 196 taps
// that change source pixel with every k: conv_mfma.hip KIND 3 gathers them one scalar load per element and runs at
// 29-38 TFLOP/s), while the work is tiny for the matrix pipes and the layer is bound by its 64-channel output stream.
// Here a workgroup (8 waves; one per CU, walking over the tiles) owns 8 output rows x 64 output columns of all 64 channels:
//   * the input patch (22 rows x 134 columns per channel) is read ONCE with coalesced row loads, split into hi = fp16(x),
//     lo = fp16(x - hi) planes and kept in LDS as [plane][channel][row][column] halfs;
//   * K is ordered (channel, dy pair, dy parity, dx 0..7) with the 8th row / column tap a zero weight: a lane's B
//     fragment of one K-block -- 8 consecutive k = 8 consecutive dx of one patch row at column 2*ow -- is 16 contiguous
//     bytes of the patch (four ds_read_b32: the address is 4-byte aligned), lanes 0-31 take the even dy of the pair,
//     lanes 32-63 the odd one; no im2col, no per-element address arithmetic: every fragment address is a lane constant
//     plus an immediate;
//   * the weights (hi / lo planes of w 2^e, deva_stem_pack) sit in LDS in the fragment layout of conv_f16.hip
//     ([k/8][plane][64][8]); each K-block issues hi.hi + hi.lo + lo.hi into the fp32 accumulators (C*4 K-blocks: 36 / 48
//     MFMAs per 32x32 output block against 84 / 112 fp32 MFMAs at 1/16 of the rate);
//   * wave w computes output row w of the tile (2 column blocks x 2 channel blocks) and writes it with bias / ReLU.
// An input beyond the fp16 range makes an accumulator non-finite; the workgroup that sees one recomputes ITS tile with
// plain fp32 FMAs from the fp32 weights (one thread per output pixel; never taken for normalised images and mask
// probabilities) and raises the caller's flag for the statistics -- no second launch.
#include "common.h"

namespace deva {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int STEM_COUT = 64;
constexpr int TR = 8, TC = 64;           // output rows x columns of a workgroup
constexpr int PR = 2 * TR + 6;           // 22 patch rows: 2r + dy - 3, dy = 0..7 (the 8th tap is a zero weight)
constexpr int PC = 2 * TC + 6;           // 134 patch columns
constexpr int PCP = 136;                 // row pitch (halfs)
constexpr int STEM_THREADS = 64 * TR;

struct StemArgs {
  const float* in0;   // [b0][c0][H][W], b0 = 1 (broadcast) or batch
  const float* in1;   // [batch][c1][H][W] or null
  int64_t bs0, bs1;   // batch strides (elements; bs0 = 0 broadcasts)
  int c0, c1;
  int H, W, OH, OW;
  const uint16_t* w16;  // [C*8][2][64][8] halfs (deva_stem_pack)
  const float* w32;     // [C*49][64] fp32 (k = (c*7 + dy)*7 + dx), the in-kernel fall-back
  const float* bias;    // [64] or null
  float out_scale;
  int relu;
  float* out;           // [batch][64][OH][OW]
  int tiles_x, tiles_y, total_tiles;  // total = tiles_x * tiles_y * batch: the workgroups walk over them
  int* flag;
};

template <int C>
__global__ __launch_bounds__(STEM_THREADS, 1) void stem7x7_kernel(const StemArgs p) {
  constexpr int KB = C * 4;                       // K-blocks of 16
  constexpr int W_HALFS = KB * 2 * 2 * STEM_COUT * 8;
  constexpr int P_PLANE = C * PR * PCP;           // halfs of one patch plane
  __shared__ __attribute__((aligned(16))) _Float16 s_w[W_HALFS];
  __shared__ __attribute__((aligned(16))) _Float16 s_p[2 * P_PLANE];
  __shared__ int s_bad;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;

  // ---- weights -> LDS once per workgroup (the fragment layout as packed); the workgroup then walks over its tiles
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(p.w16);
    u32x4* dst = reinterpret_cast<u32x4*>(s_w);
    constexpr int V = W_HALFS / 8;
    for (int i = tid; i < V; i += STEM_THREADS) dst[i] = src[i];
  }

  // the input patch of a tile: pairs of adjacent columns per thread (one 4-byte LDS store per plane); the loads of the NEXT
  // tile are issued before the MFMAs of the current one and land in registers while those run
  constexpr int PAIRS = PCP / 2;  // 68 column pairs per row (the last one is padding)
  constexpr int TASKS = C * PR * PAIRS;
  constexpr int ITERS = (TASKS + STEM_THREADS - 1) / STEM_THREADS;
  float v0[ITERS], v1[ITERS];
  const int tiles_per_image = p.tiles_x * p.tiles_y;
  auto load_patch = [&](int tile) {
    const int b = tile / tiles_per_image;
    const int rem = tile - b * tiles_per_image;
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const int row_in0 = 2 * ty * TR - 3, col_in0 = 2 * tx * TC - 3;
    int tv = tid;
    asm volatile("" : "+v"(tv));  // (opaque per call: the per-task indices are recomputed per tile -- a few VALU instructions --
                                  // instead of being hoisted out of the tile loop, where they cost 60 spilled registers)
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int t = tv + it * STEM_THREADS;
      const int c = t / (PR * PAIRS);
      const int r2 = t - c * (PR * PAIRS);
      const int pr = r2 / PAIRS, pp = r2 - pr * PAIRS;
      const int ih = row_in0 + pr, iw = col_in0 + 2 * pp;
      const float* src = c < p.c0 ? p.in0 + (int64_t)b * p.bs0 + (int64_t)c * p.H * p.W
                                  : p.in1 + (int64_t)b * p.bs1 + (int64_t)(c - p.c0) * p.H * p.W;
      const bool row_ok = t < TASKS && (unsigned)ih < (unsigned)p.H;
      const bool ok0 = row_ok && (unsigned)iw < (unsigned)p.W && 2 * pp < PC;
      const bool ok1 = row_ok && (unsigned)(iw + 1) < (unsigned)p.W && 2 * pp + 1 < PC;
      v0[it] = ok0 ? src[(int64_t)ih * p.W + iw] : 0.0f;
      v1[it] = ok1 ? src[(int64_t)ih * p.W + iw + 1] : 0.0f;
    }
  };
  auto store_patch = [&]() {  // -> hi / lo planes, [c][pr][2 pp] in the order of the tasks
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int t = tid + it * STEM_THREADS;
      if (t >= TASKS) break;
      const h2 hi = {(_Float16)v0[it], (_Float16)v1[it]};
      const h2 lo = {(_Float16)(v0[it] - (float)hi[0]), (_Float16)(v1[it] - (float)hi[1])};
      _Float16* at = s_p + 2 * t;
      *reinterpret_cast<h2*>(at) = hi;
      *reinterpret_cast<h2*>(at + P_PLANE) = lo;
    }
  };

  const _Float16* a_rd = s_w + (half * 2 * STEM_COUT + l31) * 8;     // + ((kb*2*2 + plane) * 64 + 32 i) * 8
  const _Float16* b_rd = s_p + ((2 * wave + half) * PCP + 2 * l31);  // + (c*PR + 2 dyp) * PCP + 64 j, + plane * P_PLANE
  int tile = blockIdx.x;
  if (tile < p.total_tiles) load_patch(tile);
  for (; tile < p.total_tiles; tile += gridDim.x) {
    const int b = tile / tiles_per_image;
    const int rem = tile - b * tiles_per_image;
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const int r0 = ty * TR, c0 = tx * TC;
    if (tid == 0) s_bad = 0;
    store_patch();
    __syncthreads();
    if (tile + (int)gridDim.x < p.total_tiles) load_patch(tile + gridDim.x);
    __builtin_amdgcn_sched_barrier(0);  // (the loads are issued here; their address arithmetic does not drift into the MFMAs)

    // ---- MFMAs: wave w = output row r0 + w; column blocks j = 0, 1 (32 columns each), channel blocks i = 0, 1
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int c = kb >> 2, dyp = kb & 3;
      h8 fa[2][2], fb[2][2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[pl][i] = *reinterpret_cast<const h8*>(a_rd + ((kb * 4 + pl) * STEM_COUT + 32 * i) * 8);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const _Float16* q = b_rd + pl * P_PLANE + (c * PR + 2 * dyp) * PCP + 64 * j;
          u32x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = *reinterpret_cast<const unsigned*>(q + 2 * e);
          fb[pl][j] = __builtin_bit_cast(h8, w);
        }
      }
#pragma unroll
      for (int term = 0; term < 3; ++term)  // hi.hi, hi.lo, lo.hi
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[term == 2 ? 1 : 0][i], fb[term == 1 ? 1 : 0][j], acc[i][j], 0, 0, 0);
      if (kb & 1) __builtin_amdgcn_sched_barrier(0);  // (fragments of at most two K-blocks in flight: the full unroll otherwise spills)
    }

    bool bad = false;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) bad |= (__builtin_bit_cast(unsigned, acc[i][j][r]) & 0x7f800000u) == 0x7f800000u;
    if (__builtin_amdgcn_ballot_w64(bad) && lane == 0) s_bad = 1;
    __syncthreads();
    const int oh = r0 + wave;
    if (s_bad == 0) {
      if (oh < p.OH) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ow = c0 + 32 * j + l31;
          if (ow >= p.OW) continue;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int m = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
              float v = acc[i][j][r] * p.out_scale + (p.bias ? p.bias[m] : 0.0f);
              if (p.relu) v = fmaxf(v, 0.0f);
              p.out[(((int64_t)b * STEM_COUT + m) * p.OH + oh) * p.OW + ow] = v;
            }
        }
      }
    } else {
      // ---- an input of this tile lies beyond the fp16 range: the tile again in plain fp32 (thread = output pixel)
      if (tid == 0 && p.flag) atomicOr(p.flag, 1);
      const int ow = c0 + lane;
      if (oh < p.OH && ow < p.OW) {
#pragma nounroll
        for (int m0 = 0; m0 < STEM_COUT; m0 += 16) {  // 16 channels at a time: the path is rare, its registers are not
          float out[16];
#pragma unroll
          for (int m = 0; m < 16; ++m) out[m] = p.bias ? p.bias[m0 + m] : 0.0f;
#pragma nounroll
          for (int c = 0; c < C; ++c) {
            const float* src = c < p.c0 ? p.in0 + (int64_t)b * p.bs0 + (int64_t)c * p.H * p.W
                                        : p.in1 + (int64_t)b * p.bs1 + (int64_t)(c - p.c0) * p.H * p.W;
#pragma nounroll
            for (int dy = 0; dy < 7; ++dy) {
              const int ih = 2 * oh + dy - 3;
#pragma nounroll
              for (int dx = 0; dx < 7; ++dx) {
                const int iw = 2 * ow + dx - 3;
                const float x = ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) ? src[(int64_t)ih * p.W + iw] : 0.0f;
                const float* wk = p.w32 + (int64_t)((c * 7 + dy) * 7 + dx) * STEM_COUT + m0;
#pragma unroll
                for (int m = 0; m < 16; ++m) out[m] = __builtin_fmaf(x, wk[m], out[m]);
              }
            }
          }
#pragma unroll
          for (int m = 0; m < 16; ++m)
            p.out[(((int64_t)b * STEM_COUT + m0 + m) * p.OH + oh) * p.OW + ow] = p.relu ? fmaxf(out[m], 0.0f) : out[m];
        }
      }
      // (the prefetched patch is loaded again here so that its registers are free during the 64 accumulators above)
      if (tile + (int)gridDim.x < p.total_tiles) load_patch(tile + gridDim.x);
    }
    __syncthreads();  // every wave is done with this tile's patch (and with s_bad)
  }
}

}  // namespace
}  // namespace deva

using namespace deva;

// Host-side packing of a stem's weights (model load): w_oihw [64][cin][7][7] (BatchNorm folded), cin = 3 or 4 ->
//   planes [cin*8][2][64][8] uint16: hi / lo fp16 of w * 2^e at k = ((c*4 + dy/2)*2 + dy%2)*8 + dx (dy, dx = 7: zero),
//   w32    [cin*49][64] fp32 at k = (c*7 + dy)*7 + dx (the in-kernel fp32 fall-back),
//   *scale_log2 = e with max|w| 2^e in [2^13, 2^14).
// Returns the number of uint16 elements of `planes` (planes == NULL: size query), -1 on bad arguments.
extern "C" int64_t deva_stem_pack(const float* w_oihw, int cin, uint16_t* planes, float* w32, int* scale_log2) {
  if (!w_oihw || (cin != 3 && cin != 4) || !scale_log2) {
    set_error("deva_stem_pack: 64 x {3, 4} x 7 x 7 weights expected");
    return -1;
  }
  const int64_t elems = (int64_t)cin * 8 * 2 * STEM_COUT * 8;
  float wmax = 0.0f;
  for (int64_t i = 0; i < (int64_t)STEM_COUT * cin * 49; ++i) {
    const float v = fabsf(w_oihw[i]);
    if (!(v <= 3.0e38f)) {
      set_error("deva_stem_pack: non-finite weight");
      return -1;
    }
    if (v > wmax) wmax = v;
  }
  int e = 0;
  if (wmax > 0.0f) {
    int x;
    frexpf(wmax, &x);
    e = 14 - x;
    if (e > 120) e = 120;
    if (e < -120) e = -120;
  }
  *scale_log2 = e;
  if (!planes) return elems;
  if (!w32) {
    set_error("deva_stem_pack: w32 is null");
    return -1;
  }
  for (int64_t i = 0; i < elems; ++i) planes[i] = 0;
  for (int m = 0; m < STEM_COUT; ++m)
    for (int c = 0; c < cin; ++c)
      for (int dy = 0; dy < 7; ++dy)
        for (int dx = 0; dx < 7; ++dx) {
          const float w = w_oihw[(((int64_t)m * cin + c) * 7 + dy) * 7 + dx];
          w32[(int64_t)((c * 7 + dy) * 7 + dx) * STEM_COUT + m] = w;
          const float ws = ldexpf(w, e);
          const _Float16 hi = (_Float16)ws;
          const _Float16 lo = (_Float16)(ws - (float)hi);
          uint16_t bh, bl;
          __builtin_memcpy(&bh, &hi, 2);
          __builtin_memcpy(&bl, &lo, 2);
          const int64_t k8 = (int64_t)(c * 4 + dy / 2) * 2 + dy % 2;
          planes[((k8 * 2 + 0) * STEM_COUT + m) * 8 + dx] = bh;
          planes[((k8 * 2 + 1) * STEM_COUT + m) * 8 + dx] = bl;
        }
  return elems;
}

extern "C" int deva_stem7x7(const float* in0, int64_t in0_batch_stride, int c0, const float* in1, int64_t in1_batch_stride, int c1,
                            int batch, int height, int width, const uint16_t* planes, const float* w32, int scale_log2,
                            const float* bias, int relu, float* out, int32_t* flag, void* stream) {
  DEVA_REQUIRE(in0 && planes && w32 && out && batch > 0 && height > 0 && width > 0, "deva_stem7x7: bad arguments");
  DEVA_REQUIRE((c0 == 3 && (c1 == 0 || c1 == 1)) && (c1 == 0 || in1), "deva_stem7x7: 3 (+ 1) input channels expected");
  DEVA_REQUIRE(height % 2 == 0 && width % 2 == 0, "deva_stem7x7: even input size expected (frames are padded to x16)");
  DEVA_REQUIRE(scale_log2 >= -120 && scale_log2 <= 120, "deva_stem7x7: scale_log2 out of range");
  StemArgs a;
  a.in0 = in0;
  a.in1 = in1;
  a.bs0 = in0_batch_stride;
  a.bs1 = in1_batch_stride;
  a.c0 = c0;
  a.c1 = c1;
  a.H = height;
  a.W = width;
  a.OH = height / 2;
  a.OW = width / 2;
  a.w16 = planes;
  a.w32 = w32;
  a.bias = bias;
  a.out_scale = ldexpf(1.0f, -scale_log2);
  a.relu = relu;
  a.out = out;
  a.tiles_x = (int)ceil_div(a.OW, TC);
  a.tiles_y = (int)ceil_div(a.OH, TR);
  a.flag = flag;
  const int64_t total = (int64_t)a.tiles_x * a.tiles_y * batch;
  DEVA_REQUIRE(total < (1ll << 30), "deva_stem7x7: too many tiles");
  a.total_tiles = (int)total;
  static const int cus = [] {  // (one device type per process: asked once)
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  // one workgroup per CU (112 KB of LDS): persistent, the weights are staged once and the patch loads of the next tile
  // overlap the MFMAs of the current one
  const dim3 grid((unsigned)(total < cus ? total : cus));
  if (c0 + c1 == 3) {
    hipLaunchKernelGGL(stem7x7_kernel<3>, grid, dim3(STEM_THREADS), 0, (hipStream_t)stream, a);
  } else {
    hipLaunchKernelGGL(stem7x7_kernel<4>, grid, dim3(STEM_THREADS), 0, (hipStream_t)stream, a);
  }
  return check_launch("deva_stem7x7");
}
