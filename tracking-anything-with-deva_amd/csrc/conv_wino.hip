// 3x3 / stride 1 / pad 1 convolutions as Winograd F(2x2, 3x3) on the fp32 matrix pipes (v_mfma_f32_32x32x2_f32).
//
// Why: the fp32 MFMA runs at 1/16 of the f16 rate and is THE bound of the fp32 frame (the five big 3x3 layers of the
// decoder / value encoder are 76 % of the 480p / 5-object frame at 0.86 - 0.87 of the matrix peak: nothing left to
// schedule).  F(2x2, 3x3) computes a 2x2 output tile from 16 instead of 36 multiply-adds per input channel: 2.25x fewer
// MFMAs, with transforms whose constants are 0, +-1, +-1/2.  A quarter of the accumulated terms: the error against fp64 is
// BELOW the direct kernels' on the layer shapes of the network (2.5e-7 - 1.4e-6 of the output range against 4.3e-7 - 4.8e-6;
// bound 2e-5 in tests/test_gpu_a_conv.py).  Unlike on the f16 pipes the transform is cheap here:
// ~40 VALU instructions per 32 MFMAs (2 048 matrix-pipe cycles) and wave.
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        per (output channel, 2x2 tile); g 3x3, d the 4x4 input patch
//
// GEMM view: 16 independent GEMMs (one per transform position p = 4 i + l), M = cout, N = tiles (batch-major, row-major
// inside an image), K = input channels.  A workgroup = 8 waves = 64 output channels x 64 tiles; K advances in steps of 8
// channels through two double-buffered LDS tiles (2 x 2 x 32 KB):
//   * transformed weights U (deva_conv_pack_wino: [c/8][p][c%2][cout_pad][c%8/2], i.e. the four k values a lane feeds to the
//     four MFMAs of a position are one 16-byte read) go global -> LDS as they are;
//   * activations: thread (channel, tile) loads the 4x4 patch of its tile (four unaligned 16-byte buffer loads from
//     guard-banded inputs; rows outside the image are out-of-range offsets and come back as zeros), applies ReLU-on-load
//     and B^T d B in registers (16 packed additions) and writes the 16 transformed values as [p][k parity][k/2][tile] (a wave
//     = one channel of all 64 tiles: contiguous loads, conflict-free stores; the B fragment of a position is four 4-byte reads);
//   * per position: one ds_read_b128 (A) + two ds_read2st64_b32 (B) + four MFMAs.
// Output stage: the position sums of a (channel, tile) pair sit in ONE lane (same register index of the accumulators):
// A^T M A is additions in registers, then bias / residual / activation and two 8-byte stores per output channel.
#include <cstdlib>

#include "conv_args.h"

namespace deva {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// packed fp32 additions, written out: on scalars the compiler prefers 2 x v_add_f32 (and builds shuffled pairs with moves)
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

constexpr int WM = 64, WN = 64;  // output channels x tiles of a workgroup
constexpr int KC = 8;            // channels per K step
constexpr int TILE_FLOATS = 16 * 2 * 64 * 4;  // one operand tile of a K step: [p][k parity][row / column][4]

struct WinoArgs {
  const float* in0;
  const float* in1;
  int64_t bs0, bs1;
  int c0, ctot;
  int H, W;
  int tiles_x, tiles_per_img, n_tiles;  // 2x2 output tiles
  const float* u;  // transformed weights
  const float* bias;
  int cout, cout_pad;
  int relu_in;
  const float* res;
  int64_t res_bs;
  int act;
  float* out;
  int blocks_m;
  int by_tiles;  // grid numbered XCD-major (see the kernel)
};


// TWO waves per SIMD (8 waves, 512 threads), and why: a single wave cannot keep the fp32 matrix pipe busy.  A register-only
// stream of v_mfma_f32_32x32x2_f32 from one wave per SIMD measures 0.79 busy (profiles/pmc_r06/effective_clock.json: the
// probe); the first form of this kernel -- 4 waves, each with all SIXTEEN accumulators of its quadrant in 256 AGPRs -- ran at
// exactly that rate with everything but its MFMAs switched off and at 0.58 busy as a whole; the direct kernels (two and more
// waves per SIMD) reach 0.89.  Here the 16 transform positions are split between the two waves of a SIMD: wave (ph, wq)
// holds the EIGHT accumulators of positions 8 ph .. 8 ph + 7 (rows 2 ph, 2 ph + 1 of M) for quadrant wq (32 channels x 32
// tiles) -- 128 accumulator registers + ~85 others.  No operand is read twice (the LDS tiles are indexed by position).
// Measured against the 4-wave form on one box (tools/convlab --wino, us): up_8_4 256 -> 256 at 120x216 x5 725 -> 667 (742 ->
// 675 with a residual, 786 -> 659 with ReLU-on-load), GRU 1024 -> 1536 1 011 -> 925, fuser 512 -> 512 175 -> 161, up_16_8
// 363 -> 315 / 348 -> 319.
// Staging: thread = (ONE channel = wave, tile = lane): four patch rows, 16 packed additions, 16 LDS stores; four 16-byte
// weight chunks.  One register set: the loads of step s + 1 go out at the top of step s and are transformed / stored behind
// its MFMAs -- the other wave of the SIMD covers what the wait costs.  Measured and dropped: the two waves of a SIMD running a
// step in opposite order, one staging first and one last (0 - 5 % slower).  RELU / RES are compile-time: ReLU-on-load is ONE instruction per
// element (median of x, 0, limit: limit = +inf, or 0 for a masked column -- no NaN canonicalisation in front of it as with
// v_max), and with it the staging pays behind the third MFMA group instead of the last (2 % on those layers).
// Output stage: A^T M A is linear in the rows of M, so each half reduces its own rows to a partial 2x2 output in registers,
// the ph = 1 waves hand theirs over through LDS (64 KB, the weight tiles' space) and the ph = 0 waves add, apply bias /
// residual / activation and store.  Residual and bias of eight channels are fetched together, up front: a load issued
// between the stores is waited for in full (`out` may alias `res`, so the compiler keeps every load behind the stores in
// front of it -- in the first form that was 16 exposed round trips per workgroup, +20 % on a 32-step layer).
// Grid: cout blocks fastest -- the workgroups that share an activation tile run side by side.  Workgroup b runs on XCD b % 8,
// each with its own L2: as it is, XCD x sees the cout blocks = x (mod 8) of EVERY tile block -- 1/8 of the weights, all the
// activations.  Right when the weights are the bigger operand (GRU: 100 MB of U against 33 MB); when the activations are
// (up_8_4: 132 MB against 4 MB) p.by_tiles renumbers the grid so that an XCD gets a contiguous range of TILE blocks with all
// their cout blocks, one after the other: an activation tile enters one L2, not blocks_m of them (1 - 4 %).
template <bool RELU, bool RES>
__global__ __launch_bounds__(512, 1) void conv_wino_kernel(const WinoArgs p) {
  __shared__ __attribute__((aligned(16))) float sA[2][TILE_FLOATS];
  __shared__ __attribute__((aligned(16))) float sB[2][TILE_FLOATS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int wq = wave & 3, ph = wave >> 2;
  const int wm = wq >> 1, wn = wq & 1;
  int lb = blockIdx.x;
  if (p.by_tiles) {
    const int nb = gridDim.x, x = lb & 7, i = lb >> 3;
    lb = x * (nb >> 3) + min(x, nb & 7) + i;
  }
  const int block_m = lb % p.blocks_m, block_n = lb / p.blocks_m;
  const int m0 = block_m * WM, n0 = block_n * WN;

  // ---- activation staging: thread = (channel c8 = wave of the step's eight, tile st = lane)
  const int c8 = wave, st = lane;
  const int sm = c8 >> 1, sh = c8 & 1;  // k index inside the 16-byte weight chunk, k parity (the MFMA's two k values)
  const int64_t HW = (int64_t)p.H * p.W;
  int poff[4];             // byte offsets of the four patch rows inside a step's first channel plane (out of range: zeros)
  int poff_b0, poff_b1;    // batch item + first column inside in0 / in1 (may be -4: the guard band)
  bool lcol, rcol;
  {
    const int n = min(n0 + st, p.n_tiles - 1);
    const int b = n / p.tiles_per_img;
    const int r = n - b * p.tiles_per_img;
    const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
    lcol = x0 >= 0;
    rcol = x0 + 3 < p.W;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      poff[i] = (unsigned)(y0 + i) < (unsigned)p.H ? (int)(((int64_t)c8 * HW + (int64_t)(y0 + i) * p.W) * 4) : (int)0x80000000;
    poff_b0 = (int)(((int64_t)b * p.bs0 + x0) * 4);
    poff_b1 = (int)(((int64_t)b * p.bs1 + x0) * 4);
  }
  float pinf = __builtin_inff();
  asm("" : "+v"(pinf));  // (a limit the compiler cannot see through: median(x, 0, +inf) folds to a max WITH the canonicalising max in front)
  const float llim = lcol ? __builtin_inff() : 0.0f, rlim = rcol ? __builtin_inff() : 0.0f;
  const int aoff = ((tid >> 6) * p.cout_pad + (tid & 63)) * 16;  // segment (p, k parity) = tid / 64 + 8 i, chunk tid % 64
  const int astride = 8 * p.cout_pad * 16;

  f32x16 acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;

  const int ksteps = p.ctot / KC;
  f32x4 ra[4], rb[4];
  auto load_step = [&](int s) {
    const float* ub = p.u + ((int64_t)s * 32 * p.cout_pad + m0) * 4;
    const __amdgpu_buffer_rsrc_t ru = make_rsrc(ub, 0x7fffffff);
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = buf_load4(ru, aoff, i * astride);
    const int c = s * KC;
    const bool first = c < p.c0;
    const float* base = (first ? p.in0 + (int64_t)c * HW : p.in1 + (int64_t)(c - p.c0) * HW) - 4;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(base, 0x7fffffff);
    const int pb = (first ? poff_b0 : poff_b1) + 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) rb[i] = buf_load4(rx, pb + poff[i], 0);
  };
  auto store_step = [&](int buf) {
    float* a = sA[buf];
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(a + (tid + 512 * i) * 4) = ra[i];
    float* bdst = sB[buf] + (sh * 4 + sm) * 64 + st;  // position q at + q * 8 * 64
    f32x2 dl[4], dr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 v = rb[i];
      if (RELU) {
        dl[i] = f32x2{__builtin_amdgcn_fmed3f(v[0], 0.0f, llim), __builtin_amdgcn_fmed3f(v[1], 0.0f, pinf)};
        dr[i] = f32x2{__builtin_amdgcn_fmed3f(v[2], 0.0f, pinf), __builtin_amdgcn_fmed3f(v[3], 0.0f, rlim)};
      } else {
        dl[i] = f32x2{lcol ? v[0] : 0.0f, v[1]};
        dr[i] = f32x2{v[2], rcol ? v[3] : 0.0f};
      }
    }
    f32x2 wl[4], wr[4];
    wl[0] = pk_sub(dl[0], dl[2]);
    wl[1] = pk_add(dl[1], dl[2]);
    wl[2] = pk_sub(dl[2], dl[1]);
    wl[3] = pk_sub(dl[1], dl[3]);
    wr[0] = pk_sub(dr[0], dr[2]);
    wr[1] = pk_add(dr[1], dr[2]);
    wr[2] = pk_sub(dr[2], dr[1]);
    wr[3] = pk_sub(dr[1], dr[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x2 lo, hi;  // (w0 - w2, w1 + w2), (w2 - w1, w1 - w3) of the row (w0, w1 | w2, w3)
      asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(lo) : "v"(wl[i]), "v"(wr[i]));
      asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(hi) : "v"(wr[i]), "v"(wl[i]));
      bdst[(4 * i + 0) * 8 * 64] = lo[0];
      bdst[(4 * i + 1) * 8 * 64] = lo[1];
      bdst[(4 * i + 2) * 8 * 64] = hi[0];
      bdst[(4 * i + 3) * 8 * 64] = hi[1];
    }
  };

  // fragments of position 8 ph + j: A = 16 bytes (the four k values of the lane's channel and k parity), B = four floats
  f32x4 fa[2][2], fb[2][2];
  auto read_frag = [&](int buf, int set, int g) {
    const float* a_rd = sA[buf] + ((8 * ph) * 2 * 64 + half * 64 + wm * 32 + l31) * 4;
    const float* b_rd = sB[buf] + ((8 * ph) * 2 + half) * 4 * 64 + wn * 32 + l31;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = 2 * g + j;
      fa[set][j] = *reinterpret_cast<const f32x4*>(a_rd + q * 2 * 64 * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) fb[set][j][e] = b_rd[(q * 2 * 4 + e) * 64];
    }
  };
  auto multiply = [&](int set, int g) {  // positions 2 g, 2 g + 1: consecutive MFMAs never share an accumulator
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[2 * g + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][j][e], fb[set][j][e], acc[2 * g + j], 0, 0, 0);
  };
  // A step runs ACROSS the workgroup barrier: the fragments of its last two positions are in registers before the barrier,
  // their MFMAs are issued behind it and cover the LDS latency of the next step's first fragments (both waves of a SIMD
  // arrive at the barrier together: without this the matrix pipe idles for a round trip at every step)
  load_step(0);
  store_step(0);
  __syncthreads();
  read_frag(0, 0, 0);
  for (int s = 0; s < ksteps; ++s) {
    const int buf = s & 1;
    const bool more = s + 1 < ksteps;
    if (more) load_step(s + 1);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      read_frag(buf, (g + 1) & 1, g + 1);
      multiply(g & 1, g);
      if (RELU && g == 2 && more) store_step(buf ^ 1);
    }
    if (!RELU && more) store_step(buf ^ 1);
    __syncthreads();
    if (more) read_frag(buf ^ 1, 0, 0);
    multiply(1, 3);
  }

  // ---- output stage.  Rows of M held by this half -> partial t0 = (A^T M)[0], t1 = (A^T M)[1], then the column pass:
  //   ph = 0: t0 = M0 + M1, t1 = M1;      ph = 1: t0 = M2, t1 = -M2 - M3
  float* xch = sA[0] + (wq * 16 * 4) * 64 + lane;  // [quadrant][r][a * 2 + j][lane]: 64 KB = both weight tiles
  const int n = n0 + wn * 32 + l31;
  const bool n_ok = n < p.n_tiles;
  if (ph == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float y[2][2];
      {
        const float a0 = acc[0][r], a1 = acc[1][r], a2 = acc[2][r], a3 = acc[3][r];
        y[0][0] = a0 + a1 + a2;
        y[0][1] = a1 - a2 - a3;
        const float b0 = -acc[0][r] - acc[4][r], b1 = -acc[1][r] - acc[5][r], b2 = -acc[2][r] - acc[6][r], b3 = -acc[3][r] - acc[7][r];
        y[1][0] = b0 + b1 + b2;
        y[1][1] = b1 - b2 - b3;
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 2; ++j) xch[(r * 4 + a * 2 + j) * 64] = y[a][j];
    }
  }
  __syncthreads();
  if (ph == 1 || !n_ok) return;
  int64_t o_base;
  int r_off;
  bool row1;  // the tile's second output row exists (odd heights: the last tile row has one)
  {
    const int b = n / p.tiles_per_img;
    const int rr = n - b * p.tiles_per_img;
    const int ty = rr / p.tiles_x, tx = rr - ty * p.tiles_x;
    const int64_t pix = (int64_t)(2 * ty) * p.W + 2 * tx;
    o_base = (int64_t)b * p.cout * HW + pix;
    r_off = (int)(((int64_t)b * p.res_bs + pix) * 4);
    row1 = 2 * ty + 1 < p.H;
  }
  const __amdgpu_buffer_rsrc_t rres = make_rsrc(RES ? p.res : p.out, RES ? 0x7fffffff : 0);
#pragma unroll
  for (int rh = 0; rh < 2; ++rh) {  // eight channels at a time: their residual rows are fetched together
    f32x2 rv[8][2];
    float bv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = rh * 8 + k;
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      bv[k] = p.bias ? p.bias[min(m, p.cout - 1)] : 0.0f;
      if (RES) {
        const int mo = m < p.cout ? r_off + (int)((int64_t)m * HW * 4) : (int)0x80000000;
#pragma unroll
        for (int a = 0; a < 2; ++a)
          rv[k][a] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rres, (a == 0 || row1) ? mo + a * p.W * 4 : (int)0x80000000, 0, 0));
      } else {
        rv[k][0] = rv[k][1] = f32x2{0.0f, 0.0f};
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = rh * 8 + k;
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (m >= p.cout) continue;
      float y[2][2];
      {
        const float a0 = acc[0][r] + acc[4][r], a1 = acc[1][r] + acc[5][r], a2 = acc[2][r] + acc[6][r], a3 = acc[3][r] + acc[7][r];
        y[0][0] = a0 + a1 + a2;
        y[0][1] = a1 - a2 - a3;
        const float b0 = acc[4][r], b1 = acc[5][r], b2 = acc[6][r], b3 = acc[7][r];
        y[1][0] = b0 + b1 + b2;
        y[1][1] = b1 - b2 - b3;
      }
      const int64_t o = o_base + (int64_t)m * HW;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        f32x2 v;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float x = (y[a][j] + xch[(r * 4 + a * 2 + j) * 64]) + bv[k] + rv[k][a][j];
          if (p.act == DEVA_ACT_RELU) {
            x = fmaxf(x, 0.0f);
          } else if (p.act == DEVA_ACT_SIGMOID) {
            x = sigmoidf_(x);
          } else if (p.act == DEVA_ACT_SQUARE_PLUS_ONE) {
            x = x * x + 1.0f;
          }
          v[j] = x;
        }
        if (a == 0 || row1) *reinterpret_cast<f32x2*>(p.out + o + (int64_t)a * p.W) = v;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace

// -> 0 launched, 1 launch error, -1 not eligible (the caller runs the direct kernels)
int launch_conv_wino(const ConvArgs& a, const float* u, hipStream_t st) {
  if (!u || !a.vec_ok || a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || (a.W & 1) || a.W < 4) return -1;  // (odd heights: a last tile row of one output row)
  if (a.c0 % KC || a.ctot % KC || a.cout < 32) return -1;
  if (a.in0_span >= (1ll << 29) || a.in1_span >= (1ll << 29)) return -1;  // 32-bit byte offsets inside a source (buffer loads)
  if (a.res && (int64_t)(a.n_total / a.OHW) * a.res_bs >= (1ll << 29)) return -1;  // ... and inside the residual
  WinoArgs p;
  p.in0 = a.in0;
  p.in1 = a.in1 ? a.in1 : a.in0;
  p.bs0 = a.bs0;
  p.bs1 = a.bs1;
  p.c0 = a.c0;
  p.ctot = a.ctot;
  p.H = a.H;
  p.W = a.W;
  p.tiles_x = a.W / 2;
  p.tiles_per_img = ((a.H + 1) / 2) * (a.W / 2);
  const int batch = a.n_total / a.OHW;
  p.n_tiles = batch * p.tiles_per_img;
  p.u = u;
  p.bias = a.bias;
  p.cout = a.cout;
  p.cout_pad = (a.cout + 63) / 64 * 64;
  p.relu_in = a.relu_in;
  p.res = a.res;
  p.res_bs = a.res_bs;
  p.act = a.act;
  p.out = a.out;
  p.blocks_m = p.cout_pad / WM;
  p.by_tiles = (int64_t)batch * a.HW > 16ll * p.cout_pad;  // activation elements per channel > transformed weights per channel
#if defined(DEVA_CONV_PROBES) || defined(DEVA_WINO_TUNE)  // (`make EXTRA=-DDEVA_WINO_TUNE`: the threshold alone, kernels as shipped)
  static const int min_blocks_probe = [] {
    const char* e = getenv("DEVA_WINO_MIN_BLOCKS");
    return e ? atoi(e) : 160;
  }();
  const int min_blocks = min_blocks_probe;
  static const int by_tiles_probe = [] {
    const char* e = getenv("DEVA_WINO_BY_TILES");
    return e ? atoi(e) : -1;
  }();
  if (by_tiles_probe >= 0) p.by_tiles = by_tiles_probe;
#else
  const int min_blocks = 160;
#endif
  const int64_t blocks = (int64_t)p.blocks_m * ceil_div(p.n_tiles, WN);
  if (blocks < min_blocks) return -1;  // one workgroup per CU: fewer than ~2/3 of the CUs and the direct kernels' split-K wins
  const dim3 grid((unsigned)blocks), block(512);
  if (p.relu_in) {
    if (p.res) {
      hipLaunchKernelGGL((conv_wino_kernel<true, true>), grid, block, 0, st, p);
    } else {
      hipLaunchKernelGGL((conv_wino_kernel<true, false>), grid, block, 0, st, p);
    }
  } else {
    if (p.res) {
      hipLaunchKernelGGL((conv_wino_kernel<false, true>), grid, block, 0, st, p);
    } else {
      hipLaunchKernelGGL((conv_wino_kernel<false, false>), grid, block, 0, st, p);
    }
  }
  return check_launch("deva_conv2d (Winograd F(2x2, 3x3))");
}

}  // namespace deva

// Host-side weight transform (model load): w_oihw [cout][cin][3][3] (BatchNorm folded), cin % 8 == 0 -> U = G g G^T in the
// layout the kernel stages: element (c, p = 4 i + l, m) at ((((c/8)*16 + p)*2 + c%2)*cout_pad64 + m)*4 + (c%8)/2, cout padded
// to a multiple of 64 with zeros; computed in fp64, rounded once.  Returns the number of floats (out == NULL: size query), -1 when
// the layer is not eligible.
extern "C" int64_t deva_conv_pack_wino(const float* w_oihw, float* out, int cout, int cin) {
  using namespace deva;
  if (!w_oihw || cout <= 0 || cin <= 0) {
    set_error("deva_conv_pack_wino: bad arguments");
    return -1;
  }
  if (cin % 8) return -1;
  const int cout_pad = (cout + 63) / 64 * 64;
  const int64_t elems = (int64_t)(cin / 8) * 16 * 2 * cout_pad * 4;
  if (!out) return elems;
  for (int64_t i = 0; i < elems; ++i) out[i] = 0.0f;
  static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  for (int m = 0; m < cout; ++m)
    for (int c = 0; c < cin; ++c) {
      const float* g = w_oihw + ((int64_t)m * cin + c) * 9;
      double t[4][3];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0 * 3 + j] + G[i][1] * g[1 * 3 + j] + G[i][2] * g[2 * 3 + j];
      for (int i = 0; i < 4; ++i)
        for (int l = 0; l < 4; ++l) {
          const double u = t[i][0] * G[l][0] + t[i][1] * G[l][1] + t[i][2] * G[l][2];
          const int q = 4 * i + l;
          out[((((int64_t)(c / 8) * 16 + q) * 2 + (c & 1)) * cout_pad + m) * 4 + (c % 8) / 2] = (float)u;
        }
    }
  return elems;
}
