// 3x3 / stride 1 / pad 1 convolutions as Winograd F(2x2, 3x3) on the fp32 matrix pipes (v_mfma_f32_32x32x2_f32).
//
// Why: the fp32 MFMA runs at 1/16 of the f16 rate and is THE bound of the fp32 frame (the five big 3x3 layers of the
// decoder / value encoder are 76 % of the 480p / 5-object frame at 0.86 - 0.87 of the matrix peak: nothing left to
// schedule).  F(2x2, 3x3) computes a 2x2 output tile from 16 instead of 36 multiply-adds per input channel: 2.25x fewer
// MFMAs, with transforms whose constants are 0, +-1, +-1/2 (error against fp64 ~2x the direct kernel's: 5e-7 of the output
// range on the layer shapes of the network, bound 2e-5 in tests/test_gpu_a_conv.py).  Unlike on the f16 pipes (DESIGN.md
// section 8) the transform is cheap here: ~100 VALU cycles per 1 024 MFMA cycles.
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        per (output channel, 2x2 tile); g 3x3, d the 4x4 input patch
//
// GEMM view: 16 independent GEMMs (one per transform position p = 4 i + l), M = cout, N = tiles (batch-major, row-major
// inside an image), K = input channels.  A workgroup = 4 waves = 64 output channels x 64 tiles; a wave holds the SIXTEEN
// 32x32 accumulators of its 32 channels x 32 tiles (256 registers: one wave per SIMD).  K advances in steps of 8 channels:
//   * transformed weights U (deva_conv_pack_wino: [c/8][p][c%2][cout_pad][c%8/2], i.e. the four k values a lane feeds to the
//     four MFMAs of a position are one 16-byte read) go global -> LDS as they are;
//   * activations: thread (tile, channel pair) loads the 4x4 patch of its tile for two channels (four unaligned 16-byte
//     loads each from guard-banded inputs, zeroed outside the image), applies ReLU-on-load and B^T d B in registers
//     (32 additions per channel) and writes the 16 transformed values as [p][k parity][k/2][tile] (a wave = one channel pair
//     of all 64 tiles: contiguous loads, conflict-free stores; the B fragment of a position is four 4-byte reads);
//   * per position: two ds_read_b128 (A, B) + four MFMAs; both tiles are double-buffered, the loads of step s+1 are in flight
//     under the MFMAs of step s.
// Output stage: the 16 position sums of a (channel, tile) pair sit in ONE lane (same register index of the 16 accumulators):
// A^T M A is 24 additions in registers, then bias / residual / activation and two 8-byte stores per output channel.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "conv_args.h"

namespace deva {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef f32x4 f32x4_u __attribute__((aligned(4)));

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

// f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}): loop bodies whose index is a compile-time
// constant (register-set and buffer indices)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
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
  int ablate;  // `make PROBES=1` builds only (DEVA_WINO_ABLATE): timing runs with parts of the K loop switched off
};

#ifdef DEVA_CONV_PROBES  // timing-only ablations (results are wrong): 1 activation transform + stores, 2 weight stores, 4 global loads, 8 barrier, 16 fragment reads
#define WINO_ABL(bit) (p.ablate & (bit))
#else
#define WINO_ABL(bit) false
#endif

// RELU: ReLU on the input elements (F.relu before the convolution); RES: a residual is added.  The two choose between code
// paths that compute the same thing: the kernel is ONE wave per SIMD at the register limit and its speed follows the exact
// schedule -- each variant keeps the form that measured fastest for it (tools/convlab --wino).
template <bool RELU, bool RES>
__global__ __launch_bounds__(256, 1) void conv_wino_kernel(const WinoArgs p) {
  __shared__ __attribute__((aligned(16))) float sA[2][TILE_FLOATS];
  __shared__ __attribute__((aligned(16))) float sB[2][TILE_FLOATS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;  // 32 channels x 32 tiles of the wave
  // cout blocks fastest: the workgroups that share an activation tile run side by side.  Workgroup b runs on XCD b % 8, each
  // with its own L2: as it is, XCD x sees the cout blocks = x (mod 8) of EVERY tile block -- 1/8 of the weights, all the
  // activations.  Right when the weights are the bigger operand (GRU: 100 MB of U against 33 MB); when the activations
  // are (up_8_4: 132 MB against 4 MB) p.by_tiles renumbers the grid so that an XCD gets a contiguous range of TILE blocks
  // with all their cout blocks, one after the other: an activation tile enters one L2, not blocks_m of them
  int lb = blockIdx.x;
  if (p.by_tiles) {
    const int nb = gridDim.x, x = lb & 7, i = lb >> 3;
    lb = x * (nb >> 3) + min(x, nb & 7) + i;
  }
  const int block_m = lb % p.blocks_m, block_n = lb / p.blocks_m;
  const int m0 = block_m * WM, n0 = block_n * WN;

  // ---- activation staging: thread = (channel pair sm = wave, tile st = lane): channels 8 s + 2 sm, 8 s + 2 sm + 1
  const int sm = tid >> 6, st = tid & 63;  // a wave loads ONE channel pair for the 64 tiles: every load is one contiguous row segment
  int64_t s_off0, s_off1;  // element offsets of the patch's first row (clamped) and first column inside in0 / in1 (channel 0)
  unsigned rmask = 0;      // validity of the four patch rows
  bool lcol = true, rcol = true;  // patch columns 0 / 3 inside the image (columns 1, 2 always are)
  int y0;
  {
    const int n = min(n0 + st, p.n_tiles - 1);
    const int b = n / p.tiles_per_img;
    const int r = n - b * p.tiles_per_img;
    const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
    y0 = 2 * ty - 1;
    const int x0 = 2 * tx - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) rmask |= ((unsigned)(y0 + i) < (unsigned)p.H) ? (1u << i) : 0u;
    lcol = x0 >= 0;
    rcol = x0 + 3 < p.W;
    // the 16-byte window starts at column x0 whatever it is: the inputs are guard-banded (ConvArgs::vec_ok), so the element
    // left of a row's first / right of its last is readable -- it belongs to the neighbouring row and is zeroed below
    s_off0 = (int64_t)b * p.bs0 + x0;
    s_off1 = (int64_t)b * p.bs1 + x0;
  }
  // (!RELU) interior waves (every tile of the wave has its whole patch inside the image) skip the edge selects
  const bool edge = __builtin_amdgcn_ballot_w64(rmask != 0xfu || !lcol || !rcol) != 0;
  float pinf = __builtin_inff();
  asm("" : "+v"(pinf));  // (a limit the compiler cannot see through: median(x, 0, +inf) folds to a max WITH the canonicalising max in front)
  const float llim = lcol ? __builtin_inff() : 0.0f, rlim = rcol ? __builtin_inff() : 0.0f;  // (ReLU-on-load variant)
  const int64_t HW = (int64_t)p.H * p.W;
  // thread-constant 32-bit element offsets of the eight patch loads (channel of the pair, patch row) and of the weight chunk:
  // a K step only moves the wave-uniform bases (no per-load address arithmetic in the loop)
  // (BYTE offsets for buffer loads: wave-uniform base in the descriptor, thread offset in a register, no address arithmetic)
  int poff[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // RELU variant: a patch row outside the image gets an offset beyond the descriptor's range -- the load returns zeros;
      // the other variant clamps the row and zeroes it with the column selects
      const bool oob = RELU && !((rmask >> i) & 1u);
      poff[h][i] = oob ? (int)0x80000000 : (int)(((int64_t)(2 * sm + h) * HW + (int64_t)min(max(y0 + i, 0), p.H - 1) * p.W) * 4);
    }
  const int aoff = ((tid >> 6) * p.cout_pad + (tid & 63)) * 16;  // chunk t; chunk t + 256 i is 4 i segments further
  const int astride = 4 * p.cout_pad * 16;
  const int poff_b0 = (int)(s_off0 * 4), poff_b1 = (int)(s_off1 * 4);  // batch item + first column (may be -4: the guard band)

  f32x16 acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;

  const int ksteps = p.ctot / KC;
  // staged operands, TWO sets: the loads of step s + 2 are issued while step s computes and step s + 1's set is written to
  // LDS (one wave per SIMD: a load that is not back when its data is wanted stalls the matrix pipe, nobody else runs)
  f32x4 ra[2][8];      // weight tile: 8 x 16 bytes per thread
  f32x4 rb[2][2][4];   // activation patches: 2 channels x 4 rows

  // load `idx` (0..7: weight chunk, 8..15: patch row (idx - 8) % 4 of channel (idx - 8) / 4) of step s into set SET
  // (measured and dropped: one load behind every fourth MFMA instead of the burst at the top of a step -- 40 % slower)
  auto load_one = [&](int s, auto setc, auto idxc) {
    constexpr int SET = decltype(setc)::value, IDX = decltype(idxc)::value;
    if (WINO_ABL(4) || (IDX < 8 && WINO_ABL(32)) || (IDX >= 8 && WINO_ABL(64))) return;  // 32: no weight loads, 64: no patch loads
    if constexpr (IDX < 8) {
      const float* ub = p.u + ((int64_t)s * 32 * p.cout_pad + m0) * 4;
      ra[SET][IDX] = buf_load4(make_rsrc(ub, 0x7fffffff), aoff, IDX * astride);
    } else {
      constexpr int H = (IDX - 8) / 4, I = (IDX - 8) % 4;
      const int c = s * KC;  // (a step never straddles the two sources: c0 % 8 == 0)
      const bool first = c < p.c0;
      // the descriptor starts 16 bytes BEFORE the step's first channel plane (offsets are unsigned: a tile at column 0 of the
      // first row of the first image reads one element in front of the tensor, into the guard band)
      const float* base = (first ? p.in0 + (int64_t)c * HW : p.in1 + (int64_t)(c - p.c0) * HW) - 4;
      rb[SET][H][I] = buf_load4(make_rsrc(base, 0x7fffffff), (first ? poff_b0 : poff_b1) + poff[H][I] + 16, 0);
    }
  };
  auto load_step = [&](int s, auto setc) {
    static_for<16>([&](auto k) { load_one(s, setc, k); });
  };
  auto store_a = [&](int buf, auto setc) {
    constexpr int SET = decltype(setc)::value;
    if (WINO_ABL(2)) return;
    float* a = sA[buf];
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(a + (tid + 256 * i) * 4) = ra[SET][i];
  };
  // channel h of the thread's pair: B^T d B of its patch -> the 16 positions of the activation tile.  Two forms:
  // (RELU) 16 medians + 16 packed additions per channel
  auto store_b_pk = [&](int buf, int h, auto setc) {
    constexpr int SET = decltype(setc)::value;
    if (WINO_ABL(1)) return;
    float* bdst = sB[buf];
    // the patch rows as column pairs (0, 1) and (2, 3).  Rows outside the image came back as zeros (out-of-range offsets);
    // columns 0 / 3 outside it (the neighbouring row's element, or the guard band) are zeroed here.  ReLU on load is ONE
    // instruction per element (median of x, 0, limit: limit = +inf, or 0 for a masked column; no NaN canonicalisation in
    // front of it as with v_max), the column mask alone a select.
    f32x2 dl[4], dr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 v = rb[SET][h][i];
      if (RELU) {
        dl[i] = f32x2{__builtin_amdgcn_fmed3f(v[0], 0.0f, llim), __builtin_amdgcn_fmed3f(v[1], 0.0f, pinf)};
        dr[i] = f32x2{__builtin_amdgcn_fmed3f(v[2], 0.0f, pinf), __builtin_amdgcn_fmed3f(v[3], 0.0f, rlim)};
      } else {
        dl[i] = f32x2{lcol ? v[0] : 0.0f, v[1]};
        dr[i] = f32x2{v[2], rcol ? v[3] : 0.0f};
      }
    }
    // B^T d B, 16 packed additions per channel: rows first (on the column pairs as they are), then columns -- the second
    // pass combines elements ACROSS the two pairs of a row, which v_pk_add_f32 does in one instruction through its operand
    // selectors (the compiler builds the shuffled pairs with moves instead: written out)
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
      // position q = 4 i + l at [(q*2 + h)*4 + sm][tile]: consecutive lanes write consecutive floats
      bdst[(((4 * i + 0) * 2 + h) * 4 + sm) * 64 + st] = lo[0];
      bdst[(((4 * i + 1) * 2 + h) * 4 + sm) * 64 + st] = lo[1];
      bdst[(((4 * i + 2) * 2 + h) * 4 + sm) * 64 + st] = hi[0];
      bdst[(((4 * i + 3) * 2 + h) * 4 + sm) * 64 + st] = hi[1];
    }
  };

  // (!RELU) selects on edge waves, additions as the compiler forms them
  auto store_b_sel = [&](int buf, int h, auto setc) {
    constexpr int SET = decltype(setc)::value;
    if (WINO_ABL(1)) return;
    float* bdst = sB[buf];
    // the patch d[i][j]: zero outside the image (edge waves only), ReLU on load
    float d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = rb[SET][h][i];
      if (edge) {  // (wave-uniform)
        const bool rok = (rmask >> i) & 1u;
        v[0] = (rok && lcol) ? v[0] : 0.0f;
        v[1] = rok ? v[1] : 0.0f;
        v[2] = rok ? v[2] : 0.0f;
        v[3] = (rok && rcol) ? v[3] : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) d[i][j] = v[j];
    }
    // B^T d B on pairs of columns (v_pk_add_f32): rows first, then columns
    f32x2 w[4][2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const f32x2 d0 = {d[0][2 * jj], d[0][2 * jj + 1]}, d1 = {d[1][2 * jj], d[1][2 * jj + 1]};
      const f32x2 d2 = {d[2][2 * jj], d[2][2 * jj + 1]}, d3 = {d[3][2 * jj], d[3][2 * jj + 1]};
      w[0][jj] = d0 - d2;
      w[1][jj] = d1 + d2;
      w[2][jj] = d2 - d1;
      w[3][jj] = d1 - d3;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float w0 = w[i][0][0], w1 = w[i][0][1], w2 = w[i][1][0], w3 = w[i][1][1];
      const f32x2 lo = f32x2{w0, w1} + f32x2{-w2, w2};  // (w0 - w2, w1 + w2)
      const f32x2 hi = f32x2{w2, w1} - f32x2{w1, w3};   // (w2 - w1, w1 - w3)
      // position q = 4 i + l at [(q*2 + h)*4 + sm][tile]: consecutive lanes write consecutive floats
      bdst[(((4 * i + 0) * 2 + h) * 4 + sm) * 64 + st] = lo[0];
      bdst[(((4 * i + 1) * 2 + h) * 4 + sm) * 64 + st] = lo[1];
      bdst[(((4 * i + 2) * 2 + h) * 4 + sm) * 64 + st] = hi[0];
      bdst[(((4 * i + 3) * 2 + h) * 4 + sm) * 64 + st] = hi[1];
    }
  };

  auto store_b = [&](int buf, int h, auto setc) {
    if constexpr (RELU) {
      store_b_pk(buf, h, setc);
    } else {
      store_b_sel(buf, h, setc);
    }
  };
  using MODE_TAIL = std::integral_constant<int, 0>;
  using MODE_FULL = std::integral_constant<int, 1>;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  load_step(0, S0{});
  store_a(0, S0{});
  store_b(0, 0, S0{});
  store_b(0, 1, S0{});
  if (ksteps > 1) load_step(1, S1{});
  __syncthreads();
  // One wave per SIMD: nothing but this wave's own instruction stream hides a latency.  Four positions at a time (consecutive
  // MFMAs never share an accumulator); the fragments of the next four are read while the 16 MFMAs of the current four run;
  // the set staged for step s + 1 (loaded during step s - 1) is transformed and written behind the first three groups; the
  // loads of step s + 2 go out at the top of step s into the other set.
  f32x4 fa[2][4], fb[2][4];
  auto step = [&](int s, auto setc, auto fullc) {  // SET = the register set that holds step s + 1 (loaded during step s - 1)
    constexpr int SET = decltype(setc)::value;
    constexpr int MODE = decltype(fullc)::value;
    constexpr bool FULL = MODE == 1;  // steps s + 1 and s + 2 exist: no conditions, the step is ONE basic block
    using Sx = std::integral_constant<int, SET>;
    using Sy = std::integral_constant<int, SET ^ 1>;
    const int buf = s & 1;
    const bool more = FULL || s + 1 < ksteps;
    const float* a_rd = sA[buf] + (half * 64 + wm * 32 + l31) * 4;
    const float* b_rd = sB[buf] + half * 4 * 64 + wn * 32 + l31;  // [(q*2 + half)*4 + e][tile]
    auto read_b = [&](int q) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = b_rd[(q * 2 * 4 + e) * 64];
      return v;
    };
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      fa[0][qq] = *reinterpret_cast<const f32x4*>(a_rd + qq * 2 * 64 * 4);
      fb[0][qq] = read_b(qq);
    }
    if (FULL || s + 2 < ksteps) load_step(s + 2, Sy{});  // the other set (written to LDS during step s - 1)
    static_for<4>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      if (g + 1 < 4 && !WINO_ABL(16)) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          fa[(g + 1) & 1][qq] = *reinterpret_cast<const f32x4*>(a_rd + (4 * (g + 1) + qq) * 2 * 64 * 4);
          fb[(g + 1) & 1][qq] = read_b(4 * (g + 1) + qq);
        }
      }
      static_for<4>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
          acc[4 * g + qq] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][qq][e], fb[g & 1][qq][e], acc[4 * g + qq], 0, 0, 0);
      });
      if (more) {
        if (g == 0) store_a(buf ^ 1, Sx{});
        if (g == 1) store_b(buf ^ 1, 0, Sx{});
        if (g == 2) store_b(buf ^ 1, 1, Sx{});
      }
    });
    if (FULL) {
      // the wave is alone on its SIMD: what it issues between two MFMAs is free only if it is spread evenly -- two LDS reads, a
      // few VALU instructions, one LDS write behind every MFMA, a global load behind every fourth
#pragma unroll
      for (int j = 0; j < 64; ++j) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS read
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // VALU
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
        if ((j & 3) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
      }
    }
    if (!WINO_ABL(8)) __syncthreads();
  };
  {
    int s = 0;
    for (; s + 4 <= ksteps; s += 2) {  // steps s + 2, s + 3 exist
      step(s, S1{}, MODE_FULL{});
      step(s + 1, S0{}, MODE_FULL{});
    }
    for (; s + 2 <= ksteps; s += 2) {
      step(s, S1{}, MODE_TAIL{});
      step(s + 1, S0{}, MODE_TAIL{});
    }
    if (s < ksteps) step(s, S1{}, MODE_TAIL{});
  }
  if constexpr (RES) {
    // ---- output stage with a residual: residual and bias of all 16 channels are fetched up front (the wave is alone on its
    // SIMD: a load issued between the stores is waited for in full, and `out` may alias `res`, so the compiler keeps every
    // load behind the stores in front of it -- 16 exposed round trips per workgroup, +20 % on a 32-step layer)
    const int n = n0 + wn * 32 + l31;
    const bool n_ok = n < p.n_tiles;
    int64_t o_base;  // element offset of (batch item, channel 0, first pixel of the tile) in a [b][cout][H][W] tensor
    int r_off;       // the same in the residual, as a byte offset (out of range for a tile past the end: loads return zeros)
    {
      const int nc = min(n, p.n_tiles - 1);
      const int b = nc / p.tiles_per_img;
      const int rr = nc - b * p.tiles_per_img;
      const int ty = rr / p.tiles_x, tx = rr - ty * p.tiles_x;
      const int64_t pix = (int64_t)(2 * ty) * p.W + 2 * tx;
      o_base = (int64_t)b * p.cout * HW + pix;
      r_off = n_ok ? (int)(((int64_t)b * p.res_bs + pix) * 4) : (int)0x80000000;
    }
    f32x2 rv[16][2];
    float bv[16];
    auto fetch_res = [&]() {
      const __amdgpu_buffer_rsrc_t rres = make_rsrc(p.res ? p.res : p.out, p.res ? 0x7fffffff : 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int mo = m < p.cout ? r_off + (int)((int64_t)m * HW * 4) : (int)0x80000000;
#pragma unroll
        for (int a = 0; a < 2; ++a)
          rv[r][a] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rres, mo + a * p.W * 4, 0, 0));
        bv[r] = p.bias ? p.bias[min(m, p.cout - 1)] : 0.0f;
      }
    };
    fetch_res();
    __builtin_amdgcn_sched_barrier(0);

    // ---- output stage: Y = A^T M A per (channel, tile), bias / residual / activation, two 8-byte stores per channel
    if (!n_ok) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (m >= p.cout) continue;
      float t0[4], t1[4];
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        t0[l] = acc[0 + l][r] + acc[4 + l][r] + acc[8 + l][r];
        t1[l] = acc[4 + l][r] - acc[8 + l][r] - acc[12 + l][r];
      }
      float y[2][2];
      y[0][0] = t0[0] + t0[1] + t0[2];
      y[0][1] = t0[1] - t0[2] - t0[3];
      y[1][0] = t1[0] + t1[1] + t1[2];
      y[1][1] = t1[1] - t1[2] - t1[3];
      const int64_t o = o_base + (int64_t)m * HW;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        f32x2 v = {y[a][0] + bv[r] + rv[r][a][0], y[a][1] + bv[r] + rv[r][a][1]};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float x = v[e];
          if (p.act == DEVA_ACT_RELU) {
            x = fmaxf(x, 0.0f);
          } else if (p.act == DEVA_ACT_SIGMOID) {
            x = sigmoidf_(x);
          } else if (p.act == DEVA_ACT_SQUARE_PLUS_ONE) {
            x = x * x + 1.0f;
          }
          v[e] = x;
        }
        *reinterpret_cast<f32x2*>(p.out + o + (int64_t)a * p.W) = v;
      }
      __builtin_amdgcn_sched_barrier(0);  // one channel at a time: 16 accumulator reads each, not all 256 hoisted to the top
    }
  } else {
    // ---- output stage: Y = A^T M A per (channel, tile), bias / residual / activation, two 8-byte stores per channel
    const int n = n0 + wn * 32 + l31;
    if (n >= p.n_tiles) return;
    const int b = n / p.tiles_per_img;
    const int rr = n - b * p.tiles_per_img;
    const int ty = rr / p.tiles_x, tx = rr - ty * p.tiles_x;
    const int64_t pix = (int64_t)(2 * ty) * p.W + 2 * tx;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (m >= p.cout) continue;
      float t0[4], t1[4];
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        t0[l] = acc[0 + l][r] + acc[4 + l][r] + acc[8 + l][r];
        t1[l] = acc[4 + l][r] - acc[8 + l][r] - acc[12 + l][r];
      }
      float y[2][2];
      y[0][0] = t0[0] + t0[1] + t0[2];
      y[0][1] = t0[1] - t0[2] - t0[3];
      y[1][0] = t1[0] + t1[1] + t1[2];
      y[1][1] = t1[1] - t1[2] - t1[3];
      const float bv = p.bias ? p.bias[m] : 0.0f;
      const int64_t o = ((int64_t)b * p.cout + m) * HW + pix;
      const int64_t ro = (int64_t)b * p.res_bs + (int64_t)m * HW + pix;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        f32x2 v = {y[a][0] + bv, y[a][1] + bv};
        if (p.res) {
          const f32x2 rv = *reinterpret_cast<const f32x2*>(p.res + ro + (int64_t)a * p.W);
          v[0] += rv[0];
          v[1] += rv[1];
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float x = v[e];
          if (p.act == DEVA_ACT_RELU) {
            x = fmaxf(x, 0.0f);
          } else if (p.act == DEVA_ACT_SIGMOID) {
            x = sigmoidf_(x);
          } else if (p.act == DEVA_ACT_SQUARE_PLUS_ONE) {
            x = x * x + 1.0f;
          }
          v[e] = x;
        }
        *reinterpret_cast<f32x2*>(p.out + o + (int64_t)a * p.W) = v;
      }
    }
  }
}

}  // namespace

// -> 0 launched, 1 launch error, -1 not eligible (the caller runs the direct kernels)
int launch_conv_wino(const ConvArgs& a, const float* u, hipStream_t st) {
  if (!u || !a.vec_ok || a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || (a.H & 1) || (a.W & 1) || a.W < 4) return -1;
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
  p.tiles_per_img = (a.H / 2) * (a.W / 2);
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
  p.ablate = 0;
#ifdef DEVA_CONV_PROBES
  {
    static const int abl = [] {
      const char* e = getenv("DEVA_WINO_ABLATE");
      return e ? atoi(e) : 0;
    }();
    p.ablate = abl;
  }
#endif
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
  if (blocks < min_blocks) return -1;  // one workgroup per CU at a time: fewer than that and the direct kernels' split-K wins
  const dim3 grid((unsigned)blocks), block(256);
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
