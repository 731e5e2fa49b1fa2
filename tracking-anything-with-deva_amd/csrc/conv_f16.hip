// Implicit-GEMM convolution on the f16 matrix pipes of gfx950 (v_mfma_f32_32x32x16_f16: 16x the fp32 matrix rate), fp32
// accumulation.  Two precisions, one tile machinery:
//
// PREC 1 -- fp16 OPERANDS (the opt-in `--amp` path; deva/inference/eval_args.py:17, evaluation/eval_vos.py:137: the
//   reference wraps its frame loop in fp16 autocast): activations stay fp32 in HBM and are rounded to fp16 (RNE) while
//   they are staged, the weights are packed as fp16 once, products are exact in fp32, sums / bias / residual /
//   activation / output are fp32 -- the arithmetic the oracle's amp mode restates.
//
// PREC 2 -- fp32-ACCURATE on the f16 pipes (hi/lo operand split; the arithmetic of nn.Conv2d in fp32, big_modules.py:
//   54-212, modules.py:81-169, to fp32 round-off of max(|x|, 2^-3) |w| per product -- see the error model in
//   include/deva_hip.h: the activations are not pre-scaled, their lo plane has an absolute floor of 2^-25): every fp32
//   activation x is split while it is staged into
//   hi = fp16(x), lo = fp16(x - hi) (x - hi is exact in fp32; |x - hi - lo| <= max(2^-22 |x|, 2^-25)), the weights are
//   packed once as hi / lo fp16 planes of w * 2^e (e per layer: the largest weight lands in [2^13, 2^14], which keeps
//   the lo plane of every weight that matters out of the fp16 subnormals), and each K-block issues THREE MFMAs into the
//   same fp32 accumulator: hi.hi + hi.lo + lo.hi (fp16 x fp16 products are exact in fp32; the dropped lo.lo term is
//   <= 2^-22 |x w|).  The accumulators are scaled by 2^-e (exact) before bias / residual / activation.  An input beyond
//   the fp16 range (|x| > 65504) or a non-finite one turns hi into inf / NaN, which reaches every accumulator that
//   reads it: the kernel raises a device flag when an accumulator is not finite, and the caller runs the fp32 kernels
//   behind this launch, gated on that flag (ConvArgs::gate), so the result is the fp32 kernels' in that case.
//
// Same structure as conv_mfma.hip (buffer-addressed staging two K steps ahead, barrier before the last MFMA group, row
// reuse of the 3x3 path with the zero column, compile-time buffer indices), with
//   * BKH-deep K steps: 64 for PREC 1 (an fp16 K step of 32 would be 128 matrix-pipe cycles per wave, less than the
//     barrier and the staging around it cost), 32 for PREC 2 (three MFMAs per K-block: 384 cycles per wave and step on
//     the 64x32 wave tile, 768 on the 128x32 one, at the LDS footprint of the amp tile);
//   * weights W16[K/8][NPL][cout_pad][8] fp16 (NPL = 1: DEVA_KLAYOUT_H8; NPL = 2: hi plane, lo plane): eight consecutive
//     k of one output channel = 16 bytes = the A fragment of one lane for one K-block; K ordered in BKH-channel slabs
//     for kernels larger than 1x1;
//   * the activation tile as Bs[NPL][k/2][pixel][2] fp16 (k-pair rows, 4 bytes per pixel): every thread gathers TWO
//     channels x FOUR consecutive pixels with two 16-byte loads (a vector-memory instruction costs the same ~16 cycles
//     of the CU's address path whatever its width -- tools/probe/vmem_width_probe.hip: 120 wave-instructions per us and
//     CU for 4, 8 or 16 bytes per lane -- and the 4-byte gathers of the first version kept that path 50 % busy),
//     converts (splits) pairs of channels into packed halfs, and writes one 16-byte quad per plane; a lane's B
//     fragment of a K-block is four 4-byte reads per plane (rows 8 kb + 4 half + 0..3, its own pixel column).
// Kinds: 0 = 1x1 stride 1, 1 = 3x3 stride 1 pad 1 (row reuse); both on guard-banded inputs whose channel counts are
// multiples of BKH.  Everything else stays on the fp32 kernels (the caller falls back).
#include <cstdlib>
#include <type_traits>

#include "conv_epilogue.h"

namespace deva {
namespace {

#ifdef DEVA_CONV_PROBES  // timing-only ablations of the K loop (results are wrong): 1 LDS stores, 2 global loads (16: activations only, 32: weights only), 4 barrier, 8 fragments
#define DEVA_ABL(bit) (p.ablate & (bit))
#else
#define DEVA_ABL(bit) false
#endif

typedef conv_f32x16 f32x16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int KIND, int MINW, int BKH, int PREC>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, MINW) void conv_f16_kernel(const ConvArgs p) {
  constexpr int THREADS = 64 * WAVES_M * WAVES_N;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile");
  constexpr bool ROW = KIND == 1;
  constexpr bool SPLIT = PREC == 2;
  constexpr int NPL = SPLIT ? 2 : 1;               // operand planes: (hi, lo) or the rounded value alone
  constexpr int OCT = BKH / 8;                     // k-octets per K step
  constexpr int NKB = BKH / 16;                    // MFMA K-blocks per K step
  static_assert(NKB == 2 || NKB == 4, "K step of 32 or 64");
  constexpr int AROWS = OCT * NPL;                 // 16-byte rows of the weight tile per output channel and K step
  constexpr int A_V4 = AROWS * BM / THREADS;       // 16-byte loads of the weight tile per thread and K step
  constexpr int A_PASS = THREADS / BM;             // rows per pass
  static_assert(A_V4 >= 1 && A_V4 * THREADS == AROWS * BM && A_PASS * BM == THREADS, "weight tile geometry");
  constexpr int BNP = ROW ? BN + 8 : BN;           // ROW: columns 3 .. BN+4 hold pixels n0-1 .. n0+BN, column 0 stays zero
  constexpr int KP = BKH / 2;                      // k-pair rows of the activation tile
  constexpr int NQ4 = BN / 4;                      // pixel quads per row
  constexpr int TPT = KP * NQ4 / THREADS;          // gather tasks (k-pair row, pixel quad) per thread
  static_assert(TPT >= 1 && TPT * THREADS == KP * NQ4 && THREADS % NQ4 == 0, "activation gather geometry");
  constexpr int RSTEP = THREADS / NQ4;             // k-pair rows between the tasks of one thread
  constexpr int A_HALFS = AROWS * BM * 8, B_HALFS = NPL * KP * BNP * 2;

  __shared__ __attribute__((aligned(16))) _Float16 smem[2 * A_HALFS + 2 * B_HALFS];
  _Float16* const sA = smem;
  _Float16* const sB = smem + 2 * A_HALFS;

  if (p.gate && *p.gate == 0) return;  // (never set for these kernels today; same contract as the fp32 kernels)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave / WAVES_N) * WM;
  const int wn0 = (wave % WAVES_N) * WN;
  const int l31 = lane & 31;
  const int half = lane >> 5;

  int tile_m, tile_n;
  conv_tile_coords(p, tile_m, tile_n);
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  // ---- weights: thread t loads row t / BM (+ A_PASS per pass) of the step's AROWS rows, output channel m0 + t % BM
  const int a_voff = ((tid / BM) * p.cout_pad + m0 + (tid % BM)) * 16;
  const int a_pass_bytes = A_PASS * p.cout_pad * 16;
  const int a_step_bytes = AROWS * p.cout_pad * 16;
  const int a_total_bytes = ((p.K + 7) >> 3) * NPL * p.cout_pad * 16;

  // ---- gather tasks of this thread: k-pair rows kr0 + i * RSTEP (channels 2 r, 2 r + 1), pixels n0 + 4 * vq .. + 3
  const int kr0 = tid / NQ4, vq = tid % NQ4;
  int b_voff0 = 0, b_voff1 = 0;
  {
    const int n4 = n0 + 4 * vq;
    const int nn = (n4 < p.n_total) ? n4 : 0;  // OHW % 4 == 0: a quad never straddles images or the end
    const int b = nn / p.OHW;
    const int pix = nn - b * p.OHW;
    b_voff0 = (int)(((int64_t)b * p.bs0 + (int64_t)2 * kr0 * p.HW + pix) * 4);
    b_voff1 = (int)(((int64_t)b * p.bs1 + (int64_t)2 * kr0 * p.HW + pix) * 4);
  }
  const int b_row_bytes = (int)(p.HW * 4);
  // ROW: halo pixels (n0-1, n0+BN) of every channel of the step: 2 * BKH scalar loads, one each on the first threads
  const bool has_halo = ROW && tid < 2 * BKH;
  const int h_k = tid % BKH, h_side = (tid / BKH) & 1;
  int h_voff0 = 0, h_voff1 = 0;
  unsigned cmask[TN];  // 9-bit validity masks (bit dy*3+dx) of the TN pixels this lane consumes
#pragma unroll
  for (int j = 0; j < TN; ++j) cmask[j] = 0;
  if (ROW) {
    int nh = h_side ? n0 + BN : n0 - 1;
    nh = min(max(nh, 0), p.n_total - 1);
    const int b = nh / p.OHW;
    const int pix = nh - b * p.OHW;
    h_voff0 = (int)(((int64_t)b * p.bs0 + (int64_t)h_k * p.HW + pix) * 4);
    h_voff1 = (int)(((int64_t)b * p.bs1 + (int64_t)h_k * p.HW + pix) * 4);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn0 + 32 * j + l31;
      if (n < p.n_total) {
        const int px = n % p.OHW;
        const int oh = px / p.OW, ow = px - oh * p.OW;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const bool ok = ((unsigned)(oh + t / 3 - 1) < (unsigned)p.H) && ((unsigned)(ow + t % 3 - 1) < (unsigned)p.W);
          cmask[j] |= ok ? (1u << t) : 0u;
        }
      }
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int ksteps_total = (p.K + BKH - 1) / BKH;
  int ks0 = 0, ksteps = ksteps_total;
  if (p.splits > 1) {
    ks0 = (int)blockIdx.y * p.per_split;
    ksteps = max(0, min(ksteps - ks0, p.per_split));
  }
  const int ks_last = ks0 + max(ksteps, 1) - 1;

  // ---- staging registers: two sets for the weights (loads two K steps ahead), one for the activation gather
  constexpr int ASETS = (SPLIT && ROW && BM < 256) ? 1 : 2;  // (the 128-wide split row kind is register-bound: weights one K step ahead)
  f32x4 ra[ASETS][A_V4];
  f32x4 rb[TPT][2];       // per task: two channels x four pixels
  int rb_rows = BKH;      // channels of the staged activation tile that exist (KIND 0: a partial last step)
  float rh = 0.0f;        // halo threads: one channel of one halo pixel

  auto load_a = [&](int t_raw, auto setc) {
    constexpr int SET = decltype(setc)::value % ASETS;
    if (DEVA_ABL(2 | 32)) return;
    const int t = min(t_raw, ks_last);
    const int off = t * a_step_bytes;
    const __amdgpu_buffer_rsrc_t r = make_rsrc(reinterpret_cast<const char*>(p.w16) + off, max(a_total_bytes - off, 0));
#pragma unroll
    for (int i = 0; i < A_V4; ++i) ra[SET][i] = buf_load4(r, a_voff, i * a_pass_bytes);
  };
  // activation tile of K step t (KIND 0) / the (BKH-channel slab, dy) row tile that starts at step t (KIND 1)
  auto load_b = [&](int t_raw) {
    if (DEVA_ABL(2 | 16)) return;
    const int t = min(t_raw, ks_last);
    int cbase = t * BKH, shift = 0;
    if (ROW) {
      const int chunk = t / 9;
      cbase = chunk * BKH;
      shift = ((t - chunk * 9) / 3 - 1) * p.W;
    }
    const bool first = cbase < p.c0;
    const int lc = first ? cbase : cbase - p.c0;  // first channel of the step inside its source
    const float* base = (first ? p.in0 : p.in1) + ((int64_t)lc * p.HW + shift);
    int range = 0x7fffffff;
    if constexpr (!ROW) {
      // 1x1 layers may end in a partial step (513 = 512 + 1 channels): the rows beyond the source's last channel are not
      // loaded from beyond the tensor (the descriptor ends where the tensor ends: such loads return 0) and are zeroed
      // before they are staged (rb_rows); the weights of those k are zero as well
      rb_rows = min(BKH, (first ? p.c0 : p.c1) - lc);
      if (rb_rows < BKH) range = (int)max((int64_t)0, ((first ? p.in0_span : p.in1_span) - (int64_t)lc * p.HW) * 4);
    }
    const __amdgpu_buffer_rsrc_t r = make_rsrc(base, range);
#pragma unroll
    for (int i = 0; i < TPT; ++i)
#pragma unroll
      for (int c = 0; c < 2; ++c) rb[i][c] = buf_load4(r, first ? b_voff0 : b_voff1, (2 * i * RSTEP + c) * b_row_bytes);
    if (has_halo) rh = buf_load1(r, first ? h_voff0 : h_voff1, 0);
  };
  auto store_a = [&](auto setc) {
    constexpr int BUF = decltype(setc)::value, SET = BUF % ASETS;
    if (DEVA_ABL(1)) return;
    _Float16* a = sA + BUF * A_HALFS + tid * 8;
#pragma unroll
    for (int i = 0; i < A_V4; ++i) *reinterpret_cast<f32x4*>(a + i * THREADS * 8) = ra[SET][i];
  };
  // fp32 -> fp16 (round to nearest even, like torch's .half()), relu-on-load first; per task one 16-byte quad (four pixels
  // x the channel pair) per plane.  SPLIT: hi = fp16(v), lo = fp16(v - hi)
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  auto store_b = [&](int buf) {
    if (DEVA_ABL(1)) return;
    _Float16* bt = sB + buf * B_HALFS;
    if (!ROW && rb_rows < BKH) {  // (wave-uniform, the last step of a 513- / 257-channel layer only)
#pragma unroll
      for (int i = 0; i < TPT; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          if (2 * (kr0 + i * RSTEP) + c >= rb_rows) rb[i][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
#pragma unroll
    for (int i = 0; i < TPT; ++i) {
      u32x4 hi4, lo4;
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        float v0 = rb[i][0][px], v1 = rb[i][1][px];
        if (p.relu_in) {
          v0 = fmaxf(v0, 0.0f);
          v1 = fmaxf(v1, 0.0f);
        }
        const h2 hi = {(_Float16)v0, (_Float16)v1};
        hi4[px] = __builtin_bit_cast(unsigned, hi);
        if constexpr (SPLIT) {
          const h2 lo = {(_Float16)(v0 - (float)hi[0]), (_Float16)(v1 - (float)hi[1])};
          lo4[px] = __builtin_bit_cast(unsigned, lo);
        }
      }
      _Float16* at = bt + ((kr0 + i * RSTEP) * BNP + (ROW ? 4 : 0) + 4 * vq) * 2;
      *reinterpret_cast<u32x4*>(at) = hi4;
      if constexpr (SPLIT) *reinterpret_cast<u32x4*>(at + KP * BNP * 2) = lo4;
    }
    if (has_halo) {
      const float v = p.relu_in ? fmaxf(rh, 0.0f) : rh;
      const _Float16 hi = (_Float16)v;
      _Float16* at = bt + ((h_k >> 1) * BNP + (h_side ? BN + 4 : 3)) * 2 + (h_k & 1);
      at[0] = hi;
      if constexpr (SPLIT) at[KP * BNP * 2] = (_Float16)(v - (float)hi);
    }
  };

  // ---- fragments: lane (row or pixel l31, k-group half) holds k = 16*kb + 8*half + 0..7 of K-block kb: weights octet
  // 2*kb + half, one ds_read_b128 per plane; activations k-pair rows 8*kb + 4*half + 0..3 of its pixel column, four
  // 4-byte reads per plane
  const _Float16* const a_rd0 = sA + (half * NPL * BM + wm0 + l31) * 8;
  const _Float16* const b_rd0 = sB + (4 * half * BNP + wn0 + l31 + (ROW ? 3 : 0)) * 2;
  const _Float16* const b_zero0 = sB + (4 * half * BNP) * 2;
  h8 fa[2][NPL][TM], fb[2][NPL][TN];
  struct BRd {  // read bases of the lane's TN pixels (own column + dx, or the zero column for a padded tap)
    const _Float16* q[TN];
  };
  auto frag_load = [&](int set, const _Float16* a_rd, const BRd& b_rd, int kb) {
    if (DEVA_ABL(8)) return;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[set][pl][i] = *reinterpret_cast<const h8*>(a_rd + ((2 * kb * NPL + pl) * BM + 32 * i) * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        u32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = *reinterpret_cast<const unsigned*>(b_rd.q[j] + ((pl * KP + 8 * kb + e) * BNP) * 2);
        fb[set][pl][j] = __builtin_bit_cast(h8, w);
      }
    }
  };
  // the MFMAs of one K-block, in the order hi.hi (every row block), hi.lo, lo.hi; `part`: 0 = all, 1 = first half, 2 = rest
  constexpr int MF = (SPLIT ? 3 : 1) * TM * TN;
  auto mfma_block = [&](int set, int part = 0) {
#pragma unroll
    for (int q = 0; q < MF; ++q) {
      if ((part == 1 && q >= MF / 2) || (part == 2 && q < MF / 2)) continue;
      const int term = q / (TM * TN), i = q % TM, j = q / TM % TN;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][term == 2 ? NPL - 1 : 0][i], fb[set][term == 1 ? NPL - 1 : 0][j],
                                                         acc[i][j], 0, 0, 0);
    }
  };

  auto taps_of = [&](int j, int t) { return (cmask[j] >> (t % 9 / 3 * 3)) & 7u; };
  unsigned m3[TN];
  BRd b_cur;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    m3[j] = ROW ? taps_of(j, ks0) : 0u;
    b_cur.q[j] = (ROW && !(m3[j] & 1u)) ? b_zero0 : b_rd0 + 32 * j * 2;
  }

  // One K step of NKB K-blocks; buffer / register-set indices are compile-time (see conv_mfma.hip).  The staged next
  // tile is written to LDS beside the second-to-last MFMA group, the barrier sits before the last one.
  auto step = [&](int s, auto dxc, auto parc, auto gbc) {
    constexpr int DX = decltype(dxc)::value, PAR = decltype(parc)::value, GB = decltype(gbc)::value;
    constexpr int DXN = ROW ? (DX + 1) % 3 : 0;
    constexpr int GBN = (ROW && DX == 2) ? (GB ^ 1) : GB;
    const int t = ks0 + s;
    const _Float16* a_rd = a_rd0 + PAR * A_HALFS;
    const _Float16* a_nx = a_rd0 + (PAR ^ 1) * A_HALFS;
    BRd b_nx;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (ROW) {
        if (DX == 2) m3[j] = taps_of(j, t + 1);
        b_nx.q[j] = ((m3[j] >> DXN) & 1u) ? (b_rd0 + GBN * B_HALFS + 2 * DXN + 32 * j * 2) : (b_zero0 + GBN * B_HALFS);
      } else {
        b_nx.q[j] = b_rd0 + (PAR ^ 1) * B_HALFS + 32 * j * 2;
      }
    }
    if constexpr (NKB == 4) {
      frag_load(1, a_rd, b_cur, 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(0);
      __builtin_amdgcn_sched_barrier(0);
      frag_load(0, a_rd, b_cur, 2);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(1);
      __builtin_amdgcn_sched_barrier(0);
      frag_load(1, a_rd, b_cur, 3);
      store_a(std::integral_constant<int, PAR ^ 1>{});
      if (ROW) {
        if (DX == 2) store_b(GB ^ 1);
      } else {
        store_b(PAR ^ 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(0);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      load_a(t + 1 + ASETS, std::integral_constant<int, PAR ^ 1>{});
      if (ROW) {
        if (DX == 0) load_b(t + 3);
      } else {
        load_b(t + 2);
      }
      frag_load(0, a_nx, b_nx, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(1);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      // Two K-blocks of MF MFMAs per step, software-pipelined INSIDE the wave: the waves of a workgroup run in lock
      // step (one barrier per step), so bursts of fragment reads never meet another wave's MFMA phase -- the LDS and
      // the matrix pipe would take turns (measured: 0.45 MFMA utilisation, LDS 0.3 busy).  Every MFMA is followed by
      // one LDS read of the next K-block / one LDS write or global load of the staged tiles.
      // Region 1 (up to the barrier): fragments of K-block 1, MFMAs of K-block 0 and the first half of K-block 1, the
      // staged tile -> LDS, the global loads of the tile after next (into the registers the stores have just freed).
      frag_load(1, a_rd, b_cur, 1);
      mfma_block(0);
      store_a(std::integral_constant<int, PAR ^ 1>{});
      if (ROW) {
        if (DX == 2) store_b(GB ^ 1);
      } else {
        store_b(PAR ^ 1);
      }
      load_a(t + 1 + ASETS, std::integral_constant<int, PAR ^ 1>{});
      if (ROW) {
        if (DX == 0) load_b(t + 3);
      } else {
        load_b(t + 2);
      }
      mfma_block(1, 1);
      constexpr int NFR = (TM + 4 * TN) * NPL;  // LDS reads of one fragment set (the compiler pairs the 4-byte ones)
#pragma unroll
      for (int j = 0; j < MF; ++j) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // MFMA
        if (j < NFR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
      }
#pragma unroll
      for (int j = 0; j < MF / 2; ++j) {
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);  // DS write
        __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);  // VMEM read
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!DEVA_ABL(4)) __syncthreads();
      // Region 2: fragments of the next step's K-block 0 under the second half of K-block 1
      frag_load(0, a_nx, b_nx, 0);
      mfma_block(1, 2);
#pragma unroll
      for (int j = 0; j < MF - MF / 2; ++j) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (NFR + MF - MF / 2 - 1) / (MF - MF / 2), 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    b_cur = b_nx;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;

  // ---- prologue
  if (ROW) {  // the zero column (column 0) of every octet row, both buffers
    constexpr int ZR = NPL * KP;
    for (int i = tid; i < 2 * ZR * 2; i += THREADS) sB[(i / (ZR * 2)) * B_HALFS + ((i / 2) % ZR) * BNP * 2 + (i & 1)] = (_Float16)0.0f;
  }
  load_a(ks0, I0{});
  load_b(ks0);
  store_a(I0{});
  store_b(0);
  __syncthreads();
  load_a(ks0 + 1, I1{});
  if (ASETS == 2) load_a(ks0 + 2, I0{});
  if (!ROW) load_b(ks0 + 1);
  frag_load(0, a_rd0, b_cur, 0);

  if (ROW) {
    int s = 0;
    for (; s + 6 <= ksteps; s += 6) {
      step(s, I0{}, I0{}, I0{});
      step(s + 1, I1{}, I1{}, I0{});
      step(s + 2, I2{}, I0{}, I0{});
      step(s + 3, I0{}, I1{}, I1{});
      step(s + 4, I1{}, I0{}, I1{});
      step(s + 5, I2{}, I1{}, I1{});
    }
    if (s < ksteps) {
      step(s, I0{}, I0{}, I0{});
      step(s + 1, I1{}, I1{}, I0{});
      step(s + 2, I2{}, I0{}, I0{});
    }
  } else {
    int s = 0;
    for (; s + 2 <= ksteps; s += 2) {
      step(s, I0{}, I0{}, I0{});
      step(s + 1, I0{}, I1{}, I0{});
    }
    if (s < ksteps) step(s, I0{}, I0{}, I0{});
  }

  if constexpr (SPLIT) {
    // an operand beyond the fp16 range shows as a non-finite accumulator; undo the weight scale (exact: a power of two)
    bool bad = false;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          bad |= (__builtin_bit_cast(unsigned, acc[i][j][r]) & 0x7f800000u) == 0x7f800000u;
          acc[i][j][r] *= p.out_scale;
        }
    if (bad && p.flag) atomicOr(p.flag, 1);
  }

  if (p.vec_out && p.splits == 1) {  // (split-K partial sums: the direct stores measured 3-4 % faster on the layers that split)
    static_assert((THREADS / 64) * 4096 <= (2 * A_HALFS + 2 * B_HALFS) * 2, "one 4 KB output-stage scratch per wave fits the dead tile buffers");
    __syncthreads();  // every wave is done with the tiles
    conv_store_tile_vec<TM, TN>(p, acc, m0 + wm0, n0 + wn0, lane, reinterpret_cast<float*>(smem) + wave * 1024);
    return;
  }
  conv_store_tile<TM, TN>(p, acc, m0, wm0, n0, wn0, l31, half);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int MINW, int BKH, int PREC>
int launch_tile_f16(const ConvArgs& a, int kind, hipStream_t st) {
  ConvArgs p = a;
  p.gate = nullptr;
#ifdef DEVA_CONV_PROBES
  {
    static const int abl = [] {
      const char* e = getenv("DEVA_SPLIT_ABLATE");
      return e ? atoi(e) : 0;
    }();
    p.ablate = abl;
  }
#endif
  p.tiles_m = (int)ceil_div(a.cout, BM);
  p.tiles_n = (int)ceil_div(a.n_total, BN);
  const int ksteps_total = (int)ceil_div(a.K, BKH);
  p.per_split = ksteps_total;
  p.splits = 1;
  const int64_t blocks = (int64_t)p.tiles_m * p.tiles_n;
  p.group_m = conv_group_m(a.KH * a.KW, a.stride, BM, BN, blocks);
  constexpr int STEPS_MIN = 384 / BKH;  // K steps a split must keep (6 at BKH 64)
  if (a.ws && blocks < 192 && ksteps_total >= 2 * STEPS_MIN) {  // few tiles, long K: deterministic split-K like the fp32 kernels
    int64_t sp = ceil_div(512, blocks);
    if (sp > ksteps_total / STEPS_MIN) sp = ksteps_total / STEPS_MIN;
    if (sp > 16) sp = 16;
    const int64_t fit = a.ws_elems / ((int64_t)a.cout * a.n_total);
    if (sp > fit) sp = fit;
    if (sp >= 2) {
      int per = (int)ceil_div(ksteps_total, sp);
      if (kind == 1) per = (per + 2) / 3 * 3;
      p.splits = (int)ceil_div(ksteps_total, per);
      p.per_split = per;
    }
  }
  const dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)p.splits), block(64 * WAVES_M * WAVES_N);
  if (kind == 0) {
    hipLaunchKernelGGL((conv_f16_kernel<BM, BN, WAVES_M, WAVES_N, 0, MINW, BKH, PREC>), grid, block, 0, st, p);
  } else {
    hipLaunchKernelGGL((conv_f16_kernel<BM, BN, WAVES_M, WAVES_N, 1, MINW, BKH, PREC>), grid, block, 0, st, p);
  }
  if (p.splits > 1) return launch_splitk_reduce(p, st);
  return check_launch(PREC == 2 ? "deva_conv2d (fp16 hi/lo split)" : "deva_conv2d (fp16 operands)");
}

}  // namespace

// -> 0 launched, 1 launch error, -1 not eligible (the caller runs the fp32 kernels)
int launch_conv_f16(const ConvArgs& a, hipStream_t st) {
  const int bkh = a.prec == 2 ? 32 : 64;
  if (!a.w16 || !a.vec_ok || a.stride != 1 || a.cout < 64) return -1;
  const bool is1x1 = a.KH == 1 && a.KW == 1;
  // whole K steps per source; the split 1x1 kind also takes a partial LAST step (the second source's tail, or the only
  // source's: sensory_compress 512 + 1, g4_conv 256 + 1): deva_conv_pack_split pads those weights with zeros
  const bool tail_ok = a.prec == 2 && is1x1 && (a.c1 > 0 ? a.c0 % bkh == 0 : true);
  if (!tail_ok && (a.c0 % bkh || a.c1 % bkh)) return -1;
  int kind;
  if (is1x1) {
    kind = 0;
  } else if (a.KH == 3 && a.KW == 3 && a.pad == 1) {
    kind = 1;
  } else {
    return -1;
  }
  const int64_t blocks128 = ceil_div(a.cout, 128) * ceil_div(a.n_total, 128);
  if (a.prec == 2) {
#ifdef DEVA_CONV_PROBES  // `make PROBES=1`: A/B runs of the tile policy (tools/convlab)
    static const int forced = [] {
      const char* e = getenv("DEVA_SPLIT_TILE");
      return e ? atoi(e) : 0;
    }();
    if (forced == 256 && a.cout >= 256) return launch_tile_f16<256, 128, 2, 4, 2, 32, 2>(a, kind, st);
    if (forced == 128 && a.cout >= 128) return launch_tile_f16<128, 128, 2, 4, 4, 32, 2>(a, kind, st);
    if (forced == 1284 && a.cout >= 128) return launch_tile_f16<128, 128, 2, 2, 2, 32, 2>(a, kind, st);
    if (forced == 64) return launch_tile_f16<64, 64, 2, 2, 2, 32, 2>(a, kind, st);
#endif
    // 128x128 tiles, 8 waves (wave tile 64x32), two workgroups per CU.  Measured against it on the layers of the 480p / 5-object
    // and the 1080p / 11-object frames (tools/convlab, DEVA_SPLIT_TILE in `make PROBES=1` builds; profiles/r05/lab): 256x128
    // tiles (wave tile 128x32, one workgroup per CU) -1..4 %, 128x128 on four waves (wave tile 64x64) +-1 %: under real
    // operand data the kernels run at the chip's power limit (all-zero activations: +25..34 % at an unchanged instruction
    // stream), so fewer LDS bytes per MFMA buy nothing.
    // (few tiles but a long K loop -- the 512 -> 512 3x3 image part of the fusers at 30x54: 52 tiles, 144 steps -- also takes
    // the 128x128 tile: its split-K fills the chip, 45 against 72 us on 64x64 tiles)
    // ... unless the 128x128 tiles would leave a quarter or more of the CUs without a workgroup where 64x64 tiles give every
    // CU one (the batch-1 layers of the key encoder at 1/16 of a 1080p frame: 128 tiles of 128x128)
    const int64_t blocks64 = ceil_div(a.cout, 64) * ceil_div(a.n_total, 64);
    const bool half_empty = blocks128 >= 64 && blocks128 < 192 && blocks64 >= 256;
    if (a.cout >= 128 && !half_empty && (blocks128 >= 64 || (blocks128 >= 32 && a.K >= 128 * 32)))
      return launch_tile_f16<128, 128, 2, 4, 4, 32, 2>(a, kind, st);
    return launch_tile_f16<64, 64, 2, 2, 2, 32, 2>(a, kind, st);
  }
  if (a.cout >= 128 && blocks128 >= 64) return launch_tile_f16<128, 128, 2, 4, 4, 64, 1>(a, kind, st);
  return launch_tile_f16<64, 64, 2, 2, 2, 64, 1>(a, kind, st);
}

}  // namespace deva
