// Implicit-GEMM convolution on v_mfma_f32_32x32x2_f32, second kernel generation: a main loop without VALU work.
//
// Why: on gfx950 the fp32 MFMA runs at the fp32 VALU rate and does not co-execute with VALU instructions
// (SQ_VALU_MFMA_COEXEC_CYCLES = 0 on every conv dispatch, profiles/r04a): every v_cndmask / v_add / 64-bit address
// computation in the K loop is matrix-pipe time.  The first generation (conv_igemm.hip) issues ~2 VALU instructions
// per MFMA in its loop and keeps the pipe 77-80 % busy on the big layers.  Here
//   * operands are fetched with BUFFER loads: per-thread byte offsets are computed once, the K step advances the
//     scalar base of the resource descriptor (SALU), the tile tails read zeros through the range check -- no address
//     arithmetic and no selects in the loop;
//   * the weights are stored k-quad interleaved, Wq[k/4][cout_pad][4] (DEVA_KLAYOUT_Q4): a 16-byte load brings the
//     four k values one lane needs for four consecutive MFMAs, the LDS image is lane-linear and the A fragments of a
//     wave are ONE ds_read_b128 per 32x(8 k) block instead of four ds_read_b32;
//   * the zero padding of the 3x3 row-reuse path (input rows of a (32-channel slab, dy) pair staged once for the
//     three dx taps, padding applied per consumer pixel) costs ONE select per K step: a lane whose tap falls into
//     the padding reads an always-zero column of the row tile instead of its own pixel (its LDS base address
//     changes, the row immediates stay) -- the first generation selected per MFMA pair;
//   * the activation tile is stored with K rows interleaved in pairs, Bs[k/2][pixel][k%2]: a lane reads two k values
//     with one ds_read_b64 at base + immediate (ds_read2_b32 with an 8-bit offset field needed a v_add per pair at
//     the 136-float row pitch); the pair interleave is free at the write (each thread gathers two adjacent K rows
//     and stores 16-byte pieces {r0[p], r1[p], r0[p+1], r1[p+1]});
//   * the barrier of a K step sits before the LAST MFMA group: the tile of step s+1 is written to LDS after group 1,
//     the fragments of the last group are already in registers when the wave reaches the barrier, and the first
//     fragments of step s+1 are read behind it, under the MFMAs of the last group.
//
// GEMM view as before:  M = cout, N = batch*OH*OW (pixels, batch-major), K = KH*KW*(C0+C1).
// MFMA k assignment inside a 32-deep K step: group q (0..3), MFMA e (0..3): lanes 0-31 feed k = 8q + e, lanes 32-63
// feed k = 8q + 4 + e  (the order of the fp32 FMA chain of one output; deterministic).
#include <cstdlib>
#include <type_traits>

#include "conv_epilogue.h"

#ifdef DEVA_CONV_PROBES
// `make PROBES=1` builds only: wall-clock stamps (100 MHz) of wave 0 of every workgroup at four points of the kernel
__device__ unsigned long long* g_conv_probe = nullptr;
#define DEVA_STAMP(i)                                                                                       \
  do {                                                                                                      \
    if (g_conv_probe && threadIdx.x == 0 && blockIdx.y == 0) g_conv_probe[blockIdx.x * 4 + (i)] = wall_clock64(); \
  } while (0)
#else
#define DEVA_STAMP(i)
#endif

namespace deva {
namespace {

typedef conv_f32x16 f32x16;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// KIND 0: 1x1, stride 1, guard-banded inputs, c0 % 32 == 0 (vector gathers of 4 pixels; K tail of the second source allowed)
// KIND 1: 3x3, stride 1, pad 1, 32-channel-slab K order, guard-banded inputs (row reuse)
// KIND 2: any kernel with c0 and c0+c1 multiples of 32 (tap and source uniform per K step), scalar gathers
// KIND 3: anything (per-element decode through a table: the 2/3/4-channel stems, odd channel splits)
//
// WK > 1: K slices inside the workgroup.  The workgroup is WK independent groups of WAVES_M x WAVES_N waves; group k
// owns its own LDS tiles and the k-th part of the workgroup's K range, all groups share the barriers, and the partial
// sums meet in LDS in a fixed order (k = 1, 2, ...) before group 0 runs the epilogue.  For layers with few output
// tiles (batch-1 key encoder at 30x54) this is the split-K that fills the SIMDs without partial sums in HBM and
// without a reduction launch.
//
// PERSIST: the gated fp32 re-run behind a split launch (conv_f16.hip PREC 2, ConvArgs::gate).  Such a launch does its work
// only when the split kernel met an input beyond the fp16 range -- practically never --, so what it costs is the dispatch
// of its workgroups, each of which reads the flag and leaves: 4.5-6 us per layer at 1080p / 11 objects with one workgroup
// per tile (~8 000 of them, at one or two per CU for the LDS they reserve).  The persistent form is launched with at most
// PERSIST_MAX_WGS workgroups that walk over the tiles (stride gridDim.x, a multiple of 8: a workgroup stays on its XCD's
// tiles); same arithmetic, same K order as every other WK = 1 / unsplit variant.
template <int BM, int BN, int WAVES_M, int WAVES_N, int KIND, int MINW, int WK, int RELU, bool PERSIST = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N * WK, (KIND <= 1 || MINW < 2) ? MINW : 2) void conv_mfma_kernel(const ConvArgs p) {
  static_assert(!PERSIST || (WK == 1 && KIND <= 1), "persistent form: vector kinds, no K slices");
  // RELU: 0 = the input is taken as it is, 2 = ReLU on every input element, 1 = p.relu_in decides at run time (one
  // VALU instruction per loaded element either way).  The 3x3 row kind is built as 0 and 2 (36 of the 97 VALU
  // instructions of its loop; worth 0.3-0.7 % on the GRU / fuser layers, tools/convlab --rounds 7); the 1x1 kind keeps
  // the run-time flag -- its loop came out 4 % slower without the instruction (register allocation).
  if (p.gate && *p.gate == 0) return;  // the fp32 re-run behind a split launch (conv_f16.hip) that raised no flag
  const bool relu_in = RELU == 2 ? true : (RELU == 1 ? (p.relu_in != 0) : false);
  // (the scalar-gather kinds carry 64-bit pointers and per-element validity: two waves per SIMD, no spills)
  constexpr int THREADS = 64 * WAVES_M * WAVES_N;  // threads of one K-slice group
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile");
  constexpr bool ROW = KIND == 1;
  constexpr bool VEC = KIND <= 1;
  static_assert(!ROW || TN == 1, "row reuse: one pixel per lane");
  constexpr int A_V4 = 8 * BM / THREADS;        // 16-byte loads of the weight tile per thread and K step
  constexpr int A_PASS = THREADS / BM;          // k-quad rows per pass
  static_assert(A_V4 >= 1 && A_V4 * THREADS == 8 * BM, "weight tile geometry");
  constexpr int BNP = ROW ? BN + 8 : BN;        // ROW: columns 3 .. BN+4 hold pixels n0-1 .. n0+BN
  constexpr int A_FLOATS = BK * BM, B_FLOATS = BK * BNP;
  constexpr int NQ = BN / 4;                    // pixel quads per tile row
  constexpr int KGV = THREADS / NQ;             // K rows per pass (vector gather)
  constexpr int B_V4 = VEC ? BK / KGV : 2;      // rows per thread: PAIRS adjacent row pairs (2a, 2a+1), a = vk + i*KGV
  constexpr int PAIRS = B_V4 / 2;
  static_assert(!VEC || (PAIRS >= 1 && PAIRS * 2 * KGV == BK), "vector gather geometry");
  constexpr int KG = THREADS / BN;              // K rows per pass (scalar gather)
  constexpr int B_PT = VEC ? 1 : BK / KG;
  constexpr int KTAB = 1024;

  constexpr int REGION = 2 * A_FLOATS + 2 * B_FLOATS;
  __shared__ __attribute__((aligned(16))) float smem[WK * REGION];
  __shared__ unsigned s_ktab[KIND == 3 ? KTAB : 1];
  const int slice = WK > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / THREADS) : 0;
  float* const sA = smem + slice * REGION;
  float* const sB = sA + 2 * A_FLOATS;

  const int tid = WK > 1 ? (int)threadIdx.x % THREADS : (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave / WAVES_N) * WM;
  const int wn0 = (wave % WAVES_N) * WN;
  const int l31 = lane & 31;
  const int half = lane >> 5;

  DEVA_STAMP(0);
  const bool ktab_ok = KIND == 3 && p.K <= KTAB && p.ctot < 65536 && p.KH < 256 && p.KW < 256;
  if (KIND == 3 && ktab_ok) {
    for (int k = threadIdx.x; k < p.K; k += THREADS * WK) {
      const int tap = k / p.ctot;
      const int dy = tap / p.KW;
      s_ktab[k] = (unsigned)(k - tap * p.ctot) | ((unsigned)dy << 16) | ((unsigned)(tap - dy * p.KW) << 24);
    }
    __syncthreads();
  }

  const int n_tiles = PERSIST ? p.tiles_m * p.tiles_n : (int)gridDim.x;
  int tile_id = blockIdx.x;
  do {  // (one pass unless PERSIST)
    int tile_m, tile_n;
    conv_tile_coords(p, tile_id, n_tiles, tile_m, tile_n);
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- weights: thread t loads quad row t / BM (+ A_PASS per pass), output channel m0 + t % BM
    const int a_voff = ((tid / BM) * p.cout_pad + m0 + (tid % BM)) * 16;
    const int a_pass_bytes = A_PASS * p.cout_pad * 16;
    const int a_step_bytes = 8 * p.cout_pad * 16;
    const int a_total_bytes = ((p.K + 3) >> 2) * p.cout_pad * 16;

    // ---- vector gather: 4 consecutive pixels of K row vk (+ KGV per pass)
    const int vq = tid % NQ, vk = tid / NQ;
    int b_voff0 = 0, b_voff1 = 0;
    if (VEC) {
      const int n4 = n0 + 4 * vq;
      const int nn = (n4 < p.n_total) ? n4 : 0;  // OHW % 4 == 0: a quad never straddles images or the end
      const int b = nn / p.OHW;
      const int pix = nn - b * p.OHW;
      b_voff0 = (int)(((int64_t)b * p.bs0 + (int64_t)2 * vk * p.HW + pix) * 4);
      b_voff1 = (int)(((int64_t)b * p.bs1 + (int64_t)2 * vk * p.HW + pix) * 4);
    }
    const int b_row_bytes = (int)(p.HW * 4);
    const int b_pass_bytes = (int)(2 * KGV * p.HW * 4);
    // ROW: halo pixel (n0-1 or n0+BN) of K row h_row, loaded by every thread (branch-free), stored by wave 0
    const int h_side = tid & 1, h_row = (tid >> 1) & (BK - 1);
    int h_voff0 = 0, h_voff1 = 0;
    unsigned cmask = 0;  // 9-bit validity mask (bit dy*3+dx) of the pixel this lane consumes
    if (ROW) {
      int nh = h_side ? n0 + BN : n0 - 1;
      nh = min(max(nh, 0), p.n_total - 1);
      const int b = nh / p.OHW;
      const int pix = nh - b * p.OHW;
      h_voff0 = (int)(((int64_t)b * p.bs0 + (int64_t)h_row * p.HW + pix) * 4);
      h_voff1 = (int)(((int64_t)b * p.bs1 + (int64_t)h_row * p.HW + pix) * 4);
      const int n = n0 + wn0 + l31;
      if (n < p.n_total) {
        const int px = n % p.OHW;
        const int oh = px / p.OW, ow = px - oh * p.OW;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const bool ok = ((unsigned)(oh + t / 3 - 1) < (unsigned)p.H) && ((unsigned)(ow + t % 3 - 1) < (unsigned)p.W);
          cmask |= ok ? (1u << t) : 0u;
        }
      }
    }

    // ---- scalar gather (KIND 2, 3): this thread always gathers pixel n0 + tid % BN
    const int bn_local = tid % BN, bk_group = tid / BN;
    bool n_ok = false;
    int ih0 = 0, iw0 = 0;
    const float* src0 = p.in0;
    const float* src1 = p.in0;
    if (!VEC) {
      const int n_g = n0 + bn_local;
      n_ok = n_g < p.n_total;
      const int nn = n_ok ? n_g : 0;
      const int b = nn / p.OHW;
      const int pix = nn - b * p.OHW;
      const int oh = pix / p.OW;
      const int ow = pix - oh * p.OW;
      ih0 = oh * p.stride - p.pad;
      iw0 = ow * p.stride - p.pad;
      src0 = p.in0 + (int64_t)b * p.bs0;
      src1 = p.in1 ? p.in1 + (int64_t)b * p.bs1 : p.in0;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // split-K: this workgroup accumulates K steps [ks0, ks0 + ksteps) (its slice group: the slice-th part of them)
    const int ksteps_total = (p.K + BK - 1) / BK;
    int ks0 = 0, ksteps = ksteps_total;
    if (p.splits > 1) {
      ks0 = (int)blockIdx.y * p.per_split;
      ksteps = max(0, min(ksteps - ks0, p.per_split));
    }
    int loop_steps = ksteps;  // steps every slice group walks through (the barriers are shared)
    if (WK > 1) {
      int per = (ksteps + WK - 1) / WK;
      if (ROW) per = (per + 2) / 3 * 3;
      loop_steps = per;
      ks0 += slice * per;
      ksteps = max(0, min(ksteps - slice * per, per));
    }
    const int ks_end = ks0 + ksteps;                                          // steps from here on contribute zeros
    const int ks_last = min(ks0 + max(ksteps, 1), ksteps_total) - 1;          // last step whose addresses are loaded

    // ---- staging registers: TWO sets for the per-step tiles (the loads of step s+3 are issued while those of step s+2
    // are still in flight: two K steps of latency cover instead of one -- a workgroup alone on its CU has nobody to
    // hide an HBM miss behind), one for the row tile of the 3x3 path (loaded a whole group ahead)
    f32x4 ra[2][A_V4];
    f32x4 rbv[ROW ? 1 : 2][B_V4];
    float rh = 0.0f;
    float rb[B_PT];  // scalar gathers (KIND 2, 3): one set, loaded one step ahead
    unsigned ok_b = 0;
    int rows_valid[2] = {BK, BK};  // KIND 0: rows of the staged tile that exist in their source

    // global -> registers (set SET) for K step t (clamped to the last step of this workgroup: the tail re-loads, unused)
    auto load_issue = [&](int t_raw, auto setc, auto row_tile, int tb_raw) {
      constexpr int SET = decltype(setc)::value;
      constexpr bool WITH_ROWS = decltype(row_tile)::value;  // ROW: the (slab, dy) row tile that starts at step t
      const int t = min(t_raw, ks_last);
      {
        const int off = t * a_step_bytes;
        // weights of a step beyond this slice's range read as zeros (empty range)
        const __amdgpu_buffer_rsrc_t r =
            make_rsrc(reinterpret_cast<const char*>(p.w) + off, (WK > 1 && t_raw >= ks_end) ? 0 : max(a_total_bytes - off, 0));
#pragma unroll
        for (int i = 0; i < A_V4; ++i) ra[SET][i] = buf_load4(r, a_voff, i * a_pass_bytes);
      }
      if (KIND == 0) {
        const int cbase = t * BK;
        const bool first = cbase < p.c0;
        const int c = first ? cbase : cbase - p.c0;
        rows_valid[SET] = min(BK, (first ? p.c0 : p.c1) - c);
        const float* base = (first ? p.in0 : p.in1) + (int64_t)c * p.HW;
        const int64_t span = (first ? p.in0_span : p.in1_span) - (int64_t)c * p.HW;
        const __amdgpu_buffer_rsrc_t r = make_rsrc(base, (int)min(span * 4, (int64_t)0x7fffffff));
        const int voff = first ? b_voff0 : b_voff1;
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
          rbv[ROW ? 0 : SET][2 * i] = buf_load4(r, voff, i * b_pass_bytes);
          rbv[ROW ? 0 : SET][2 * i + 1] = buf_load4(r, voff, i * b_pass_bytes + b_row_bytes);
        }
      } else if (KIND == 1) {
        if (WITH_ROWS) {
          const int chunk = t / 9;
          const int dy = (t - chunk * 9) / 3;
          const int cbase = chunk * BK;
          const bool first = cbase < p.c0;
          const float* base = (first ? p.in0 : p.in1) + ((int64_t)(first ? cbase : cbase - p.c0) * p.HW + (dy - 1) * p.W);
          const __amdgpu_buffer_rsrc_t r = make_rsrc(base, 0x7fffffff);
          const int voff = first ? b_voff0 : b_voff1;
#pragma unroll
          for (int i = 0; i < PAIRS; ++i) {
            rbv[0][2 * i] = buf_load4(r, voff, i * b_pass_bytes);
            rbv[0][2 * i + 1] = buf_load4(r, voff, i * b_pass_bytes + b_row_bytes);
          }
          rh = buf_load1(r, first ? h_voff0 : h_voff1, 0);
        }
      } else if (tb_raw >= 0) {
        unsigned okb = 0;
        const int tb = min(tb_raw, ks_last);
        const int k0 = tb * BK;
        if (KIND == 2) {
          int tap = 0, cbase = k0;
          if (p.KH * p.KW > 1) {
            if ((p.k_layout & 0xf) == DEVA_KLAYOUT_CHUNK32) {
              const int taps = p.KH * p.KW;
              const int chunk = tb / taps;
              tap = tb - chunk * taps;
              cbase = chunk * BK;
            } else {
              tap = k0 / p.ctot;
              cbase = k0 - tap * p.ctot;
            }
          }
          const int dy = tap / p.KW;
          const int ih = ih0 + dy, iw = iw0 + (tap - dy * p.KW);
          const bool first = cbase < p.c0;
          const bool okp = n_ok && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
          const float* sp = (first ? (src0 + (int64_t)cbase * p.HW) : (src1 + (int64_t)(cbase - p.c0) * p.HW)) +
                            (okp ? (ih * p.W + iw) : 0);
#pragma unroll
          for (int i = 0; i < B_PT; ++i) {
            const int ci = bk_group + i * KG;
            const bool kin = k0 + ci < p.K;
            rb[i] = sp[kin ? (int64_t)ci * p.HW : 0];
            okb |= (okp && kin) ? (1u << i) : 0u;
          }
        } else {
#pragma unroll
          for (int i = 0; i < B_PT; ++i) {
            const int k = k0 + bk_group + i * KG;
            int c, dy, dx;
            if (ktab_ok) {
              const unsigned e = s_ktab[min(k, KTAB - 1)];
              c = (int)(e & 0xffffu);
              dy = (int)((e >> 16) & 0xffu);
              dx = (int)(e >> 24);
            } else {
              const int tap = k / p.ctot;
              c = k - tap * p.ctot;
              dy = tap / p.KW;
              dx = tap - dy * p.KW;
            }
            const int ih = ih0 + dy, iw = iw0 + dx;
            const bool ok = n_ok && (k < p.K) && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
            const bool first = ok ? (c < p.c0) : true;
            const float* sp = first ? src0 : src1;
            const int64_t off = ok ? ((int64_t)(first ? c : (c - p.c0)) * p.HW + (ih * p.W + iw)) : 0;
            rb[i] = sp[off];
            okb |= ok ? (1u << i) : 0u;
          }
        }
        ok_b = okb;
      }
    };

    // registers (set SET) -> LDS buffers SET (per-step tiles) / GB (row tile).  Activation tile: element (K row r,
    // column c) lives at ((r/2)*BNP + c)*2 + r%2.
    auto relu4 = [](f32x4& v) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_fmed3f(v[j], 0.0f, __builtin_inff());  // one v_med3_f32
    };
    auto store_pairs = [&](const f32x4* rv, float* bt, int col0) {  // rows (2a, 2a+1), a = vk + i*KGV; pixels col0 + 4*vq .. +3
#pragma unroll
      for (int i = 0; i < PAIRS; ++i) {
        const f32x4 r0 = rv[2 * i], r1 = rv[2 * i + 1];
        float* d = bt + ((vk + i * KGV) * BNP + col0 + 4 * vq) * 2;
        *reinterpret_cast<f32x4*>(d) = f32x4{r0[0], r1[0], r0[1], r1[1]};
        *reinterpret_cast<f32x4*>(d + 4) = f32x4{r0[2], r1[2], r0[3], r1[3]};
      }
    };
    auto lds_store = [&](auto setc, auto row_tile, auto gbc) {
      constexpr int SET = decltype(setc)::value;
      constexpr bool WITH_ROWS = decltype(row_tile)::value;
      constexpr int GB = decltype(gbc)::value;
      float* a = sA + SET * A_FLOATS + tid * 4;
#pragma unroll
      for (int i = 0; i < A_V4; ++i) *reinterpret_cast<f32x4*>(a + i * THREADS * 4) = ra[SET][i];
      if (KIND == 0) {
        f32x4* rv = rbv[ROW ? 0 : SET];
        if (relu_in) {
#pragma unroll
          for (int i = 0; i < B_V4; ++i) relu4(rv[i]);
        }
        if (rows_valid[SET] < BK) {
#pragma unroll
          for (int i = 0; i < B_V4; ++i)
            if (2 * (vk + (i >> 1) * KGV) + (i & 1) >= rows_valid[SET]) rv[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        store_pairs(rv, sB + SET * B_FLOATS, 0);
      } else if (KIND == 1) {
        if (WITH_ROWS) {
          float* bt = sB + GB * B_FLOATS;
          if (relu_in) {
#pragma unroll
            for (int i = 0; i < B_V4; ++i) relu4(rbv[0][i]);
            rh = __builtin_amdgcn_fmed3f(rh, 0.0f, __builtin_inff());
          }
          store_pairs(rbv[0], bt, 4);
          if (tid < 64) bt[((h_row >> 1) * BNP + (h_side ? BN + 4 : 3)) * 2 + (h_row & 1)] = rh;
        }
      } else {
        float* b = sB + SET * B_FLOATS + bn_local * 2;
#pragma unroll
        for (int i = 0; i < B_PT; ++i) {
          const int r = bk_group + i * KG;
          float v = rb[i];
          if (relu_in) v = __builtin_amdgcn_fmed3f(v, 0.0f, __builtin_inff());
          b[(r >> 1) * BNP * 2 + (r & 1)] = (ok_b & (1u << i)) ? v : 0.0f;
        }
      }
    };

    // ---- fragment reads: A one ds_read_b128 per (32 rows x 8 k), B two ds_read_b64 per (8 k x 32 pixels)
    const float* const a_rd0 = sA + (half * BM + wm0 + l31) * 4;
    const float* const b_rd0 = sB + (2 * half * BNP + wn0 + l31 + (ROW ? 3 : 0)) * 2;
    const float* const b_zero0 = sB + (2 * half * BNP) * 2;  // ROW: column 0 of every row pair stays zero
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x4 fa[2][TM];
    f32x2 fb[2][TN][2];
    auto frag_load = [&](int set, const float* a_rd, const float* b_rd, int q) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[set][i] = *reinterpret_cast<const f32x4*>(a_rd + (2 * q * BM + 32 * i) * 4);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2)
          fb[set][j][e2] = *reinterpret_cast<const f32x2*>(b_rd + ((4 * q + e2) * BNP + 32 * j) * 2);
    };
    auto mfma_group = [&](int set) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int i = 0; i < TM; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i][e], fb[set][j][e >> 1][e & 1], acc[i][j], 0, 0, 0);
    };

    // ROW: 3-bit validity (dx = 0..2) of this lane's pixel for the dy of K step t; the lane's read base of a step is its
    // own column (+ dx) or, when the tap falls into the zero padding, the zero column
    auto taps_of = [&](int t) { return (cmask >> (t % 9 / 3 * 3)) & 7u; };
    unsigned m3 = ROW ? taps_of(ks0) : 0u;
    const float* b_cur = ROW ? ((m3 & 1u) ? b_rd0 : b_zero0) : b_rd0;  // read base of the current step

    // One K step; all buffer / register-set indices are compile-time (the loop is unrolled over them):
    // DX: dx tap of the step (ROW: steps come in (slab, dy) groups of three; splits start on group boundaries),
    // PAR: parity of the step within this workgroup, GB: parity of the row-tile group.
    auto step = [&](int s, auto dxc, auto parc, auto gbc) {
      constexpr int DX = decltype(dxc)::value, PAR = decltype(parc)::value, GB = decltype(gbc)::value;
      constexpr int DXN = ROW ? (DX + 1) % 3 : 0;
      constexpr int GBN = (ROW && DX == 2) ? (GB ^ 1) : GB;
      const int t = ks0 + s;
      const float* a_rd = a_rd0 + PAR * A_FLOATS;
      const float* a_nx = a_rd0 + (PAR ^ 1) * A_FLOATS;
      const float* b_nx;
      if (ROW) {
        if (DX == 2) m3 = taps_of(t + 1);
        b_nx = ((m3 >> DXN) & 1u) ? (b_rd0 + GBN * B_FLOATS + 2 * DXN) : (b_zero0 + GBN * B_FLOATS);
      } else {
        b_nx = b_rd0 + (PAR ^ 1) * B_FLOATS;
      }
      // every segment: LDS reads of the NEXT group first (they return under the MFMAs of this one), then the 8 MFMAs
      frag_load(1, a_rd, b_cur, 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(0);
      __builtin_amdgcn_sched_barrier(0);
      frag_load(0, a_rd, b_cur, 2);
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(1);
      __builtin_amdgcn_sched_barrier(0);
      frag_load(1, a_rd, b_cur, 3);
      lds_store(std::integral_constant<int, PAR ^ 1>{}, std::integral_constant<bool, DX == 2>{}, std::integral_constant<int, GB ^ 1>{});
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(0);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      load_issue(t + 3, std::integral_constant<int, PAR ^ 1>{}, std::integral_constant<bool, DX == 0>{}, t + 2);
      frag_load(0, a_nx, b_nx, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(1);
      __builtin_amdgcn_sched_barrier(0);
      b_cur = b_nx;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    // ---- prologue: zero column, tile of the first step, loads of the next two, first fragments
    if (ROW) {
      for (int i = tid; i < 2 * BK; i += THREADS) sB[(i >> 5) * B_FLOATS + ((i & 31) >> 1) * BNP * 2 + (i & 1)] = 0.0f;
    }
    load_issue(ks0, I0{}, std::true_type{}, ks0);
    lds_store(I0{}, std::true_type{}, I0{});
    __syncthreads();
    DEVA_STAMP(1);
    load_issue(ks0 + 1, I1{}, std::false_type{}, ks0 + 1);
    load_issue(ks0 + 2, I0{}, std::false_type{}, -1);
    frag_load(0, a_rd0, b_cur, 0);

    if (ROW) {
      int s = 0;
      for (; s + 6 <= loop_steps; s += 6) {
        step(s, I0{}, I0{}, I0{});
        step(s + 1, I1{}, I1{}, I0{});
        step(s + 2, I2{}, I0{}, I0{});
        step(s + 3, I0{}, I1{}, I1{});
        step(s + 4, I1{}, I0{}, I1{});
        step(s + 5, I2{}, I1{}, I1{});
      }
      if (s < loop_steps) {
        step(s, I0{}, I0{}, I0{});
        step(s + 1, I1{}, I1{}, I0{});
        step(s + 2, I2{}, I0{}, I0{});
      }
    } else {
      int s = 0;
      for (; s + 2 <= loop_steps; s += 2) {
        step(s, I0{}, I0{}, I0{});
        step(s + 1, I0{}, I1{}, I0{});
      }
      if (s < loop_steps) step(s, I0{}, I0{}, I0{});
    }

    DEVA_STAMP(2);
    if (WK > 1) {
      // ---- the slice groups' partial sums meet in LDS (the tiles are dead), added in the fixed order 1, 2, ...
      __syncthreads();
      constexpr int NACC = TM * TN * 16;
      static_assert((WK - 1) * NACC * THREADS <= WK * REGION, "reduction scratch fits the tile regions");
      if (slice > 0) {
        float* red = smem + (slice - 1) * NACC * THREADS + tid;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((i * TN + j) * 16 + r) * THREADS] = acc[i][j][r];
      }
      __syncthreads();
      if (slice > 0) return;
      for (int k = 1; k < WK; ++k) {
        const float* red = smem + (k - 1) * NACC * THREADS + tid;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += red[((i * TN + j) * 16 + r) * THREADS];
      }
    }

    if (p.vec_out && p.splits == 1) {  // (split-K partial sums: the direct stores measured 3-4 % faster on the layers that split)
      static_assert((THREADS / 64) * 1024 <= REGION, "one 4 KB output-stage scratch per wave fits the dead tile buffers");
      __syncthreads();  // every wave is done with the tiles (and with the slice groups' partial sums)
      conv_store_tile_vec<TM, TN>(p, acc, m0 + wm0, n0 + wn0, lane, smem + wave * 1024);
    } else {
      conv_store_tile<TM, TN>(p, acc, m0, wm0, n0, wn0, l31, half);
    }
    DEVA_STAMP(3);
    if (PERSIST) __syncthreads();  // the output stage is done with the LDS before the next tile is staged
  } while (PERSIST && (tile_id += (int)gridDim.x) < n_tiles);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int MINW, int WK = 1>
int launch_tile_q4(const ConvArgs& a, hipStream_t st) {
  ConvArgs p = a;
  const bool uniform = a.ctot % BK == 0 && a.c0 % BK == 0;  // tap and source uniform per K step
  int kind;
  if (a.vec_ok && a.KH == 1 && a.KW == 1 && a.c0 % BK == 0) {
    kind = 0;
  } else if (a.vec_ok && uniform && a.KH == 3 && a.KW == 3 && a.pad == 1 && (a.k_layout & 0xf) == DEVA_KLAYOUT_CHUNK32 &&
             BN / WAVES_N == 32) {
    kind = 1;
  } else if (uniform) {
    kind = 2;
  } else {
    kind = 3;
  }
  if ((a.k_layout & 0xf) == DEVA_KLAYOUT_CHUNK32 && !uniform) {
    set_error("deva_conv2d: 32-channel-slab weights need c0 and c1 to be multiples of 32");
    return 2;
  }
  p.tiles_m = (int)ceil_div(a.cout, BM);
  p.tiles_n = (int)ceil_div(a.n_total, BN);
  const int ksteps_total = (int)ceil_div(a.K, BK);
  p.per_split = ksteps_total;
  const int blocks_eff_scale = WK;  // every workgroup already holds WK slice groups
  const int64_t blocks = (int64_t)p.tiles_m * p.tiles_n;
  p.group_m = conv_group_m(a.KH * a.KW, a.stride, BM, BN, blocks);
  p.splits = 1;
  int64_t target_blocks = 512;
  // measured (tools/convlab sweep, profiles/r04a): from ~190 tiles up the split (+ its reduction pass) loses
  bool want_split = a.ws && blocks < 192 && ksteps_total >= (blocks >= 128 ? 32 : 8);
#ifdef DEVA_CONV_PROBES
  {
    static const int forced = [] {
      const char* e = getenv("DEVA_CONV_SPLIT_TARGET");  // blocks to aim for; 0 = no split-K at all
      return e ? atoi(e) : -1;
    }();
    static const int forced_group = [] {
      const char* e = getenv("DEVA_CONV_GROUP_M");  // cout tiles per tile-order group; 0 = all
      return e ? atoi(e) : -1;
    }();
    if (forced_group >= 0) p.group_m = forced_group;
    if (forced == 0) want_split = false;
    if (forced > 0) {
      target_blocks = forced;
      want_split = a.ws && blocks < forced;
    }
  }
#endif
  if (want_split) {
    int64_t sp = ceil_div(target_blocks, blocks * blocks_eff_scale);
    if (sp > ksteps_total / (4 * WK)) sp = ksteps_total / (4 * WK);
    if (sp > 16) sp = 16;
    const int64_t fit = a.ws_elems / ((int64_t)a.cout * a.n_total);
    if (sp > fit) sp = fit;
    if (sp >= 2) {
      int per = (int)ceil_div(ksteps_total, sp);
      if (kind == 1) per = (per + 3 * WK - 1) / (3 * WK) * (3 * WK);  // row reuse: whole (slab, dy) groups per slice group
      sp = ceil_div(ksteps_total, per);
      p.splits = (int)sp;
      p.per_split = per;
    }
    if (p.splits < 2) p.splits = 1;
  }
  const dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)p.splits), block(64 * WAVES_M * WAVES_N * WK);
  if constexpr (BM == 128 && WK > 1) {
    if (kind >= 2) return launch_tile_q4<BM, BN, WAVES_M, WAVES_N, MINW, 1>(a, st);  // scalar kinds: no K-slice build at 1024 threads
  }
  switch (kind) {
    case 0: hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WAVES_M, WAVES_N, 0, MINW, WK, 1>), grid, block, 0, st, p); break;
    case 1:
      if constexpr (BN / WAVES_N == 32) {
        if (p.relu_in) hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WAVES_M, WAVES_N, 1, MINW, WK, 2>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WAVES_M, WAVES_N, 1, MINW, WK, 0>), grid, block, 0, st, p);
      }
      break;
    case 2:
      if constexpr (!(BM == 128 && WK > 1)) hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WAVES_M, WAVES_N, 2, MINW, WK, 1>), grid, block, 0, st, p);
      break;
    default:
      if constexpr (!(BM == 128 && WK > 1)) hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WAVES_M, WAVES_N, 3, MINW, WK, 1>), grid, block, 0, st, p);
      break;
  }
  if (p.splits > 1) return launch_splitk_reduce(p, st);
  return check_launch("deva_conv2d");
}

constexpr int PERSIST_MAX_WGS = 1024;  // workgroups of a gated re-run (a multiple of 8; 4 per CU fit: 35 KB of LDS each)

}  // namespace

// The fp32 re-run behind a split launch, gated on the flag the split kernel raises (a.gate): persistent 64x64 tiles for
// the two kinds the split kernels take (1x1 and 3x3 stride 1 on guard-banded inputs); -1 = not one of those, the caller
// launches the regular kernels with the gate.  No split-K, no K slices: when the gate opens, the result is the fp32
// kernels' in their plain K order (bit-identical to every unsplit WK = 1 variant).
int launch_conv_q4_gated(const ConvArgs& a, hipStream_t st) {
  constexpr int BM = 64, BN = 64;
  if (!a.gate || !a.vec_ok || a.cout <= 32 || a.c0 % BK) return -1;
  int kind;
  if (a.KH == 1 && a.KW == 1) {
    kind = 0;
  } else if (a.ctot % BK == 0 && a.KH == 3 && a.KW == 3 && a.pad == 1 && (a.k_layout & 0xf) == DEVA_KLAYOUT_CHUNK32) {
    kind = 1;
  } else {
    return -1;
  }
  ConvArgs p = a;
  p.tiles_m = (int)ceil_div(a.cout, BM);
  p.tiles_n = (int)ceil_div(a.n_total, BN);
  p.per_split = (int)ceil_div(a.K, BK);
  p.splits = 1;
  const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n;
  p.group_m = conv_group_m(a.KH * a.KW, a.stride, BM, BN, tiles);
  const dim3 grid((unsigned)(tiles < PERSIST_MAX_WGS ? tiles : PERSIST_MAX_WGS)), block(256);
  if (kind == 0) {
    hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, 2, 2, 0, 1, 1, 1, true>), grid, block, 0, st, p);
  } else if (p.relu_in) {
    hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, 2, 2, 1, 1, 1, 2, true>), grid, block, 0, st, p);
  } else {
    hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, 2, 2, 1, 1, 1, 0, true>), grid, block, 0, st, p);
  }
  return check_launch("deva_conv2d (gated fp32 re-run)");
}

int launch_conv_q4(const ConvArgs& a, hipStream_t st) {
#ifdef DEVA_CONV_PROBES  // `make PROBES=1`: A/B runs of the tile policy (tools/convlab)
  {
    static const int forced = [] {
      const char* e = getenv("DEVA_CONV_TILE");
      return e ? atoi(e) : 0;
    }();
    switch (forced) {
      case 128: if (a.cout >= 64) return launch_tile_q4<128, 128, 2, 4, 4>(a, st); break;
      case 64: if (a.cout > 32) return launch_tile_q4<64, 64, 2, 2, 1>(a, st); break;
      case 642: if (a.cout > 32) return launch_tile_q4<64, 64, 2, 2, 1, 2>(a, st); break;
      case 644: if (a.cout > 32) return launch_tile_q4<64, 64, 2, 2, 1, 4>(a, st); break;
      case 1282: if (a.cout >= 64) return launch_tile_q4<128, 128, 2, 4, 4, 2>(a, st); break;
      case 12864: if (a.cout >= 64) return launch_tile_q4<128, 64, 2, 2, 2>(a, st); break;
      case 64128: if (a.cout > 32) return launch_tile_q4<64, 128, 1, 4, 2>(a, st); break;
      default: break;
    }
  }
#endif
  if (a.cout <= 32) return launch_tile_q4<32, 128, 1, 4, 1>(a, st);
  // Tile policy (warm sweeps over the layers of the 480p frame, tools/convlab/sweep.sh, profiles/r04a):
  //  * 128x128 (8 waves, wave tile 64x32) from 64 tiles up -- unless it would leave most CUs empty while 64x64 tiles
  //    fill the chip; with at most one tile per CU the workgroup carries two K-slice groups (16 waves per CU);
  //  * 64x64 (4 waves) otherwise; layers with few tiles and a long K loop run 2 or 4 K-slice groups per workgroup.
  const int64_t blocks128 = ceil_div(a.cout, 128) * ceil_div(a.n_total, 128);
  const int64_t blocks64 = ceil_div(a.cout, 64) * ceil_div(a.n_total, 64);
  const int ksteps = (int)ceil_div(a.K, BK);
  const bool vec_kind = a.vec_ok && a.c0 % BK == 0 &&
                        ((a.KH == 1 && a.KW == 1) || (a.ctot % BK == 0 && a.KH == 3 && a.KW == 3 && a.pad == 1 &&
                                                      (a.k_layout & 0xf) == DEVA_KLAYOUT_CHUNK32));
  if (a.cout >= 128 && blocks128 >= 64 && !(blocks128 < 192 && blocks64 >= 256)) {
    if (vec_kind && blocks128 <= 256 && ksteps >= 16) return launch_tile_q4<128, 128, 2, 4, 4, 2>(a, st);
    return launch_tile_q4<128, 128, 2, 4, 4>(a, st);
  }
  if (ksteps >= 32 && blocks64 >= 64 && blocks64 <= 208) return launch_tile_q4<64, 64, 2, 2, 1, 4>(a, st);
  if ((ksteps >= 16 && blocks64 <= 208) || (ksteps >= 32 && blocks64 <= 512)) return launch_tile_q4<64, 64, 2, 2, 1, 2>(a, st);
  return launch_tile_q4<64, 64, 2, 2, 1>(a, st);
}

}  // namespace deva

#ifdef DEVA_CONV_PROBES
extern "C" int deva_conv_set_probe(unsigned long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_probe), &buf, sizeof(buf));
}
#endif
