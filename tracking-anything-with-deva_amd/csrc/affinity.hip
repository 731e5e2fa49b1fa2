// Memory read of DEVA on gfx950: anisotropic-L2 similarity -> exact top-k -> softmax (-> usage),
// fused so that the N x HW similarity matrix is never written (the reference materialises it
// ~12 times per frame, memory_utils.py:29-74).
//
// Similarity (memory_utils.py:29-43), per memory token n and query q:
//     A = sum_c mk[n][c]^2 * qe[c][q]          B = sum_c mk[n][c] * (qk[c][q]*qe[c][q])
//     sim = ((-A + 2B) - bsq[q]) * ms[n] / sqrt(64)
// A and B run on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 FMA chains) with the
// channels in natural order, each in its own accumulator and combined in the reference's order, so
// the scores agree with an fp32 FMA GEMM to the last bits -- top-k is discontinuous, a 1e-5
// relative error flips memory tokens in and out of the softmax support.
//
// Work decomposition: one wave owns 32 queries (the MFMA N dimension) and streams a range of
// memory tokens in tiles of 32 (the MFMA M dimension); the query operand lives in registers for
// the whole kernel, the key rows are read straight from the token-major bank and prefetched one
// tile ahead (a 32x64 fp32 tile per 4096 matrix-pipe cycles -- operand traffic is irrelevant here,
// the kernel is bound by the fp32 MFMA rate).  Per query the wave keeps a candidate list in LDS
// (LCAP slots of 6 bytes) with its length and threshold in registers: scores >= the running lower bound
// of the k-th best are appended (~k*(1+ln(n/k)) appends per query over n tokens); a list that could
// overflow is pruned to the entries >= the k-th largest of its 64 per-lane maxima (at least k entries
// are >= that value, so the final result stays exact).  That k-th largest is found by rank counting
// over the 64 lane values through a 256-B LDS row (128 independent compare / add pairs) instead of a
// ballot bisection, whose ~10-30 dependent VALU->SALU->branch round trips cost ~3 700 cycles per list
// and, at three prune rounds of 32 lists per range, a third of the kernel at the N = 10 000 shape
// (round-1 profile); the exact bisection on unique (score, token) keys remains as the fallback for
// banks full of identical scores.  grid.y splits the bank into token ranges so small frames still
// fill the chip; every range hands over its (pruned, <= CAP entries) lists with their lengths and a
// second kernel (one wave per query) selects the exact top-k over all ranges, applies exp/normalise
// and accumulates the usage counters.
//
// What the round-2 measurements say about this part (profiles/r02b_affinity_shapes.txt, tools/probe):
// fp32 MFMAs and VALU work do not overlap -- kernel time ~ MFMA cycles + VALU issue cycles + exposed
// waits at any occupancy -- and the exposed waits are the prunes' LDS round trips (per-wave lists) or
// the skew collected by workgroup barriers (shared lists).  Hence the kernel shapes (see
// deva_affinity_force_shape in the header; the automatic choice is by bank size):
//   affinity_topk_wg_kernel   short banks (<= 20 000 - 30 000 tokens): the waves of a workgroup share 32
//                             queries' lists (LDS-atomic appends, one threshold per query, two barriers per
//                             tile) -- fewest appends and prunes.  Four waves, 352 slots, two workgroups per
//                             CU on large frames; eight waves, 704 slots, one workgroup per CU on small
//                             frames (half the token ranges, so half the lists to merge afterwards);
//   affinity_topk_kernel      longer banks: four waves x four query groups, 100-slot per-wave lists, two
//                             workgroups per CU, no barriers, key rows prefetched after the last MFMA
//                             and read in place;
//   affinity_topk_pp_kernel   (A/B only) eight waves in two groups alternating a matrix and a scoring phase:
//                             bit-identical, slower -- the fp32 MFMAs share the VALU data path, so there is
//                             nothing for the other group's scoring to overlap with.
// Common to all: the prefetched key rows keep their registers reserved until the MFMAs read them
// (DEVA_KEEP_ROWS: otherwise the scoring phase waits for the loads it is supposed to cover); rows are filed only when some lane passes, accumulators stay in VGPRs
// (-amdgpu-mfma-vgpr-form, see the Makefile), operand rows are loaded with a per-half-lane offset instead
// of being selected, appends store raw fp32 bits (ordered only when a list is pruned / handed over), and
// the scrambled tile order is advanced incrementally (a 64-bit modulo per tile was ~300 scalar
// instructions on the critical path).
#include <math.h>

#include <type_traits>

#include "common.h"

#pragma clang fp contract(off)

namespace deva {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4_u __attribute__((aligned(4)));  // 16-B load from a dword-aligned address

constexpr int CK = 64;
constexpr int QT = 32;            // queries per wave
constexpr int CAP = 64;           // candidate slots per (range, query) handed to the merge kernel (one per lane)
// in-kernel candidate lists: LCAP slots of 6 bytes (order-preserving score bits + 16-bit token offset
// inside the range); row stride LCAP + 1 (odd: spreads the LDS banks).
[[maybe_unused]] constexpr int LCAP_WIDE = 176;    // one workgroup per CU (136 KiB of lists; probe-build shapes 1 and 3)
constexpr int LCAP_DUAL = 100;    // two workgroups per CU (2 x 76 KiB)
constexpr int K_MAX = 32;          // top-k supported by the list / hand-over sizing below
constexpr int MAX_SPLITS = 32;    // one 64-bit key per lane and range in the merge kernel
constexpr int WAVES = 4;
constexpr int TOKT = 32;          // tokens per tile

__device__ __forceinline__ uint32_t orderable(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_orderable(uint32_t o) {
  const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}
// 64-bit candidate keys handed to the merge kernel: order-preserving score bits << 32 | ~token index,
// so a larger key is a better candidate (higher score first, then lower token index).

// Hand-over point of cross-lane communication through LDS inside one wave: a wavefront-scope acquire-release fence (the
// LDS pipeline executes a wave's accesses in order, so the fence costs no instruction; it is what makes the
// ordering part of the program instead of an assumption about the compiler) plus a wave barrier for the scheduler.
#define DEVA_COMPILER_FENCE()                               \
  do {                                                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                        \
  } while (0)

// The prefetched key rows sit in registers for a whole tile while their loads are in flight.  Only two of
// the four floats of a 16-B piece are MFMA operands, and the register allocator would hand the other two to
// unrelated values of the scoring phase -- writing such a register has to wait (s_waitcnt vmcnt) for the load
// that targets it, which exposes the memory latency the prefetch is there to hide.  Naming every piece as an
// asm input right before its first use keeps all four registers reserved until then (no instruction emitted).
// For the same reason nothing is computed from the prefetched shrinkage until it is stored to LDS.
#define DEVA_KEEP_ROWS(rows)                                     \
  _Pragma("unroll") for (int j_ = 0; j_ < CK / 4; ++j_) {        \
    asm volatile("" ::"v"(rows[j_]));                            \
  }


__device__ __forceinline__ int wave_count(bool pred) { return __popcll(__builtin_amdgcn_ballot_w64(pred)); }
// number of set bits of a wave ballot below this lane
__device__ __forceinline__ int prefix_below(unsigned long long b) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
}

// Exact k-th largest of the unique non-zero 64-bit keys held E per lane (0 = empty slot) by bitwise
// bisection with wave ballots: 32 steps on the score half; the index half only if the k-th score is
// tied.  Requires >= k non-zero keys.  Everything >= the returned key is the top-k set.
template <int E>
__device__ __forceinline__ uint64_t kth_largest(const uint64_t (&e)[E], int n_live, int k) {
  uint32_t T = 0;
  for (int b = 31; b >= 0; --b) {
    const uint32_t trial = T | (1u << b);
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < E; ++i)
      if (i < n_live) cnt += wave_count((uint32_t)(e[i] >> 32) >= trial);
    if (cnt >= k) {
      T = trial;
      // exactly k keys at or above the trial: it separates the top-k set, no need to resolve the
      // remaining bits (typically reached after ~10 of the 32 steps)
      if (cnt == k) return (uint64_t)T << 32;
    }
  }
  int above = 0, ties = 0;
#pragma unroll
  for (int i = 0; i < E; ++i)
    if (i < n_live) {
      above += wave_count((uint32_t)(e[i] >> 32) > T);
      ties += wave_count((uint32_t)(e[i] >> 32) == T);
    }
  const int need = k - above;  // ties to keep: the ones with the largest low half (lowest token index)
  uint32_t L = 0;
  if (ties > need) {
    for (int b = 31; b >= 0; --b) {
      const uint32_t trial = L | (1u << b);
      int cnt = 0;
#pragma unroll
      for (int i = 0; i < E; ++i)
        if (i < n_live) cnt += wave_count((uint32_t)(e[i] >> 32) == T && (uint32_t)e[i] >= trial);
      if (cnt >= need) L = trial;
    }
  }
  return ((uint64_t)T << 32) | L;
}

// Lower bound of the k-th largest of the 64 lane values `m` (order-preserving score bits, 0 = empty
// lane; needs >= k non-empty lanes) by rank counting: the low 6 bits are replaced by the lane number so
// that the values are unique, the 64 values go through a 256-B LDS scratch row, every lane reads them
// back as 16 broadcast 16-B reads and counts the values above its own (128 independent compare / add
// pairs -- no readlane hazards and no dependent VALU -> SALU -> branch chain as in a bisection), and the
// lane of rank k-1 publishes its value with the low bits cleared: at least k lanes have m >= the result.
__device__ __forceinline__ uint32_t kth_lane_value(uint32_t m, int k, int lane, uint32_t* scratch) {
  const uint32_t u = (m & ~63u) | (uint32_t)lane;
  scratch[lane] = u;
  DEVA_COMPILER_FENCE();
  int rank = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint4 o = *reinterpret_cast<const uint4*>(scratch + 4 * j);
    rank += (o.x > u) ? 1 : 0;
    rank += (o.y > u) ? 1 : 0;
    rank += (o.z > u) ? 1 : 0;
    rank += (o.w > u) ? 1 : 0;
  }
  DEVA_COMPILER_FENCE();
  const unsigned long long b = __builtin_amdgcn_ballot_w64(rank == k - 1);  // exactly one lane: the values are unique
  return (uint32_t)__builtin_amdgcn_readlane((int)u, __ffsll(b) - 1) & ~63u;
}

// Prune one candidate list (wave-cooperative, 64 <= c <= 64*E entries).  Threshold = (a lower bound of)
// the k-th largest of the 64 per-lane maxima: at least k entries of the list are >= it, so nothing below
// it can belong to the top-k -- the result stays exact.  At most E*k entries survive (k lanes hold a
// maximum >= the threshold, each with <= E entries), typically k .. 1.5k.  Returns the threshold (score
// bits); *kept = number of survivors (compacted to the front, unsorted).
template <int E>
__device__ __forceinline__ uint32_t prune_list(uint32_t* sc, uint16_t* tk, uint32_t c, int k, int lane, int* kept,
                                               uint32_t* scratch) {
  uint32_t raw[E], e[E], t[E];  // the lists hold raw fp32 bits (cheap appends); ordered here
  uint32_t m = 0u;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const uint32_t j = (uint32_t)lane + 64u * i;
    raw[i] = (j < c) ? sc[j] : 0u;
    e[i] = (j < c) ? orderable(__uint_as_float(raw[i])) : 0u;
    t[i] = (j < c) ? (uint32_t)tk[j] : 0u;
    m = e[i] > m ? e[i] : m;
  }
  const uint32_t thr = kth_lane_value(m, k, lane, scratch);
  const uint32_t thr1 = thr ? thr : 1u;  // empty slots (0) never pass: one compare per entry
  DEVA_COMPILER_FENCE();
  int base = 0;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const bool keep = e[i] >= thr1;
    const unsigned long long b = __builtin_amdgcn_ballot_w64(keep);
    if (keep) {
      const int w = base + prefix_below(b);
      sc[w] = raw[i];
      tk[w] = (uint16_t)t[i];
    }
    base += __popcll(b);
  }
  DEVA_COMPILER_FENCE();
  *kept = base;
  return thr;
}

// Exact variant (slow path): threshold = the exact k-th largest of the 64 per-lane maxima of the UNIQUE
// 64-bit keys (score bits, token), by ballot bisection.  Exactly k lanes hold a maximum >= it, so at
// most E*k entries survive whatever the scores are -- including banks full of identical keys, where the
// rank-counting prune (which compares 26 score bits) cannot separate anything.
template <int E>
__device__ __forceinline__ uint32_t prune_list_exact(uint32_t* sc, uint16_t* tk, uint32_t c, int k, int lane,
                                                     int* kept) {
  uint64_t e[E];
  uint64_t m[1] = {0ull};
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const uint32_t j = (uint32_t)lane + 64u * i;
    e[i] = (j < c) ? (((uint64_t)orderable(__uint_as_float(sc[j])) << 32) | (uint64_t)(0xffffu - tk[j])) : 0ull;
    m[0] = e[i] > m[0] ? e[i] : m[0];
  }
  const uint64_t thr = kth_largest<1>(m, 1, k);
  DEVA_COMPILER_FENCE();
  int base = 0;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const bool keep = e[i] >= thr && e[i] != 0ull;
    const unsigned long long b = __builtin_amdgcn_ballot_w64(keep);
    if (keep) {
      const int w = base + prefix_below(b);
      sc[w] = __float_as_uint(from_orderable((uint32_t)(e[i] >> 32)));
      tk[w] = (uint16_t)(0xffffu - (uint32_t)(e[i] & 0xffffu));
    }
    base += __popcll(b);
  }
  DEVA_COMPILER_FENCE();
  *kept = base;
  return (uint32_t)(thr >> 32);
}

struct AffArgs {
  const float* key_long;
  const float* shr_long;
  int n_long;
  const float* key_work;
  const float* shr_work;
  int n_total;
  const float* qk;
  const float* qe;
  int hw;
  int k;
  int splits;
  int total_tiles;
  uint64_t* part;     // [splits][hw][CAP] candidate keys
  uint32_t* part_cnt;  // [splits][hw] live entries of each list
  int ablate;          // timing probes, builds with -DDEVA_AFFINITY_PROBES only (0 otherwise): 1 = file nothing,
                       // 2 = no key loads in the loop, 4 = no scoring, 8 = no MFMAs; results are meaningless when non-zero
  uint64_t* probe;     // probe builds: cycle stamps of the first workgroups' phases (tools/probe/affinity_phases.py)
  const uint32_t* guard;  // != NULL: run only if *guard != 0 (the fp32 path as the fall-back of the fp16 pre-filter)
};

#ifdef DEVA_AFFINITY_PROBES
#define DEVA_ABLATE(bit) ((p.ablate & (bit)) != 0)
#else
#define DEVA_ABLATE(bit) false  // the probe paths are compiled out of the product build
#endif

#ifdef DEVA_AFFINITY_PROBES
#define DEVA_STAMP(slot)                                                                    \
  do {                                                                                      \
    if (pb && it < 64 && lane == 0) pb[it * 8 + (slot)] = __builtin_readcyclecounter();     \
  } while (0)
#else
#define DEVA_STAMP(slot) \
  do {                   \
  } while (0)
#endif

// key tile of the SHARED variant in LDS: [buffer][channel parity][token row][TROW floats]; a row holds
// the 32 even (or odd) channels of one token; 36-float stride: 16-B aligned rows whose bank groups rotate
constexpr int TROW = 36;

// LCAP: list slots per query; MINB: workgroups per CU the register / LDS budget is sized for.
// SHARED: the four waves of a workgroup (same token range, different queries) load every key tile ONCE,
// coalesced (8 KiB contiguous: 2 x 16 B per lane), and pass it through LDS, de-interleaved into even /
// odd channels so that each lane then reads exactly the 32 operands it feeds to the MFMAs with eight
// 16-B LDS reads.  Without it every lane reads its own 256-B row (16 loads touching 64 cache lines
// each, four times per workgroup): ~1 000 L1 line accesses per wave and tile against 4 100 MFMA
// cycles -- the round-1 counters show the waves of that kernel waiting on memory for 34-51 % of their
// cycles.  One s_barrier per tile (double-buffered tile).
template <int LCAP, int MINB, bool SHARED, bool LATE = false>
__global__ __launch_bounds__(WAVES * 64, MINB) void affinity_topk_kernel(const AffArgs p) {
  if (p.guard && *p.guard == 0u) return;  // uniform over the grid
  constexpr int LSTRIDE = LCAP + 1;
  constexpr int E = (LCAP + 63) / 64;  // list entries per lane in a prune
  static_assert(LCAP - TOKT >= 64, "a list is pruned only when every lane holds an entry");
  static_assert(LCAP < 65536 && CAP == 64, "hand-over: one key per lane");
  static_assert(E * K_MAX <= LCAP - TOKT, "one exact prune (<= E*k survivors) must get below the in-loop limit");
  static_assert(2 * K_MAX <= CAP, "an exact prune of a two-entries-per-lane list must fit the hand-over");
  __shared__ uint32_t s_sc[WAVES][QT][LSTRIDE];  // candidate scores (fp32 bits)
  __shared__ uint16_t s_tk[WAVES][QT][LSTRIDE];  // candidate tokens (offset inside this range)
  __shared__ __attribute__((aligned(16))) float s_ms[WAVES][TOKT];  // shrinkage / 8 of the current tile
  __shared__ __attribute__((aligned(16))) uint32_t s_rank[WAVES][64];  // scratch row of the prune
  __shared__ __attribute__((aligned(16))) float s_tile[SHARED ? 2 : 1][2][SHARED ? TOKT : 1][SHARED ? TROW : 4];
  __shared__ __attribute__((aligned(16))) float s_tms[2][TOKT];  // SHARED: shrinkage / 8, per tile buffer

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int l31 = lane & 31;
  const int half = lane >> 5;
  const int q0 = (blockIdx.x * WAVES + wave) * QT;
  // a wave without queries (ragged last query block): idle, except that in the SHARED variant it still
  // loads its quarter of every tile and takes part in the barriers
  const bool active = q0 < p.hw;
  if (!SHARED && !active) return;
  const int split = blockIdx.y;

  // NB plain (non-volatile) LDS accesses: hipcc puts `s_waitcnt vmcnt(0)` next to every volatile
  // access, which would drain the key-row prefetch at each list operation.  Program order on may-alias
  // LDS locations plus the in-order LDS pipeline give the cross-lane visibility needed inside a wave;
  // DEVA_COMPILER_FENCE() marks the hand-over points.
  uint32_t* csc = &s_sc[wave][0][0];
  uint16_t* ctk = &s_tk[wave][0][0];
  float* msl = &s_ms[wave][0];
  uint32_t* srow = csc + l31 * LSTRIDE;
  uint16_t* trow = ctk + l31 * LSTRIDE;

  // ---- query operand (registers, whole kernel).  MFMA t consumes channels 2t (lanes 0-31) and
  // 2t+1 (lanes 32-63): natural channel order in the accumulation chain.
  const int q = min(q0 + l31, p.hw - 1);
  float bqe[CK / 2], bqk[CK / 2];
  // bsq = sum_c qe*qk^2 in the order ATen's CPU sum uses for this reduction (four 16-channel
  // partial sums, then ((s0+s1)+s2)+s3) -- probed bit-equal on >99% of queries
  float bs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int t = 0; t < CK / 2; ++t) {
    const float e0 = p.qe[(int64_t)(2 * t) * p.hw + q], e1 = p.qe[(int64_t)(2 * t + 1) * p.hw + q];
    const float k0 = p.qk[(int64_t)(2 * t) * p.hw + q], k1 = p.qk[(int64_t)(2 * t + 1) * p.hw + q];
    bs[t >> 3] += e0 * (k0 * k0);
    bs[t >> 3] += e1 * (k1 * k1);
    bqe[t] = half ? e1 : e0;
    bqk[t] = half ? (k1 * e1) : (k0 * e0);
  }
  const float bsq = ((bs[0] + bs[1]) + bs[2]) + bs[3];
  const f32x2 bsq2 = {bsq, bsq};

  // Token ranges are tile-cyclic (range s owns tiles s, s+S, s+2S, ...) and every range visits its
  // tiles in a multiplicatively scrambled order.  The filter only works while the threshold is
  // representative of what is still to come: a video memory is ordered in time and space, scores
  // drift upwards towards the best-matching frame / region, and a front-to-back scan of a contiguous
  // range then appends nearly every token it meets (measured: 3x slower than on shuffled keys).
  const int n_my = (p.total_tiles - split + p.splits - 1) / p.splits;  // tiles of this range
  int stride = 61;  // a prime that does not divide n_my: c -> (c * stride) % n_my is a permutation
  if (n_my % 61 == 0) stride = (n_my % 59 == 0) ? 53 : 59;
  // the cyclic index of visit i is (i * stride) % n_my, advanced by one conditional subtraction per tile
  const int step = (n_my > 0) ? stride % n_my : 0;
  auto advance = [&](int c) {
    c += step;
    return c >= n_my ? c - n_my : c;
  };
  // candidate tokens are stored as 16 bits: (cyclic tile index << 5) | row

  // ---- per-query state in registers (the same value in both half-lanes of a query): list length and
  // the running lower bound of the k-th best score
  uint32_t cnt = 0;
  float tau = DEVA_ABLATE(1) ? INFINITY : -INFINITY;

  // Key rows are software-prefetched one tile ahead: lane (l31, half) reads the 256-B row of token
  // n_base + l31 while the matrix pipe works on the previous tile.  Lanes of the upper half start one
  // float later, so that x / z of every 16-B piece are exactly the channels 2t+half the lane feeds to
  // the MFMAs -- no per-element select (VALU work does not overlap the wave's own MFMAs, every
  // instruction saved here is time saved; tools/probe/README.md).  The last piece is read aligned
  // (a shifted read would touch the next row) and selected.
  f32x4 xbuf[CK / 4];
  float ms_buf;
  auto prefetch = [&](int cyc) {
    const int tile = split + p.splits * cyc;
    const int n_mine = min(tile * TOKT + l31, p.n_total - 1);
    const float* krow = (n_mine < p.n_long) ? (p.key_long + (int64_t)n_mine * CK)
                                            : (p.key_work + (int64_t)(n_mine - p.n_long) * CK);
    // 1/sqrt(CK) folded into the shrinkage: (x * ms) * 0.125 == x * (ms * 0.125) exactly
    ms_buf = ((n_mine < p.n_long) ? p.shr_long[n_mine] : p.shr_work[n_mine - p.n_long]);  // scaled when stored: see DEVA_KEEP_ROWS
    const float* shifted = krow + half;
#pragma unroll
    for (int j = 0; j < CK / 4 - 1; ++j) xbuf[j] = *reinterpret_cast<const f32x4_u*>(shifted + 4 * j);
    xbuf[CK / 4 - 1] = *reinterpret_cast<const f32x4*>(krow + CK - 4);
  };
  // SHARED: wave w loads token rows 8w .. 8w+7 of a tile, lane L the 32 bytes (channels 8c .. 8c+7,
  // c = L & 7) of row 8w + (L >> 3); wave 0 also loads the 32 shrinkage values
  f32x4 g0, g1;
  float g_ms = 0.0f;
  auto load_shared = [&](int cyc_) {
    const int tile = split + p.splits * cyc_;
    const int n_mine = min(tile * TOKT + 8 * wave + (lane >> 3), p.n_total - 1);
    const float* krow = (n_mine < p.n_long) ? (p.key_long + (int64_t)n_mine * CK)
                                            : (p.key_work + (int64_t)(n_mine - p.n_long) * CK);
    const f32x4* src = reinterpret_cast<const f32x4*>(krow + 8 * (lane & 7));
    g0 = src[0];
    g1 = src[1];
    if (wave == 0 && lane < TOKT) {
      const int n_s = min(tile * TOKT + lane, p.n_total - 1);
      g_ms = (n_s < p.n_long) ? p.shr_long[n_s] : p.shr_work[n_s - p.n_long];  // scaled when stored: no wait here
    }
  };
  auto store_shared = [&](int buf) {
    const int r = 8 * wave + (lane >> 3), c4 = 4 * (lane & 7);
    *reinterpret_cast<f32x4*>(&s_tile[buf][0][r][c4]) = f32x4{g0[0], g0[2], g1[0], g1[2]};  // channels 8c, +2, +4, +6
    *reinterpret_cast<f32x4*>(&s_tile[buf][1][r][c4]) = f32x4{g0[1], g0[3], g1[1], g1[3]};  // channels 8c+1, +3, +5, +7
    if (wave == 0 && lane < TOKT) s_tms[buf][lane] = g_ms * 0.125f;  // 1/sqrt(CK) folded in (exact)
  };
  int cyc = 0;  // cyclic tile index of the current visit
  if (n_my > 0) {
    if (SHARED) {
      load_shared(cyc);
    } else {
      prefetch(cyc);
    }
  }

  // fast = rank-counting prune first; the exact prune runs if that left the list above the limit (or
  // alone if !fast).  One exact prune leaves <= E*k <= limit entries in the tile loop.
  auto prune_over = [&](uint32_t limit, bool fast) {
    uint64_t need = __builtin_amdgcn_ballot_w64(cnt > limit) & 0xffffffffull;
    while (need) {
      const int qq = __ffsll((unsigned long long)need) - 1;
      need &= need - 1;
      const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cnt, qq);
      int kept = (int)c;
      uint32_t thr = 0u;
      if (fast) thr = prune_list<E>(csc + qq * LSTRIDE, ctk + qq * LSTRIDE, c, p.k, lane, &kept, &s_rank[wave][0]);
      if ((uint32_t)kept > limit) {
        const uint32_t thr2 =
            prune_list_exact<E>(csc + qq * LSTRIDE, ctk + qq * LSTRIDE, (uint32_t)kept, p.k, lane, &kept);
        thr = thr2 > thr ? thr2 : thr;
      }
      if (l31 == qq) {
        cnt = (uint32_t)kept;
        tau = from_orderable(thr);
      }
    }
  };

  for (int it = 0; it < n_my; ++it) {
    const int tile = split + p.splits * cyc;
    const int n_base = tile * TOKT;
    const uint32_t tok0 = (uint32_t)(cyc * TOKT + 4 * half);

    float a_op[CK / 2];
    if (SHARED) {
      // ---- publish the tile loaded during the previous iteration, start loading the next one
      store_shared(it & 1);
      __syncthreads();
      if (it + 1 < n_my) {
        cyc = advance(cyc);
        if (!DEVA_ABLATE(2)) load_shared(cyc);
      }
      if (!active) continue;
      prune_over((uint32_t)(LCAP - TOKT), true);
      const float* arow = &s_tile[it & 1][half][l31][0];
#pragma unroll
      for (int j = 0; j < CK / 8; ++j) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(arow + 4 * j);
        a_op[4 * j] = v[0];
        a_op[4 * j + 1] = v[1];
        a_op[4 * j + 2] = v[2];
        a_op[4 * j + 3] = v[3];
      }
      msl = &s_tms[it & 1][0];
    } else {
      // ---- prune lists that could overflow during this tile (at most 32 appends per query per tile)
      prune_over((uint32_t)(LCAP - TOKT), true);

      // ---- this tile's operand: channel 2t + half of this lane's token.  Early prefetch (default): copy
      // the operands out and start the next loads at once (a whole tile of latency cover); LATE: the MFMAs
      // read the prefetched rows in place (no copies) and the next loads start after the last MFMA.
      DEVA_KEEP_ROWS(xbuf);
#pragma unroll
      for (int j = 0; j < CK / 4 - 1; ++j) {
        a_op[2 * j] = xbuf[j][0];
        a_op[2 * j + 1] = xbuf[j][2];
      }
      a_op[CK / 2 - 2] = half ? xbuf[CK / 4 - 1][1] : xbuf[CK / 4 - 1][0];
      a_op[CK / 2 - 1] = half ? xbuf[CK / 4 - 1][3] : xbuf[CK / 4 - 1][2];
      if (lane < TOKT) msl[lane] = ms_buf * 0.125f;  // 1/sqrt(CK) folded in (exact)
      DEVA_COMPILER_FENCE();
      if (!LATE) {
        if (it + 1 < n_my) cyc = advance(cyc);
        if (!DEVA_ABLATE(2)) prefetch(cyc);
      }
    }

    f32x16 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      accA[r] = 0.0f;
      accB[r] = 0.0f;
    }
    if (!DEVA_ABLATE(8)) {
#pragma unroll
      for (int t = 0; t < CK / 2; ++t) {
        const float a = a_op[t];
        accA = __builtin_amdgcn_mfma_f32_32x32x2f32(a * a, bqe[t], accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bqk[t], accB, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        accA[r] = a_op[r];
        accB[r] = a_op[r + 16];
      }
    }
    if (!SHARED && LATE) {
      DEVA_COMPILER_FENCE();
      if (it + 1 < n_my) cyc = advance(cyc);
      if (!DEVA_ABLATE(2)) prefetch(cyc);
    }

    // ---- scores of this lane: query l31, tokens n_base + (r&3) + 8*(r>>2) + 4*half, two accumulator
    // rows per packed-fp32 instruction.  A row is filed only if some lane of the wave passes its
    // threshold (about every other row once the thresholds have settled); the position in the list the
    // two half-lanes of a query share comes from the ballot.
    float4 ms4[4];  // scaled shrinkage in accumulator-row order: rows 4g..4g+3 <-> tokens 8g+4*half..+3
#pragma unroll
    for (int g = 0; g < 4; ++g) ms4[g] = *reinterpret_cast<const float4*>(&msl[8 * g + 4 * half]);
    const int rows_left = p.n_total - n_base;  // >= TOKT except in the last tile of the bank
    auto file_rows = [&](auto full) {
      constexpr bool full_tile = decltype(full)::value;
#pragma unroll
      for (int r2 = 0; r2 < 8; ++r2) {
        const f32x2 a2 = {accA[2 * r2], accA[2 * r2 + 1]};
        const f32x2 b2 = {accB[2 * r2], accB[2 * r2 + 1]};
        const float4 m4 = ms4[r2 >> 1];
        const f32x2 m2 = (r2 & 1) ? f32x2{m4.z, m4.w} : f32x2{m4.x, m4.y};
        f32x2 v2 = ((b2 + b2) - a2) - bsq2;  // == (-A + 2B) - bsq, every step correctly rounded
        v2 = v2 * m2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = 2 * r2 + h;
          const int j0 = (r & 3) + 8 * (r >> 2);  // + 4 * half
          const float v = v2[h];
          bool ok = v >= tau;
          if (!full_tile) ok = ok && (j0 + 4 * half < rows_left);
          const unsigned long long b = __builtin_amdgcn_ballot_w64(ok);
          if (b) {
            // the two half-lanes of query l31 are lanes l31 and 32 + l31: one 32-bit bit-field extract each
            const uint32_t ok_lo = ((uint32_t)b >> l31) & 1u, ok_hi = ((uint32_t)(b >> 32) >> l31) & 1u;
            if (ok) {
              const uint32_t pos = cnt + (half ? ok_lo : 0u);
              srow[pos] = __float_as_uint(v);  // raw bits: ordered when a list is pruned / handed over
              trow[pos] = (uint16_t)(tok0 + j0);
            }
            cnt += ok_lo + ok_hi;
          }
        }
      }
    };
    if (DEVA_ABLATE(4)) {  // keep the accumulators alive without scoring them
      if (accA[0] + accB[15] == 12345.678f) cnt += 1;
    } else if (rows_left >= TOKT) {
      file_rows(std::true_type{});
    } else {
      file_rows(std::false_type{});
    }
    DEVA_COMPILER_FENCE();
  }

  // ---- hand-over: every list is pruned to at most CAP = 64 entries (one per lane) and written with its
  // length.  A rank-counting round first; exact rounds only for lists still above CAP (an exact round
  // leaves <= 64 * ceil(c/64) * k / 64 entries: <= 3k from 176, <= 2k <= 64 from 96).  The exact top-k
  // selection over all ranges happens in the merge kernel, where one wave per query gives thousands of
  // independent waves -- here it would run serially, 32 lists per wave.
  static_assert(LCAP <= 192, "two exact rounds must reach CAP");
  if (!active) return;
  prune_over((uint32_t)CAP, true);
  prune_over((uint32_t)CAP, false);
  DEVA_COMPILER_FENCE();
  const int nq = min(QT, p.hw - q0);
  uint64_t* dst = p.part + ((int64_t)split * p.hw + q0) * CAP;
  if (lane < nq) p.part_cnt[(int64_t)split * p.hw + q0 + lane] = cnt;  // lanes 0..31 hold query l31 = lane
  for (int ql = 0; ql < nq; ++ql) {
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cnt, ql);
    if ((uint32_t)lane < c) {
      const uint32_t off = (uint32_t)ctk[ql * LSTRIDE + lane];
      const uint32_t token = ((off >> 5) * (uint32_t)p.splits + (uint32_t)split) * TOKT + (off & 31u);
      dst[(int64_t)ql * CAP + lane] =
          ((uint64_t)orderable(__uint_as_float(csc[ql * LSTRIDE + lane])) << 32) | (uint64_t)(~token);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Workgroup-shared candidate lists.  The four waves of a workgroup own the SAME 32 queries and split the
// token tiles of the range among themselves (wave w visits tiles w, w+4, ... of the scrambled order);
// they append to one set of 32 lists through LDS atomics and filter against one threshold per query.
// Compared with four waves x four different query groups this
//  * tightens every threshold four times faster (fewer appends: ~k(1+ln(N/k)) per query for the WHOLE
//    range instead of per wave) and needs a quarter of the lists per workgroup, so each list gets four
//    times the slots (LCAP 352 with two workgroups per CU, 704 with one) and is pruned 2-3 times per
//    range instead of 4-6 -- the round-2 ablation showed appends + prunes costing 124 us of 376 us at
//    N = 10 000 x 8 160 (one third of the kernel), ~1 us per pruned list;
//  * gives one workgroup per 32 queries: 255 workgroups at 1080p without splitting the bank at all.
// Appends: a lane that passes reserves a slot with ds_add_rtn (16 independent atomics per tile are issued
// first, the entries written afterwards, so no append waits for its atomic).  Lists are checked between
// two barriers once per tile: wave w prunes lists 8w .. 8w+7 that could overflow during the next tile
// (at most 4 x 32 appends per query per tile).
template <int LCAP, int MINB, int NW>
__global__ __launch_bounds__(NW * 64, MINB) void affinity_topk_wg_kernel(const AffArgs p) {
  if (p.guard && *p.guard == 0u) return;  // uniform over the grid
  constexpr int LSTRIDE = LCAP + 1;
  constexpr int E = (LCAP + 63) / 64;          // list entries per lane in a prune
  constexpr int BURST = NW * TOKT;          // appends per query between two maintenance points
  constexpr int QW = QT / NW;               // lists maintained by one wave
  constexpr bool MS_EARLY = (E <= 6);
  static_assert(LCAP - BURST >= 64, "a list is pruned only when every lane holds an entry");
  static_assert(E * K_MAX <= LCAP - BURST, "one exact prune (<= E*k survivors) must get below the in-loop limit");
  static_assert(2 * K_MAX <= CAP && CAP == 64, "hand-over: one key per lane");
  __shared__ uint32_t s_sc[QT][LSTRIDE];  // candidate scores (fp32 bits)
  __shared__ uint16_t s_tk[QT][LSTRIDE];  // candidate tokens (offset inside this range)
  __shared__ uint32_t s_cnt[QT];
  __shared__ float s_tau[QT];
  __shared__ __attribute__((aligned(16))) float s_ms[NW][TOKT];
  __shared__ __attribute__((aligned(16))) uint32_t s_rank[NW][64];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int l31 = lane & 31;
  const int half = lane >> 5;
  const int q0 = blockIdx.x * QT;
  const int split = blockIdx.y;
  uint32_t* srow = &s_sc[l31][0];
  uint16_t* trow = &s_tk[l31][0];
  float* msl = &s_ms[wave][0];

  // ---- query operand (registers, whole kernel), identical in the four waves
  const int q = min(q0 + l31, p.hw - 1);
  float bqe[CK / 2], bqk[CK / 2];
  float bs[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // bsq in ATen's summation order (see affinity_topk_kernel)
#pragma unroll
  for (int t = 0; t < CK / 2; ++t) {
    const float e0 = p.qe[(int64_t)(2 * t) * p.hw + q], e1 = p.qe[(int64_t)(2 * t + 1) * p.hw + q];
    const float k0 = p.qk[(int64_t)(2 * t) * p.hw + q], k1 = p.qk[(int64_t)(2 * t + 1) * p.hw + q];
    bs[t >> 3] += e0 * (k0 * k0);
    bs[t >> 3] += e1 * (k1 * k1);
    bqe[t] = half ? e1 : e0;
    bqk[t] = half ? (k1 * e1) : (k0 * e0);
  }
  const float bsq = ((bs[0] + bs[1]) + bs[2]) + bs[3];

  // ---- this range's tiles in scrambled order (see affinity_topk_kernel); wave w takes visits w, w+4, ...
  const int n_my = (p.total_tiles - split + p.splits - 1) / p.splits;
  int stride = 61;
  if (n_my % 61 == 0) stride = (n_my % 59 == 0) ? 53 : 59;
  const int n_vis = (n_my > wave) ? (n_my - wave + NW - 1) / NW : 0;  // visits of this wave
  const int n_iter = (n_my + NW - 1) / NW;                             // of the busiest wave
  int cyc = (n_my > 0) ? (int)(((int64_t)wave * stride) % n_my) : 0;
  const int step = (n_my > 0) ? (int)(((int64_t)NW * stride) % n_my) : 0;

  if (threadIdx.x < QT) {
    s_cnt[threadIdx.x] = 0u;
    s_tau[threadIdx.x] = DEVA_ABLATE(1) ? INFINITY : -INFINITY;
  }

  f32x4 xbuf[CK / 4];
  float ms_buf;
  auto prefetch = [&](int cyc_) __attribute__((always_inline)) {
    const int tile = split + p.splits * cyc_;
    const int n_mine = min(tile * TOKT + l31, p.n_total - 1);
    const float* krow = (n_mine < p.n_long) ? (p.key_long + (int64_t)n_mine * CK)
                                            : (p.key_work + (int64_t)(n_mine - p.n_long) * CK);
    ms_buf = ((n_mine < p.n_long) ? p.shr_long[n_mine] : p.shr_work[n_mine - p.n_long]);  // scaled when stored: see DEVA_KEEP_ROWS
    const float* shifted = krow + half;
#pragma unroll
    for (int j = 0; j < CK / 4 - 1; ++j) xbuf[j] = *reinterpret_cast<const f32x4_u*>(shifted + 4 * j);
    xbuf[CK / 4 - 1] = *reinterpret_cast<const f32x4*>(krow + CK - 4);
  };
  if (n_vis > 0) prefetch(cyc);

  // after the rank-counting prune of list qq: exact rounds while it is above `limit`, then publish length / threshold
  auto finish_prune = [&](int qq, int kept, uint32_t thr, uint32_t limit) __attribute__((always_inline)) {
    while ((uint32_t)kept > limit) {
      const uint32_t thr2 = prune_list_exact<E>(&s_sc[qq][0], &s_tk[qq][0], (uint32_t)kept, p.k, lane, &kept);
      thr = thr2 > thr ? thr2 : thr;
    }
    if (lane == 0) {
      s_cnt[qq] = (uint32_t)kept;
      const float t_new = from_orderable(thr);
      if (t_new > s_tau[qq]) s_tau[qq] = t_new;
    }
    DEVA_COMPILER_FENCE();
  };
  auto prune_one = [&](int qq, uint32_t c, uint32_t limit) __attribute__((always_inline)) {
    int kept = (int)c;
    const uint32_t thr = prune_list<E>(&s_sc[qq][0], &s_tk[qq][0], c, p.k, lane, &kept, &s_rank[wave][0]);
    finish_prune(qq, kept, thr, limit);
  };
  // the lists of `todo` this wave takes (every NW-th, starting with the wave-th: any wave can prune any list of the
  // workgroup); lengths from lane qq of c_l.  (Pruning two lists at a time, interleaved to cover each other's LDS round
  // trips, was measured: bit-identical, no faster -- profiles/r02e_affinity_shapes.txt item 11.)
  auto prune_share = [&](uint32_t todo, uint32_t c_l, uint32_t limit) __attribute__((always_inline)) {
    int nth = 0;
    while (todo) {
      const int qq = __ffs((int)todo) - 1;
      todo &= todo - 1;
      if ((nth++ % NW) == wave) prune_one(qq, (uint32_t)__builtin_amdgcn_readlane((int)c_l, qq), limit);
    }
  };

#ifdef DEVA_AFFINITY_PROBES
  uint64_t* pb = (p.probe && blockIdx.x < 8 && blockIdx.y == 0) ? p.probe + (size_t)(blockIdx.x * NW + wave) * 64 * 8 : nullptr;
#endif
  for (int it = 0; it < n_iter; ++it) {
    DEVA_STAMP(0);
    __syncthreads();  // every append of the previous tile has landed: the list lengths are final
    // every wave reads all 32 lengths and thresholds (one LDS access each) and takes the same decision
    const uint32_t c_l = s_cnt[l31];
    float tau = s_tau[l31];
    const uint32_t need = (uint32_t)__builtin_amdgcn_ballot_w64(c_l > (uint32_t)(LCAP - BURST));
    // every wave has read this tile's list lengths (and taken the same pruning decision) before anyone appends
    // again: without this barrier a fast wave's appends could change a slow wave's decision.  It follows the
    // first barrier directly, so the skew of a whole tile (MFMAs, scoring, appends) is collected once per tile.
    DEVA_COMPILER_FENCE();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the reads above
    __builtin_amdgcn_s_barrier();
    DEVA_COMPILER_FENCE();
    DEVA_STAMP(1);
    if (need) {  // uniform over the workgroup: some list could overflow during the next tile
      prune_share(need, c_l, (uint32_t)(LCAP - BURST));
      __syncthreads();  // pruned lists / raised thresholds are visible
      tau = s_tau[l31];
    }
    DEVA_STAMP(2);
    const bool work = it < n_vis;  // the ragged last round: a wave without a tile only keeps the barriers
    const int tile = split + p.splits * cyc;
    const int n_base = tile * TOKT;
    const uint32_t tok0 = (uint32_t)(cyc * TOKT + 4 * half);
    f32x16 accA, accB;
    float4 ms4[4];  // scaled shrinkage in accumulator-row order: rows 4g..4g+3 <-> tokens 8g+4*half..+3
    if (work) {
      // 1/sqrt(CK) folded in (exact); NaN past the end of the bank: such a score fails every threshold test, so
      // the ragged last tile needs no per-row bound check
      if (lane < TOKT) msl[lane] = (n_base + lane < p.n_total) ? ms_buf * 0.125f : __builtin_nanf("");
      DEVA_COMPILER_FENCE();
      // the MFMAs read the prefetched rows in place (channel 2t + half of this lane's token is x / z of the
      // 16-B pieces); the next tile's loads are issued after the last MFMA, under the scoring
      DEVA_KEEP_ROWS(xbuf);
      const float a30 = half ? xbuf[CK / 4 - 1][1] : xbuf[CK / 4 - 1][0];
      const float a31 = half ? xbuf[CK / 4 - 1][3] : xbuf[CK / 4 - 1][2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        accA[r] = 0.0f;
        accB[r] = 0.0f;
      }
      if (!DEVA_ABLATE(8)) {
#ifdef DEVA_AFFINITY_PROBES  // issue-priority experiments (profiles/r02e_affinity_shapes.txt item 10)
        if (DEVA_ABLATE(32)) {
          if (blockIdx.y & 1) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1);
        } else if (DEVA_ABLATE(64)) {
          __builtin_amdgcn_s_setprio(3);
        } else if (DEVA_ABLATE(128)) {
          if (blockIdx.y & 1) __builtin_amdgcn_s_setprio(3);
        }
#endif
#pragma unroll
        for (int t = 0; t < CK / 2; ++t) {
          const float a = (t == CK / 2 - 2) ? a30 : (t == CK / 2 - 1) ? a31 : xbuf[t >> 1][(t & 1) * 2];
          accA = __builtin_amdgcn_mfma_f32_32x32x2f32(a * a, bqe[t], accA, 0, 0, 0);
          accB = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bqk[t], accB, 0, 0, 0);
        }
#ifdef DEVA_AFFINITY_PROBES
        if (p.ablate & (32 | 64 | 128)) __builtin_amdgcn_s_setprio(0);
#endif
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          accA[r] = xbuf[r >> 2][r & 3];
          accB[r] = xbuf[4 + (r >> 2)][r & 3];
        }
      }
      DEVA_COMPILER_FENCE();
      DEVA_STAMP(3);
      if (it + 1 < n_vis) {
        cyc += step;
        cyc = cyc >= n_my ? cyc - n_my : cyc;
        if (!DEVA_ABLATE(2)) prefetch(cyc);
      }
      // this wave's own shrinkage row, read back before the barrier so that its LDS latency is not exposed after
      // it (the 8-wave instantiation has no registers to spare for that and reads it after the barrier)
      if (MS_EARLY) {
#pragma unroll
        for (int g = 0; g < 4; ++g) ms4[g] = *reinterpret_cast<const float4*>(&msl[8 * g + 4 * half]);
      }
    }
    DEVA_COMPILER_FENCE();
    DEVA_STAMP(4);
    DEVA_STAMP(5);
    if (!work) continue;
    if (DEVA_ABLATE(4)) {
      if (accA[0] + accB[15] == 12345.678f) s_cnt[0] = 1u;
      continue;
    }

    // ---- scores of this lane: query l31, tokens n_base + (r&3) + 8*(r>>2) + 4*half.  Phase A: compare
    // and reserve list slots (one LDS atomic per passing lane and row, none waited for); phase B: write.
    if (!MS_EARLY) {
#pragma unroll
      for (int g = 0; g < 4; ++g) ms4[g] = *reinterpret_cast<const float4*>(&msl[8 * g + 4 * half]);
    }
    float v[16];
    uint32_t pos[16];
    unsigned long long okm[16];
    const f32x2 bsq2 = {bsq, bsq};
#pragma unroll
    for (int r2 = 0; r2 < 8; ++r2) {  // two accumulator rows per packed-fp32 instruction
      const f32x2 a2 = {accA[2 * r2], accA[2 * r2 + 1]};
      const f32x2 b2 = {accB[2 * r2], accB[2 * r2 + 1]};
      const float4 m4 = ms4[r2 >> 1];
      const f32x2 m2 = (r2 & 1) ? f32x2{m4.z, m4.w} : f32x2{m4.x, m4.y};
      const f32x2 v2 = (((b2 + b2) - a2) - bsq2) * m2;  // == ((-A + 2B) - bsq) * ms / 8, every step correctly rounded
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = 2 * r2 + h;
        v[r] = v2[h];
        const bool ok = v[r] >= tau;
        okm[r] = __builtin_amdgcn_ballot_w64(ok);
        // lanes that do not pass write to the spare slot LCAP of their row (never read): no predicate in phase B
        pos[r] = (uint32_t)LCAP;
        if (ok) pos[r] = atomicAdd(&s_cnt[l31], 1u);
      }
    }
    DEVA_STAMP(6);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (okm[r]) {
        const int j0 = (r & 3) + 8 * (r >> 2);
        srow[pos[r]] = __float_as_uint(v[r]);
        trow[pos[r]] = (uint16_t)(tok0 + j0);
      }
    }
    DEVA_COMPILER_FENCE();
    DEVA_STAMP(7);
  }

  // ---- hand-over: every list down to at most CAP entries (exact rounds, if needed, shrink 704 -> 352 -> 192 -> 96 ->
  // 64 at worst), then wave w writes lists w*QW .. with their lengths
  __syncthreads();
  {
    const uint32_t c_l = s_cnt[l31];
    const uint32_t over = (uint32_t)__builtin_amdgcn_ballot_w64(c_l > (uint32_t)CAP);
    prune_share(over, c_l, (uint32_t)CAP);
  }
  __syncthreads();  // lists pruned by other waves are visible
  DEVA_COMPILER_FENCE();
  for (int qq = wave * QW; qq < wave * QW + QW; ++qq) {
    if (q0 + qq >= p.hw) break;
    const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_cnt[qq]);
    const int64_t list = (int64_t)split * p.hw + q0 + qq;
    if (lane == 0) p.part_cnt[list] = c;
    if ((uint32_t)lane < c) {
      const uint32_t off = (uint32_t)s_tk[qq][lane];
      const uint32_t token = ((off >> 5) * (uint32_t)p.splits + (uint32_t)split) * TOKT + (off & 31u);
      p.part[list * CAP + lane] = ((uint64_t)orderable(__uint_as_float(s_sc[qq][lane])) << 32) | (uint64_t)(~token);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Ping-pong variant of the workgroup-shared lists: EIGHT waves (two per SIMD) share 32 queries; waves 0-3
// (group A) and 4-7 (group B) alternate between a matrix phase (P: 64 MFMAs of one tile) and a scoring
// phase (Q: scores, appends) separated by workgroup barriers, so that in every slot each SIMD runs the
// MFMAs of one wave beside the scoring of the other:
//      slot 1:  A: P(tile a_i)   B: Q(tile b_i-1)   | barrier |   slot 2:  A: Q(tile a_i)   B: P(tile b_i)   | barrier
// Each group owns its own set of 32 lists (two lists per query are handed over per range), which gives
// every list a quiescent window: a group's lists are pruned at the start of its P phase, when none of its
// waves appends (the other group appends to its own lists), and the result is picked up after the barrier
// that precedes its Q phase.  List lengths are kept in registers (V, identical in the waves of a group);
// the appends of phase Q(j) count into s_delta[group][j & 1], read at the start of P(j+1) and cleared at
// the start of P(j+2).  The filter threshold of a query is the larger of the two groups' bounds.
constexpr int PP_WAVES = 8;
template <int LCAP>
__global__ __launch_bounds__(PP_WAVES * 64, 1) void affinity_topk_pp_kernel(const AffArgs p) {
  if (p.guard && *p.guard == 0u) return;
  constexpr int LSTRIDE = LCAP + 1;
  constexpr int E = (LCAP + 63) / 64;
  constexpr int GW = PP_WAVES / 2;             // waves per group
  constexpr int BURST = GW * TOKT;             // appends per query and group in one Q phase
  constexpr int QW = QT / GW;                  // lists maintained by one wave of a group
  static_assert(LCAP - BURST >= 64, "a list is pruned only when every lane holds an entry");
  static_assert(2 * K_MAX <= CAP && CAP == 64, "hand-over: one key per lane");
  __shared__ uint32_t s_sc[2][QT][LSTRIDE];
  __shared__ uint16_t s_tk[2][QT][LSTRIDE];
  __shared__ uint32_t s_delta[2][2][QT];
  __shared__ uint32_t s_len[2][QT];
  __shared__ float s_tau[2][QT];
  __shared__ __attribute__((aligned(16))) float s_ms[PP_WAVES][TOKT];
  __shared__ __attribute__((aligned(16))) uint32_t s_rank[PP_WAVES][64];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int grp = wave / GW;
  const int gw = wave % GW;
  const int l31 = lane & 31;
  const int half = lane >> 5;
  const int q0 = blockIdx.x * QT;
  const int split = blockIdx.y;          // token range; its two lists are 2*split + grp
  const int ranges = p.splits / 2;
  uint32_t* srow = &s_sc[grp][l31][0];
  uint16_t* trow = &s_tk[grp][l31][0];
  float* msl = &s_ms[wave][0];

  const int q = min(q0 + l31, p.hw - 1);
  float bqe[CK / 2], bqk[CK / 2];
  float bs[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // bsq in ATen's summation order (see affinity_topk_kernel)
#pragma unroll
  for (int t = 0; t < CK / 2; ++t) {
    const float e0 = p.qe[(int64_t)(2 * t) * p.hw + q], e1 = p.qe[(int64_t)(2 * t + 1) * p.hw + q];
    const float k0 = p.qk[(int64_t)(2 * t) * p.hw + q], k1 = p.qk[(int64_t)(2 * t + 1) * p.hw + q];
    bs[t >> 3] += e0 * (k0 * k0);
    bs[t >> 3] += e1 * (k1 * k1);
    bqe[t] = half ? e1 : e0;
    bqk[t] = half ? (k1 * e1) : (k0 * e0);
  }
  const float bsq = ((bs[0] + bs[1]) + bs[2]) + bs[3];

  // tiles of this range in scrambled order; wave w takes visits w, w+8, ...
  const int n_my = (p.total_tiles - split + ranges - 1) / ranges;
  int stride = 61;
  if (n_my % 61 == 0) stride = (n_my % 59 == 0) ? 53 : 59;
  const int n_vis = (n_my > wave) ? (n_my - wave + PP_WAVES - 1) / PP_WAVES : 0;
  const int n_iter = (n_my + PP_WAVES - 1) / PP_WAVES;
  int cyc = (n_my > 0) ? (int)(((int64_t)wave * stride) % n_my) : 0;
  const int step = (n_my > 0) ? (int)(((int64_t)PP_WAVES * stride) % n_my) : 0;

  if (threadIdx.x < 2 * QT) {
    const int g = threadIdx.x / QT, qq = threadIdx.x % QT;
    s_delta[g][0][qq] = 0u;
    s_delta[g][1][qq] = 0u;
    s_tau[g][qq] = DEVA_ABLATE(1) ? INFINITY : -INFINITY;
  }
  uint32_t V = 0u;          // length of this group's list l31 (identical in the four waves of the group)
  uint32_t pruned = 0u;     // lists of this group pruned in the last P phase

  f32x4 xbuf[CK / 4];
  float ms_buf;
  auto prefetch = [&](int cyc_) __attribute__((always_inline)) {
    const int tile = split + ranges * cyc_;
    const int n_mine = min(tile * TOKT + l31, p.n_total - 1);
    const float* krow = (n_mine < p.n_long) ? (p.key_long + (int64_t)n_mine * CK)
                                            : (p.key_work + (int64_t)(n_mine - p.n_long) * CK);
    ms_buf = ((n_mine < p.n_long) ? p.shr_long[n_mine] : p.shr_work[n_mine - p.n_long]);  // scaled when stored: see DEVA_KEEP_ROWS
    const float* shifted = krow + half;
#pragma unroll
    for (int j = 0; j < CK / 4 - 1; ++j) xbuf[j] = *reinterpret_cast<const f32x4_u*>(shifted + 4 * j);
    xbuf[CK / 4 - 1] = *reinterpret_cast<const f32x4*>(krow + CK - 4);
  };
  if (n_vis > 0) prefetch(cyc);

  auto prune_one = [&](int qq, uint32_t c, uint32_t limit) __attribute__((always_inline)) {
    int kept = (int)c;
    uint32_t thr = prune_list<E>(&s_sc[grp][qq][0], &s_tk[grp][qq][0], c, p.k, lane, &kept, &s_rank[wave][0]);
    while ((uint32_t)kept > limit) {
      const uint32_t thr2 = prune_list_exact<E>(&s_sc[grp][qq][0], &s_tk[grp][qq][0], (uint32_t)kept, p.k, lane, &kept);
      thr = thr2 > thr ? thr2 : thr;
    }
    if (lane == 0) {
      s_len[grp][qq] = (uint32_t)kept;
      const float t_new = from_orderable(thr);
      if (t_new > s_tau[grp][qq]) s_tau[grp][qq] = t_new;
    }
    DEVA_COMPILER_FENCE();
  };
  auto prune_mine = [&](uint32_t need, uint32_t limit) __attribute__((always_inline)) {
    uint32_t mine = (need >> (gw * QW)) & ((1u << QW) - 1u);
    while (mine) {
      const int qq = gw * QW + __ffs((int)mine) - 1;
      mine &= mine - 1;
      prune_one(qq, (uint32_t)__builtin_amdgcn_readlane((int)V, qq), limit);
    }
  };

  f32x16 accA, accB;
  int tile_cyc = 0;  // cyclic index of the tile whose scores sit in the accumulators

  // ---- matrix phase of visit j
  auto phase_p = [&](int j) __attribute__((always_inline)) {
    if (j > 0) V += s_delta[grp][(j - 1) & 1][l31];                    // appends of Q(j-1): final
    if (gw == 0 && lane < QT) s_delta[grp][j & 1][lane] = 0u;         // read at P(j-1), next used by Q(j)
    // the phase after the last visit prunes for the hand-over
    const uint32_t limit = (j == n_iter) ? (uint32_t)CAP : (uint32_t)(LCAP - BURST);
    const uint32_t need = (uint32_t)__builtin_amdgcn_ballot_w64(V > limit);
    pruned = need;
    if (need) prune_mine(need, limit);                                 // nobody appends to this group's lists now
    if (j >= n_vis) return;
    if (lane < TOKT) msl[lane] = ms_buf * 0.125f;  // 1/sqrt(CK) folded in (exact)
    tile_cyc = cyc;
    DEVA_KEEP_ROWS(xbuf);
    const float a30 = half ? xbuf[CK / 4 - 1][1] : xbuf[CK / 4 - 1][0];
    const float a31 = half ? xbuf[CK / 4 - 1][3] : xbuf[CK / 4 - 1][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      accA[r] = 0.0f;
      accB[r] = 0.0f;
    }
#pragma unroll
    for (int t = 0; t < CK / 2; ++t) {
      const float a = (t == CK / 2 - 2) ? a30 : (t == CK / 2 - 1) ? a31 : xbuf[t >> 1][(t & 1) * 2];
      accA = __builtin_amdgcn_mfma_f32_32x32x2f32(a * a, bqe[t], accA, 0, 0, 0);
      accB = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bqk[t], accB, 0, 0, 0);
    }
    DEVA_COMPILER_FENCE();
    if (j + 1 < n_vis) {
      cyc += step;
      cyc = cyc >= n_my ? cyc - n_my : cyc;
      prefetch(cyc);
    }
  };
  // ---- scoring phase of visit j
  auto phase_q = [&](int j) __attribute__((always_inline)) {
    if ((pruned >> l31) & 1u) V = s_len[grp][l31];
    pruned = 0u;
    if (j >= n_vis) return;
    const float tau = fmaxf(s_tau[0][l31], s_tau[1][l31]);
    const int tile = split + ranges * tile_cyc;
    const int rows_left = p.n_total - tile * TOKT;
    const uint32_t tok0 = (uint32_t)(tile_cyc * TOKT + 4 * half);
    uint32_t* dcount = &s_delta[grp][j & 1][l31];
    float4 ms4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) ms4[g] = *reinterpret_cast<const float4*>(&msl[8 * g + 4 * half]);
    float v[16];
    uint32_t pos[16];
    unsigned long long okm[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 m4 = ms4[r >> 2];
      const float m = (r & 3) == 0 ? m4.x : (r & 3) == 1 ? m4.y : (r & 3) == 2 ? m4.z : m4.w;
      const float b = accB[r];
      v[r] = (((b + b) - accA[r]) - bsq) * m;  // == ((-A + 2B) - bsq) * ms / 8, every step correctly rounded
      const int j0 = (r & 3) + 8 * (r >> 2);
      const bool ok = (v[r] >= tau) && (j0 + 4 * half < rows_left);
      okm[r] = __builtin_amdgcn_ballot_w64(ok);
      pos[r] = 0u;
      if (ok) pos[r] = V + atomicAdd(dcount, 1u);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (okm[r]) {
        const int j0 = (r & 3) + 8 * (r >> 2);
        if ((okm[r] >> lane) & 1ull) {
          srow[pos[r]] = __float_as_uint(v[r]);
          trow[pos[r]] = (uint16_t)(tok0 + j0);
        }
      }
    }
    DEVA_COMPILER_FENCE();
  };

  __syncthreads();  // list state initialised
  // group B runs one slot behind group A; phase 2j is P(j), phase 2j+1 is Q(j), phase 2*n_iter the hand-over prune
  for (int s = 0; s <= 2 * n_iter + 1; ++s) {
    const int ph = s - grp;
    if (ph >= 0 && ph <= 2 * n_iter) {
      if (ph & 1) {
        phase_q(ph >> 1);
      } else {
        phase_p(ph >> 1);
      }
    }
    // LDS traffic of this slot retired, but NOT the key rows just requested (__syncthreads waits for those too)
    DEVA_COMPILER_FENCE();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    DEVA_COMPILER_FENCE();
  }

  // ---- hand-over: every wave writes the lists it maintains (at most CAP entries each after the last prune)
  if ((pruned >> l31) & 1u) V = s_len[grp][l31];
  DEVA_COMPILER_FENCE();
  for (int qq = gw * QW; qq < gw * QW + QW; ++qq) {
    if (q0 + qq >= p.hw) break;
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)V, qq);
    const int64_t list = ((int64_t)(2 * split + grp)) * p.hw + q0 + qq;
    if (lane == 0) p.part_cnt[list] = c;
    if ((uint32_t)lane < c) {
      const uint32_t off = (uint32_t)s_tk[grp][qq][lane];
      const uint32_t token = ((off >> 5) * (uint32_t)ranges + (uint32_t)split) * TOKT + (off & 31u);
      p.part[list * CAP + lane] = ((uint64_t)orderable(__uint_as_float(s_sc[grp][qq][lane])) << 32) | (uint64_t)(~token);
    }
  }
}

// one wave per query: exact top-k over the candidate lists of all ranges (lane l holds entry l of every
// range's list: ME >= splits keys per lane), sorted by rank counting, then exp / normalise / usage.
// With out_keys != NULL the sorted top-k is instead written back in the hand-over format (token index
// shifted by token_offset): the per-shard selection of a token-sharded bank, merged by a second pass of
// this kernel over the gathered lists of all shards.
template <int ME>
__global__ __launch_bounds__(256) void affinity_finalize_kernel(const uint64_t* __restrict__ part,
                                                                const uint32_t* __restrict__ part_cnt, int hw, int k,
                                                                int splits, int32_t* __restrict__ idx,
                                                                float* __restrict__ weight,
                                                                unsigned long long* __restrict__ usage_fix,
                                                                uint64_t* __restrict__ out_keys,
                                                                uint32_t* __restrict__ out_cnt, uint32_t token_offset,
                                                                const uint32_t* __restrict__ guard) {
  if (guard && *guard == 0u) return;  // fall-back of the fp16 pre-filter: nothing to do
  __shared__ uint64_t s_buf[4][2][64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + wave;
  if (q >= hw) return;
  volatile uint64_t* unsorted = &s_buf[wave][0][0];
  volatile uint64_t* sorted = &s_buf[wave][1][0];

  const int n_live = splits;
  uint64_t e[ME];
#pragma unroll
  for (int i = 0; i < ME; ++i) {
    uint64_t v = 0ull;
    if (i < splits) {  // key and length loads are independent (one memory round trip): slots past the length are
      const int64_t list = (int64_t)i * hw + q;  // allocated workspace, read and discarded
      const uint64_t key = part[list * CAP + lane];
      v = ((uint32_t)lane < part_cnt[list]) ? key : 0ull;
    }
    e[i] = v;
  }
  const uint64_t thr = kth_largest<ME>(e, n_live, k);
  // compact the k survivors into LDS (any order), then sort them by rank counting
  int base = 0;
#pragma unroll
  for (int i = 0; i < ME; ++i) {
    if (i < n_live) {
      const bool keep = e[i] >= thr && e[i] != 0ull;
      const unsigned long long b = __builtin_amdgcn_ballot_w64(keep);
      if (keep) unsorted[base + prefix_below(b)] = e[i];
      base += __popcll(b);
    }
  }
  DEVA_COMPILER_FENCE();
  // (fewer than k survivors only if scores are NaN -- a NaN fails every comparison of the selection; the reference's
  // topk propagates NaN there: the missing slots get weight NaN / token 0 instead of uninitialised LDS contents)
  const bool live = lane < k;
  // placeholder of a missing slot: unique, below every real key, score bits of a NaN, token = lane (in range)
  const uint64_t cand = (live && lane < base) ? unsorted[lane] : (uint64_t)(0xffffffffu - (uint32_t)lane);
  int rank = 0;
  for (int j = 0; j < k; ++j) {  // lane j's key, broadcast through SGPRs (j is wave-uniform)
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cand, j);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cand >> 32), j);
    rank += ((((uint64_t)hi << 32) | lo) > cand) ? 1 : 0;
  }
  DEVA_COMPILER_FENCE();
  if (live) sorted[rank] = cand;
  DEVA_COMPILER_FENCE();
  const uint64_t mine = live ? sorted[lane] : 0ull;  // lane r holds the r-th best

  if (out_keys) {
    if (live) out_keys[(int64_t)q * CAP + lane] = (mine & 0xffffffff00000000ull) | (uint64_t)(~(~(uint32_t)mine + token_offset));
    if (lane == 0) out_cnt[q] = (uint32_t)k;
    return;
  }
  const float score = from_orderable((uint32_t)(mine >> 32));
  const uint32_t token = ~(uint32_t)mine;
  const float ex = live ? expf(score) : 0.0f;
  float sum = 0.0f;
  for (int r = 0; r < k; ++r)  // sequential, like torch.sum over the sorted top-k
    sum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ex), r));
  const float w = ex / sum;
  if (live) {
    idx[(int64_t)q * k + lane] = (int32_t)token;
    weight[(int64_t)q * k + lane] = w;
    if (usage_fix && w == w) {
      atomicAdd(&usage_fix[token], (unsigned long long)(w * 1099511627776.0f));  // w * 2^40, exact scaling
    }
  }
}

__global__ void usage_update_kernel(unsigned long long* __restrict__ usage_fix, int64_t offset,
                                    float* __restrict__ use, float* __restrict__ life, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long f = usage_fix[offset + i];
  usage_fix[offset + i] = 0ull;
  if (use) use[i] += (float)((double)f * (1.0 / 1099511627776.0));
  if (life) life[i] += 1.0f;
}

// ------------------------------------------------------------------ sparse readout
// block = 256 threads = 4 waves; tile = 8 queries x 256 channels.  A wave gathers the k value rows
// of 2 queries (each lane a float4 of the 1-KiB row slab), accumulates in registers, and the tile
// is transposed through LDS so each [cv][hw] output row is written as one 32-B segment.
constexpr int RQ = 8;     // queries per block (small tiles: a 480p frame still yields ~400 workgroups)
constexpr int RC = 256;   // channels per block

__global__ __launch_bounds__(256) void readout_sparse_kernel(const int32_t* __restrict__ idx,
                                                             const float* __restrict__ weight, int hw, int k,
                                                             const float* __restrict__ val_long, int n_long,
                                                             const float* __restrict__ val_work, int cv,
                                                             float* __restrict__ out, int tok_lo, int tok_hi,
                                                             const int32_t* __restrict__ map_long,
                                                             const int32_t* __restrict__ map_work) {
  __shared__ float tile[RC][RQ + 1];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int q0 = blockIdx.x * RQ;
  const int c0 = blockIdx.y * RC;
  const int cl = lane * 4;  // channel offset inside the slab
  const bool c_ok = (c0 + cl) < cv;  // cv is a multiple of 4
  for (int qi = 0; qi < RQ / 4; ++qi) {
    const int ql = wave * (RQ / 4) + qi;
    const int q = q0 + ql;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < hw) {  // (wave-uniform)
      // lane j resolves term j of the query once: arena row and weight (0 for a term this launch does not add:
      // token of another bank shard, row stored on another rank); then the rows are fetched eight at a time --
      // independent 16-byte loads in flight instead of an index -> row dependency chain per term
      int row = 0, is_long = 0;
      float wj = 0.0f;
      bool live = false;
      if (lane < k) {
        const int t = idx[(int64_t)q * k + lane];
        wj = weight[(int64_t)q * k + lane];
        is_long = (t < n_long) ? 1 : 0;
        row = is_long ? t : t - n_long;
        live = t >= tok_lo && t < tok_hi;
        const int32_t* map = is_long ? map_long : map_work;
        if (live && map) {
          row = map[row];
          live = row >= 0;
        }
        if (!live) {
          row = 0;
          wj = 0.0f;
          is_long = n_long > 0 ? is_long : 0;
        }
      }
      // a term this launch does not add is SKIPPED, not added with weight 0: its stand-in row (row 0 of an arena that may be
      // uninitialised on a rank without local rows) never enters the sum, so 0 * NaN cannot either
      const uint64_t live_bits = __builtin_amdgcn_ballot_w64(live);
      const float* col = nullptr;
      for (int j0 = 0; j0 < k; j0 += 8) {
        float4 v[8];
        float w8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = min(j0 + u, k - 1);
          const int r = __builtin_amdgcn_readlane(row, j);
          const int lg = __builtin_amdgcn_readlane(is_long, j);
          w8[u] = (j0 + u < k) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wj), j)) : 0.0f;
          col = (lg ? val_long : val_work) + (int64_t)r * cv + c0 + cl;
          v[u] = c_ok ? *reinterpret_cast<const float4*>(col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {  // terms in index order, like the one-at-a-time loop: same sums
          if ((live_bits >> (j0 + u)) & 1ull) {
            acc.x += w8[u] * v[u].x;
            acc.y += w8[u] * v[u].y;
            acc.z += w8[u] * v[u].z;
            acc.w += w8[u] * v[u].w;
          }
        }
      }
    }
    tile[cl + 0][ql] = acc.x;
    tile[cl + 1][ql] = acc.y;
    tile[cl + 2][ql] = acc.z;
    tile[cl + 3][ql] = acc.w;
  }
  __syncthreads();
  const int tq = threadIdx.x % RQ;
  const int tc = threadIdx.x / RQ;  // 0 .. 256/RQ - 1
  if (q0 + tq < hw) {
    for (int c = tc; c < RC; c += 256 / RQ) {
      if (c0 + c < cv) out[(int64_t)(c0 + c) * hw + q0 + tq] = tile[c][tq];
    }
  }
}

// ===================================================================================================
// fp16 pre-filter with exact fp32 re-scoring (VERDICT r2 "next" 3b).
//
// The fused fp32 kernels above are bound by the fp32 matrix rate (1/16 of the f16 rate on gfx950) plus
// the VALU / LDS work of keeping exact candidate lists.  This path scores every (token, query) pair with
// v_mfma_f32_32x32x16_f16 on fp16 copies of the operands, carries a RIGOROUS error bound per score, and
// re-scores only the ~k+5 tokens per query that the bound cannot exclude with the natural-order fp32 FMA
// chain (bit-identical to v_mfma_f32_32x32x2_f32, MI355X_MICROARCH.md) -- selections, weights and usage
// counters are bit-identical to the fp32 kernels by construction, not by tolerance.
//
//   sim(n, q) = -m_n * (A - 2B + bsq),  m_n = ms_n / 8,  A = sum mk^2 qe,  B = sum mk qk qe,  bsq = sum qk^2 qe
//   P = m (A + bsq) >= 0 (needs qe >= 0: checked, else fall-back), Q = 2 m B, sim = Q - P.
//   Cauchy-Schwarz + AM-GM: sum_c |2 m mk qk qe| <= 2 m sqrt(A bsq) <= m (A + bsq) = P, so with d the relative
//   error of a product of two fp16-rounded operands (2^-10, + fp32 accumulation; d = 1.3e-3 is used) both
//   chains are off by at most d * P_true (+ an absolute term for operands below the fp16 normal range):
//       lo = Q~ - (1 + d2) P~ - ABS  <=  sim_fp32 * S  <=  Q~ - (1 - d2) P~ + ABS = hi,   d2 = 2 d / (1 - d)
//   (S = power-of-two scale of the query, see pf_query_operand; the d margin of 33 % over 2^-10 absorbs the fp32
//   round-off of the reference chain itself, ~70 * 2^-24 relative to P).
//
// Centering.  On real clips the keys share a large common component (the best-matching tokens have mk ~ qk), so
// A, B and bsq are ~200x the score they cancel to and a 2^-10 relative bound on P would let thousands of tokens
// through (measured on the 1080p clip: 480 - 6 200 candidates per query).  The distance is invariant under a
// common shift, sum qe (mk - qk)^2 = sum qe ((mk - mu) - (qk - mu))^2, so both sides are centred on the bank's mean
// key mu before they are rounded to fp16 (fp32 subtractions: relative error 2^-24 of the CENTRED value): P shrinks
// 35x on that clip and the same bound admits ~37 candidates.  The fp32 round-off of the REFERENCE chain is relative
// to its uncentred terms, P_unc <= 2 P + 4 m sum qe mu^2: the first part is inside d2, the second is the per-query
// constant E_q = 4e-5 * max m * sum_c qe_c mu_c^2 (8e-6 relative round-off of the 64-step fp32 chains x 4, + margin)
// subtracted from the filter threshold twice like ABS.
//
// Kernels (no LDS lists, no prune rounds, no workgroup barriers):
//   pf_mean     per-block partial channel sums of the bank -> mu;
//   pf_stats    max of mk^2 m, 2|mk| m, m over the bank -> power-of-two scales that put the largest operand
//               just under 2^15 (fp16 max 65 504); non-finite input -> fall-back flag;
//   pf_prep     the bank as fp16 MFMA A-operands, tile-major [tile][9 K-blocks][64 lanes][8 halfs]: every
//               operand load of the passes is one fully coalesced 1-KiB instruction;
//   pass A      group maxima of lo: per (token range, query) PF_GROUPS = 32 groups (one per token slot of the
//               tiles) of DISTINCT tokens, so the k-th largest group maximum over all ranges is a valid
//               lower bound of the k-th best score -- with 512 groups at 1080p it sits within a few per
//               cent of the true k-th best (expected ~k+2 tokens above it); 2 VALU per score;
//   pf_tau      one wave per query: k-th largest of its group maxima -> filter threshold;
//   pass B      hi >= threshold -> candidate (token, hi) into the private sub-list of the (range, query,
//               half-lane): plain predicated stores, no atomics; 2 VALU per score;
//   pf_rescore  one wave per query: gathers the candidates of all ranges (~35), re-scores each with the
//               fp32 FMA chain from its key row, selects the exact top-k on (score, index) keys and
//               finishes like affinity_finalize_kernel (softmax, usage, or the hand-over format).
// Anything the bound does not cover (negative selection, non-finite or out-of-range operands, a sub-list
// or the re-score buffer overflowing -- flat "near-tie" banks) raises a device flag and the fp32 kernels
// run as the fall-back in the same stream (they return at once when the flag is clear).
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int PF_KB = 9;                      // K-blocks of 16 halfs per token: 4 (P: mk^2 m) + 1 (P: m x bsq) + 4 (Q)
constexpr int PF_TILE_BYTES = PF_KB * 64 * 16;  // [kb][lane][8 halfs]
constexpr int PF_QW = 4;                      // waves (32 queries each) per workgroup, all on the same token range
constexpr int PF_SUB = 64;                    // candidate slots per (range, query, half-lane)
constexpr int PF_MAX_SPLITS = 32;
constexpr int PF_GROUPS = 32;                 // group maxima per (range, query): one per token slot of the tiles
constexpr int PF_RESC_MAX = 2048;             // candidates re-scored per query (up to 32 rounds of 64)
constexpr float PF_D2 = 2.63e-3f;             // 2 d / (1 - d), d = 1.3e-3, + 2e-5 for the reference's fp32 round-off on 2 P
constexpr float PF_ABS = 600.0f;              // operands below 2^-14 (flushed or subnormal): 2 chains x 2^-14 x 2 x 65 x 2^15

struct PfState {     // device block, written by the prep kernel of every read
  uint32_t flag;     // != 0: the fp32 kernels take over
  uint32_t max_p;    // float bits: max (mk - mu)^2 m
  uint32_t max_q;    // max 2 |mk - mu| m
  uint32_t max_m;    // max m
  float mu[CK];      // mean key of the bank (the common shift of both operand sides)
  uint32_t bank_flag;  // the flag as the bank alone sets it (non-finite key / shrinkage): what a read on cached operands starts from
};
static_assert(sizeof(PfState) <= 512, "the state block is 512 bytes of the scratch / of a prepared-bank buffer");
constexpr float PF_EQ = 4e-5f;  // reference fp32 round-off carried by the shift: E_q = PF_EQ * max m * sum qe mu^2

// power of two P with x * P in [2^14, 2^15)
__device__ __forceinline__ float pf_scale(float x) {
  if (!(x > 0.0f) || !(x < INFINITY)) return 1.0f;
  int e;
  (void)frexpf(x, &e);  // x = f * 2^e, f in [0.5, 1)
  e = 15 - e;
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return ldexpf(1.0f, e);
}

__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

struct PfBank {
  const float* key_long;
  const float* shr_long;
  int n_long;
  const float* key_work;
  const float* shr_work;
  int n_total;
};

__device__ __forceinline__ const float* pf_row(const PfBank& b, int n, float* ms) {
  if (n < b.n_long) {
    *ms = b.shr_long[n];
    return b.key_long + (int64_t)n * CK;
  }
  *ms = b.shr_work[n - b.n_long];
  return b.key_work + (int64_t)(n - b.n_long) * CK;
}

constexpr int PF_STAT_BLOCKS = 256;

// grid-stride over tokens: per-block partial channel sums of the keys -> sums[block][64] (thread = channel x 4 token lanes)
__global__ __launch_bounds__(256) void affinity_pf_mean_kernel(const PfBank b, float* __restrict__ sums) {
  __shared__ float s_part[4][CK];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  float acc = 0.0f;
  for (int n = blockIdx.x * 4 + g; n < b.n_total; n += PF_STAT_BLOCKS * 4) {
    float ms;
    acc += pf_row(b, n, &ms)[c];
  }
  s_part[g][c] = acc;
  __syncthreads();
  if (g == 0) sums[blockIdx.x * CK + c] = (s_part[0][c] + s_part[1][c]) + (s_part[2][c] + s_part[3][c]);
}

// grid-stride over (token, 16 channels); every block first forms mu from the partial sums, then leaves its partial
// maxima of the CENTRED operands (and a bad-input mark) in part[block][4] -- no atomics, nothing to zero beforehand
__global__ __launch_bounds__(256) void affinity_pf_stats_kernel(const PfBank b, const float* __restrict__ sums,
                                                                 uint32_t* __restrict__ part, float* __restrict__ mu_out) {
  __shared__ float s_red[4][4];
  __shared__ float s_mu[CK];
  if (threadIdx.x < CK) {
    float t = 0.0f;
    for (int i = 0; i < PF_STAT_BLOCKS; ++i) t += sums[i * CK + threadIdx.x];
    t = t / (float)b.n_total;
    t = (t == t && fabsf(t) < INFINITY) ? t : 0.0f;  // (non-finite banks fall back anyway)
    s_mu[threadIdx.x] = t;
    if (blockIdx.x == 0) mu_out[threadIdx.x] = t;
  }
  __syncthreads();
  float mp = 0.0f, mq = 0.0f, mm = 0.0f;
  bool bad = false;
  const int total = b.n_total * 4;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += PF_STAT_BLOCKS * 256) {
    const int n = i >> 2;
    float ms;
    const float* row = pf_row(b, n, &ms) + 16 * (i & 3);
    const float m = ms * 0.125f;
    mm = fmaxf(mm, m);
    bad = bad || !(m >= 0.0f && m < INFINITY);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * j);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        bad = bad || !(fabsf(v[u]) < INFINITY);
        const float a = fabsf(v[u] - s_mu[16 * (i & 3) + 4 * j + u]);
        mp = fmaxf(mp, a * a * m);
        mq = fmaxf(mq, 2.0f * a * m);
      }
    }
  }
  mp = wave_max_f(mp);
  mq = wave_max_f(mq);
  mm = wave_max_f(mm);
  bad = bad || !(mp < INFINITY) || !(mq < INFINITY);
  const float fb = __builtin_amdgcn_ballot_w64(bad) ? 1.0f : 0.0f;
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_red[wave][0] = mp;
    s_red[wave][1] = mq;
    s_red[wave][2] = mm;
    s_red[wave][3] = fb;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const float v = fmaxf(fmaxf(s_red[0][threadIdx.x], s_red[1][threadIdx.x]), fmaxf(s_red[2][threadIdx.x], s_red[3][threadIdx.x]));
    part[blockIdx.x * 4 + threadIdx.x] = __float_as_uint(v);
  }
}

// one thread per (token slot of the padded bank, half-lane): writes the 9 x 16 B this MFMA lane will load
__global__ __launch_bounds__(256) void affinity_pf_prep_kernel(const PfBank b, const uint32_t* __restrict__ part,
                                                                PfState* st, PfState* keep, int n_pad, uint8_t* __restrict__ a16) {
  // every block reduces the partial maxima of the stats kernel itself (one wave); block 0 publishes
  // them together with the cleared fall-back flag for the kernels that follow in the stream
  __shared__ float s_max[4];
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float m = 0.0f;
      for (int i = lane; i < PF_STAT_BLOCKS; i += 64) m = fmaxf(m, __uint_as_float(part[i * 4 + c]));
      v[c] = wave_max_f(m);
    }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) s_max[c] = v[c];
      if (blockIdx.x == 0) {
        st->flag = st->bank_flag = v[3] > 0.0f ? 1u : 0u;
        st->max_p = __float_as_uint(v[0]);
        st->max_q = __float_as_uint(v[1]);
        st->max_m = __float_as_uint(v[2]);
      }
    }
    // a prepared-bank buffer keeps the state beside the operands: reads of the unchanged bank start from this image
    if (blockIdx.x == 0 && keep) {
      keep->mu[lane] = st->mu[lane];
      if (lane == 0) {
        keep->flag = keep->bank_flag = v[3] > 0.0f ? 1u : 0u;
        keep->max_p = __float_as_uint(v[0]);
        keep->max_q = __float_as_uint(v[1]);
        keep->max_m = __float_as_uint(v[2]);
      }
    }
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = i >> 1, half = i & 1;
  if (n >= n_pad) return;
  const float sp = pf_scale(s_max[0]);
  const float sq = pf_scale(s_max[1]);
  const float sm = pf_scale(s_max[2]);
  uint8_t* dst = a16 + (int64_t)(n >> 5) * PF_TILE_BYTES + ((n & 31) + 32 * half) * 16;
  h8 out[PF_KB];
#pragma unroll
  for (int kb = 0; kb < PF_KB; ++kb)
#pragma unroll
    for (int e = 0; e < 8; ++e) out[kb][e] = (_Float16)0.0f;
  if (n < b.n_total) {
    float ms;
    const float* row = pf_row(b, n, &ms);
    const float m = ms * 0.125f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(row + 16 * kb + 8 * half);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(row + 16 * kb + 8 * half + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = (e < 4 ? v0[e] : v1[e - 4]) - st->mu[16 * kb + 8 * half + e];  // centred on the bank's mean key
        out[kb][e] = (_Float16)(a * a * m * sp);
        out[5 + kb][e] = (_Float16)(2.0f * a * m * sq);
      }
    }
    if (half == 0) out[4][0] = (_Float16)(m * sm);
  }
#pragma unroll
  for (int kb = 0; kb < PF_KB; ++kb) *reinterpret_cast<h8*>(dst + kb * 1024) = out[kb];
}

struct PfArgs {
  const uint8_t* bq16;  // [ceil(hw/32)][9][64][8 halfs] query operands (affinity_pf_query_kernel)
  const uint8_t* a16;
  int n_total;
  int total_tiles;
  const float* qk;
  const float* qe;
  int hw;
  int splits;
  PfState* st;
  float* gmax;         // [splits][hw][32] group maxima of lo (pass A)
  const float* thr;    // [hw] filter threshold (pass B)
  uint64_t* cand;      // [hw][splits][2][PF_SUB]: hi bits << 32 | token
  uint32_t* cand_cnt;  // [hw][splits][2]
};

// The query side of the MFMAs for lane (l31, half): channel 16 kb + 8 half + e of query q, scaled by the
// per-query powers of two that bring P~ and Q~ to the common scale S_q = min over the three operand groups
// of (bank scale x largest query scale that keeps the group below 2^15).  Raises the fall-back flag for a
// negative or non-finite selection / key.
__device__ __forceinline__ void pf_query_operand(const float* __restrict__ qk, const float* __restrict__ qe, int hw, int q,
                                                 int half, const PfState* st, h8 (&bq)[PF_KB], bool* bad_out,
                                                 float* eq_out) {
  float e_[32], p_[32];
  float bsq = 0.0f, e_max = 0.0f, p_max = 0.0f, mq2 = 0.0f;
  bool bad = false;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = 16 * kb + 8 * half + e;
      const float ev = qe[(int64_t)c * hw + q], kraw = qk[(int64_t)c * hw + q];
      const float mu_c = st->mu[c];
      const float kv = kraw - mu_c;  // centred like the bank
      mq2 += ev * (mu_c * mu_c);
      bad = bad || !(ev >= 0.0f && ev < INFINITY) || !(fabsf(kraw) < INFINITY);
      e_[8 * kb + e] = ev;
      p_[8 * kb + e] = kv * ev;
      bsq += ev * (kv * kv);
      e_max = fmaxf(e_max, ev);
      p_max = fmaxf(p_max, fabsf(kv * ev));
    }
  // both half-lanes of a query end up with the same values
  const float bsq_o = __shfl_xor(bsq, 32, 64), e_o = __shfl_xor(e_max, 32, 64), p_o = __shfl_xor(p_max, 32, 64);
  mq2 += __shfl_xor(mq2, 32, 64);
  bsq = half ? (bsq_o + bsq) : (bsq + bsq_o);
  e_max = fmaxf(e_max, e_o);
  p_max = fmaxf(p_max, p_o);
  bad = bad || !(bsq < INFINITY) || !(p_max < INFINITY);
  const float sp = pf_scale(__uint_as_float(st->max_p));
  const float sq = pf_scale(__uint_as_float(st->max_q));
  const float sm = pf_scale(__uint_as_float(st->max_m));
  const float S = fminf(fminf(sp * pf_scale(e_max), sq * pf_scale(p_max)), sm * pf_scale(bsq));
  const float te = S / sp, tp = S / sq, tb = S / sm;  // powers of two: exact
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bq[kb][e] = (_Float16)(e_[8 * kb + e] * te);
      bq[5 + kb][e] = (_Float16)(p_[8 * kb + e] * tp);
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) bq[4][e] = (_Float16)0.0f;
  if (half == 0) bq[4][0] = (_Float16)(bsq * tb);
  *bad_out = bad || !(mq2 < INFINITY);
  *eq_out = PF_EQ * __uint_as_float(st->max_m) * mq2 * S;  // in the query's scaled units, like ABS
}

// one wave per 32 queries: the fp16 query operands of both passes, computed ONCE per read (every (range, pass) wave
// used to rebuild them: 64 strided loads and the scale logic per query group, which at ~20 tiles per wave cost as
// much as the tiles).  Layout like the bank operand: [query group][9 K-blocks][64 lanes][8 halfs].
// `src`: where the bank's state is read from -- the scratch's own block (`st`, just written by the prep kernel) or, on a read
// of cached bank operands, the image in the prepared-bank buffer, which workgroup 0 then copies to the scratch for the
// kernels that follow (round 5 had a one-wave launch of its own for that copy).  A bad query does not touch st->flag here
// (workgroup 0 may be writing it): every query group leaves its mark in qbad[group], which the check kernel -- the next
// writer of the flag, ahead of all its readers -- folds in.
__global__ __launch_bounds__(256) void affinity_pf_query_kernel(const float* __restrict__ qk, const float* __restrict__ qe,
                                                                 int hw, const PfState* __restrict__ src, PfState* st,
                                                                 uint8_t* __restrict__ bq16, float* __restrict__ eq,
                                                                 uint32_t* __restrict__ qbad) {
  const int lane = threadIdx.x & 63;
  const int group = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (src != st && blockIdx.x == 0 && threadIdx.x < 64) {
    st->mu[lane] = src->mu[lane];
    if (lane == 0) {
      st->flag = st->bank_flag = src->bank_flag;
      st->max_p = src->max_p;
      st->max_q = src->max_q;
      st->max_m = src->max_m;
    }
  }
  const int q0 = group * QT;
  if (q0 >= hw) return;
  const int l31 = lane & 31, half = lane >> 5;
  h8 bq[PF_KB];
  bool bad;
  float e_q;
  pf_query_operand(qk, qe, hw, min(q0 + l31, hw - 1), half, src, bq, &bad, &e_q);
  const bool any_bad = __builtin_amdgcn_ballot_w64(bad && q0 + l31 < hw) != 0;
  if (lane == 0) qbad[group] = any_bad ? 1u : 0u;
  if (half == 0 && q0 + l31 < hw) eq[q0 + l31] = e_q;
  uint8_t* dst = bq16 + (int64_t)group * PF_TILE_BYTES + lane * 16;
#pragma unroll
  for (int kb = 0; kb < PF_KB; ++kb) *reinterpret_cast<h8*>(dst + kb * 1024) = bq[kb];
}

// PASS 0: group maxima of lo; PASS 1: candidates with hi >= threshold.
// QG query groups (of 32) per wave share every token tile the wave loads: the vector-memory path delivers 64 B per
// clock and CU, a 32x32x16 MFMA (32 clocks) consumes a 1-KiB A fragment, so with one query group per wave the
// passes are bound by operand delivery at ~3x the matrix time (measured: 840 clocks per tile against 288 of
// MFMAs); two groups halve the operand bytes per MFMA.  Small frames keep QG = 1 (more workgroups).
template <int PASS, int QG>
__global__ __launch_bounds__(PF_QW * 64, 2) void affinity_pf_pass_kernel(const PfArgs p) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int l31 = lane & 31;
  const int half = lane >> 5;
  const int q0 = (blockIdx.x * PF_QW + wave) * (QT * QG);
  if (q0 >= p.hw) return;
  const int split = blockIdx.y;
  h8 bq[QG][PF_KB];
  bool q_ok[QG];
  int q[QG];
  float thr[QG];
  uint32_t ccnt[QG];
  uint64_t* my_list[QG];
  float g[QG][16];
#pragma unroll
  for (int u = 0; u < QG; ++u) {
    q_ok[u] = q0 + QT * u + l31 < p.hw;
    q[u] = min(q0 + QT * u + l31, p.hw - 1);
    // (a query group past the end of the frame re-reads the last one: its results are discarded)
    const int group = min(q0 / QT + u, (p.hw - 1) / QT);
    const uint8_t* src = p.bq16 + (int64_t)group * PF_TILE_BYTES + lane * 16;
#pragma unroll
    for (int kb = 0; kb < PF_KB; ++kb) bq[u][kb] = *reinterpret_cast<const h8*>(src + kb * 1024);
    // a lane without a query (ragged last group) must never file a candidate into the clamped query's sub-list
    thr[u] = PASS ? (q_ok[u] ? p.thr[q[u]] : INFINITY) : 0.0f;
    ccnt[u] = 0u;
    my_list[u] = p.cand + (((int64_t)q[u] * p.splits + split) * 2 + half) * PF_SUB;
#pragma unroll
    for (int r = 0; r < 16; ++r) g[u][r] = -INFINITY;
  }

  // token ranges are TILE-CYCLIC (range s owns tiles s, s + S, ...): a video memory holds the same location once per
  // memory frame, hw tokens apart -- with contiguous ranges those near-duplicates (the best matches of a query) share
  // one (range, slot) group, whose single maximum then says little about the k-th best, and one half-lane sub-list
  const int n_my = (p.total_tiles - split + p.splits - 1) / p.splits;
  const int t0 = 0, t1 = n_my;  // visit index; tile = split + visit * splits
  const uint8_t* mine = p.a16 + lane * 16;
  auto load = [&](h8 (&x)[PF_KB], int visit) __attribute__((always_inline)) {
    const uint8_t* base = mine + (int64_t)(split + visit * p.splits) * PF_TILE_BYTES;
#pragma unroll
    for (int kb = 0; kb < PF_KB; ++kb) x[kb] = *reinterpret_cast<const h8*>(base + kb * 1024);
  };
  const float c_lo = -(1.0f + PF_D2), c_hi = -(1.0f - PF_D2);

  // full: every token slot of the tile is inside the bank (all tiles but possibly the last one of the bank)
  auto process = [&](const h8 (&x)[PF_KB], int visit, auto full) __attribute__((always_inline)) {
    constexpr bool full_tile = decltype(full)::value;
    const int tile = split + visit * p.splits;
    const int rows_left = p.n_total - tile * TOKT;
    const uint32_t tok0 = (uint32_t)(tile * TOKT + 4 * half);
#pragma unroll
    for (int u = 0; u < QG; ++u) {
      f32x16 accP, accQ;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        accP[r] = 0.0f;
        accQ[r] = 0.0f;
      }
#pragma unroll
      for (int kb = 0; kb < 5; ++kb) accP = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[kb], bq[u][kb], accP, 0, 0, 0);
#pragma unroll
      for (int kb = 5; kb < PF_KB; ++kb) accQ = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[kb], bq[u][kb], accQ, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j0 = (r & 3) + 8 * (r >> 2);  // + 4 * half: token slot of accumulator row r
        if (PASS == 0) {
          float lo = __builtin_fmaf(accP[r], c_lo, accQ[r]);
          if (!full_tile && j0 + 4 * half >= rows_left) lo = -INFINITY;
          // plain v_max_f32 (fmaxf would first canonicalise both inputs: one more VALU instruction per score)
          asm("v_max_f32 %0, %1, %2" : "=v"(g[u][r]) : "v"(g[u][r]), "v"(lo));
        } else {
          const float hi = __builtin_fmaf(accP[r], c_hi, accQ[r]);
          bool ok = hi >= thr[u];
          if (!full_tile) ok = ok && (j0 + 4 * half < rows_left);
          if (ok) {
            if (ccnt[u] < (uint32_t)PF_SUB)
              my_list[u][ccnt[u]] = ((uint64_t)__float_as_uint(hi) << 32) | (uint64_t)(tok0 + j0);
            ccnt[u] += 1u;
          }
        }
      }
      // one accumulator set: the next group's MFMAs must not be hoisted above this group's scoring (two sets in
      // flight push pass A over the 256-register budget of two waves per SIMD -> scratch spills)
      if (PASS == 0 && QG > 1) __builtin_amdgcn_sched_barrier(0);
    }
  };

  // the bank's ragged last tile (if this range holds it) is peeled off the loop
  const bool ragged = (n_my > 0) && (split + (n_my - 1) * p.splits == p.total_tiles - 1) && (p.n_total % TOKT != 0);
  const int t1f = ragged ? t1 - 1 : t1;
  h8 xa[PF_KB], xb[PF_KB];
  int t = t0;
  if (t < t1f) load(xa, t);
  while (t < t1f) {
    // (a workgroup barrier here, to keep the four waves on the same tile so that the L1 could merge their operand
    // loads, was measured: 130.9 vs 128.3 us at 10 000 x 8 160 -- the passes are not bound by L2 bandwidth)
    if (t + 1 < t1f) load(xb, t + 1);
    process(xa, t, std::true_type{});
    if (t + 1 >= t1f) break;
    if (t + 2 < t1f) load(xa, t + 2);
    process(xb, t + 1, std::true_type{});
    t += 2;
  }
  if (ragged) {
    load(xa, t1f);
    process(xa, t1f, std::false_type{});
  }

#pragma unroll
  for (int u = 0; u < QG; ++u) {
    if (!q_ok[u]) continue;
    if (PASS == 0) {
      float* dst = p.gmax + ((int64_t)split * p.hw + q[u]) * PF_GROUPS;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(r & 3) + 8 * (r >> 2) + 4 * half] = g[u][r];
    } else {
      p.cand_cnt[((int64_t)q[u] * p.splits + split) * 2 + half] = ccnt[u];
      if (ccnt[u] > (uint32_t)PF_SUB) atomicOr(&p.st->flag, 4u);
    }
  }
}

// one wave per query: threshold = the k-th largest of its splits x 32 group maxima, resolved to the top 18 bits of
// the order-preserving score bits (rounded DOWN: still a valid lower bound, at most 2^-9 relative below the exact
// value -- a fraction of the bound's own width --, usually exact through the early exit) minus the absolute slack
constexpr int PF_TAU_E = PF_MAX_SPLITS * PF_GROUPS / 64;
__global__ __launch_bounds__(256) void affinity_pf_tau_kernel(const float* __restrict__ gmax, const float* __restrict__ eq,
                                                              int hw, int k, int splits, float* __restrict__ thr) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= hw) return;
  const int total = splits * PF_GROUPS;
  uint32_t e[PF_TAU_E];
#pragma unroll
  for (int i = 0; i < PF_TAU_E; ++i) {
    const int gi = i * 64 + lane;
    uint32_t key = 0u;
    if (gi < total) {
      float v = gmax[((int64_t)(gi / PF_GROUPS) * hw + q) * PF_GROUPS + (gi % PF_GROUPS)];
      v = (v == v) ? v : -INFINITY;
      key = orderable(v);
    }
    e[i] = key;
  }
  const int n_live = (total + 63) / 64;
  uint32_t T = 0u;
  for (int b = 31; b >= 14; --b) {
    const uint32_t trial = T | (1u << b);
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < PF_TAU_E; ++i)
      if (i < n_live) cnt += wave_count(e[i] >= trial);
    if (cnt >= k) {
      T = trial;
      if (cnt == k) break;  // separates exactly k groups: the remaining bits cannot raise it past the k-th value
    }
  }
  // T == 0 (fewer than k groups hold tokens): -inf, every score becomes a candidate and the sub-lists overflow
  if (lane == 0) thr[q] = (T == 0u ? -INFINITY : from_orderable(T)) - 2.0f * (PF_ABS + eq[q]);
}

// Candidate totals per query, checked BEFORE the re-score kernel commits anything: a query with more candidates than
// the re-score buffer holds (or fewer than k) raises the fall-back flag here, so that the re-score kernel -- which
// writes idx / weight and adds to the usage counters -- either runs for every query or for none (it reads the flag
// at entry; raised from inside it, the flag let earlier workgroups' usage additions stand and the fp32 fall-back
// counted those queries twice).
__global__ __launch_bounds__(256) void affinity_pf_check_kernel(const uint32_t* __restrict__ cand_cnt, int hw, int k,
                                                                int n_sub, const uint32_t* __restrict__ qbad, PfState* st) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  bool bad = false, bad_query = false;
  if (q < hw) {
    const uint32_t* c = cand_cnt + (int64_t)q * n_sub;
    uint32_t total = 0u;
    for (int i = 0; i < n_sub; ++i) total += c[i] < (uint32_t)PF_SUB ? c[i] : (uint32_t)PF_SUB;
    bad = total > (uint32_t)PF_RESC_MAX || total < (uint32_t)k;
    bad_query = qbad[q / QT] != 0u;  // a negative / non-finite selection or key in the query's group (affinity_pf_query_kernel)
  }
  const uint32_t bits = (__builtin_amdgcn_ballot_w64(bad) ? 8u : 0u) | (__builtin_amdgcn_ballot_w64(bad_query) ? 2u : 0u);
  if (bits && (threadIdx.x & 63) == 0) atomicOr(&st->flag, bits);
}

struct PfRescoreArgs {
  PfBank bank;
  const float* qk;
  const float* qe;
  int hw;
  int k;
  int splits;
  PfState* st;
  const uint64_t* cand;
  const uint32_t* cand_cnt;
  int32_t* idx;
  float* weight;
  unsigned long long* usage_fix;
  uint64_t* out_keys;
  uint32_t* out_cnt;
  uint32_t token_offset;
};

// one wave per query (four queries per workgroup): gather candidates, exact fp32 scores (the FMA chain of
// v_mfma_f32_32x32x2_f32: channels in natural order, mk^2 rounded before it enters the chain, qk*qe rounded
// likewise), exact top-k, softmax / usage
__global__ __launch_bounds__(256) void affinity_pf_rescore_kernel(const PfRescoreArgs p) {
  __shared__ uint64_t s_key[4][PF_RESC_MAX];  // candidate tokens, overwritten in place by their exact (score, token) keys
  __shared__ __attribute__((aligned(16))) float s_qe[4][CK];
  __shared__ __attribute__((aligned(16))) float s_qp[4][CK];
  __shared__ float s_term[4][CK];
  __shared__ uint64_t s_buf[4][2][64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + wave;
  const int k = p.k;
  {  // the four queries' operands, loaded by the whole workgroup: 16 B per channel row instead of 4 x 4 B
    const int c = threadIdx.x >> 2, jq = threadIdx.x & 3;
    const int qq = min(blockIdx.x * 4 + jq, p.hw - 1);
    const float ev = p.qe[(int64_t)c * p.hw + qq], kv = p.qk[(int64_t)c * p.hw + qq];
    s_qe[jq][c] = ev;
    s_qp[jq][c] = kv * ev;
    s_term[jq][c] = ev * (kv * kv);
  }
  __syncthreads();
  if (q >= p.hw) return;
  if (p.st->flag != 0u) return;  // the fp32 kernels produce this read

  // ---- bsq in ATen's summation order: four 16-channel partial sums (one per lane 0..3), then ((s0+s1)+s2)+s3
  float part = 0.0f;
  if (lane < 4) {
#pragma unroll
    for (int c = 0; c < 16; ++c) part += s_term[wave][16 * lane + c];
  }
  const float bs0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(part), 0));
  const float bs1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(part), 1));
  const float bs2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(part), 2));
  const float bs3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(part), 3));
  const float bsq = ((bs0 + bs1) + bs2) + bs3;

  // ---- gather: lane l owns sub-list l = (range, half-lane) of this query (<= 64 of them); its entries go to
  // s_tok[prefix(l) .. prefix(l) + count(l))
  const int n_sub = p.splits * 2;
  const int64_t sub0 = (int64_t)q * n_sub;
  uint32_t mine = 0u;
  if (lane < n_sub) {
    mine = p.cand_cnt[sub0 + lane];
    mine = mine < (uint32_t)PF_SUB ? mine : (uint32_t)PF_SUB;
  }
  uint32_t incl = mine;  // inclusive prefix sum over the lanes
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
    if (lane >= o) incl += up;
  }
  const int total = __builtin_amdgcn_readlane((int)incl, 63);
  const uint32_t base = incl - mine;
  uint32_t longest = mine;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const uint32_t other = (uint32_t)__shfl_xor((int)longest, o, 64);
    longest = other > longest ? other : longest;
  }
  if (total > PF_RESC_MAX || total < k) return;  // unreachable: affinity_pf_check_kernel raised the flag (same totals)
  const uint64_t* my_sub = p.cand + (sub0 + lane) * PF_SUB;
  for (uint32_t s0 = 0; s0 < longest; s0 += 4) {  // four independent loads in flight per lane
    uint64_t ent[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ent[u] = (s0 + u < mine) ? my_sub[s0 + u] : 0ull;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (s0 + u < mine) s_key[wave][base + s0 + u] = ent[u] & 0xffffffffull;
  }
  DEVA_COMPILER_FENCE();

  // ---- exact scores, 64 candidates per round (a real loop: the keys go to LDS, the registers of a round are reused)
  for (int r0 = 0; r0 < total; r0 += 64) {
    const int c = r0 + lane;
    const bool live_c = c < total;
    const uint32_t tok = live_c ? (uint32_t)s_key[wave][c] : 0u;
    float ms;
    const float* row = pf_row(p.bank, (int)tok, &ms);
    float accA = 0.0f, accB = 0.0f;
#pragma unroll
    for (int j = 0; j < CK / 4; ++j) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(row + 4 * j);
      const f32x4 qe4 = *reinterpret_cast<const f32x4*>(&s_qe[wave][4 * j]);
      const f32x4 qp4 = *reinterpret_cast<const f32x4*>(&s_qp[wave][4 * j]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float a = x[u];
        accA = __builtin_fmaf(a * a, qe4[u], accA);
        accB = __builtin_fmaf(a, qp4[u], accB);
      }
    }
    const float v = (((accB + accB) - accA) - bsq) * (ms * 0.125f);
    s_key[wave][c] = live_c ? (((uint64_t)orderable(v) << 32) | (uint64_t)(~tok)) : 0ull;
  }
  DEVA_COMPILER_FENCE();
  const int rounds = (total + 63) / 64;
  uint64_t e[PF_RESC_MAX / 64];
#pragma unroll
  for (int rr = 0; rr < PF_RESC_MAX / 64; ++rr) e[rr] = (rr < rounds) ? s_key[wave][rr * 64 + lane] : 0ull;
  volatile uint64_t* unsorted = &s_buf[wave][0][0];
  volatile uint64_t* sorted = &s_buf[wave][1][0];
  uint64_t best;  // lane r: the r-th best key
  const bool live = lane < k;
  if (total <= 64) {
    // one round (the usual case): rank counting over the <= 64 unique keys selects AND sorts
    const uint64_t key = e[0];
    int rank = 0;
    for (int j = 0; j < total; ++j) {
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, j);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), j);
      rank += ((((uint64_t)hi << 32) | lo) > key) ? 1 : 0;
    }
    if (lane < total && rank < k) sorted[rank] = key;
    DEVA_COMPILER_FENCE();
    best = live ? sorted[lane] : 0ull;
  } else {
    const uint64_t thr = kth_largest<PF_RESC_MAX / 64>(e, rounds, k);
    int base_k = 0;
#pragma unroll
    for (int i = 0; i < PF_RESC_MAX / 64; ++i) {
      if (i < rounds) {
        const bool keep = e[i] >= thr && e[i] != 0ull;
        const unsigned long long b = __builtin_amdgcn_ballot_w64(keep);
        if (keep) unsorted[base_k + prefix_below(b)] = e[i];
        base_k += __popcll(b);
      }
    }
    DEVA_COMPILER_FENCE();
    const uint64_t cand = live ? unsorted[lane] : 0ull;
    int rank = 0;
    for (int j = 0; j < k; ++j) {
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cand, j);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cand >> 32), j);
      rank += ((((uint64_t)hi << 32) | lo) > cand) ? 1 : 0;
    }
    DEVA_COMPILER_FENCE();
    if (live) sorted[rank] = cand;
    DEVA_COMPILER_FENCE();
    best = live ? sorted[lane] : 0ull;
  }
  // ---- from here on: affinity_finalize_kernel's tail
  if (p.out_keys) {
    if (live) p.out_keys[(int64_t)q * CAP + lane] = (best & 0xffffffff00000000ull) | (uint64_t)(~(~(uint32_t)best + p.token_offset));
    if (lane == 0) p.out_cnt[q] = (uint32_t)k;
    return;
  }
  const float score = from_orderable((uint32_t)(best >> 32));
  const uint32_t token = ~(uint32_t)best;
  const float ex = live ? expf(score) : 0.0f;
  float sum = 0.0f;
  for (int r = 0; r < k; ++r) sum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ex), r));
  const float w = ex / sum;
  if (live) {
    p.idx[(int64_t)q * k + lane] = (int32_t)token;
    p.weight[(int64_t)q * k + lane] = w;
    if (p.usage_fix && w == w) atomicAdd(&p.usage_fix[token], (unsigned long long)(w * 1099511627776.0f));
  }
}

}  // namespace
}  // namespace deva

using namespace deva;

extern "C" int64_t deva_affinity_workspace(int hw, int k, int splits) {
  (void)k;
  // [splits][hw][CAP] 64-bit candidate keys, then [splits][hw] 32-bit list lengths
  return (int64_t)splits * hw * CAP + ((int64_t)splits * hw + 1) / 2;
}

// kernel shapes: 1 = 176-slot lists, one workgroup per CU, every lane loads its own key row;
// 2 = 100-slot lists, two workgroups per CU (small frames: twice the resident workgroups);
// 3 = 176-slot lists, one workgroup per CU, key tiles loaded once per workgroup through LDS;
// 4 = workgroup-shared lists (4 waves x the same 32 queries), 352 slots, two workgroups per CU;
// 5 = workgroup-shared lists, 704 slots, one workgroup per CU;
// 6 = shape 2 with the early prefetch; 7 = ping-pong (8 waves in two groups alternating matrix / scoring phases);
// 8 = workgroup-shared lists, EIGHT waves x the same 32 queries, 704 slots, one workgroup per CU.
// DEVA_AFFINITY_SHAPE overrides the choice (tuning / A-B measurements only).
static int g_forced_shape = -1;  // -1: not initialised (DEVA_AFFINITY_SHAPE is read on first use)

static int affinity_shape(int n_total, int hw) {
  if (g_forced_shape < 0) {
    const char* e = getenv("DEVA_AFFINITY_SHAPE");
    const int v = e ? atoi(e) : 0;
#ifdef DEVA_AFFINITY_PROBES
    g_forced_shape = (v >= 1 && v <= 8) ? v : 0;
#else
    g_forced_shape = (v == 2 || v == 4 || v == 8) ? v : 0;
#endif
  }
  if (g_forced_shape) return g_forced_shape;
  // measured (profiles/r02e_affinity_shapes.txt, total us of filter + finalize):
  //   few query blocks (480p, hw = 1 620): the 8-wave workgroups need half the token ranges for the same number
  //   of workgroups, i.e. half the lists to merge: 42 vs 71 (N = 1 620), 83 vs 94 (8 100), 192 vs 200 (24 580);
  //   from ~30 000 tokens on the 4-wave shape is ahead (286 vs 300 at 40 000) and stays ahead of the per-wave lists
  //   (518 vs 624 at 83 440);
  //   many query blocks (1080p / 4K): shared lists while appends / prunes dominate (312 vs 362 at 10 000 x 8 160,
  //   1 114 vs 1 195 at 10 000 x 32 400), per-wave lists without barriers on longer banks (829 vs 859 at 30 000,
  //   1 039 vs 1 118 at 40 000, 1 873 vs 2 128 at 83 440 x 8 160).
  if (hw <= 4096) return n_total <= 30000 ? 8 : 4;
  return n_total <= 20000 ? 4 : 2;
}

extern "C" int deva_affinity_force_shape(int shape) {
  DEVA_REQUIRE(shape >= 0 && shape <= 8, "deva_affinity_force_shape: shape must be 0 (automatic) .. 8");
#ifndef DEVA_AFFINITY_PROBES
  // the product library carries the three shapes the automatic choice uses; 1, 3, 5, 6, 7 are A/B variants
  // (bit-identical, slower: profiles/r02e_affinity_shapes.txt) of `make PROBES=1` builds
  DEVA_REQUIRE(shape == 0 || shape == 2 || shape == 4 || shape == 8,
               "deva_affinity_force_shape: shape %d is an A/B variant of probe builds (make PROBES=1)", shape);
#endif
  g_forced_shape = shape;
  return 0;
}

#ifdef DEVA_AFFINITY_PROBES
static uint64_t* g_probe = nullptr;
// probe builds only (not part of the ABI): device buffer of 8 workgroups x 8 waves x 64 tiles x 8 cycle stamps
extern "C" int deva_affinity_set_probe(uint64_t* buf) {
  g_probe = buf;
  return 0;
}
#endif

extern "C" int deva_affinity_default_splits(int n_total, int hw) {
  // aim at one resident set of workgroups: 256 CUs x (1 or 2) four-wave workgroups
  const int shape = affinity_shape(n_total, hw);
  const int slots = (shape == 2 || shape == 4 || shape == 6) ? 512 : 256;
  const bool wg_lists = shape == 4 || shape == 5 || shape == 7 || shape == 8;
  const int qblocks = (int)ceil_div(hw, wg_lists ? QT : WAVES * QT);
  if (shape == 8) {
    // one 8-wave workgroup per CU, every token range hands over one list per query
    const int tiles8 = (int)ceil_div(n_total, TOKT);
    int r = (256 + qblocks / 2) / qblocks;
    if (r > tiles8 / 16) r = tiles8 / 16;  // >= 2 tiles per wave and range
    if (r > MAX_SPLITS) r = MAX_SPLITS;
    if (r < 1) r = 1;
    while (r < MAX_SPLITS && ceil_div(tiles8, r) > 2047) ++r;
    return r;
  }
  if (shape == 7) {
    // ping-pong kernel: one 8-wave workgroup per CU; every token range hands over TWO lists per query, and
    // `splits` counts lists
    const int tiles7 = (int)ceil_div(n_total, TOKT);
    int r = (256 + qblocks / 2) / qblocks;
    if (r > tiles7 / 16) r = tiles7 / 16;  // >= 2 tiles per wave and range
    if (r > MAX_SPLITS / 2) r = MAX_SPLITS / 2;
    if (r < 1) r = 1;
    while (r < MAX_SPLITS / 2 && ceil_div(tiles7, r) > 2047) ++r;
    return 2 * r;
  }
  const int tiles = (int)ceil_div(n_total, TOKT);
  // workgroup-shared lists: the grid should be a whole number of resident sets (round, do not overshoot)
  int s = wg_lists ? (slots + qblocks / 2) / qblocks : (int)ceil_div(slots, qblocks);
  if (s > tiles / 4) s = tiles / 4;  // keep >= 4 tiles (128 tokens) per range
  if (s > MAX_SPLITS) s = MAX_SPLITS;
  if (s < 1) s = 1;
  // small banks (first memory frames of a clip): ranges of <= CAP tokens hand every score over without
  // building a threshold or pruning
  const int s_nofilter = (int)ceil_div(tiles, CAP / TOKT);
  if (s_nofilter <= MAX_SPLITS && s_nofilter > s) s = s_nofilter;
  while (s < MAX_SPLITS && ceil_div(tiles, s) > 2047) ++s;  // 16-bit token offsets inside a range
  return s;
}

static int topk_fp32(const float* key_long, const float* shr_long, int n_long, const float* key_work,
                     const float* shr_work, int n_work, const float* qk, const float* qe, int hw, int k, int splits,
                     uint64_t* part_keys, void* stream, const uint32_t* guard) {
  DEVA_REQUIRE(qk && qe && part_keys && hw > 0, "deva_affinity_topk: bad query args");
  DEVA_REQUIRE(n_long >= 0 && n_work >= 0, "deva_affinity_topk: negative bank size");
  DEVA_REQUIRE(n_long == 0 || (key_long && shr_long), "deva_affinity_topk: null long-term segment");
  DEVA_REQUIRE(n_work == 0 || (key_work && shr_work), "deva_affinity_topk: null working segment");
  DEVA_REQUIRE(k >= 1 && k <= K_MAX, "deva_affinity_topk: k=%d unsupported (1..%d)", k, K_MAX);
  const int64_t n_total = (int64_t)n_long + n_work;
  DEVA_REQUIRE(n_total >= k, "deva_affinity_topk: selected index k out of range (bank has %lld tokens, k=%d)",
               (long long)n_total, k);
  DEVA_REQUIRE(n_total < (1ll << 31), "deva_affinity_topk: bank too large");
  DEVA_REQUIRE(splits >= 1 && splits <= MAX_SPLITS, "deva_affinity_topk: splits must be 1..%d", MAX_SPLITS);
  AffArgs a;
  a.key_long = key_long ? key_long : key_work;
  a.shr_long = shr_long ? shr_long : shr_work;
  a.n_long = n_long;
  a.key_work = key_work ? key_work : key_long;
  a.shr_work = shr_work ? shr_work : shr_long;
  a.n_total = (int)n_total;
  a.qk = qk;
  a.qe = qe;
  a.hw = hw;
  a.k = k;
  a.splits = splits;
  a.total_tiles = (int)ceil_div(n_total, TOKT);
  DEVA_REQUIRE(ceil_div(a.total_tiles, splits) <= 2047,
               "deva_affinity_topk: %d tokens per range exceed the 16-bit in-range token offset; use more splits",
               (int)ceil_div(a.total_tiles, splits) * TOKT);
  a.part = part_keys;
  a.part_cnt = reinterpret_cast<uint32_t*>(part_keys + (int64_t)splits * hw * CAP);
#ifdef DEVA_AFFINITY_PROBES  // `make PROBES=1`: timing probes for the ablation table (profiles/r02b_affinity_shapes.txt)
  static const int ablate = [] {
    const char* e = getenv("DEVA_AFFINITY_ABLATE");
    return e ? atoi(e) : 0;
  }();
  a.ablate = ablate;
  a.probe = g_probe;
#else
  a.ablate = 0;
  a.probe = nullptr;
#endif
  a.guard = guard;
  dim3 grid((unsigned)ceil_div(hw, WAVES * QT), (unsigned)splits);
  const dim3 grid_wg((unsigned)ceil_div(hw, QT), (unsigned)splits);
  int shape = affinity_shape((int)n_total, hw);
  if (shape == 7 && (splits % 2 != 0 || ceil_div(a.total_tiles, splits / 2) > 2047)) shape = 4;  // needs list pairs
  switch (shape) {
#ifdef DEVA_AFFINITY_PROBES  // A/B variants: bit-identical, slower (profiles/r02e_affinity_shapes.txt)
    case 7:
      hipLaunchKernelGGL((affinity_topk_pp_kernel<352>), dim3((unsigned)ceil_div(hw, QT), (unsigned)(splits / 2)),
                         dim3(PP_WAVES * 64), 0, (hipStream_t)stream, a);
      break;
    case 6:  // shape 2 with the early prefetch (operands copied out, next loads issued before the MFMAs)
      hipLaunchKernelGGL((affinity_topk_kernel<LCAP_DUAL, 2, false, false>), grid, dim3(WAVES * 64), 0, (hipStream_t)stream, a);
      break;
    case 5:
      hipLaunchKernelGGL((affinity_topk_wg_kernel<704, 1, WAVES>), grid_wg, dim3(WAVES * 64), 0, (hipStream_t)stream, a);
      break;
    case 1:
      hipLaunchKernelGGL((affinity_topk_kernel<LCAP_WIDE, 1, false>), grid, dim3(WAVES * 64), 0, (hipStream_t)stream, a);
      break;
    case 3:
      hipLaunchKernelGGL((affinity_topk_kernel<LCAP_WIDE, 1, true>), grid, dim3(WAVES * 64), 0, (hipStream_t)stream, a);
      break;
#endif
    case 8:
      hipLaunchKernelGGL((affinity_topk_wg_kernel<704, 1, 8>), grid_wg, dim3(8 * 64), 0, (hipStream_t)stream, a);
      break;
    case 4:
      hipLaunchKernelGGL((affinity_topk_wg_kernel<352, 2, WAVES>), grid_wg, dim3(WAVES * 64), 0, (hipStream_t)stream, a);
      break;
    default:  // 2: key rows prefetched after the MFMAs, read in place (2-6 % faster than the early prefetch + copies)
      hipLaunchKernelGGL((affinity_topk_kernel<LCAP_DUAL, 2, false, true>), grid, dim3(WAVES * 64), 0, (hipStream_t)stream, a);
  }
  return check_launch("deva_affinity_topk");
}

extern "C" int deva_affinity_topk(const float* key_long, const float* shr_long, int n_long, const float* key_work,
                                  const float* shr_work, int n_work, const float* qk, const float* qe, int hw,
                                  int k, int splits, uint64_t* part_keys, void* stream) {
  return topk_fp32(key_long, shr_long, n_long, key_work, shr_work, n_work, qk, qe, hw, k, splits, part_keys, stream,
                   nullptr);
}

static int launch_merge(const uint64_t* keys, const uint32_t* cnt, int hw, int k, int lists, int32_t* idx, float* weight,
                        uint64_t* usage_fix, uint64_t* out_keys, uint32_t* out_cnt, uint32_t token_offset,
                        void* stream, const char* what, const uint32_t* guard = nullptr) {
  const dim3 grid((unsigned)ceil_div(hw, 4));
#define DEVA_MERGE(ME)                                                                                            \
  hipLaunchKernelGGL(affinity_finalize_kernel<ME>, grid, dim3(256), 0, (hipStream_t)stream, keys, cnt, hw, k, lists, \
                     idx, weight, (unsigned long long*)usage_fix, out_keys, out_cnt, token_offset, guard)
  if (lists <= 4) {
    DEVA_MERGE(4);
  } else if (lists <= 8) {
    DEVA_MERGE(8);
  } else if (lists <= 16) {
    DEVA_MERGE(16);
  } else {
    DEVA_MERGE(32);
  }
#undef DEVA_MERGE
  return check_launch(what);
}

extern "C" int deva_affinity_finalize(const uint64_t* part_keys, int hw, int k, int splits, int32_t* idx,
                                      float* weight, uint64_t* usage_fix, void* stream) {
  DEVA_REQUIRE(part_keys && idx && weight && hw > 0, "deva_affinity_finalize: bad args");
  DEVA_REQUIRE(k >= 1 && k <= K_MAX && splits >= 1 && splits <= MAX_SPLITS,
               "deva_affinity_finalize: k/splits out of range");
  const uint32_t* cnt = reinterpret_cast<const uint32_t*>(part_keys + (int64_t)splits * hw * CAP);
  return launch_merge(part_keys, cnt, hw, k, splits, idx, weight, usage_fix, nullptr, nullptr, 0u, stream,
                      "deva_affinity_finalize");
}

extern "C" int deva_affinity_select(const uint64_t* part_keys, int hw, int k, int splits, int64_t token_offset,
                                    uint64_t* out_keys, uint32_t* out_counts, void* stream) {
  DEVA_REQUIRE(part_keys && out_keys && out_counts && hw > 0, "deva_affinity_select: bad args");
  DEVA_REQUIRE(k >= 1 && k <= K_MAX && splits >= 1 && splits <= MAX_SPLITS,
               "deva_affinity_select: k/splits out of range");
  DEVA_REQUIRE(token_offset >= 0 && token_offset < (1ll << 31), "deva_affinity_select: bad token offset");
  const uint32_t* cnt = reinterpret_cast<const uint32_t*>(part_keys + (int64_t)splits * hw * CAP);
  return launch_merge(part_keys, cnt, hw, k, splits, nullptr, nullptr, nullptr, out_keys, out_counts,
                      (uint32_t)token_offset, stream, "deva_affinity_select");
}

extern "C" int deva_affinity_merge(const uint64_t* keys, const uint32_t* counts, int hw, int k, int lists, int32_t* idx,
                                   float* weight, uint64_t* usage_fix, void* stream) {
  DEVA_REQUIRE(keys && counts && idx && weight && hw > 0, "deva_affinity_merge: bad args");
  DEVA_REQUIRE(k >= 1 && k <= K_MAX && lists >= 1 && lists <= MAX_SPLITS,
               "deva_affinity_merge: k/lists out of range");
  return launch_merge(keys, counts, hw, k, lists, idx, weight, usage_fix, nullptr, nullptr, 0u, stream,
                      "deva_affinity_merge");
}

extern "C" int deva_usage_update(uint64_t* usage_fix, int64_t offset, float* use, float* life, int n,
                                 void* stream) {
  DEVA_REQUIRE(usage_fix && n >= 0 && offset >= 0, "deva_usage_update: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(usage_update_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (unsigned long long*)usage_fix, offset, use, life, n);
  return check_launch("deva_usage_update");
}

extern "C" int deva_readout_sparse(const int32_t* idx, const float* weight, int hw, int k, const float* val_long,
                                   int n_long, const float* val_work, int cv, float* out, int tok_lo, int tok_hi,
                                   const int32_t* map_long, const int32_t* map_work, void* stream) {
  DEVA_REQUIRE(idx && weight && out && hw > 0 && k > 0 && cv > 0, "deva_readout_sparse: bad args");
  DEVA_REQUIRE(k <= 64, "deva_readout_sparse: k=%d unsupported (one term per lane: 1..64)", k);
  DEVA_REQUIRE(cv % 4 == 0, "deva_readout_sparse: value dim must be a multiple of 4");
  DEVA_REQUIRE(n_long == 0 || val_long, "deva_readout_sparse: null long-term values");
  const float* vl = val_long ? val_long : val_work;
  const float* vw = val_work ? val_work : val_long;
  DEVA_REQUIRE(vl && vw, "deva_readout_sparse: no value segment");
  dim3 grid((unsigned)ceil_div(hw, RQ), (unsigned)ceil_div(cv, RC));
  hipLaunchKernelGGL(readout_sparse_kernel, grid, dim3(256), 0, (hipStream_t)stream, idx, weight, hw, k, vl, n_long,
                     vw, cv, out, tok_lo, tok_hi, map_long, map_work);
  return check_launch("deva_readout_sparse");
}

// ------------------------------------------------------------------ fp16 pre-filter + exact re-scoring (host side)
static int g_prefilter = -1;  // -1: DEVA_AFFINITY_PREFILTER not read yet; 0 = never, 1 = automatic (default)
// below these the six launches cost more than the fp32 kernels need (profiles/r03_affinity_read.txt: 65 vs 40 us at
// 2 048 x 1 620, 81 vs 89 us at 10 000 x 1 620 before the query operands were hoisted)
constexpr int PF_MIN_TOKENS = 4096;
constexpr int64_t PF_MIN_SCORES = 8000000;

static int pf_qg(int hw) { return hw >= 4096 ? 2 : 1; }  // query groups per wave (large frames: half the operand traffic)

static int pf_splits(int n_total, int hw) {
  const int tiles = (int)ceil_div(n_total, TOKT);
  const int qblocks = (int)ceil_div(hw, PF_QW * QT * pf_qg(hw));
  int s = (int)ceil_div(512, qblocks);  // two 4-wave workgroups per CU
  // >= 16 tile-cyclic ranges: 512 groups per query keep the k-th largest group maximum tight, and the near-duplicate
  // tokens of consecutive memory frames spread over the ranges (measured on the 4K clip: 5 ranges -> sub-lists of 185)
  if (s < 16) s = 16;
  if (s > tiles / 2) s = tiles / 2;     // >= 2 tiles per range
  if (s > PF_MAX_SPLITS) s = PF_MAX_SPLITS;
  if (s < 1) s = 1;
  return s;
}

struct PfLayout {
  int splits, tiles, old_splits;
  int64_t off_state, off_a16, off_bq16, off_gmax, off_thr, off_eq, off_cand, off_cnt, off_part, bytes;
};

static PfLayout pf_layout(int n_total, int hw, int k) {
  PfLayout L;
  L.splits = pf_splits(n_total, hw);
  L.tiles = (int)ceil_div(n_total, TOKT);
  L.old_splits = deva_affinity_default_splits(n_total, hw);
  auto align = [](int64_t b) { return (b + 255) / 256 * 256; };
  int64_t o = 0;
  L.off_state = o;
  o += 512 + PF_STAT_BLOCKS * 16 + PF_STAT_BLOCKS * CK * 4;  // PfState | [blocks][4] partial maxima | [blocks][64] channel sums
  L.off_a16 = o;
  o += align((int64_t)L.tiles * PF_TILE_BYTES);
  L.off_bq16 = o;
  o += align(ceil_div(hw, QT) * PF_TILE_BYTES);
  L.off_gmax = o;
  o += align((int64_t)L.splits * hw * PF_GROUPS * 4);
  L.off_thr = o;
  o += align((int64_t)hw * 4);
  L.off_eq = o;
  o += align((int64_t)hw * 4);
  L.off_cand = o;
  o += align((int64_t)L.splits * hw * 2 * PF_SUB * 8);
  L.off_cnt = o;
  o += align((int64_t)L.splits * hw * 2 * 4);
  L.off_part = o;
  o += align(deva_affinity_workspace(hw, k, L.old_splits) * 8);
  L.bytes = o;
  return L;
}

// ------------------------------------------------------------------ dense read (32 < k <= 64)
// The list kernels above size their per-range hand-over for k <= K_MAX = 32.  For the rare larger top_k
// (eval_args.py:40 leaves it free) the read runs on this one kernel: lane = query (64 queries per one-wave workgroup,
// query operands in registers), the bank streams through wave-uniform rows, every score is the same natural-order fp32
// FMA chain as in affinity_pf_rescore_kernel (bit-identical scores, hence the same selection as the list kernels for any
// k both can serve), and each lane keeps its k best (score, ~token) keys in LDS -- unsorted, smallest tracked, replaced
// on insert (~k ln(N/k) inserts per query).  Finish per query exactly like affinity_finalize_kernel.  VALU-bound:
// 192 instructions per (token, query); ~4 ms at N = 10 000 x 8 160 queries -- a correct path, not a fast one.
constexpr int DK_MAX = 64;

struct DenseArgs {
  PfBank bank;
  const float* qk;
  const float* qe;
  int hw, k;
  int32_t* idx;
  float* weight;
  unsigned long long* usage_fix;
};

__global__ __launch_bounds__(64) void affinity_dense_kernel(const DenseArgs p) {
  __shared__ uint64_t s_list[DK_MAX][64];  // [entry][query lane]
  __shared__ uint64_t s_sort[64];
  const int lane = threadIdx.x;
  const int q0 = blockIdx.x * 64;
  const int qq = min(q0 + lane, p.hw - 1);
  const int k = p.k, n = p.bank.n_total;
  float qe[CK], qp[CK];
  float bs[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // bsq in ATen's summation order (see affinity_topk_kernel)
#pragma unroll
  for (int c = 0; c < CK; ++c) {
    const float ev = p.qe[(int64_t)c * p.hw + qq], kv = p.qk[(int64_t)c * p.hw + qq];
    qe[c] = ev;
    qp[c] = kv * ev;
    bs[c >> 4] += ev * (kv * kv);
  }
  const float bsq = ((bs[0] + bs[1]) + bs[2]) + bs[3];

  uint64_t kmin = ~0ull;
  int pmin = 0;
  for (int t = 0; t < n; ++t) {  // t is wave-uniform: the row and its shrinkage are scalar loads
    float ms;
    const float* row = pf_row(p.bank, t, &ms);
    float accA = 0.0f, accB = 0.0f;
#pragma unroll
    for (int c = 0; c < CK; ++c) {
      const float a = row[c];
      accA = __builtin_fmaf(a * a, qe[c], accA);
      accB = __builtin_fmaf(a, qp[c], accB);
    }
    const float v = (((accB + accB) - accA) - bsq) * (ms * 0.125f);
    const uint64_t key = ((uint64_t)orderable(v) << 32) | (uint64_t)(~(uint32_t)t);
    if (t < k) {  // (uniform) the first k tokens fill the list
      s_list[t][lane] = key;
      if (key < kmin) {
        kmin = key;
        pmin = t;
      }
    } else if (key > kmin) {
      s_list[pmin][lane] = key;
      kmin = ~0ull;
      for (int e = 0; e < k; ++e) {
        const uint64_t o = s_list[e][lane];
        if (o < kmin) {
          kmin = o;
          pmin = e;
        }
      }
    }
  }
  __syncthreads();

  // ---- per query of the tile: sort the k keys by rank counting, exp / normalise / usage (affinity_finalize_kernel's tail)
  const bool live = lane < k;
  for (int j = 0; j < 64 && q0 + j < p.hw; ++j) {
    const int q = q0 + j;
    const uint64_t cand = live ? s_list[live ? lane : 0][j] : 0ull;
    int rank = 0;
    for (int r = 0; r < k; ++r) {
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cand, r);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cand >> 32), r);
      rank += ((((uint64_t)hi << 32) | lo) > cand) ? 1 : 0;
    }
    __syncthreads();  // (one wave: orders the LDS traffic of consecutive queries)
    if (live) s_sort[rank] = cand;
    __syncthreads();
    const uint64_t mine = live ? s_sort[lane] : 0ull;  // lane r holds the r-th best
    const float score = from_orderable((uint32_t)(mine >> 32));
    const uint32_t token = ~(uint32_t)mine;
    const float ex = live ? expf(score) : 0.0f;
    float sum = 0.0f;
    for (int r = 0; r < k; ++r)  // sequential, like torch.sum over the sorted top-k
      sum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ex), r));
    const float w = ex / sum;
    if (live) {
      p.idx[(int64_t)q * k + lane] = (int32_t)token;
      p.weight[(int64_t)q * k + lane] = w;
      if (p.usage_fix && w == w) atomicAdd(&p.usage_fix[token], (unsigned long long)(w * 1099511627776.0f));
    }
  }
}

extern "C" int deva_affinity_dense(const float* key_long, const float* shr_long, int n_long, const float* key_work,
                                   const float* shr_work, int n_work, const float* qk, const float* qe, int hw, int k,
                                   int32_t* idx, float* weight, uint64_t* usage_fix, void* stream) {
  DEVA_REQUIRE(qk && qe && idx && weight && hw > 0, "deva_affinity_dense: bad args");
  DEVA_REQUIRE(n_long >= 0 && n_work >= 0, "deva_affinity_dense: negative bank size");
  DEVA_REQUIRE(n_long == 0 || (key_long && shr_long), "deva_affinity_dense: null long-term segment");
  DEVA_REQUIRE(n_work == 0 || (key_work && shr_work), "deva_affinity_dense: null working segment");
  DEVA_REQUIRE(k >= 1 && k <= DK_MAX, "deva_affinity_dense: k=%d unsupported (1..%d)", k, DK_MAX);
  const int64_t n_total = (int64_t)n_long + n_work;
  DEVA_REQUIRE(n_total >= k, "deva_affinity_dense: selected index k out of range (bank has %lld tokens, k=%d)",
               (long long)n_total, k);
  DEVA_REQUIRE(n_total < (1ll << 31) - 64, "deva_affinity_dense: bank too large");
  DenseArgs a;
  a.bank.key_long = key_long ? key_long : key_work;
  a.bank.shr_long = shr_long ? shr_long : shr_work;
  a.bank.n_long = n_long;
  a.bank.key_work = key_work ? key_work : key_long;
  a.bank.shr_work = shr_work ? shr_work : shr_long;
  a.bank.n_total = (int)n_total;
  a.qk = qk;
  a.qe = qe;
  a.hw = hw;
  a.k = k;
  a.idx = idx;
  a.weight = weight;
  a.usage_fix = (unsigned long long*)usage_fix;
  hipLaunchKernelGGL(affinity_dense_kernel, dim3((unsigned)ceil_div(hw, 64)), dim3(64), 0, (hipStream_t)stream, a);
  return check_launch("deva_affinity_dense");
}

extern "C" int deva_affinity_prefilter_enabled(int n_total, int hw, int k) {
  if (g_prefilter < 0) {
    const char* e = getenv("DEVA_AFFINITY_PREFILTER");
    g_prefilter = (e && atoi(e) == 0) ? 0 : 1;
  }
  return g_prefilter && n_total >= PF_MIN_TOKENS && (int64_t)n_total * hw >= PF_MIN_SCORES && k >= 1 && k <= K_MAX;
}

extern "C" int deva_affinity_force_prefilter(int mode) {
  DEVA_REQUIRE(mode == 0 || mode == 1, "deva_affinity_force_prefilter: 0 = fp32 kernels only, 1 = automatic");
  g_prefilter = mode;
  return 0;
}

extern "C" int64_t deva_affinity_read_scratch(int n_total, int hw, int k) {
  if (k > K_MAX) return 64;  // the dense kernel (32 < k <= 64) needs no scratch; a token size keeps callers uniform
  return pf_layout(n_total, hw, k).bytes / 8;
}

extern "C" int64_t deva_affinity_bank_prep_bytes(int n_total) {
  if (n_total <= 0) return 512;
  return 512 + (ceil_div(n_total, TOKT) * (int64_t)PF_TILE_BYTES + 255) / 256 * 256;
}

extern "C" int deva_affinity_read(const float* key_long, const float* shr_long, int n_long, const float* key_work,
                                  const float* shr_work, int n_work, const float* qk, const float* qe, int hw, int k,
                                  uint64_t* scratch, int32_t* idx, float* weight, uint64_t* usage_fix,
                                  uint64_t* out_keys, uint32_t* out_counts, int64_t token_offset, void* stream) {
  return deva_affinity_read_prepared(key_long, shr_long, n_long, key_work, shr_work, n_work, qk, qe, hw, k, scratch, idx, weight,
                                     usage_fix, out_keys, out_counts, token_offset, nullptr, 0, stream);
}

extern "C" int deva_affinity_read_prepared(const float* key_long, const float* shr_long, int n_long, const float* key_work,
                                           const float* shr_work, int n_work, const float* qk, const float* qe, int hw, int k,
                                           uint64_t* scratch, int32_t* idx, float* weight, uint64_t* usage_fix,
                                           uint64_t* out_keys, uint32_t* out_counts, int64_t token_offset,
                                           uint64_t* bank_prep, int bank_prep_valid, void* stream) {
  DEVA_REQUIRE(qk && qe && scratch && hw > 0, "deva_affinity_read: bad query args");
  DEVA_REQUIRE((idx && weight && !out_keys && !out_counts) || (out_keys && out_counts && !idx && !weight && !usage_fix),
               "deva_affinity_read: pass either idx + weight (+ usage_fix) or out_keys + out_counts");
  DEVA_REQUIRE(n_long >= 0 && n_work >= 0, "deva_affinity_read: negative bank size");
  DEVA_REQUIRE(n_long == 0 || (key_long && shr_long), "deva_affinity_read: null long-term segment");
  DEVA_REQUIRE(n_work == 0 || (key_work && shr_work), "deva_affinity_read: null working segment");
  if (k > K_MAX) {  // beyond the list kernels: one dense kernel, no hand-over format (a token-sharded bank cannot use it)
    DEVA_REQUIRE(idx && weight, "deva_affinity_read: k=%d > %d is served by the dense kernel, which has no hand-over format "
                 "(out_keys / out_counts)", k, K_MAX);
    return deva_affinity_dense(key_long, shr_long, n_long, key_work, shr_work, n_work, qk, qe, hw, k, idx, weight, usage_fix,
                               stream);
  }
  DEVA_REQUIRE(k >= 1, "deva_affinity_read: k=%d unsupported", k);
  const int64_t n_total = (int64_t)n_long + n_work;
  DEVA_REQUIRE(n_total >= k, "deva_affinity_read: selected index k out of range (bank has %lld tokens, k=%d)",
               (long long)n_total, k);
  DEVA_REQUIRE(n_total < (1ll << 31) - 64, "deva_affinity_read: bank too large");
  DEVA_REQUIRE(token_offset >= 0 && token_offset < (1ll << 31), "deva_affinity_read: bad token offset");
  hipStream_t st = (hipStream_t)stream;
  const PfLayout L = pf_layout((int)n_total, hw, k);
  uint8_t* base = reinterpret_cast<uint8_t*>(scratch);
  PfState* state = reinterpret_cast<PfState*>(base + L.off_state);
  uint64_t* part = reinterpret_cast<uint64_t*>(base + L.off_part);

  if (deva_affinity_prefilter_enabled((int)n_total, hw, k)) {
    PfBank b;
    b.key_long = key_long ? key_long : key_work;
    b.shr_long = shr_long ? shr_long : shr_work;
    b.n_long = n_long;
    b.key_work = key_work ? key_work : key_long;
    b.shr_work = shr_work ? shr_work : shr_long;
    b.n_total = (int)n_total;
    uint32_t* stat_part = reinterpret_cast<uint32_t*>(base + L.off_state + 512);  // [PF_STAT_BLOCKS][4], after the state
    float* sums = reinterpret_cast<float*>(base + L.off_state + 512 + PF_STAT_BLOCKS * 16);  // [PF_STAT_BLOCKS][64]
    // the bank side of the operands (mean key, scales, fp16 fragments) depends on the bank alone: with a prepared-bank
    // buffer it lives there, and a read that the caller declares to be on the UNCHANGED bank (same rows, same n_long /
    // n_work as the read that filled the buffer) skips the three bank kernels
    uint8_t* const a16 = bank_prep ? reinterpret_cast<uint8_t*>(bank_prep) + 512 : base + L.off_a16;
    PfState* const keep = bank_prep ? reinterpret_cast<PfState*>(bank_prep) : nullptr;
    const bool cached = keep && bank_prep_valid;
    if (!cached) {
      hipLaunchKernelGGL(affinity_pf_mean_kernel, dim3(PF_STAT_BLOCKS), dim3(256), 0, st, b, sums);
      hipLaunchKernelGGL(affinity_pf_stats_kernel, dim3(PF_STAT_BLOCKS), dim3(256), 0, st, b, sums, stat_part, state->mu);
      const int n_pad = L.tiles * TOKT;
      hipLaunchKernelGGL(affinity_pf_prep_kernel, dim3((unsigned)ceil_div((int64_t)n_pad * 2, 256)), dim3(256), 0, st, b,
                         stat_part, state, keep, n_pad, a16);
    }
    // (the channel sums are dead once the prep kernel has run: their block holds the query groups' bad-input marks)
    uint32_t* const qbad = reinterpret_cast<uint32_t*>(sums);
    DEVA_REQUIRE(ceil_div(hw, QT) <= PF_STAT_BLOCKS * CK, "deva_affinity_read: frame too large for the pre-filter's scratch");
    hipLaunchKernelGGL(affinity_pf_query_kernel, dim3((unsigned)ceil_div(hw, 4 * QT)), dim3(256), 0, st, qk, qe, hw,
                       cached ? keep : state, state, base + L.off_bq16, reinterpret_cast<float*>(base + L.off_eq), qbad);
    PfArgs a;
    a.bq16 = base + L.off_bq16;
    a.a16 = a16;
    a.n_total = (int)n_total;
    a.total_tiles = L.tiles;
    a.qk = qk;
    a.qe = qe;
    a.hw = hw;
    a.splits = L.splits;
    a.st = state;
    a.gmax = reinterpret_cast<float*>(base + L.off_gmax);
    a.thr = reinterpret_cast<const float*>(base + L.off_thr);
    a.cand = reinterpret_cast<uint64_t*>(base + L.off_cand);
    a.cand_cnt = reinterpret_cast<uint32_t*>(base + L.off_cnt);
    const int qg = pf_qg(hw);
    const dim3 grid((unsigned)ceil_div(hw, PF_QW * QT * qg), (unsigned)L.splits);
    if (qg == 2) {
      hipLaunchKernelGGL((affinity_pf_pass_kernel<0, 2>), grid, dim3(PF_QW * 64), 0, st, a);
    } else {
      hipLaunchKernelGGL((affinity_pf_pass_kernel<0, 1>), grid, dim3(PF_QW * 64), 0, st, a);
    }
    hipLaunchKernelGGL(affinity_pf_tau_kernel, dim3((unsigned)ceil_div(hw, 4)), dim3(256), 0, st, a.gmax,
                       reinterpret_cast<const float*>(base + L.off_eq), hw, k, L.splits,
                       reinterpret_cast<float*>(base + L.off_thr));
    if (qg == 2) {
      hipLaunchKernelGGL((affinity_pf_pass_kernel<1, 2>), grid, dim3(PF_QW * 64), 0, st, a);
    } else {
      hipLaunchKernelGGL((affinity_pf_pass_kernel<1, 1>), grid, dim3(PF_QW * 64), 0, st, a);
    }
    hipLaunchKernelGGL(affinity_pf_check_kernel, dim3((unsigned)ceil_div(hw, 256)), dim3(256), 0, st, a.cand_cnt, hw, k,
                       L.splits * 2, qbad, state);
    PfRescoreArgs r;
    r.bank = b;
    r.qk = qk;
    r.qe = qe;
    r.hw = hw;
    r.k = k;
    r.splits = L.splits;
    r.st = state;
    r.cand = a.cand;
    r.cand_cnt = a.cand_cnt;
    r.idx = idx;
    r.weight = weight;
    r.usage_fix = (unsigned long long*)usage_fix;
    r.out_keys = out_keys;
    r.out_cnt = out_counts;
    r.token_offset = (uint32_t)token_offset;
    hipLaunchKernelGGL(affinity_pf_rescore_kernel, dim3((unsigned)ceil_div(hw, 4)), dim3(256), 0, st, r);
    if (check_launch("deva_affinity_read (pre-filter)")) return 1;
    // fall-back: the fp32 kernels, which return at once while the flag is clear
    if (int rc = topk_fp32(key_long, shr_long, n_long, key_work, shr_work, n_work, qk, qe, hw, k, L.old_splits, part, stream,
                           &state->flag))
      return rc;
    const uint32_t* cnt = reinterpret_cast<const uint32_t*>(part + (int64_t)L.old_splits * hw * CAP);
    return launch_merge(part, cnt, hw, k, L.old_splits, idx, weight, usage_fix, out_keys, out_counts, (uint32_t)token_offset,
                        stream, "deva_affinity_read (fall-back)", &state->flag);
  }
  if (int rc = topk_fp32(key_long, shr_long, n_long, key_work, shr_work, n_work, qk, qe, hw, k, L.old_splits, part, stream,
                         nullptr))
    return rc;
  const uint32_t* cnt = reinterpret_cast<const uint32_t*>(part + (int64_t)L.old_splits * hw * CAP);
  return launch_merge(part, cnt, hw, k, L.old_splits, idx, weight, usage_fix, out_keys, out_counts, (uint32_t)token_offset,
                      stream, "deva_affinity_read");
}

// test hook: the fall-back flag of the last deva_affinity_read on this scratch (device -> host copy, synchronises)
extern "C" int deva_affinity_read_flag(const uint64_t* scratch, void* stream) {
  uint32_t flag = 0;
  if (hipMemcpyAsync(&flag, scratch, sizeof(flag), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return -1;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
  return (int)flag;
}

// test / tuning hook: candidate statistics of the last pre-filtered read on `scratch` (synchronises the stream):
// out[0] = fall-back flag, out[1] = largest sub-list count (capacity PF_SUB = 64 per (range, query, half-lane)), out[2] = largest
// number of candidates of one query, out[3] = mean candidates per query x 1000, out[4] = ranges (splits)
extern "C" int deva_affinity_read_stats(const uint64_t* scratch, int n_total, int hw, int k, int64_t* out, void* stream) {
  DEVA_REQUIRE(scratch && out && hw > 0, "deva_affinity_read_stats: bad args");
  const PfLayout L = pf_layout(n_total, hw, k);
  const uint8_t* base = reinterpret_cast<const uint8_t*>(scratch);
  const size_t n_cnt = (size_t)L.splits * hw * 2;
  uint32_t* host = (uint32_t*)malloc(n_cnt * sizeof(uint32_t));
  uint32_t flag = 0;
  if (!host) return 1;
  hipStream_t st = (hipStream_t)stream;
  bool ok = hipMemcpyAsync(&flag, base + L.off_state, 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
            hipMemcpyAsync(host, base + L.off_cnt, n_cnt * 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
            hipStreamSynchronize(st) == hipSuccess;
  if (ok) {
    uint32_t max_sub = 0, max_q = 0;
    uint64_t total = 0;
    for (int q = 0; q < hw; ++q) {
      uint32_t sum = 0;
      for (int s = 0; s < L.splits * 2; ++s) {
        const uint32_t c = host[(size_t)q * L.splits * 2 + s];
        max_sub = c > max_sub ? c : max_sub;
        sum += c;
      }
      max_q = sum > max_q ? sum : max_q;
      total += sum;
    }
    out[0] = flag;
    out[1] = max_sub;
    out[2] = max_q;
    out[3] = (int64_t)(total * 1000 / (uint64_t)hw);
    out[4] = L.splits;
  }
  free(host);
  return ok ? 0 : 1;
}
