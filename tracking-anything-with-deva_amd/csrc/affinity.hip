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
// (176 slots of 6 bytes): scores >= the running k-th best are appended (~k*(1+ln(n/k)) appends per
// query over n tokens), and a list that could overflow is pruned back to its exact best k by a
// ballot-driven bitwise bisection.  grid.y splits the bank into token ranges so small frames
// still fill 256 CUs; the ranges hand over their lists as they are and a second kernel (one wave
// per query) selects the exact top-k over all ranges, applies exp/normalise and accumulates the
// usage counters.
#include <math.h>

#include "common.h"

#pragma clang fp contract(off)

namespace deva {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CK = 64;
constexpr int QT = 32;            // queries per wave
constexpr int CAP = 128;          // candidate slots per (range, query) handed to the merge kernel
// in-kernel candidate lists: 176 slots of 6 bytes (order-preserving score bits + 16-bit token offset
// inside the range).  A range of up to ~1 000 tokens then never has to prune (k*(1+ln(n/k)) appends
// expected), and the merge kernel does the only exact selection.
constexpr int LCAP = 176;
constexpr int LSTRIDE = LCAP + 1;  // odd row stride: spreads the LDS banks
constexpr int MAX_SPLITS = 16;    // splits * CAP <= 2048 = 64 lanes x 32 keys in the merge kernel
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

#define DEVA_COMPILER_FENCE() asm volatile("" ::: "memory")

__device__ __forceinline__ int wave_count(bool pred) { return __popcll(__ballot(pred)); }
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

// prune one candidate list (wave-cooperative, c <= 192 entries, c >= k) to its exact best k
// (unsorted); returns the k-th best key (score bits << 32 | 0xffff - token offset)
__device__ __forceinline__ uint64_t prune_list(uint32_t* sc, uint16_t* tk, uint32_t c, int k,
                                               int lane) {
  uint64_t e[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const uint32_t j = (uint32_t)lane + 64u * i;
    e[i] = (j < c) ? (((uint64_t)sc[j] << 32) | (uint64_t)(0xffffu - tk[j])) : 0ull;
  }
  const uint64_t thr = kth_largest<3>(e, 3, k);
  DEVA_COMPILER_FENCE();
  int base = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const bool keep = e[i] >= thr && e[i] != 0ull;
    const unsigned long long b = __ballot(keep);
    if (keep) {
      const int w = base + prefix_below(b);
      sc[w] = (uint32_t)(e[i] >> 32);
      tk[w] = (uint16_t)(0xffffu - (uint32_t)(e[i] & 0xffffu));
    }
    base += __popcll(b);
  }
  DEVA_COMPILER_FENCE();
  return thr;
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
  int tiles_per_split;
  int total_tiles;
  uint64_t* part;
};

__global__ __launch_bounds__(WAVES * 64) void affinity_topk_kernel(const AffArgs p) {
  __shared__ uint32_t s_sc[WAVES][QT][LSTRIDE];  // candidate scores (order-preserving bits)
  __shared__ uint16_t s_tk[WAVES][QT][LSTRIDE];  // candidate tokens (offset inside this range)
  __shared__ uint32_t s_cnt[WAVES][QT];
  __shared__ float s_tau[WAVES][QT];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int l31 = lane & 31;
  const int half = lane >> 5;
  const int q0 = (blockIdx.x * WAVES + wave) * QT;
  if (q0 >= p.hw) return;  // whole wave idle (no block-level barrier is used in this kernel)
  const int split = blockIdx.y;

  // NB plain (non-volatile) LDS accesses: hipcc puts `s_waitcnt vmcnt(0)` next to every volatile
  // access, which would drain the key-row prefetch at each list operation.  Program order on may-alias
  // LDS locations plus the in-order LDS pipeline give the cross-lane visibility needed inside a wave;
  // DEVA_COMPILER_FENCE() marks the hand-over points.
  uint32_t* csc = &s_sc[wave][0][0];
  uint16_t* ctk = &s_tk[wave][0][0];
  uint32_t* cnt = &s_cnt[wave][0];
  float* tau = &s_tau[wave][0];

  if (lane < QT) {
    cnt[lane] = 0;
    tau[lane] = -INFINITY;
  }
  DEVA_COMPILER_FENCE();

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

  const int t_begin = split * p.tiles_per_split;
  const int t_end = min(p.total_tiles, t_begin + p.tiles_per_split);
  const int n_range0 = t_begin * TOKT;  // candidate tokens are stored as 16-bit offsets from here

  // key rows are software-prefetched one tile ahead: lane (l31, half) reads the whole 256-B row of
  // token n_base + l31 into xbuf while the matrix pipe works on the previous tile
  float4 xbuf[CK / 4];
  float ms_buf;
  auto prefetch = [&](int tile) {
    const int n_mine = min(tile * TOKT + l31, p.n_total - 1);
    const float* krow = (n_mine < p.n_long) ? (p.key_long + (int64_t)n_mine * CK)
                                            : (p.key_work + (int64_t)(n_mine - p.n_long) * CK);
    ms_buf = (n_mine < p.n_long) ? p.shr_long[n_mine] : p.shr_work[n_mine - p.n_long];
#pragma unroll
    for (int j = 0; j < CK / 4; ++j) xbuf[j] = reinterpret_cast<const float4*>(krow)[j];
  };
  if (t_begin < t_end) prefetch(t_begin);

  for (int tile = t_begin; tile < t_end; ++tile) {
    const int n_base = tile * TOKT;

    // ---- prune lists that could overflow during this tile (at most 32 appends per query per tile)
    {
      const uint32_t c_mine = cnt[l31];
      uint64_t need = __ballot(c_mine > (uint32_t)(LCAP - TOKT)) & 0xffffffffull;
      while (need) {
        const int qq = __ffsll((unsigned long long)need) - 1;
        need &= need - 1;
        const uint64_t thr = prune_list(csc + qq * LSTRIDE, ctk + qq * LSTRIDE, cnt[qq], p.k, lane);
        if (lane == 0) {
          cnt[qq] = (uint32_t)p.k;
          tau[qq] = from_orderable((uint32_t)(thr >> 32));
        }
        DEVA_COMPILER_FENCE();
      }
    }
    const float tau_l = tau[l31];

    // ---- this tile's operand: channel 2t + half of this lane's token, then start the next loads
    float a_op[CK / 2];
#pragma unroll
    for (int t = 0; t < CK / 2; ++t) {
      const float4 v = xbuf[t >> 1];
      const float lo = (t & 1) ? v.z : v.x;
      const float hi = (t & 1) ? v.w : v.y;
      a_op[t] = half ? hi : lo;
    }
    const float ms_mine = ms_buf;
    prefetch(min(tile + 1, t_end - 1));

    f32x16 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      accA[r] = 0.0f;
      accB[r] = 0.0f;
    }
#pragma unroll
    for (int t = 0; t < CK / 2; ++t) {
      const float a = a_op[t];
      accA = __builtin_amdgcn_mfma_f32_32x32x2f32(a * a, bqe[t], accA, 0, 0, 0);
      accB = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bqk[t], accB, 0, 0, 0);
    }

    // ---- scores of this lane: query l31, tokens n_base + (r&3) + 8*(r>>2) + 4*half
    uint32_t pass = 0;
    float sc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = (r & 3) + 8 * (r >> 2) + 4 * half;
      const float ms = __shfl(ms_mine, j);
      float v = (-accA[r] + 2.0f * accB[r]) - bsq;
      v = v * ms * 0.125f;
      sc[r] = v;
      const bool ok = (n_base + j < p.n_total) && (v >= tau_l);
      pass |= ok ? (1u << r) : 0u;
    }
    const int np = __popc(pass);
    if (np) {
      uint32_t pos = __hip_atomic_fetch_add((uint32_t*)&s_cnt[wave][l31], (uint32_t)np, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_WORKGROUP);
      uint32_t* srow = csc + l31 * LSTRIDE;
      uint16_t* trow = ctk + l31 * LSTRIDE;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (pass & (1u << r)) {
          const int j = (r & 3) + 8 * (r >> 2) + 4 * half;
          srow[pos] = orderable(sc[r]);
          trow[pos] = (uint16_t)(n_base - n_range0 + j);
          ++pos;
        }
      }
    }
    DEVA_COMPILER_FENCE();
  }

  // ---- the candidate lists of this range go to global memory as they are (zero padded to CAP keys per
  // query): the exact top-k selection over all ranges happens in the merge kernel, where one wave per
  // query gives thousands of independent waves -- here it would run serially, 32 lists per wave.
  // Only a list that outgrew the CAP hand-over slots is pruned first.
  {
    const uint32_t c_mine = cnt[l31];
    uint64_t need = __ballot(c_mine > (uint32_t)CAP) & 0xffffffffull;
    while (need) {
      const int qq = __ffsll((unsigned long long)need) - 1;
      need &= need - 1;
      prune_list(csc + qq * LSTRIDE, ctk + qq * LSTRIDE, cnt[qq], p.k, lane);
      if (lane == 0) cnt[qq] = (uint32_t)p.k;
      DEVA_COMPILER_FENCE();
    }
  }
  const int nq = min(QT, p.hw - q0);
  uint64_t* dst = p.part + ((int64_t)split * p.hw + q0) * CAP;
  for (int e = lane; e < nq * CAP; e += 64) {
    const int ql = e / CAP;
    const int r = e - ql * CAP;
    uint64_t key = 0ull;
    if ((uint32_t)r < cnt[ql]) {
      const uint32_t token = (uint32_t)n_range0 + (uint32_t)ctk[ql * LSTRIDE + r];
      key = ((uint64_t)csc[ql * LSTRIDE + r] << 32) | (uint64_t)(~token);
    }
    dst[e] = key;
  }
}

// one wave per query: exact top-k of the splits*CAP candidate slots (ME keys per lane), sorted by
// rank counting, then exp / normalise / usage
template <int ME>
__global__ __launch_bounds__(256) void affinity_finalize_kernel(const uint64_t* __restrict__ part, int hw, int k,
                                                                int splits, int32_t* __restrict__ idx,
                                                                float* __restrict__ weight,
                                                                unsigned long long* __restrict__ usage_fix) {
  __shared__ uint64_t s_buf[4][2][64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + wave;
  if (q >= hw) return;
  volatile uint64_t* unsorted = &s_buf[wave][0][0];
  volatile uint64_t* sorted = &s_buf[wave][1][0];

  const int total = splits * CAP;
  const int n_live = (total + 63) >> 6;
  uint64_t e[ME];
#pragma unroll
  for (int i = 0; i < ME; ++i) {
    const int c = lane + 64 * i;
    uint64_t v = 0ull;
    if (c < total) {
      const int sp = c / CAP;
      v = part[((int64_t)sp * hw + q) * CAP + (c - sp * CAP)];
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
      const unsigned long long b = __ballot(keep);
      if (keep) unsorted[base + prefix_below(b)] = e[i];
      base += __popcll(b);
    }
  }
  DEVA_COMPILER_FENCE();
  const bool live = lane < k;
  const uint64_t cand = live ? unsorted[lane] : 0ull;
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
                                                             float* __restrict__ out) {
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
    if (q < hw && c_ok) {
      for (int j = 0; j < k; ++j) {
        const int t = idx[(int64_t)q * k + j];
        const float w = weight[(int64_t)q * k + j];
        const float* row = (t < n_long) ? (val_long + (int64_t)t * cv) : (val_work + (int64_t)(t - n_long) * cv);
        const float4 v = *reinterpret_cast<const float4*>(row + c0 + cl);
        acc.x += w * v.x;
        acc.y += w * v.y;
        acc.z += w * v.z;
        acc.w += w * v.w;
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

}  // namespace
}  // namespace deva

using namespace deva;

extern "C" int64_t deva_affinity_workspace(int hw, int k, int splits) {
  (void)k;
  return (int64_t)splits * hw * CAP;  // one zero-padded candidate list per (range, query)
}

extern "C" int deva_affinity_default_splits(int n_total, int hw) {
  // one 4-wave workgroup per CU is resident (133 KB of candidate lists): aim at ~256 workgroups
  const int qblocks = (int)ceil_div(hw, WAVES * QT);
  const int tiles = (int)ceil_div(n_total, TOKT);
  int s = (int)ceil_div(256, qblocks);
  if (s > tiles / 4) s = tiles / 4;  // keep >= 4 tiles (128 tokens) per range
  if (s > MAX_SPLITS) s = MAX_SPLITS;
  if (s < 1) s = 1;
  // small banks (first memory frames of a clip): ranges of <= CAP tokens hand every score over without
  // building a threshold or pruning (a prune round of 32 lists costs ~50 us, measured)
  const int s_nofilter = (int)ceil_div(tiles, CAP / TOKT);
  if (s_nofilter <= MAX_SPLITS && s_nofilter > s) s = s_nofilter;
  while (s < MAX_SPLITS && ceil_div(tiles, s) > 2047) ++s;  // 16-bit token offsets inside a range
  return s;
}

extern "C" int deva_affinity_topk(const float* key_long, const float* shr_long, int n_long, const float* key_work,
                                  const float* shr_work, int n_work, const float* qk, const float* qe, int hw,
                                  int k, int splits, uint64_t* part_keys, void* stream) {
  DEVA_REQUIRE(qk && qe && part_keys && hw > 0, "deva_affinity_topk: bad query args");
  DEVA_REQUIRE(n_long >= 0 && n_work >= 0, "deva_affinity_topk: negative bank size");
  DEVA_REQUIRE(n_long == 0 || (key_long && shr_long), "deva_affinity_topk: null long-term segment");
  DEVA_REQUIRE(n_work == 0 || (key_work && shr_work), "deva_affinity_topk: null working segment");
  DEVA_REQUIRE(k >= 1 && k <= 32, "deva_affinity_topk: k=%d unsupported (1..32)", k);
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
  a.tiles_per_split = (int)ceil_div(a.total_tiles, splits);
  DEVA_REQUIRE(a.tiles_per_split <= 2047,
               "deva_affinity_topk: %d tokens per range exceed the 16-bit in-range token offset; use more splits",
               a.tiles_per_split * TOKT);
  a.part = part_keys;
  dim3 grid((unsigned)ceil_div(hw, WAVES * QT), (unsigned)splits);
  hipLaunchKernelGGL(affinity_topk_kernel, grid, dim3(WAVES * 64), 0, (hipStream_t)stream, a);
  return check_launch("deva_affinity_topk");
}

extern "C" int deva_affinity_finalize(const uint64_t* part_keys, int hw, int k, int splits, int32_t* idx,
                                      float* weight, uint64_t* usage_fix, void* stream) {
  DEVA_REQUIRE(part_keys && idx && weight && hw > 0, "deva_affinity_finalize: bad args");
  DEVA_REQUIRE(k >= 1 && k <= 32 && splits >= 1 && splits <= MAX_SPLITS,
               "deva_affinity_finalize: k/splits out of range");
  if (splits * CAP <= 64 * 8) {
    hipLaunchKernelGGL(affinity_finalize_kernel<8>, dim3((unsigned)ceil_div(hw, 4)), dim3(256), 0,
                       (hipStream_t)stream, part_keys, hw, k, splits, idx, weight, (unsigned long long*)usage_fix);
  } else {
    hipLaunchKernelGGL(affinity_finalize_kernel<32>, dim3((unsigned)ceil_div(hw, 4)), dim3(256), 0,
                       (hipStream_t)stream, part_keys, hw, k, splits, idx, weight, (unsigned long long*)usage_fix);
  }
  return check_launch("deva_affinity_finalize");
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
                                   int n_long, const float* val_work, int cv, float* out, void* stream) {
  DEVA_REQUIRE(idx && weight && out && hw > 0 && k > 0 && cv > 0, "deva_readout_sparse: bad args");
  DEVA_REQUIRE(cv % 4 == 0, "deva_readout_sparse: value dim must be a multiple of 4");
  DEVA_REQUIRE(n_long == 0 || val_long, "deva_readout_sparse: null long-term values");
  const float* vl = val_long ? val_long : val_work;
  const float* vw = val_work ? val_work : val_long;
  DEVA_REQUIRE(vl && vw, "deva_readout_sparse: no value segment");
  dim3 grid((unsigned)ceil_div(hw, RQ), (unsigned)ceil_div(cv, RC));
  hipLaunchKernelGGL(readout_sparse_kernel, grid, dim3(256), 0, (hipStream_t)stream, idx, weight, hw, k, vl, n_long,
                     vw, cv, out);
  return check_launch("deva_readout_sparse");
}
