// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
//   out[b][m][oh][ow] = act( sum_k Wp[k][m] * X(k; b,oh,ow) + bias[m] + residual[b][m][oh][ow] )
//
// GEMM view:  M = cout, N = batch*OH*OW (pixels, batch-major), K = KH*KW*(C0+C1), k = tap*Ctot + c.
// A = packed weights, k-major [K][cout_pad]  (cout contiguous)  -> float4 global loads along M
// B = im2col of the NCHW input, gathered on the fly; for a fixed k consecutive pixels are
//     consecutive addresses (stride-1 convs), so the scalar gathers of a wave are coalesced.
//
// fp32-in MFMA runs at the fp32 vector rate (256 FLOP/clk/CU), i.e. a 128x128 block tile needs
// only ~8 B/clk/CU of operand traffic: the kernel is matrix-pipe bound, so the staging path is
// kept simple (register-staged, double-buffered LDS, one barrier per K step) and exact fp32.
//
// Each of the 4 waves owns a (WM x WN) sub-tile made of 32x32 MFMA tiles.  Operand fragments for
// v_mfma_f32_32x32x2_f32:  A: lane l holds A[i = l&31][k = l>>5],  B: B[k = l>>5][j = l&31],
// D: reg r of lane l is D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
#include "conv_args.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace deva {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4_u __attribute__((aligned(4)));  // 4 consecutive pixels, dword-aligned only



// MODE 0: 1x1 kernel with c0 a multiple of BK (source uniform per K step, K tail allowed);
// MODE 1: k x k kernel with c0 and c0+c1 multiples of BK (tap and source uniform per K step);
// MODE 2: anything (per-element decode: the 2/3/4-channel stems, odd channel splits).
// ROW (3x3, pad 1, 32-channel-slab weights, VEC): the input rows of a (slab, dy) pair are staged ONCE,
// unmasked and with one halo pixel either side, and serve the three dx taps with a column offset on
// the LDS fragment read; the zero padding is applied per consumer pixel at that read.
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int MODE, int SPREAD, int VEC, int ROW>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (WAVES_M * WAVES_N == 8) ? 4 : 1) void conv_igemm_kernel(
    const ConvArgs p) {
  constexpr int THREADS = 64 * WAVES_M * WAVES_N;
  static_assert(WAVES_M * WAVES_N == 4 || WAVES_M * WAVES_N == 8, "4 or 8 waves per block");
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile");
  constexpr int A_V4 = (BK * BM / 4 + THREADS - 1) / THREADS;  // float4 loads per thread
  constexpr int KG = THREADS / BN;                              // k rows covered per pass
  constexpr int B_PT = BK / KG;                                 // scalar gathers per thread

  __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
  constexpr int BNP = ROW ? BN + 8 : BN;  // ROW: columns 3 .. BN+4 hold pixels n0-1 .. n0+BN
  static_assert(ROW == 0 || (VEC == 1 && MODE == 1 && TN == 1), "row reuse builds on the vector gather");
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][BNP];
  constexpr int KTAB = 1024;  // MODE 2: k -> (channel | dy << 16 | dx << 24)
  __shared__ unsigned s_ktab[MODE == 2 ? KTAB : 1];

  if (p.gate && *p.gate == 0) return;  // the fp32 re-run behind a split launch (conv_f16.hip) that raised no flag
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm0 = (wave / WAVES_N) * WM;
  const int wn0 = (wave % WAVES_N) * WN;
  const int l31 = lane & 31;
  const int half = lane >> 5;
  const bool ktab_ok = MODE == 2 && p.K <= KTAB && p.ctot < 65536 && p.KH < 256 && p.KW < 256;
  if (MODE == 2 && ktab_ok) {
    for (int k = tid; k < p.K; k += THREADS) {
      const int tap = k / p.ctot;
      const int dy = tap / p.KW;
      s_ktab[k] = (unsigned)(k - tap * p.ctot) | ((unsigned)dy << 16) | ((unsigned)(tap - dy * p.KW) << 24);
    }
    __syncthreads();
  }

  // Tile order: cout tiles fastest (the workgroups sharing one pixel tile run together), and an
  // XCD-aware remap -- hardware places workgroup b on XCD b % 8, so XCD x gets a CONTIGUOUS range
  // of logical tiles and neighbouring pixel tiles (shared halo rows, shared B tile) meet in one L2.
  int logical;
  {
    const int nb = gridDim.x, b = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = b & 7;
    logical = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int tile_n = logical / p.tiles_m;
  const int tile_m = logical - tile_n * p.tiles_m;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  // ---- per-thread constants of the B gather: this thread always gathers pixel n0 + (tid % BN)
  const int bn_local = tid % BN;
  const int bk_group = tid / BN;
  const int n_g = n0 + bn_local;
  const bool n_ok = n_g < p.n_total;
  int ih0, iw0;
  const float* src0;
  const float* src1;
  {
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

  // ---- vector gather (VEC): stride-1 "same" convolutions read 4 consecutive pixels of one channel
  // with a single dword-aligned 16-B load (flat input pixel = flat output pixel + tap offset); pixels
  // that fall into the zero padding are masked when the tile is written to LDS.  Needs readable
  // guard bands of pad*(W+1)+4 floats around the inputs (checked by the host, see deva_hip.h).
  constexpr int NQ = BN / 4;          // pixel quads per tile row
  constexpr int KGV = THREADS / NQ;   // K rows covered per pass
  constexpr int B_V4 = BK / KGV;      // 16-B gathers per thread
  static_assert(VEC == 0 || (B_V4 >= 1 && B_V4 * KGV == BK), "vector gather geometry");
  const int vq = tid % NQ, vk = tid / NQ;
  bool vq_ok = false;
  int v_oh[4], v_ow[4];
  const float* vsrc0 = nullptr;
  const float* vsrc1 = nullptr;
  int v_pix0 = 0;
  if (VEC) {
    const int n4 = n0 + 4 * vq;
    vq_ok = n4 < p.n_total;  // OHW % 4 == 0: a quad never straddles images or the end
    const int nn = vq_ok ? n4 : 0;
    const int b = nn / p.OHW;
    v_pix0 = nn - b * p.OHW;
    const int oh = v_pix0 / p.OW, ow = v_pix0 - oh * p.OW;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool wrap = ow + j >= p.OW;  // OW >= 4: at most one row wrap inside a quad
      v_oh[j] = oh + (wrap ? 1 : 0);
      v_ow[j] = ow + j - (wrap ? p.OW : 0);
    }
    vsrc0 = p.in0 + (int64_t)b * p.bs0;
    vsrc1 = p.in1 ? p.in1 + (int64_t)b * p.bs1 : p.in0;
  }
  f32x4 rbv[VEC ? B_V4 : 1];
  unsigned v_mask = 0;

  // ---- ROW: halo pixel (n0-1 or n0+BN) of K row h_row, loaded by every thread (kept branch-free),
  // stored by wave 0; 9-bit validity mask (bit dy*3+dx) of the pixel this lane consumes
  const int h_side = tid & 1, h_row = (tid >> 1) & (BK - 1);
  const float* hsrc0 = nullptr;
  const float* hsrc1 = nullptr;
  int h_pix = 0;
  unsigned cmask = 0;
  float rh = 0.0f;
  const float* st_hptr = nullptr;
  if (ROW) {
    int nh = h_side ? n0 + BN : n0 - 1;
    nh = min(max(nh, 0), p.n_total - 1);
    const int b = nh / p.OHW;
    h_pix = nh - b * p.OHW;
    hsrc0 = p.in0 + (int64_t)b * p.bs0;
    hsrc1 = p.in1 ? p.in1 + (int64_t)b * p.bs1 : p.in0;
    const int n = n0 + wn0 + l31;
    if (n < p.n_total) {
      const int pix = n % p.OHW;
      const int oh = pix / p.OW, ow = pix - oh * p.OW;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const bool ok = ((unsigned)(oh + t / 3 - 1) < (unsigned)p.H) && ((unsigned)(ow + t % 3 - 1) < (unsigned)p.W);
        cmask |= ok ? (1u << t) : 0u;
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

  // staged operands: raw loaded values + validity bits; select/relu happen in store_tiles, AFTER
  // the MFMA block, so the loads stay in flight while the matrix pipe works on the current tile
  float4 ra[A_V4];
  float rb[B_PT];
  unsigned ok_a = 0, ok_b = 0;

  // ---- staging of one K step, split into pieces so the main loop can interleave them with the
  // MFMAs of the current step (an MFMA occupies the matrix pipe for 64 cycles while the wave keeps
  // issuing independent VALU / memory instructions in its shadow)
  const float* st_ptr = nullptr;  // fast path: per-step gather base of this thread
  bool st_okp = false;
  int st_k0 = 0;

  auto stage_begin = [&](int k0) {
    st_k0 = k0;
    ok_a = 0;
    ok_b = 0;
    if (MODE != 2) {
      // the tap and the source tensor are uniform over the K step (channel counts are multiples
      // of BK): one pointer per step + a channel stride per element
      int tap = 0, cbase = k0;
      if (MODE == 1) {
        if (p.k_layout == DEVA_KLAYOUT_CHUNK32) {  // k = ((c/32)*taps + tap)*32 + c%32, BK == 32
          const int slab = k0 / BK, taps = p.KH * p.KW;
          const int chunk = slab / taps;
          tap = slab - chunk * taps;
          cbase = chunk * BK;
        } else {  // k = tap*ctot + c
          tap = k0 / p.ctot;
          cbase = k0 - tap * p.ctot;
        }
      }
      int ih = ih0, iw = iw0;
      if (MODE == 1) {
        const int dy = tap / p.KW;
        ih += dy;
        iw += tap - dy * p.KW;
      }
      const bool first = cbase < p.c0;
      if (ROW) {
        // B comes from the row tile staged by stage_begin_row
      } else if (VEC) {
        const int dy = ih - ih0 - p.pad, dx = iw - iw0 - p.pad;  // tap offset relative to the centre
        v_mask = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = vq_ok && ((unsigned)(v_oh[j] + dy) < (unsigned)p.H) &&
                          ((unsigned)(v_ow[j] + dx) < (unsigned)p.W);
          v_mask |= ok ? (1u << j) : 0u;
        }
        st_ptr = (first ? (vsrc0 + (int64_t)cbase * p.HW) : (vsrc1 + (int64_t)(cbase - p.c0) * p.HW)) +
                 (v_pix0 + dy * p.W + dx);
      } else {
        st_okp = n_ok && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
        st_ptr = (first ? (src0 + (int64_t)cbase * p.HW) : (src1 + (int64_t)(cbase - p.c0) * p.HW)) +
                 (st_okp ? (ih * p.W + iw) : 0);
      }
    }
  };

  // ROW: pointers of the (slab, dy) row tile that starts at K step kg (its dx == 0 step)
  auto stage_begin_row = [&](int kg) {
    const int slab = kg / BK;
    const int chunk = slab / 9;
    const int dy = (slab - chunk * 9) / 3;
    const int cbase = chunk * BK;
    const bool first = cbase < p.c0;
    const int64_t coff = (int64_t)(first ? cbase : cbase - p.c0) * p.HW;
    const int shift = (dy - 1) * p.W;
    st_ptr = (first ? vsrc0 : vsrc1) + coff + (v_pix0 + shift);
    st_hptr = (first ? hsrc0 : hsrc1) + coff + (int64_t)h_row * p.HW + (h_pix + shift);
    ok_b = 0;
  };

  // A: rows k0..k0+BK-1 of the packed weights, columns m0..m0+BM-1.  Loads are unconditional from a
  // clamped (always valid) address, the select happens when the tile is written to LDS: a load under
  // a branch makes hipcc wait vmcnt(0) right behind it and serialises the staging.
  auto stage_a = [&](int i) {
    const int e = tid + i * THREADS;
    const int kr = e / (BM / 4);
    const int mc = (e % (BM / 4)) * 4;
    const int k = st_k0 + kr;
    const int m = m0 + mc;
    const bool ok = (e < BK * BM / 4) && (k < p.K) && (m < p.cout_pad);
    ra[i] = *reinterpret_cast<const float4*>(p.w + (ok ? ((int64_t)k * p.cout_pad + m) : 0));
    ok_a |= ok ? (1u << i) : 0u;
  };

  // B: im2col gather of element i of this thread
  auto stage_bv = [&](int i) {  // VEC: 4 consecutive pixels of K row vk + i*KGV
    const int ci = vk + i * KGV;
    const bool kin = (MODE == 1) || (st_k0 + ci < p.K);
    rbv[i] = *reinterpret_cast<const f32x4_u*>(st_ptr + (kin ? (int64_t)ci * p.HW : 0));
    ok_b |= kin ? (1u << i) : 0u;
  };

  auto stage_b = [&](int i) {
    const int ci = bk_group + i * KG;  // row within this K step
    if (MODE == 2) {
      // generic per-element decode of k -> (tap, channel, source): 2/3/4-channel stems, odd splits.
      // The two integer divisions come from a table built once per workgroup (the staging of these
      // layers is VALU-bound); layers with more than KTAB k values divide.
      const int k = st_k0 + ci;
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
      ok_b |= ok ? (1u << i) : 0u;
    } else {
      const bool kin = (MODE == 1) || (st_k0 + ci < p.K);    // MODE 0 may have a K tail (513 channels)
      rb[i] = st_ptr[kin ? (int64_t)ci * p.HW : 0];           // clamped: never reads past the source
      ok_b |= (st_okp && kin) ? (1u << i) : 0u;
    }
  };

  auto store_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_V4; ++i) {
      const int e = tid + i * THREADS;
      if (e < BK * BM / 4) {
        const int kr = e / (BM / 4);
        const int mc = (e % (BM / 4)) * 4;
        *reinterpret_cast<float4*>(&As[buf][kr][mc]) =
            (ok_a & (1u << i)) ? ra[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto store_b = [&](int buf) {
    if (ROW) {
#pragma unroll
      for (int i = 0; i < B_V4; ++i) {
        f32x4 v = rbv[i];
        if (p.relu_in) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
        }
        *reinterpret_cast<f32x4*>(&Bs[buf][vk + i * KGV][4 + 4 * vq]) = v;
      }
      if (tid < 64) Bs[buf][h_row][h_side ? BN + 4 : 3] = p.relu_in ? fmaxf(rh, 0.0f) : rh;
      return;
    }
    if (VEC) {
#pragma unroll
      for (int i = 0; i < B_V4; ++i) {
        f32x4 v = rbv[i];
        const bool kin = ok_b & (1u << i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x = v[j];
          if (p.relu_in) x = fmaxf(x, 0.0f);
          v[j] = (kin && (v_mask & (1u << j))) ? x : 0.0f;
        }
        *reinterpret_cast<f32x4*>(&Bs[buf][vk + i * KGV][4 * vq]) = v;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < B_PT; ++i) {
      float v = rb[i];
      if (p.relu_in) v = fmaxf(v, 0.0f);
      Bs[buf][bk_group + i * KG][bn_local] = (ok_b & (1u << i)) ? v : 0.0f;
    }
  };
  auto store_tiles = [&](int buf) {
    store_a(buf);
    store_b(buf);
  };

  constexpr int NKK = BK / 2;  // MFMA groups (k pairs) per K step
  // split-K: this workgroup accumulates K steps [ks0, ks0 + ksteps) of the layer
  int ks0 = 0, ksteps = (p.K + BK - 1) / BK;
  if (p.splits > 1) {
    const int per = p.per_split;
    ks0 = (int)blockIdx.y * per;
    ksteps = max(0, min(ksteps - ks0, per));
  }
  stage_begin(ks0 * BK);
#pragma unroll
  for (int i = 0; i < A_V4; ++i) stage_a(i);
  if (ROW) {
    stage_begin_row(ks0 * BK);
#pragma unroll
    for (int i = 0; i < B_V4; ++i) stage_bv(i);
    rh = *st_hptr;
  } else if (VEC) {
#pragma unroll
    for (int i = 0; i < B_V4; ++i) stage_bv(i);
  } else {
#pragma unroll
    for (int i = 0; i < B_PT; ++i) stage_b(i);
  }
  store_tiles(0);
  __syncthreads();

  if (ROW) {
    // K steps come in (slab, dy) groups of three dx taps (splits start on group boundaries); the loop
    // is unrolled over dx so that the row-tile loads of the NEXT group sit in straight-line code of the
    // dx == 0 step (a load under a branch would be followed by s_waitcnt vmcnt(0))
    auto row_step = [&](int s, auto dxc) {
      constexpr int DX = decltype(dxc)::value;
      const int buf = s & 1;
      const int gbuf = (s / 3) & 1;
      const int tap = (ks0 + s) % 9;
      const bool okc = (cmask >> tap) & 1u;
      const int k_next = (s + 1 < ksteps) ? (ks0 + s + 1) * BK : 0;
      const int kg_next = (s + 3 < ksteps) ? (ks0 + s + 3) * BK : 0;
      const int col = wn0 + l31 + 3 + DX;
      float fa[2][TM], fb[2];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[0][i] = As[buf][half][wm0 + i * 32 + l31];
      fb[0] = Bs[gbuf][half][col];
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        if (kk + 1 < NKK) {  // fragments of the next k pair, in flight during this pair's MFMAs
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[(kk + 1) & 1][i] = As[buf][2 * (kk + 1) + half][wm0 + i * 32 + l31];
          fb[(kk + 1) & 1] = Bs[gbuf][2 * (kk + 1) + half][col];
        }
        const float bval = okc ? fb[kk & 1] : 0.0f;  // zero padding, per consumer pixel and tap
#pragma unroll
        for (int i = 0; i < TM; ++i)
          acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][i], bval, acc[i][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (kk == 0) {
          stage_begin(k_next);
          if (DX == 0) stage_begin_row(kg_next);
        }
#pragma unroll
        for (int i = 0; i < A_V4; ++i)
          if (i * (NKK / SPREAD) / A_V4 == kk) stage_a(i);
        if (DX == 0) {
#pragma unroll
          for (int i = 0; i < B_V4; ++i)
            if (i * (NKK / SPREAD) / B_V4 == kk) stage_bv(i);
          if (kk == 1) rh = *st_hptr;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      store_a(buf ^ 1);
      if (DX == 0) store_b(gbuf ^ 1);
      __syncthreads();
    };
    for (int s = 0; s < ksteps; s += 3) {
      row_step(s, std::integral_constant<int, 0>{});
      row_step(s + 1, std::integral_constant<int, 1>{});
      row_step(s + 2, std::integral_constant<int, 2>{});
    }
  } else {
  for (int s = 0; s < ksteps; ++s) {
    const int buf = s & 1;
    // the last step re-stages step 0 (valid addresses, result unused) instead of branching
    const int k_next = (s + 1 < ksteps) ? (ks0 + s + 1) * BK : 0;
    float fa[2][TM], fb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = As[buf][half][wm0 + i * 32 + l31];
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = Bs[buf][half][wn0 + j * 32 + l31];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      if (kk + 1 < NKK) {  // fragments of the next k pair, in flight during this pair's MFMAs
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[(kk + 1) & 1][i] = As[buf][2 * (kk + 1) + half][wm0 + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[(kk + 1) & 1][j] = Bs[buf][2 * (kk + 1) + half][wn0 + j * 32 + l31];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][i], fb[kk & 1][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // a slice of the next tile's staging in the shadow of the MFMAs just issued
      if (kk == 0) stage_begin(k_next);
#pragma unroll
      for (int i = 0; i < A_V4; ++i)
        if (i * (NKK / SPREAD) / A_V4 == kk) stage_a(i);
      if (VEC) {
#pragma unroll
        for (int i = 0; i < B_V4; ++i)
          if (i * (NKK / SPREAD) / B_V4 == kk) stage_bv(i);
      } else {
#pragma unroll
        for (int i = 0; i < B_PT; ++i)
          if (i * (NKK / SPREAD) / B_PT == kk) stage_b(i);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    store_tiles(buf ^ 1);
    __syncthreads();
  }
  }

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

// out[b][m][pix] = act(sum_s ws[s][m][n] + bias[m] + residual), n = b*OHW + pix
__global__ void splitk_reduce_kernel(const ConvArgs p) {
  if (p.gate && *p.gate == 0) return;  // the fp32 re-run behind a split launch that stayed inside the fp16 range
  const int64_t total = (int64_t)p.cout * p.n_total;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / p.n_total);
    const int n = (int)(i - (int64_t)m * p.n_total);
    float v = 0.0f;
    for (int s = 0; s < p.splits; ++s) v += p.ws[(int64_t)s * total + i];
    const int b = n / p.OHW;
    const int pix = n - b * p.OHW;
    if (p.bias) v += p.bias[m];
    if (p.res) v += p.res[(int64_t)b * p.res_bs + (int64_t)m * p.OHW + pix];
    if (p.act == DEVA_ACT_RELU) {
      v = fmaxf(v, 0.0f);
    } else if (p.act == DEVA_ACT_SIGMOID) {
      v = sigmoidf_(v);
    } else if (p.act == DEVA_ACT_SQUARE_PLUS_ONE) {
      v = v * v + 1.0f;
    }
    p.out[((int64_t)b * p.cout + m) * p.OHW + pix] = v;
  }
}

// SPREAD: the staging of the next K step is issued during the first 1/SPREAD of the MFMA groups, the
// rest of the step is slack for the loads to land before the LDS write.
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int SPREAD = 2>
int launch_tile(const ConvArgs& a, hipStream_t st) {
  ConvArgs p = a;
  int mode;
  if (a.KH == 1 && a.KW == 1 && a.c0 % BK == 0) {
    mode = 0;
  } else if (a.ctot % BK == 0 && a.c0 % BK == 0) {
    mode = 1;
  } else {
    mode = 2;
  }
  static_assert(BK == 32, "the 32-channel-slab K layout assumes 32-deep K steps");
  p.tiles_m = (int)ceil_div(a.cout, BM);
  p.tiles_n = (int)ceil_div(a.n_total, BN);
  if (a.k_layout == DEVA_KLAYOUT_CHUNK32 && mode != 1) {
    set_error("deva_conv2d: 32-channel-slab weights need c0 and c1 to be multiples of 32");
    return 2;
  }
  // split-K when the layer has too few tiles to fill the 256 CUs (small frames / single objects):
  // partial sums go to the caller's workspace, a second kernel reduces them deterministically
  const int ksteps_total = (int)ceil_div(a.K, BK);
  const bool row = a.vec_ok && mode == 1 && a.KH == 3 && a.KW == 3 && a.pad == 1 &&
                   a.k_layout == DEVA_KLAYOUT_CHUNK32 && BN / WAVES_N == 32;
  p.per_split = ksteps_total;
  const int64_t blocks = (int64_t)p.tiles_m * p.tiles_n;
  p.splits = 1;
  int64_t target_blocks = 512;
#ifdef DEVA_CONV_PROBES
  {
    static const int forced = [] {
      const char* e = getenv("DEVA_CONV_SPLIT_TARGET");  // 0 = no split-K at all
      return e ? atoi(e) : -1;
    }();
    if (forced >= 0) target_blocks = forced;
  }
#endif
  // measured on the batch-1 key-encoder layers (profiles/r02d_conv_small_layers.txt): the reduction launch costs
  // ~7 us, so a layer that already has >= 128 tiles is split only when its K loop is long (>= 32 steps)
  if (a.ws && blocks < 256 && ksteps_total >= (blocks >= 128 ? 32 : 8) && target_blocks > 0) {
    int64_t sp = ceil_div(target_blocks, blocks);
    if (sp > ksteps_total / 4) sp = ksteps_total / 4;
    if (sp > 16) sp = 16;
    const int64_t fit = a.ws_elems / ((int64_t)a.cout * a.n_total);
    if (sp > fit) sp = fit;
    if (sp >= 2) {
      // every split must own at least one K step (ROW: whole groups of three)
      int per = (int)ceil_div(ksteps_total, sp);
      if (row) per = (per + 2) / 3 * 3;
      sp = ceil_div(ksteps_total, per);
      p.splits = (int)sp;
      p.per_split = per;
    }
    if (p.splits < 2) p.splits = 1;
  }
  dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)p.splits);
  const bool vec = a.vec_ok && mode != 2;
  if (mode == 0) {
    if (vec) {
      hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, 0, SPREAD, 1, 0>), grid, dim3(64 * WAVES_M * WAVES_N), 0, st, p);
    } else {
      hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, 0, SPREAD, 0, 0>), grid, dim3(64 * WAVES_M * WAVES_N), 0, st, p);
    }
  } else if (mode == 1) {
    if (row) {
      if constexpr (BN / WAVES_N == 32) {
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, 1, SPREAD, 1, 1>), grid, dim3(64 * WAVES_M * WAVES_N), 0, st, p);
      }
    } else if (vec) {
      hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, 1, SPREAD, 1, 0>), grid, dim3(64 * WAVES_M * WAVES_N), 0, st, p);
    } else {
      hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, 1, SPREAD, 0, 0>), grid, dim3(64 * WAVES_M * WAVES_N), 0, st, p);
    }
  } else {
    {
      hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, 2, SPREAD, 0, 0>), grid, dim3(64 * WAVES_M * WAVES_N), 0, st, p);
    }
  }
  if (p.splits > 1) return launch_splitk_reduce(p, st);
  return check_launch("deva_conv2d");
}

}  // namespace

int launch_splitk_reduce(const ConvArgs& p, hipStream_t st) {
  const int64_t total = (int64_t)p.cout * p.n_total;
  int64_t rb = ceil_div(total, 256);
  if (rb > 4096) rb = 4096;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, st, p);
  return check_launch("deva_conv2d");
}

}  // namespace deva

extern "C" int deva_conv2d(const deva_conv_desc* d, void* stream) {
  using namespace deva;
  DEVA_REQUIRE(d != nullptr, "deva_conv2d: null descriptor");
  DEVA_REQUIRE(d->in0 && d->weight && d->out, "deva_conv2d: null tensor");
  DEVA_REQUIRE(d->c0 > 0 && d->c1 >= 0 && (d->c1 == 0 || d->in1), "deva_conv2d: bad channel split");
  DEVA_REQUIRE(d->batch > 0 && d->height > 0 && d->width > 0 && d->cout > 0, "deva_conv2d: bad shape");
  DEVA_REQUIRE(d->cout_pad % 32 == 0 && d->cout_pad >= d->cout, "deva_conv2d: cout_pad must be cout rounded up to 32");
  DEVA_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->pad >= 0, "deva_conv2d: bad kernel geometry");
  DEVA_REQUIRE((d->k_layout & ~DEVA_KLAYOUT_Q4) == DEVA_KLAYOUT_TAP_MAJOR || (d->k_layout & ~DEVA_KLAYOUT_Q4) == DEVA_KLAYOUT_CHUNK32,
               "deva_conv2d: unknown k_layout %d", d->k_layout);
  // the single-channel heads (conv_cout1.hip) read column 0 of [K][cout_pad]: deva_conv_pack never interleaves them
  DEVA_REQUIRE(!(d->cout == 1 && (d->k_layout & DEVA_KLAYOUT_Q4)), "deva_conv2d: k-quad interleaved weights need cout > 1");
  DEVA_REQUIRE(d->amp >= 0 && d->amp <= 2, "deva_conv2d: amp must be 0 (fp32), 1 (fp16 operands) or 2 (fp16 hi/lo split)");
  DEVA_REQUIRE(d->amp != 2 || !d->weight_f16 || d->split_flag, "deva_conv2d: the split path needs a device flag (split_flag)");
  {
    // the vector-gather kernels address their inputs with 32-bit byte offsets from the tensor bases: a batch whose inputs
    // span 2 GiB or more (many objects at 4K) runs as consecutive sub-batches
    const int64_t hw = (int64_t)d->height * d->width;
    const int64_t lim = (1ll << 29) - 1;
    const int64_t span0 = (int64_t)(d->batch - 1) * d->in0_batch_stride + (int64_t)d->c0 * hw;
    const int64_t span1 = d->c1 ? (int64_t)(d->batch - 1) * d->in1_batch_stride + (int64_t)d->c1 * hw : 0;
    if ((span0 > lim || span1 > lim) && d->batch > 1) {
      int64_t per = d->batch;
      if (d->in0_batch_stride > 0) per = std::min(per, (lim - (int64_t)d->c0 * hw) / d->in0_batch_stride + 1);
      if (d->c1 && d->in1_batch_stride > 0) per = std::min(per, (lim - (int64_t)d->c1 * hw) / d->in1_batch_stride + 1);
      if (per >= 1 && per < d->batch) {
        const int64_t oh = (d->height + 2 * d->pad - d->kh) / d->stride + 1, ow = (d->width + 2 * d->pad - d->kw) / d->stride + 1;
        for (int64_t b0 = 0; b0 < d->batch; b0 += per) {
          deva_conv_desc sub = *d;
          sub.batch = (int32_t)std::min<int64_t>(per, d->batch - b0);
          sub.in0 = d->in0 + b0 * d->in0_batch_stride;
          if (d->in1) sub.in1 = d->in1 + b0 * d->in1_batch_stride;
          if (d->residual) sub.residual = d->residual + b0 * d->residual_batch_stride;
          sub.out = d->out + b0 * (int64_t)d->cout * oh * ow;
          const int rc = deva_conv2d(&sub, stream);
          if (rc) return rc;
        }
        return 0;
      }
    }
  }
  ConvArgs a;
  a.in0 = d->in0;
  a.in1 = d->c1 ? d->in1 : nullptr;
  a.bs0 = d->in0_batch_stride;
  a.bs1 = d->in1_batch_stride;
  a.c0 = d->c0;
  a.c1 = d->c1;
  a.ctot = d->c0 + d->c1;
  a.H = d->height;
  a.W = d->width;
  a.OH = (d->height + 2 * d->pad - d->kh) / d->stride + 1;
  a.OW = (d->width + 2 * d->pad - d->kw) / d->stride + 1;
  DEVA_REQUIRE(a.OH > 0 && a.OW > 0, "deva_conv2d: empty output");
  a.OHW = a.OH * a.OW;
  a.HW = (int64_t)d->height * d->width;
  a.w = d->weight;
  a.bias = d->bias;
  a.cout = d->cout;
  a.cout_pad = d->cout_pad;
  a.k_layout = d->k_layout;
  a.KH = d->kh;
  a.KW = d->kw;
  a.stride = d->stride;
  a.pad = d->pad;
  a.K = d->kh * d->kw * a.ctot;
  const int64_t n_total = (int64_t)d->batch * a.OHW;
  DEVA_REQUIRE(n_total < (1ll << 31) && a.HW < (1ll << 31), "deva_conv2d: tensor too large for 32-bit pixel index");
  a.n_total = (int)n_total;
  a.relu_in = d->relu_in;
  a.res = d->residual;
  a.res_bs = d->residual_batch_stride;
  a.act = d->act;
  a.out = d->out;
  a.tiles_n = 0;
  a.tiles_m = 0;
  a.group_m = 0;
  {
    const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    a.vec_out = a.OHW % 4 == 0 && al16(d->out) && (!d->residual || (al16(d->residual) && d->residual_batch_stride % 4 == 0));
  }
  // 4-pixel vector gathers: stride-1 'same' geometry, quads never straddle images, and the caller
  // vouches for readable guard bands around both inputs
  a.vec_ok = (d->stride == 1 && a.OH == a.H && a.OW == a.W && a.OHW % 4 == 0 && a.OW >= 4 &&
              (int64_t)d->in_guard_elems >= (int64_t)d->pad * (a.W + 1) + 4)
                 ? 1
                 : 0;
  a.per_split = 0;
  a.splits = 1;
  a.ws = d->workspace;
  a.ws_elems = d->workspace ? d->workspace_elems : 0;
  a.w16 = d->amp ? d->weight_f16 : nullptr;
  a.prec = a.w16 ? d->amp : 0;
  a.out_scale = 1.0f;
  a.flag = nullptr;
  a.gate = nullptr;
  a.ablate = 0;
  a.in0_span = (int64_t)(d->batch - 1) * a.bs0 + (int64_t)a.c0 * a.HW;
  a.in1_span = a.in1 ? (int64_t)(d->batch - 1) * a.bs1 + (int64_t)a.c1 * a.HW : 0;

  hipStream_t st = (hipStream_t)stream;
  // single output channel: VALU kernels (conv_cout1.hip) -- a dot product per pixel on small maps
  // (shrinkage head, CBAM gate), a row-reusing 3x3 kernel on large guard-banded maps (mask-logit head);
  // anything else stays on the MFMA tile
  const bool rows3x3 = a.cout == 1 && a.vec_ok && a.KH == 3 && a.KW == 3 && a.pad == 1 && a.OW % 4 == 0 &&
                       a.n_total >= 16384 && a.ctot <= 1024 && (int64_t)d->in_guard_elems >= a.W + 8;
  if (rows3x3 || (a.cout == 1 && a.K <= 7168 && a.n_total < 16384)) {  // 8 bytes of LDS table per reduction index
    Cout1Args c;
    c.in0 = a.in0;
    c.in1 = a.in1;
    c.bs0 = a.bs0;
    c.bs1 = a.bs1;
    c.c0 = a.c0;
    c.ctot = a.ctot;
    c.H = a.H;
    c.W = a.W;
    c.OH = a.OH;
    c.OW = a.OW;
    c.OHW = a.OHW;
    c.HW = a.HW;
    c.w = a.w;
    c.bias = a.bias;
    c.cout_pad = a.cout_pad;
    c.k_layout = a.k_layout;
    c.KH = a.KH;
    c.KW = a.KW;
    c.stride = a.stride;
    c.pad = a.pad;
    c.n_total = a.n_total;
    c.relu_in = a.relu_in;
    c.res = a.res;
    c.res_bs = a.res_bs;
    c.act = a.act;
    c.out = a.out;
    if (rows3x3) return launch_conv3x3_cout1_rows(c, st);
    return launch_conv_cout1(c, st);
  }
  if (d->weight_wino && !a.w16 && a.cout > 1) {  // fp32 Winograd F(2x2, 3x3): big 3x3 stride-1 layers only (conv_wino.hip)
    const int rc = launch_conv_wino(a, d->weight_wino, st);
    if (rc >= 0) return rc;
  }
  if (a.w16 && a.cout > 1 && a.in0_span < (1ll << 29) && a.in1_span < (1ll << 29)) {
    // opt-in f16 matrix pipes (fp16 operands, or the fp32-accurate hi/lo split): eligible shapes only, everything else
    // -- and inputs too large for 32-bit buffer offsets -- stays on the fp32 kernels
    // amp == 2: the gated fp32 re-run reads in0 / in1 / residual AFTER the split kernel has written `out`.  An output that
    // overlaps one of them (an in-place residual add, say, which is fine on the plain fp32 path: every element is read
    // before it is written by the same thread) would make the re-run add the residual twice: such calls take the fp32
    // kernels directly (include/deva_hip.h)
    bool aliased = false;
    if (a.prec == 2) {
      const auto overlaps = [&](const float* q, int64_t elems) {
        const int64_t out_elems = (int64_t)d->batch * a.cout * a.OHW;
        return q && elems > 0 && q < a.out + out_elems && a.out < q + elems;
      };
      const int64_t res_elems = d->residual ? (int64_t)(d->batch - 1) * a.res_bs + (int64_t)a.cout * a.OHW : 0;
      aliased = overlaps(a.in0, a.in0_span) || overlaps(a.in1, a.in1_span) || overlaps(d->residual, res_elems);
    }
    if (a.prec == 2) {
      DEVA_REQUIRE(d->split_scale_log2 >= -120 && d->split_scale_log2 <= 120, "deva_conv2d: split_scale_log2 out of range");
      a.out_scale = ldexpf(1.0f, -d->split_scale_log2);
      a.flag = d->split_flag;
    }
    const int rc = aliased ? -1 : launch_conv_f16(a, st);
    if (rc > 0 || (rc == 0 && a.prec != 2)) return rc;
    // split launched: the fp32 kernels run behind it, gated on the flag it raises for inputs beyond the fp16 range
    if (rc == 0) a.gate = a.flag;
#ifdef DEVA_CONV_PROBES  // `make PROBES=1`: what the gated launch costs (tools/convlab)
    {
      static const bool nogate = getenv("DEVA_SPLIT_NOGATE") != nullptr;
      if (rc == 0 && nogate) return 0;
    }
#endif
  }
  a.w16 = nullptr;
  a.prec = 0;
  a.out_scale = 1.0f;
  a.flag = nullptr;
  if (a.k_layout & DEVA_KLAYOUT_Q4) {
    // buffer addressing: 32-bit byte offsets from the tensor bases
    DEVA_REQUIRE(a.in0_span < (1ll << 29) && a.in1_span < (1ll << 29),
                 "deva_conv2d: k-quad weights need inputs below 2 GiB (32-bit buffer offsets)");
    if (a.gate) {  // the re-run behind a split launch: a persistent kernel (what it costs is its dispatch)
      const int rc = launch_conv_q4_gated(a, st);
      if (rc >= 0) return rc;
    }
    return launch_conv_q4(a, st);
  }
  // Tile choice (all tiles run 32-deep K steps):
  const int64_t blocks128 = ceil_div(a.cout, 128) * ceil_div(a.n_total, 128);
#ifdef DEVA_CONV_PROBES  // `make PROBES=1`: A/B runs of the tile policy (tools/conv_microbench.py)
  {
    static const int forced = [] {
      const char* e = getenv("DEVA_CONV_TILE");
      return e ? atoi(e) : 0;
    }();
    if (forced == 64 && a.cout > 32) return launch_tile<64, 64, 32, 2, 2>(a, st);
    if (forced == 128 && a.cout >= 128) return launch_tile<128, 128, 32, 2, 4>(a, st);
  }
#endif
  if (a.cout <= 32) return launch_tile<32, 128, 32, 1, 4>(a, st);
  // measured: the 8-wave 128x128 tile beats the 64x64 tile from ~64 tiles up (split-K tops the grid up)
  // ... except where the 128-wide tiles would need split-K while the 64-wide ones fill the chip on their own
  // (1x1 256->1024 on a 30x54 map: 20 vs 29 us)
  const int64_t blocks64 = ceil_div(a.cout, 64) * ceil_div(a.n_total, 64);
  if (a.cout >= 128 && blocks128 >= 64 && !(blocks128 < 256 && blocks64 >= 256 && a.K <= 512))
    return launch_tile<128, 128, 32, 2, 4>(a, st);
  return launch_tile<64, 64, 32, 2, 2>(a, st);
}

// Host-side weight packing (model load, not the frame path): [cout][cin][kh][kw] -> the layout deva_conv2d reads.
extern "C" int64_t deva_conv_pack(const float* w_oihw, float* out, int cout, int cin, int kh, int kw, int want_q4,
                                  int* k_layout, int* cout_pad_out) {
  using namespace deva;
  if (!w_oihw || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0 || !k_layout || !cout_pad_out) {
    set_error("deva_conv_pack: bad arguments");
    return -1;
  }
  const int taps = kh * kw;
  const int K = taps * cin;
  const int cout_pad = (cout + 31) / 32 * 32;
  const bool chunk = taps > 1 && cin % 32 == 0;
  const bool q4 = want_q4 && cout > 1;  // the single-channel heads (conv_cout1.hip) read column 0 of [K][cout_pad]
  const int64_t rows = q4 ? (int64_t)(K + 3) / 4 * 4 : K;
  const int64_t elems = rows * cout_pad;
  *k_layout = (chunk ? DEVA_KLAYOUT_CHUNK32 : DEVA_KLAYOUT_TAP_MAJOR) | (q4 ? DEVA_KLAYOUT_Q4 : 0);
  *cout_pad_out = cout_pad;
  if (!out) return elems;
  for (int64_t i = 0; i < elems; ++i) out[i] = 0.0f;
  for (int m = 0; m < cout; ++m)
    for (int c = 0; c < cin; ++c)
      for (int t = 0; t < taps; ++t) {
        const int64_t k = chunk ? ((int64_t)(c / 32) * taps + t) * 32 + c % 32 : (int64_t)t * cin + c;
        const int64_t at = q4 ? ((k >> 2) * cout_pad + m) * 4 + (k & 3) : k * cout_pad + m;
        out[at] = w_oihw[((int64_t)m * cin + c) * taps + t];
      }
  return elems;
}

// fp16 weights of the opt-in amp path (host side, model load): element (k, m) at ((k/8)*cout_pad + m)*8 + k%8, IEEE
// binary16 bits, round to nearest even; K order: tap-major for 1x1, 64-channel slabs otherwise
// (k = ((c/64)*taps + tap)*64 + c%64; needs cin % 64 == 0, else -1: the layer stays fp32).
extern "C" int64_t deva_conv_pack_f16(const float* w_oihw, uint16_t* out, int cout, int cin, int kh, int kw, int* cout_pad_out) {
  using namespace deva;
  if (!w_oihw || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0 || !cout_pad_out) {
    set_error("deva_conv_pack_f16: bad arguments");
    return -1;
  }
  const int taps = kh * kw;
  if (cin % 64 != 0) return -1;
  const int K = taps * cin;
  const int cout_pad = (cout + 31) / 32 * 32;
  const int64_t elems = (int64_t)K * cout_pad;
  *cout_pad_out = cout_pad;
  if (!out) return elems;
  for (int64_t i = 0; i < elems; ++i) out[i] = 0;
  for (int m = 0; m < cout; ++m)
    for (int c = 0; c < cin; ++c)
      for (int t = 0; t < taps; ++t) {
        const int64_t k = taps > 1 ? ((int64_t)(c / 64) * taps + t) * 64 + c % 64 : c;
        const _Float16 h = (_Float16)w_oihw[((int64_t)m * cin + c) * taps + t];
        uint16_t bits;
        __builtin_memcpy(&bits, &h, 2);
        out[((k >> 3) * cout_pad + m) * 8 + (k & 7)] = bits;
      }
  return elems;
}

// hi / lo fp16 planes of the split path (host side, model load): with s = 2^e, e such that the largest |w| * s lies in
// [2^13, 2^14) (e = 0 for an all-zero layer), hi = fp16(w s), lo = fp16(w s - hi) (round to nearest even; w s and the
// difference are exact in fp32), element (k, plane, m) at (((k/8)*2 + plane)*cout_pad + m)*8 + k%8; K order: tap-major
// for 1x1 (any cin: K is padded with zero rows to a multiple of 32), 32-channel slabs otherwise
// (k = ((c/32)*taps + tap)*32 + c%32; needs cin % 32 == 0, else -1: the layer stays on the fp32 kernels).
// *scale_log2 = e; deva_conv2d multiplies the accumulators by 2^-e.
extern "C" int64_t deva_conv_pack_split(const float* w_oihw, uint16_t* out, int cout, int cin, int kh, int kw, int* cout_pad_out,
                                        int* scale_log2) {
  using namespace deva;
  if (!w_oihw || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0 || !cout_pad_out || !scale_log2) {
    set_error("deva_conv_pack_split: bad arguments");
    return -1;
  }
  const int taps = kh * kw;
  if (cin % 32 != 0 && taps > 1) return -1;
  const int K = (taps * cin + 31) / 32 * 32;  // 1x1 layers with a channel tail (513, 257): zero rows up to the next K step
  const int cout_pad = (cout + 31) / 32 * 32;
  const int64_t elems = (int64_t)K * 2 * cout_pad;
  *cout_pad_out = cout_pad;
  float wmax = 0.0f;
  const int64_t n = (int64_t)cout * cin * taps;
  for (int64_t i = 0; i < n; ++i) {
    const float v = fabsf(w_oihw[i]);
    if (!(v <= 3.0e38f)) {
      set_error("deva_conv_pack_split: non-finite weight");
      return -1;
    }
    if (v > wmax) wmax = v;
  }
  int e = 0;
  if (wmax > 0.0f) {
    int x;
    frexpf(wmax, &x);  // wmax = f * 2^x, f in [0.5, 1)
    e = 14 - x;
    if (e > 120) e = 120;
    if (e < -120) e = -120;
  }
  *scale_log2 = e;
  if (!out) return elems;
  for (int64_t i = 0; i < elems; ++i) out[i] = 0;
  for (int m = 0; m < cout; ++m)
    for (int c = 0; c < cin; ++c)
      for (int t = 0; t < taps; ++t) {
        const int64_t k = taps > 1 ? ((int64_t)(c / 32) * taps + t) * 32 + c % 32 : c;
        const float ws = ldexpf(w_oihw[((int64_t)m * cin + c) * taps + t], e);
        const _Float16 hi = (_Float16)ws;
        const _Float16 lo = (_Float16)(ws - (float)hi);
        uint16_t bh, bl;
        __builtin_memcpy(&bh, &hi, 2);
        __builtin_memcpy(&bl, &lo, 2);
        out[(((k >> 3) * 2 + 0) * cout_pad + m) * 8 + (k & 7)] = bh;
        out[(((k >> 3) * 2 + 1) * cout_pad + m) * 8 + (k & 7)] = bl;
      }
  return elems;
}
