// Token-major memory-bank maintenance: append / gather / export, usage ranking, least-usage
// eviction and the dense potentiation step of memory consolidation.  These run once per memory
// frame (append) or once per consolidation (everything else), so they are written for clarity and
// determinism (stable orders, no floating-point atomics), not for peak throughput.
#include <math.h>

#include "common.h"

#pragma clang fp contract(off)

namespace deva {
namespace {

// dst[c][r] = src[r][c]   (src rows x cols, row-major) via a padded 32x32 LDS tile
__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(int64_t)r * cols + c] : 0.0f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst[(int64_t)c * rows + r] = tile[tx][i];
  }
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ rows,
                                   float* __restrict__ dst, int64_t total, int channels) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / channels;
    const int c = (int)(i - r * channels);
    const int64_t sr = rows ? (int64_t)rows[r] : r;
    dst[i] = src[sr * channels + c];
  }
}

__global__ void normalize_usage_kernel(const float* __restrict__ use, const float* __restrict__ life,
                                       float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = use[i] / life[i];
}

// rank[i] = #{ j : x_j before x_i } in the order (desc ? larger first : smaller first), ties by index
__global__ __launch_bounds__(256) void rank_kernel(const float* __restrict__ x, int n, int descending,
                                                   int32_t* __restrict__ rank) {
  __shared__ float sx[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float xi = (i < n) ? x[i] : 0.0f;
  int r = 0;
  for (int j0 = 0; j0 < n; j0 += 256) {
    const int jj = j0 + threadIdx.x;
    __syncthreads();
    sx[threadIdx.x] = (jj < n) ? x[jj] : 0.0f;
    __syncthreads();
    const int lim = min(256, n - j0);
    for (int t = 0; t < lim; ++t) {
      const float xj = sx[t];
      const int j = j0 + t;
      const bool before = descending ? (xj > xi) : (xj < xi);
      r += (before || (xj == xi && j < i)) ? 1 : 0;
    }
  }
  if (i < n) rank[i] = r;
}

__global__ void rank_select_kernel(const int32_t* __restrict__ rank, int n, int k, int32_t* __restrict__ out_idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && rank[i] < k) out_idx[rank[i]] = i;
}

// single block: threshold = x at ascending rank n_remove-1; survivors (x > thr) compacted in order
__global__ __launch_bounds__(1024) void evict_select_kernel(const float* __restrict__ x,
                                                            const int32_t* __restrict__ rank_asc, int n,
                                                            int n_remove, int32_t* __restrict__ out_idx,
                                                            int32_t* __restrict__ out_count) {
  __shared__ float s_thr;
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_base = 0;
  for (int i = tid; i < n; i += 1024)
    if (rank_asc[i] == n_remove - 1) s_thr = x[i];
  __syncthreads();
  const float thr = s_thr;
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + tid;
    const bool keep = (i < n) && (x[i] > thr);
    const unsigned long long b = __ballot(keep);
    const int prefix = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(b);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    if (keep) out_idx[off + prefix] = i;
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += s_wave[w];
      s_base += tot;
    }
    __syncthreads();
  }
  if (tid == 0) out_count[0] = s_base;
}

// dense similarity of candidates vs prototypes with the reference's arithmetic (FMA chains in
// channel order, memory_utils.py:29-43).  block = 64 prototypes x 4 candidate lanes.
constexpr int SP = 64;
__global__ __launch_bounds__(256) void similarity_dense_kernel(const float* __restrict__ key,
                                                               const float* __restrict__ shr,
                                                               const float* __restrict__ sel,
                                                               const int32_t* __restrict__ proto_idx, int n_cand,
                                                               int n_proto, int ld, float* __restrict__ sim) {
  __shared__ float s_qe[64][SP];
  __shared__ float s_qkqe[64][SP];
  __shared__ float s_bsq[SP];
  const int p0 = blockIdx.x * SP;
  for (int e = threadIdx.x; e < 64 * SP; e += 256) {
    const int pl = e / 64, c = e % 64;  // consecutive threads read consecutive channels of a row
    const int p = p0 + pl;
    float qe = 0.0f, qk = 0.0f;
    if (p < n_proto) {
      const int64_t row = proto_idx[p];
      qe = sel[row * 64 + c];
      qk = key[row * 64 + c];
    }
    s_qe[c][pl] = qe;
    s_qkqe[c][pl] = qk * qe;
  }
  __syncthreads();
  if (threadIdx.x < SP) {
    const int pl = threadIdx.x;
    const int p = p0 + pl;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    if (p < n_proto) {
      const int64_t row = proto_idx[p];
      for (int c = 0; c < 64; ++c) {
        const float qk = key[row * 64 + c];
        bs[c >> 4] += s_qe[c][pl] * (qk * qk);
      }
    }
    s_bsq[pl] = ((bs[0] + bs[1]) + bs[2]) + bs[3];
  }
  __syncthreads();
  const int pl = threadIdx.x & (SP - 1);
  const int nl = threadIdx.x / SP;  // 0..3
  const int p = p0 + pl;
  for (int n = blockIdx.y * 4 + nl; n < n_cand; n += gridDim.y * 4) {
    const float* krow = key + (int64_t)n * 64;
    float a = 0.0f, b = 0.0f;
#pragma unroll 8
    for (int c = 0; c < 64; ++c) {
      const float m = krow[c];
      a = fmaf(m * m, s_qe[c][pl], a);
      b = fmaf(m, s_qkqe[c][pl], b);
    }
    float v = (-a + 2.0f * b) - s_bsq[pl];
    v = v * shr[n] * 0.125f;
    if (p < n_proto) sim[(int64_t)n * ld + p] = v;
  }
}

// softmax over n for each column p of x[n][ld]; block = 32 columns x 8 row lanes
__global__ __launch_bounds__(256) void softmax_columns_kernel(float* __restrict__ x, int n, int p, int ld) {
  __shared__ float red[8][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + tx;
  const bool ok = col < p;
  float m = -INFINITY;
  if (ok)
    for (int r = ty; r < n; r += 8) m = fmaxf(m, x[(int64_t)r * ld + col]);
  red[ty][tx] = m;
  __syncthreads();
  m = red[0][tx];
  for (int i = 1; i < 8; ++i) m = fmaxf(m, red[i][tx]);
  __syncthreads();
  float s = 0.0f;
  if (ok)
    for (int r = ty; r < n; r += 8) s += expf(x[(int64_t)r * ld + col] - m);
  red[ty][tx] = s;
  __syncthreads();
  s = red[0][tx];
  for (int i = 1; i < 8; ++i) s += red[i][tx];
  if (ok)
    for (int r = ty; r < n; r += 8) x[(int64_t)r * ld + col] = expf(x[(int64_t)r * ld + col] - m) / s;
}

}  // namespace
}  // namespace deva

using namespace deva;

static int launch_transpose(const float* src, float* dst, int rows, int cols, void* stream) {
  dim3 grid((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows, 32));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, rows, cols);
  return 0;
}

// usage counters of freshly appended tokens (kv_memory_store.py:93-95: use_count = 0, life_count = 1e-7)
__global__ void usage_init_kernel(float* __restrict__ use, float* __restrict__ life, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    use[i] = 0.0f;
    life[i] = 1e-7f;
  }
}

extern "C" int deva_usage_init(float* use, float* life, int count, void* stream) {
  DEVA_REQUIRE(use && life && count > 0, "deva_usage_init: bad args");
  hipLaunchKernelGGL(usage_init_kernel, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0, (hipStream_t)stream, use, life, count);
  return check_launch("deva_usage_init");
}

extern "C" int deva_bank_append(const float* src, float* arena, int64_t dst_row0, int channels, int count,
                                void* stream) {
  DEVA_REQUIRE(src && arena && dst_row0 >= 0 && channels > 0 && count > 0, "deva_bank_append: bad args");
  launch_transpose(src, arena + dst_row0 * channels, channels, count, stream);
  return check_launch("deva_bank_append");
}

extern "C" int deva_bank_export(const float* arena, float* dst, int channels, int count, void* stream) {
  DEVA_REQUIRE(arena && dst && channels > 0 && count > 0, "deva_bank_export: bad args");
  launch_transpose(arena, dst, count, channels, stream);
  return check_launch("deva_bank_export");
}

extern "C" int deva_bank_gather_rows(const float* src, const int32_t* rows, float* dst, int count, int channels,
                                     void* stream) {
  DEVA_REQUIRE(src && dst && count >= 0 && channels > 0, "deva_bank_gather_rows: bad args");
  if (count == 0) return 0;
  const int64_t total = (int64_t)count * channels;
  int64_t blocks = ceil_div(total, 256);
  if (blocks > 1048576) blocks = 1048576;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, rows, dst,
                     total, channels);
  return check_launch("deva_bank_gather_rows");
}

extern "C" int deva_rank(const float* x, const float* life, float* x_out, int n, int descending, int32_t* rank,
                         void* stream) {
  DEVA_REQUIRE(x && rank && n > 0, "deva_rank: bad args");
  const float* src = x;
  if (life) {
    DEVA_REQUIRE(x_out, "deva_rank: x_out required when life is given");
    hipLaunchKernelGGL(normalize_usage_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       x, life, x_out, n);
    src = x_out;
  }
  hipLaunchKernelGGL(rank_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, src, n,
                     descending, rank);
  return check_launch("deva_rank");
}

extern "C" int deva_rank_select(const int32_t* rank, int n, int k, int32_t* out_idx, void* stream) {
  DEVA_REQUIRE(rank && out_idx && n > 0 && k > 0 && k <= n, "deva_rank_select: bad args (n=%d k=%d)", n, k);
  hipLaunchKernelGGL(rank_select_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, rank,
                     n, k, out_idx);
  return check_launch("deva_rank_select");
}

extern "C" int deva_evict_select(const float* x, const int32_t* rank_asc, int n, int n_remove, int32_t* out_idx,
                                 int32_t* out_count, void* stream) {
  DEVA_REQUIRE(x && rank_asc && out_idx && out_count && n > 0, "deva_evict_select: bad args");
  DEVA_REQUIRE(n_remove >= 1 && n_remove <= n, "deva_evict_select: n_remove=%d out of range (n=%d)", n_remove, n);
  hipLaunchKernelGGL(evict_select_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, rank_asc, n, n_remove,
                     out_idx, out_count);
  return check_launch("deva_evict_select");
}

extern "C" int deva_similarity_dense(const float* key, const float* shr, const float* sel, const int32_t* proto_idx,
                                     int n_cand, int n_proto, int ld, float* sim, void* stream) {
  DEVA_REQUIRE(key && shr && sel && proto_idx && sim && n_cand > 0 && n_proto > 0 && ld >= n_proto,
               "deva_similarity_dense: bad args");
  int gy = (int)ceil_div(n_cand, 4 * 8);
  if (gy > 4096) gy = 4096;
  dim3 grid((unsigned)ceil_div(n_proto, SP), (unsigned)gy);
  hipLaunchKernelGGL(similarity_dense_kernel, grid, dim3(256), 0, (hipStream_t)stream, key, shr, sel, proto_idx,
                     n_cand, n_proto, ld, sim);
  return check_launch("deva_similarity_dense");
}

extern "C" int deva_softmax_columns(float* x, int n, int p, int ld, void* stream) {
  DEVA_REQUIRE(x && n > 0 && p > 0 && ld >= p, "deva_softmax_columns: bad args");
  hipLaunchKernelGGL(softmax_columns_kernel, dim3((unsigned)ceil_div(p, 32)), dim3(256), 0, (hipStream_t)stream, x,
                     n, p, ld);
  return check_launch("deva_softmax_columns");
}
