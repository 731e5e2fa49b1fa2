// Detection merging support (deva/inference/segment_merging.py:17-143): the reference compares every
// propagated segment with every detected segment through boolean-mask products, one host sync per
// pair.  Here one pass over the two index masks builds the joint label histogram (integer atomics:
// exact and order-independent), the host takes all IoU decisions from that small matrix, and one more
// pass paints the merged result straight into one-hot planes.
#include "common.h"

namespace deva {
namespace {

// column of a detection id (linear search: a frame has at most a few dozen detections); n_new = "none"
__device__ __forceinline__ int find_id(const int64_t* __restrict__ ids, int n, int64_t v) {
  for (int j = 0; j < n; ++j)
    if (ids[j] == v) return j;
  return n;
}

// counts[(t)*(n_new+1) + j] += 1 for every pixel with propagated tmp id t (0 = background, clamped to
// n_our) and detection column j (n_new = not a listed detection)
__global__ void label_histogram_kernel(const int64_t* __restrict__ ours, const int64_t* __restrict__ news,
                                       const int64_t* __restrict__ new_ids, int n_our, int n_new, int64_t pixels,
                                       int* __restrict__ counts) {
  extern __shared__ int hist[];  // (n_our+1)*(n_new+1)
  const int bins = (n_our + 1) * (n_new + 1);
  for (int i = threadIdx.x; i < bins; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < pixels; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = ours[i];
    if (t < 0 || t > n_our) t = 0;
    const int j = find_id(new_ids, n_new, news[i]);
    atomicAdd(&hist[(int)t * (n_new + 1) + j], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < bins; i += blockDim.x)
    if (hist[i]) atomicAdd(&counts[i], hist[i]);
}

// paint: every source (propagated tmp id t, detection column j) carries (order, label) or order < 0;
// the pixel takes the label of the source painted last; out[o][i] = (label == out_ids[o])
__global__ void merge_paint_kernel(const int64_t* __restrict__ ours, const int64_t* __restrict__ news,
                                   const int64_t* __restrict__ new_ids, int n_our, int n_new,
                                   const int* __restrict__ our_order, const int64_t* __restrict__ our_label,
                                   const int* __restrict__ new_order, const int64_t* __restrict__ new_label,
                                   const int64_t* __restrict__ out_ids, int n_out, int64_t pixels,
                                   float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < pixels; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = ours[i];
    if (t < 0 || t > n_our) t = 0;
    const int j = find_id(new_ids, n_new, news[i]);
    int order = -1;
    int64_t label = 0;
    if (t > 0 && our_order[t] >= 0) {
      order = our_order[t];
      label = our_label[t];
    }
    if (j < n_new && new_order[j] > order) {  // a matched pair shares one order: the detection wins
      order = new_order[j];
      label = new_label[j];
    } else if (j < n_new && new_order[j] == order && order >= 0) {
      label = new_label[j];
    }
    for (int o = 0; o < n_out; ++o) out[(int64_t)o * pixels + i] = (order >= 0 && label == out_ids[o]) ? 1.0f : 0.0f;
  }
}

// out[i] = lut[in[i]] for 0 <= in[i] < n, else 0  (ObjectManager.tmp_to_obj_cls, object_manager.py:112-117)
__global__ void lut_remap_kernel(const int64_t* __restrict__ in, const int64_t* __restrict__ lut, int n,
                                 int64_t pixels, int64_t* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < pixels; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = in[i];
    out[i] = (v >= 0 && v < n) ? lut[v] : 0;
  }
}

}  // namespace
}  // namespace deva

using namespace deva;

extern "C" int deva_lut_remap(const int64_t* in, const int64_t* lut, int n, int64_t pixels, int64_t* out,
                              void* stream) {
  DEVA_REQUIRE(in && lut && out && n > 0 && pixels > 0, "deva_lut_remap: bad args");
  int64_t blocks = ceil_div(pixels, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(lut_remap_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, lut, n, pixels,
                     out);
  return check_launch("deva_lut_remap");
}


extern "C" int deva_label_histogram(const int64_t* ours, const int64_t* news, const int64_t* new_ids, int n_our,
                                    int n_new, int64_t pixels, int32_t* counts, void* stream) {
  DEVA_REQUIRE(ours && news && counts && n_our >= 0 && n_new >= 0 && pixels > 0, "deva_label_histogram: bad args");
  DEVA_REQUIRE(n_new == 0 || new_ids, "deva_label_histogram: null id list");
  const size_t smem = sizeof(int) * (size_t)(n_our + 1) * (n_new + 1);
  DEVA_REQUIRE(smem <= 60 * 1024, "deva_label_histogram: %d x %d label pairs do not fit the LDS histogram", n_our + 1,
               n_new + 1);
  int64_t blocks = ceil_div(pixels, 256 * 8);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(label_histogram_kernel, dim3((unsigned)blocks), dim3(256), smem, (hipStream_t)stream, ours, news,
                     new_ids, n_our, n_new, pixels, counts);
  return check_launch("deva_label_histogram");
}

extern "C" int deva_merge_paint(const int64_t* ours, const int64_t* news, const int64_t* new_ids, int n_our, int n_new,
                                const int32_t* our_order, const int64_t* our_label, const int32_t* new_order,
                                const int64_t* new_label, const int64_t* out_ids, int n_out, int64_t pixels,
                                float* out, void* stream) {
  DEVA_REQUIRE(ours && news && our_order && our_label && out && pixels > 0 && n_out >= 0, "deva_merge_paint: bad args");
  DEVA_REQUIRE(n_new == 0 || (new_ids && new_order && new_label), "deva_merge_paint: null detection tables");
  if (n_out == 0) return 0;
  int64_t blocks = ceil_div(pixels, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(merge_paint_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ours, news, new_ids,
                     n_our, n_new, our_order, our_label, new_order, new_label, out_ids, n_out, pixels, out);
  return check_launch("deva_merge_paint");
}
