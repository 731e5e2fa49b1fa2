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

// the same without the LDS stage, for label tables beyond 60 KiB (> ~124 x 124 pairs: "segment everything"
// clips with --max_num_objects -1): integer atomics straight into `counts` (exact in any order)
__global__ void label_histogram_global_kernel(const int64_t* __restrict__ ours, const int64_t* __restrict__ news,
                                              const int64_t* __restrict__ new_ids, int n_our, int n_new,
                                              int64_t pixels, int* __restrict__ counts) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < pixels; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = ours[i];
    if (t < 0 || t > n_our) t = 0;
    const int j = find_id(new_ids, n_new, news[i]);
    atomicAdd(&counts[(int64_t)t * (n_new + 1) + j], 1);
  }
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

// Output tail (evaluation/eval_vos.py:170-181, result_utils.py:98-102, object_manager.py:112-117):
// out[y][x] = lut[ argmax_c resize(prob)[c][y][x] ] in one pass -- F.interpolate(mode='bilinear',
// align_corners=False) of every channel to (oh, ow) when the size differs, first-maximum argmax, and the
// tmp-id -> object-id table.  The (no+1)*H*W fp32 probabilities never leave the device; the host
// copies H*W labels.  Bilinear arithmetic follows ATen's upsample_bilinear2d: source coordinate
// scale*(dst+0.5)-0.5 clamped at 0, neighbour index clamped at the border, rows blended after columns.
__global__ void index_mask_kernel(const float* __restrict__ prob, int channels, int h, int w, int oh, int ow,
                                  float scale_y, float scale_x, const int64_t* __restrict__ lut, int n_lut,
                                  int64_t* __restrict__ out) {
  const int64_t total = (int64_t)oh * ow;
  const int64_t plane = (int64_t)h * w;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / ow), x = (int)(i - (int64_t)y * ow);
    int best = 0;
    if (oh == h && ow == w) {
      float bv = prob[i];
      for (int c = 1; c < channels; ++c) {
        const float v = prob[(int64_t)c * plane + i];
        if (v > bv) {
          bv = v;
          best = c;
        }
      }
    } else {
      const float sy = fmaxf(scale_y * ((float)y + 0.5f) - 0.5f, 0.0f);
      const float sx = fmaxf(scale_x * ((float)x + 0.5f) - 0.5f, 0.0f);
      const int y0 = min((int)sy, h - 1), x0 = min((int)sx, w - 1);
      const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
      const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
      const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
      float bv = -INFINITY;
      for (int c = 0; c < channels; ++c) {
        const float* pc = prob + (int64_t)c * plane;
        const float top = lx0 * pc[(int64_t)y0 * w + x0] + lx1 * pc[(int64_t)y0 * w + x1];
        const float bot = lx0 * pc[(int64_t)y1 * w + x0] + lx1 * pc[(int64_t)y1 * w + x1];
        const float v = ly0 * top + ly1 * bot;
        if (v > bv) {
          bv = v;
          best = c;
        }
      }
    }
    out[i] = lut ? ((best < n_lut) ? lut[best] : 0) : (int64_t)best;
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
  DEVA_REQUIRE((int64_t)(n_our + 1) * (n_new + 1) < (1ll << 31), "deva_label_histogram: label table too large");
  int64_t blocks = ceil_div(pixels, 256 * 8);
  if (blocks > 2048) blocks = 2048;
  if (smem <= 60 * 1024) {
    hipLaunchKernelGGL(label_histogram_kernel, dim3((unsigned)blocks), dim3(256), smem, (hipStream_t)stream, ours,
                       news, new_ids, n_our, n_new, pixels, counts);
  } else {  // the table does not fit the LDS: global integer atomics (the reference has no object limit)
    hipLaunchKernelGGL(label_histogram_global_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ours,
                       news, new_ids, n_our, n_new, pixels, counts);
  }
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

extern "C" int deva_index_mask(const float* prob, int channels, int height, int width, int out_height,
                               int out_width, const int64_t* lut, int n_lut, int64_t* out, void* stream) {
  using namespace deva;
  DEVA_REQUIRE(prob && out && channels > 0 && height > 0 && width > 0 && out_height > 0 && out_width > 0,
               "deva_index_mask: bad args");
  DEVA_REQUIRE(!lut || n_lut > 0, "deva_index_mask: empty table");
  const int64_t total = (int64_t)out_height * out_width;
  int64_t blocks = ceil_div(total, 256);
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(index_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, prob, channels,
                     height, width, out_height, out_width, (float)height / (float)out_height,
                     (float)width / (float)out_width, lut, n_lut, out);
  return check_launch("deva_index_mask");
}
