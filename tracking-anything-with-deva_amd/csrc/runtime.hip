// Error channel + version of libdeva_hip.so, and the matrix-pipe probe.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace deva {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace deva

extern "C" int deva_hip_version(void) { return DEVA_HIP_ABI_VERSION; }
extern "C" const char* deva_hip_last_error(void) { return deva::g_err; }

// ---- deva_probe_mfma_f32: what the fp32 matrix pipes sustain with nothing else in the way.  Every wave issues
// v_mfma_f32_32x32x2_f32 back to back on 4 independent accumulators (operands in registers, no memory traffic in the
// loop), 4 waves per SIMD on every CU.  bench.py times it next to the convolution roofline: the chip clocks to its
// power budget under dense MFMA work, so this -- not 256 CUs x 256 flop/cycle x 2.4 GHz -- is the rate a perfect
// kernel would reach (tools/convlab/mfma_peak.cpp is the stand-alone version).
namespace deva {
namespace {
typedef float probe_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_probe_kernel(const float* __restrict__ src, int src_mask, float* __restrict__ dst, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = src[(t * 8 + i) & src_mask];
    b[i] = src[(t * 8 + 4 + i) & src_mask];
  }
  probe_f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + u) & 3], b[i], acc[i], 0, 0, 0);
    }
  }
  float s = 0.0f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 123.456f) dst[t & src_mask] = s;  // keeps the loop alive
}
}  // namespace
}  // namespace deva

extern "C" int64_t deva_probe_mfma_f32(const float* operands, int64_t operand_elems, float* sink, int iters, void* stream) {
  if (!operands || !sink || operand_elems < 1024 || iters < 1) {
    deva::set_error("deva_probe_mfma_f32: needs >= 1024 operand floats, a sink of as many and iters >= 1");
    return -1;
  }
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    deva::set_error("deva_probe_mfma_f32: no device");
    return -1;
  }
  int mask = 1024;
  while ((int64_t)mask * 2 <= operand_elems && mask < (1 << 20)) mask *= 2;
  hipLaunchKernelGGL(deva::mfma_probe_kernel, dim3((unsigned)cus * 4), dim3(256), 0, (hipStream_t)stream, operands, mask - 1, sink, iters);
  if (deva::check_launch("deva_probe_mfma_f32")) return -1;
  return (int64_t)cus * 16 * (int64_t)iters * 16 * (2ll * 32 * 32 * 2);  // flop of this launch
}
