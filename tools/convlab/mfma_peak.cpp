// mfma_peak -- what the fp32 matrix pipes sustain on this chip with nothing else in the way: every wave issues
// v_mfma_f32_32x32x2_f32 back to back on 4 independent accumulators (operands in registers, no memory traffic in the
// loop), 4 waves per SIMD on every CU, for ~25 ms after a warm-up.  Random operands vs zeros shows how much of the
// rate is the power budget (MI355X_MICROARCH.md, "DVFS give-back").  Peak by the data sheet: 256 CUs x 256 flop/cycle
// x 2.4 GHz = 157.3 TFLOP/s.
//
//   hipcc --offload-arch=gfx950 -O2 tools/convlab/mfma_peak.cpp -o tools/convlab/mfma_peak && tools/convlab/mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ src, float* __restrict__ dst, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = src[(t * 8 + i) & 65535];
    b[i] = src[(t * 8 + 4 + i) & 65535];
  }
  f32x16 acc[4];
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
  if (s == 123.456f) dst[t] = s;  // keeps the loop alive
}

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float *src, *dst;
  HIP_OK(hipMalloc(&src, 65536 * 4));
  HIP_OK(hipMalloc(&dst, (size_t)cus * 4 * 256 * 4));
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  printf("%d CUs, data-sheet fp32 matrix peak at 2.4 GHz: %.1f TFLOP/s\n", cus, cus * 256 * 2.4e9 / 1e12);
  for (int mode = 0; mode < 2; ++mode) {
    std::vector<float> h(65536);
    srand(7);
    for (auto& v : h) v = mode == 0 ? (float)(rand() % 65536) / 32768.0f - 1.0f : 0.0f;
    HIP_OK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int iters = 20000;                   // x 16 MFMAs per wave: ~9 ms per launch at full rate
    const dim3 grid(cus * 4), block(256);      // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(mfma_loop, grid, block, 0, 0, src, dst, iters);
    HIP_OK(hipEventRecord(e0, 0));
    const int reps = 4;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop, grid, block, 0, 0, src, dst, iters);
    HIP_OK(hipEventRecord(e1, 0));
    HIP_OK(hipEventSynchronize(e1));
    float ms;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)reps * cus * 16 * (double)iters * 16 * (2.0 * 32 * 32 * 2);
    const double tf = flop / (ms * 1e-3) / 1e12;
    printf("%-16s %8.2f ms  %7.1f TFLOP/s  = %.3f of the data-sheet peak, i.e. an effective clock of %.2f GHz\n",
           mode == 0 ? "random operands" : "zero operands", ms, tf, tf / (cus * 256 * 2.4e9 / 1e12), tf * 1e12 / (cus * 256.0) / 1e9);
  }
  return 0;
}
