// How fast can ONE wave per SIMD issue dependent v_mfma_f32_32x32x2_f32 chains?  (The affinity kernel
// runs one wave per SIMD with two alternating accumulator chains.)  Prints cycles per MFMA for 1, 2
// and 4 independent chains at 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void probe(float* out, int iters, long long* cycles) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32 / CHAINS; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int CHAINS>
void run(int waves_per_simd) {
  float* out;
  long long* cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(float) * 4);
  hipMalloc(&cyc, 8);
  const int iters = 2000;
  dim3 block(64 * 4 * waves_per_simd);  // 4 SIMDs per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<CHAINS>, dim3(256), block, 0, 0, out, 10, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<CHAINS>, dim3(256), block, 0, 0, out, iters, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n_mfma = (double)iters * 32;
  const double tflops = 256.0 * 4 * waves_per_simd * n_mfma * 4096 / (ms * 1e-3) / 1e12;
  printf("chains=%d waves/SIMD=%d: %.1f us, %.1f TFLOP/s, s_memtime ticks per MFMA per wave %.2f (100 MHz ticks), ns per MFMA per wave %.1f\n",
         CHAINS, waves_per_simd, ms * 1e3, tflops, (double)c / n_mfma, ms * 1e6 / n_mfma);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<1>(w);
    run<2>(w);
    run<4>(w);
  }
  return 0;
}
