// Issue model of a lone wave per SIMD: cycles per MFMA when G independent VALU instructions sit in
// every gap between consecutive v_mfma_f32_32x32x2_f32 of two alternating accumulator chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int G, int KIND>
__global__ void probe(float* out, int iters, long long* cycles) {
  f32x16 accA, accB;
  for (int r = 0; r < 16; ++r) accA[r] = accB[r] = 0.0f;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = a + i;
  __shared__ float lds[1024];
  lds[threadIdx.x & 1023] = a;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      accA = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, accA, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[g & 7]) : "v"(b));
        if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[0]) : "v"(b));  // dependent chain
        if (KIND == 2) asm volatile("s_nop 0");
      }
      accB = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, accB, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[g & 7]) : "v"(b));
        if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[0]) : "v"(b));
        if (KIND == 2) asm volatile("s_nop 0");
      }
    }
  }
  long long t1 = clock64();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += accA[r] + accB[r];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[(threadIdx.x * 7) & 1023];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int G, int KIND>
void run(int waves_per_simd = 1) {
  float* out;
  long long* cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(float));
  hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipLaunchKernelGGL((probe<G, KIND>), dim3(256), dim3(256 * waves_per_simd), 0, 0, out, 10, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<G, KIND>), dim3(256), dim3(256 * waves_per_simd), 0, 0, out, iters, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("kind=%d gap=%2d waves/SIMD=%d: %.1f ns per MFMA per SIMD (s_memtime ticks per MFMA per wave %.1f)\n", KIND, G, waves_per_simd, ms * 1e6 / (iters * 32.0) / waves_per_simd, (double)c / (iters * 32.0));
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<0, 0>(); run<2, 0>(); run<4, 0>(); run<6, 0>(); run<8, 0>(); run<12, 0>(); run<16, 0>(); run<24, 0>();
  run<4, 1>(); run<8, 1>(); run<16, 1>();
  run<8, 2>(); run<16, 2>();
  run<0, 0>(2); run<4, 0>(2); run<8, 0>(2); run<12, 0>(2); run<16, 0>(2); run<24, 0>(2);
  run<8, 0>(4); run<16, 0>(4); run<24, 0>(4); run<32, 0>(4);
  run<8, 1>(2); run<16, 1>(2);
  return 0;
}
