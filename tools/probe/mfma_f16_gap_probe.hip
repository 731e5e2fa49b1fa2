// What does an instruction in the gap between two v_mfma_f32_32x32x16_f16 cost on gfx950?  (Round 5: the hi/lo split
// convolution kernels sit at ~0.5 MFMA utilisation whatever their tile -- which of their per-MFMA companions is it?)
// One workgroup per CU, W waves per SIMD; every wave runs CHAINS independent accumulator chains and, after every MFMA,
// G instructions of one kind: VALU (v_fma_f32 on private registers), LDS read (ds_read_b128, conflict-free, linear),
// LDS write (ds_write_b128), or nothing.  Prints ns per MFMA per SIMD (32 cycles at 2.4 GHz = 13.3 ns) and TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_f16_gap_probe.hip -o mfma_f16_gap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

enum { NONE = 0, VALU = 1, LDSR = 2, LDSW = 3 };

template <int CHAINS, int KIND, int G>
__global__ __launch_bounds__(1024) void probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384];  // 64 KB
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
  h8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(threadIdx.x * 0.001f + i);
    b[i] = (_Float16)(1.0f + threadIdx.x * 0.002f - i);
  }
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
  __syncthreads();
  float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  f32x4 ld[4] = {};
  const f32x4* rd = reinterpret_cast<const f32x4*>(lds) + (threadIdx.x & 1023);
  f32x4* wr = reinterpret_cast<f32x4*>(lds) + (threadIdx.x & 1023);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u % CHAINS], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (KIND == VALU) {
          v[(u * G + g) % 8] = __builtin_fmaf(v[(u * G + g) % 8], 1.0001f, 0.5f);
        } else if (KIND == LDSR) {
          ld[(u * G + g) % 4] = rd[((u * G + g) % 4) * 1024];
        } else if (KIND == LDSW) {
          wr[((u * G + g) % 4) * 1024] = ld[(u * G + g) % 4];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KIND == LDSR) {
#pragma unroll
      for (int q = 0; q < 4; ++q) asm volatile("" ::"v"(ld[q]));
    }
  }
  float s = 0;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int q = 0; q < 4; ++q) s += ld[q][0] + ld[q][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS, int KIND, int G>
void run(int waves_per_simd, const char* name) {
  float* out;
  hipMalloc(&out, 256 * 1024 * sizeof(float));
  const int iters = 4000;
  dim3 block(64 * 4 * waves_per_simd);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<CHAINS, KIND, G>), dim3(256), block, 0, 0, out, iters / 4);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<CHAINS, KIND, G>), dim3(256), block, 0, 0, out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n_mfma = (double)iters * 16;  // per wave
  const double tflops = 256.0 * 4 * waves_per_simd * n_mfma * 32768 / (ms * 1e-3) / 1e12;
  printf("%-5s G=%d chains=%d waves/SIMD=%d: %8.1f us  %7.1f TFLOP/s  %6.2f ns per MFMA per SIMD\n", name, G, CHAINS, waves_per_simd,
         ms * 1e3, tflops, ms * 1e6 / (n_mfma * waves_per_simd));
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<2, NONE, 0>(w, "none");
    run<4, NONE, 0>(w, "none");
    run<4, VALU, 1>(w, "valu");
    run<4, VALU, 2>(w, "valu");
    run<4, VALU, 4>(w, "valu");
    run<4, LDSR, 1>(w, "ldsr");
    run<4, LDSR, 2>(w, "ldsr");
    run<4, LDSW, 1>(w, "ldsw");
  }
  return 0;
}
