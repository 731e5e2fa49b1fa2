// How many vector-memory wave-instructions per microsecond does one CU sustain, by load width?  (Round 5: the split
// convolution kernels gather activations with 4-byte buffer loads, 64 lanes x 4 B per instruction; is the cost of a
// load its instruction or its bytes?)  Every wave streams over a small L2-resident window with coalesced loads of
// 4 / 8 / 16 bytes per lane, 8 independent loads in flight per iteration.  Prints wave-instructions per us per CU, bytes
// per clock per CU (at 2.4 GHz) for 1, 2, 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/vmem_width_probe.hip -o vmem_width_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__global__ __launch_bounds__(1024) void probe(const float* __restrict__ src, float* out, int iters, int window_floats) {
  constexpr int W = sizeof(T) / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // every wave reads rows of 64 lanes x W floats; 8 rows per iteration, rows stride through the window
  const float* base = src + ((size_t)blockIdx.x * 16 + wave) * 64 * W * 8 % window_floats;
  T acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = T{};
  int off = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const T v = *reinterpret_cast<const T*>(base + (off + i * 64 * W * 16) % (window_floats / 2) + lane * W);
      acc[i] += v;
    }
    off = (off + 64 * W * 128) % (window_floats / 2);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < W; ++j) s += reinterpret_cast<float*>(&acc[i])[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename T>
void run(const float* src, float* out, int waves_per_simd, int window_floats, const char* name) {
  const int iters = 4000;
  dim3 block(64 * 4 * waves_per_simd);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<T>), dim3(256), block, 0, 0, src, out, iters / 8, window_floats);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<T>), dim3(256), block, 0, 0, src, out, iters, window_floats);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_cu = (double)iters * 8 * 4 * waves_per_simd;
  const double bytes_per_cu = instr_per_cu * 64 * sizeof(T);
  printf("%-8s waves/SIMD=%d window %5.1f MB: %8.1f us  %7.1f wave-instr/us/CU  %6.1f B/clk/CU  %6.2f TB/s chip\n", name, waves_per_simd,
         window_floats * 4e-6, ms * 1e3, instr_per_cu / (ms * 1e3), bytes_per_cu / (ms * 1e-3) / 2.4e9, bytes_per_cu * 256 / (ms * 1e-3) / 1e12);
}

int main() {
  const int n = 64 << 20;  // 256 MB
  float *src, *out;
  hipMalloc(&src, (size_t)n * 4);
  hipMalloc(&out, 256 * 1024 * 4);
  hipMemset(src, 0, (size_t)n * 4);
  for (int window : {1 << 18, 1 << 23}) {  // 1 MB (L2 resident), 32 MB (beyond the L2s, inside the Infinity Cache)
    for (int w : {2, 4}) {
      run<float>(src, out, w, window, "dword");
      run<f32x2>(src, out, w, window, "dwordx2");
      run<f32x4>(src, out, w, window, "dwordx4");
    }
  }
  return 0;
}
