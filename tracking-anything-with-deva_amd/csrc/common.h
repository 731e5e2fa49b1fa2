// Shared helpers for libdeva_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "deva_hip.h"

namespace deva {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return 1;
  }
  return 0;
}

#define DEVA_REQUIRE(cond, ...)    \
  do {                             \
    if (!(cond)) {                 \
      deva::set_error(__VA_ARGS__); \
      return 2;                    \
    }                              \
  } while (0)

constexpr int kWave = 64;  // CDNA wavefront

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__host__ __device__ __forceinline__ int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace deva
