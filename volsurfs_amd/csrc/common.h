// Shared helpers for the volsurfs_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VSA_WAVE 64

#include "../../include/volsurfs_hip.h"

// Launch-status helper: every C-ABI entry point returns 0 or the hipError_t of
// its launch (the reference only *prints* launch errors, src/VolumeRendering.cu:64-75;
// this library surfaces them to the caller, which raises).
#define VSA_RETURN_LAUNCH_STATUS()            \
  do {                                        \
    hipError_t e__ = hipGetLastError();       \
    return e__ == hipSuccess ? VSA_OK : (int)e__; \
  } while (0)

#define VSA_HIP_TRY(expr)                     \
  do {                                        \
    hipError_t e__ = (expr);                  \
    if (e__ != hipSuccess) return (int)e__;   \
  } while (0)

typedef _Float16 half_t;

__device__ __forceinline__ float vsa_round_f16(float x) { return (float)(half_t)x; }

// Makes an fp32 value opaque to the optimiser.  hipcc folds
// `(half)(f32_a * f32_b)` into v_fma_mixlo_f16, which rounds the exact product
// ONCE to fp16; ATen rounds to fp32 first and then to fp16.  Where bit parity
// with that double rounding matters, pin the fp32 intermediate with this.
__device__ __forceinline__ float vsa_pin_f32(float x) {
  asm volatile("" : "+v"(x));
  return x;
}

static inline int vsa_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// Compute units of the current device (grid size of the persistent kernels).
static inline int vsa_cu_count(int* out) {
  static int cached = 0;
  if (!cached) {
    int dev = 0, n = 0;
    VSA_HIP_TRY(hipGetDevice(&dev));
    VSA_HIP_TRY(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cached = n > 0 ? n : 256;
  }
  *out = cached;
  return VSA_OK;
}
