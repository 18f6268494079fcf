// Micro-benchmark: LDS random scatter/gather throughput on gfx950 (addresses
// generated in registers, 128 KiB table, 1024-thread workgroups, one per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); return; } } while (0)
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, unsigned mask) {
  extern __shared__ float s[];
  for (int i = threadIdx.x; i < 32768; i += 1024) s[i] = 0.f;
  __syncthreads();
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      x = x * 1664525u + 1013904223u;
      const unsigned a = (x >> 10) & mask;
      if (MODE == 0) atomicAdd(&s[a], 1.0f);                                   // ds_add_f32
      if (MODE == 1) s[a] = 1.0f;                                              // ds_write_b32
      if (MODE == 2) atomicAdd(reinterpret_cast<int*>(&s[a]), 3);              // ds_add_u32
      if (MODE == 3) acc += s[a];                                              // ds_read_b32
      if (MODE == 4) atomicAdd(reinterpret_cast<unsigned long long*>(&s[a & ~1u]), 3ull);  // ds_add_u64
      if (MODE == 5) {   // 64-bit add as two 32-bit limbs: returned low limb -> carry into the high one
        unsigned* lo = reinterpret_cast<unsigned*>(&s[a & 16383u]);
        const unsigned v = x | 1u;
        const unsigned old = atomicAdd(lo, v);                                 // ds_add_rtn_u32
        atomicAdd(lo + 16384, (x >> 31) + (old + v < old ? 1u : 0u));          // ds_add_u32
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 32768; i += 1024) acc += s[i];
  if (acc == 12345.f) out[0] = acc;
}
template <int MODE> void run(const char* name, float* d_out, int iters, unsigned mask) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 131072, 0, d_out, iters, mask);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 131072, 0, d_out, iters, mask);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double ops = 256.0 * 1024 * iters * 8;
  printf("%-13s mask=%5u %8.3f ms %8.1f Gops/s %6.2f lanes/clk/CU @2.1GHz\n", name, mask, ms, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.1);
}
int main() {
  float* d_out; if (hipMalloc(&d_out, 4) != hipSuccess) return 1;
  const unsigned masks[3] = {32767u, 255u, 3u};
  for (unsigned m : masks) {
    run<0>("ds_add_f32", d_out, 64, m);
    run<1>("ds_write_b32", d_out, 64, m);
    run<2>("ds_add_u32", d_out, 64, m);
    run<3>("ds_read_b32", d_out, 64, m);
    run<4>("ds_add_u64", d_out, 64, m);
    run<5>("2x32 carry", d_out, 64, m);
  }
  return 0;
}
