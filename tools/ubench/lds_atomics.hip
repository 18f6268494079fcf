// Micro-benchmark: LDS scatter throughput on gfx950 (random addresses over a
// 128 KiB table, 1024-thread workgroups, one per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(1024) void k(const unsigned* __restrict__ idx, float* out, int iters) {
  extern __shared__ float s[];
  for (int i = threadIdx.x; i < 32768; i += 1024) s[i] = 0.f;
  __syncthreads();
  const unsigned* p = idx + (size_t)blockIdx.x * 1024 * iters + threadIdx.x;
  for (int it = 0; it < iters; it += 4) {
    unsigned a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = p[(size_t)(it + u) * 1024];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 0) atomicAdd(&s[a[u] & 32767], 1.0f);                       // ds_add_f32
      if (MODE == 1) s[a[u] & 32767] = 1.0f;                                  // ds_write_b32
      if (MODE == 2) atomicAdd(reinterpret_cast<unsigned*>(&s[a[u] & 32767]), 1u);  // ds_add_u32
      if (MODE == 3) { float v = s[a[u] & 32767]; s[(a[u] >> 15) & 32767] = v + 1.0f; }
    }
  }
  __syncthreads();
  float acc = 0;
  for (int i = threadIdx.x; i < 32768; i += 1024) acc += s[i];
  if (acc == 12345.f) out[0] = acc;
}
template <int MODE> void run(const char* name, unsigned* d_idx, float* d_out, int iters) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 131072, 0, d_idx, d_out, iters);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 131072, 0, d_idx, d_out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double ops = 256.0 * 1024 * iters;
  printf("%-14s %8.3f ms  %7.1f Gops/s  %.2f lanes/clk/CU (2.1GHz)\n", name, ms, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.1);
}
int main() {
  const int iters = 256;
  size_t n = 256ull * 1024 * iters;
  std::vector<unsigned> h(n);
  unsigned x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x >> 2; }
  unsigned* d_idx; float* d_out;
  hipMalloc(&d_idx, n * 4); hipMalloc(&d_out, 4);
  hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice);
  run<0>("ds_add_f32", d_idx, d_out, iters);
  run<1>("ds_write_b32", d_idx, d_out, iters);
  run<2>("ds_add_u32", d_idx, d_out, iters);
  run<3>("read+write", d_idx, d_out, iters);
  return 0;
}
