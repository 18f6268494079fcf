// VALU issue rate on gfx950 by instruction kind and waves per SIMD (1, 2, 4): what "VALU-busy"
// can reach.  Every wave runs ITER trips of 16 independent instructions of one kind; the
// elapsed shader clock (s_memtime) of the slowest wave / (instructions issued per SIMD) =
// cycles per wave-instruction per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define ITER 4096

#define BODY16(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(8) INS(9) INS(10) INS(11) INS(12) INS(13) INS(14) INS(15)

template <int KIND>
__global__ void k(unsigned long long* out, float seed) {
  float r[16];
  unsigned u[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { r[i] = seed + i + threadIdx.x; u[i] = (unsigned)(threadIdx.x * 2654435761u + i); }
  float a = seed * 1.0001f, b = seed * 0.5f;
  unsigned ua = (unsigned)threadIdx.x | 1u;
  unsigned long long t0, t1;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
      if (KIND == 1) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(u[i]) : "v"(ua));
      if (KIND == 2) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(u[i]) : "v"(ua));
      if (KIND == 3) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(r[i]));
      if (KIND == 4) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(u[i]) : "v"(ua), "v"(ua));
      if (KIND == 5) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(r[i]) : "v"(ua), "v"(b));
      if (KIND == 6) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(*(reinterpret_cast<double*>(&r[i & ~1]))) : "v"(*(reinterpret_cast<double*>(&r[(i & ~1) ^ 2]))));
      if (KIND == 7) asm volatile("v_cvt_flr_i32_f32 %0, %0" : "+v"(r[i]));
      if (KIND == 8) asm volatile("v_fract_f32 %0, %0" : "+v"(r[i]));
      if (KIND == 9) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[i]) : "v"(ua));
      if (KIND == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(ua));
      if (KIND == 11) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[i]) : "v"(ua));
      if (KIND == 12) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(ua));
      if (KIND == 13) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
      if (KIND == 14) asm volatile("v_cvt_rpi_i32_f32 %0, %0" : "+v"(r[i]));
      if (KIND == 15) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0; unsigned su = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { s += r[i]; su ^= u[i]; }
  if (s == 12345.678f && su == 77) out[0] = 1;   // keep the registers live
  if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
void run(const char* name, int cus) {
  unsigned long long* d;
  hipMalloc(&d, 8 * (1 + cus * 16));
  printf("%-22s", name);
  for (int wps : {1, 2, 4}) {
    const int threads = 256 * wps;
    hipMemset(d, 0, 8 * (1 + cus * 16));
    hipLaunchKernelGGL(k<KIND>, dim3(cus), dim3(threads), 0, 0, d, 1.25f);
    hipLaunchKernelGGL(k<KIND>, dim3(cus), dim3(threads), 0, 0, d, 1.25f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(1 + cus * 16);
    hipMemcpy(h.data(), d, 8 * h.size(), hipMemcpyDeviceToHost);
    std::vector<unsigned long long> v(h.begin() + 1, h.begin() + 1 + cus * (threads / 64));
    std::sort(v.begin(), v.end());
    const double med = (double)v[v.size() / 2];
    printf("  wps=%d: %6.2f cyc/instr/SIMD (wave sees %6.2f)", wps, med / (ITER * 16.0 * wps), med / (ITER * 16.0));
  }
  printf("\n");
  hipFree(d);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("%s, %d CUs; s_memtime ticks are shader cycles\n", p.gcnArchName, cus);
  run<0>("v_fma_f32", cus);
  run<1>("v_xor_b32", cus);
  run<9>("v_and_b32", cus);
  run<11>("v_add_u32", cus);
  run<15>("v_lshlrev_b32", cus);
  run<12>("v_mov_b32", cus);
  run<10>("v_cndmask_b32", cus);
  run<2>("v_mul_lo_u32", cus);
  run<3>("v_cvt_f16_f32", cus);
  run<7>("v_cvt_flr_i32_f32", cus);
  run<14>("v_cvt_rpi_i32_f32", cus);
  run<8>("v_fract_f32", cus);
  run<4>("v_pk_fma_f16", cus);
  run<5>("v_fma_mix_f32", cus);
  run<6>("v_pk_mul_f32", cus);
  run<13>("v_rcp_f32", cus);
  return 0;
}
