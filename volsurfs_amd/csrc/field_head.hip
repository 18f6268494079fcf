// The elementwise glue between NerfHash's two MLPs (SURVEY §8a row A10,
// /root/reference/volsurfs_py/models/nerfhash.py:72-91):
//     density  = softplus(y1[:, 0:1])
//     x2       = cat([gelu(y1[:, 1:1+F]), dirs_enc], 1)
// and its backward, each as ONE pass.  As torch ops this is a strided slice copy, a GELU, a cat, a
// softplus forward and, backward, two zero fills, two slice copies, an add, a GELU backward and a
// softplus backward — eleven launches over [samples, 64..80] floats (2.3 ms of a 2.1 M-sample
// background batch, tools/bench_bg.py).  The arithmetic is torch's, term for term:
//     GELU      x * 0.5 * (1 + erf(x / sqrt 2))                       (activation.cpp GeluKernel, 'none')
//     GELU'     dy * (cdf + x * pdf), pdf = exp(-x^2 / 2) / sqrt(2 pi)
//     softplus  x > 20 ? x : log1p(exp(x));  softplus' = x > 20 ? dy : dy * z / (z + 1), z = exp(x)
#include "common.h"
#include "gelu_fast.h"

namespace {

// (Phi to 6.6e-8 absolute in ~13 instructions, the same evaluation as the MLP kernels': gelu_fast.h)
__device__ __forceinline__ float head_gelu(float x) { return gelu_fast(x); }

__device__ __forceinline__ float head_gelu_grad(float x, float dy) {
  float cdf, pdf;
  gelu_cdf_pdf(x, cdf, pdf);
  return dy * (cdf + x * pdf);
}

// A wave walks rows: its lanes are the feature columns (64 per pass), so a row is one contiguous
// load, one GELU per lane with no column arithmetic, one contiguous store; then the first E lanes
// copy the encoded direction.  (One thread per element of x2 spent more instructions on the 64-bit
// row / column division and on waves that mix GELU and copy lanes than on the GELU: 850 us for
// 2.1 M rows against 340 us of traffic.)
constexpr int FH_BLOCK = 256;
#ifndef FH_ROWS_PER_TRIP
#define FH_ROWS_PER_TRIP 8
#endif
constexpr int FH_ROWS = FH_ROWS_PER_TRIP;

__global__ __launch_bounds__(FH_BLOCK) void field_head_fwd_kernel(const float* __restrict__ y1,
                                                                 const float* __restrict__ dirs_enc,
                                                                 long long B, int F, int E, int S1,
                                                                 float* __restrict__ x2,
                                                                 float* __restrict__ density) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * FH_BLOCK + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * FH_BLOCK) >> 6;
  const int W = F + E;
  // FH_ROWS rows per trip, all their loads issued before the first GELU: with one 4-byte load in
  // flight per lane the 32 waves of a CU keep 8 KiB on the way and the pass ran at 1.6 TB/s
  for (long long b0 = wave * FH_ROWS; b0 < B; b0 += nwaves * FH_ROWS) {
    for (int j = lane; j < F; j += 64) {
      float v[FH_ROWS];
#pragma unroll
      for (int r = 0; r < FH_ROWS; ++r) v[r] = b0 + r < B ? y1[(b0 + r) * S1 + 1 + j] : 0.f;
#pragma unroll
      for (int r = 0; r < FH_ROWS; ++r)
        if (b0 + r < B) x2[(b0 + r) * W + j] = head_gelu(v[r]);
    }
    for (int j = lane; j < E; j += 64) {
      float v[FH_ROWS];
#pragma unroll
      for (int r = 0; r < FH_ROWS; ++r) v[r] = b0 + r < B ? dirs_enc[(b0 + r) * E + j] : 0.f;
#pragma unroll
      for (int r = 0; r < FH_ROWS; ++r)
        if (b0 + r < B) x2[(b0 + r) * W + F + j] = v[r];
    }
    if (lane < FH_ROWS && b0 + lane < B) {
      const float a = y1[(b0 + lane) * S1];
      density[b0 + lane] = a > 20.0f ? a : log1pf(expf(a));
    }
  }
}

__global__ __launch_bounds__(FH_BLOCK) void field_head_bwd_kernel(const float* __restrict__ y1,
                                                                 const float* __restrict__ dx2,
                                                                 const float* __restrict__ d_density,
                                                                 long long B, int F, int E, int S1,
                                                                 float* __restrict__ dy1) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * FH_BLOCK + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * FH_BLOCK) >> 6;
  const int W = F + E;
  for (long long b0 = wave * FH_ROWS; b0 < B; b0 += nwaves * FH_ROWS) {
    for (int j = lane; j < F; j += 64) {
      float v[FH_ROWS], d[FH_ROWS];
#pragma unroll
      for (int r = 0; r < FH_ROWS; ++r) {
        const bool in = b0 + r < B;
        v[r] = in ? y1[(b0 + r) * S1 + 1 + j] : 0.f;
        d[r] = in && dx2 ? dx2[(b0 + r) * W + j] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < FH_ROWS; ++r)
        if (b0 + r < B) dy1[(b0 + r) * S1 + 1 + j] = dx2 ? head_gelu_grad(v[r], d[r]) : 0.f;
    }
    if (lane < FH_ROWS && b0 + lane < B) {
      const long long b = b0 + lane;
      float g = 0.f;
      if (d_density) {
        const float y = y1[b * S1], dy = d_density[b];
        const float z = expf(y);
        g = y > 20.0f ? dy : dy * z / (z + 1.0f);
      }
      dy1[b * S1] = g;
      // (padding columns of a wider row: zero, so that whoever reads the row as 16-byte groups sees no garbage)
      for (int c = 1 + F; c < S1; ++c) dy1[b * S1 + c] = 0.f;
    }
  }
}

}  // namespace

// persistent-ish grid: enough waves to fill the chip, each walking rows with a grid stride
static int head_grid(long long rows) {
  long long blocks = (rows + (FH_BLOCK / 64) * FH_ROWS - 1) / ((FH_BLOCK / 64) * FH_ROWS);
  if (blocks > 256 * 16) blocks = 256 * 16;
  return (int)(blocks < 1 ? 1 : blocks);
}

extern "C" int vsa_field_head_fwd(const float* y1, int y1_stride, const float* dirs_enc, long long nr_points,
                                  int nr_feat, int nr_dir, float* x2, float* density, void* stream) {
  if (nr_points < 0 || nr_feat < 1 || nr_dir < 0 || y1_stride < nr_feat + 1) return VSA_ERR_ARG;
  if (nr_points == 0) return VSA_OK;
  if (!y1 || !x2 || !density || (nr_dir > 0 && !dirs_enc)) return VSA_ERR_ARG;
  hipLaunchKernelGGL(field_head_fwd_kernel, dim3(head_grid(nr_points)), dim3(FH_BLOCK), 0, (hipStream_t)stream,
                     y1, dirs_enc, nr_points, nr_feat, nr_dir, y1_stride, x2, density);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_field_head_bwd(const float* y1, int y1_stride, const float* dx2, const float* d_density,
                                  long long nr_points, int nr_feat, int nr_dir, float* dy1,
                                  void* stream) {
  if (nr_points < 0 || nr_feat < 1 || nr_dir < 0 || y1_stride < nr_feat + 1) return VSA_ERR_ARG;
  if (nr_points == 0) return VSA_OK;
  if (!y1 || !dy1) return VSA_ERR_ARG;
  hipLaunchKernelGGL(field_head_bwd_kernel, dim3(head_grid(nr_points)), dim3(FH_BLOCK), 0, (hipStream_t)stream,
                     y1, dx2, d_density, nr_points, nr_feat, nr_dir, y1_stride, dy1);
  VSA_RETURN_LAUNCH_STATUS();
}
