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

namespace {

__device__ __forceinline__ float head_gelu(float x) {
  return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
}

__device__ __forceinline__ float head_gelu_grad(float x, float dy) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = expf(-0.5f * x * x) * 0.39894228040143267794f;
  return dy * (cdf + x * pdf);
}

// one thread per element of x2 [B][F + E]
__global__ __launch_bounds__(256) void field_head_fwd_kernel(const float* __restrict__ y1,
                                                             const float* __restrict__ dirs_enc,
                                                             int B, int F, int E,
                                                             float* __restrict__ x2,
                                                             float* __restrict__ density) {
  // 32-bit index arithmetic (the host splits batches of 2^31 elements or more): a 64-bit division
  // per element made this pass 2.4x slower than its memory traffic
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  const unsigned W = (unsigned)(F + E);
  if (t >= (unsigned)B * W) return;
  const unsigned b = t / W;
  const int j = (int)(t - b * W);
  if (j < F) {
    x2[t] = head_gelu(y1[b * (unsigned)(1 + F) + 1 + j]);
  } else {
    x2[t] = dirs_enc[b * (unsigned)E + (j - F)];
  }
  if (j == 0) {
    const float a = y1[b * (unsigned)(1 + F)];
    density[b] = a > 20.0f ? a : log1pf(expf(a));
  }
}

// one thread per element of dy1 [B][1 + F]
__global__ __launch_bounds__(256) void field_head_bwd_kernel(const float* __restrict__ y1,
                                                             const float* __restrict__ dx2,
                                                             const float* __restrict__ d_density,
                                                             int B, int F, int E,
                                                             float* __restrict__ dy1) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  const unsigned W = 1u + (unsigned)F;
  if (t >= (unsigned)B * W) return;
  const unsigned b = t / W;
  const int j = (int)(t - b * W);
  const float y = y1[t];
  if (j == 0) {
    float g = 0.f;
    if (d_density) {
      const float dy = d_density[b];
      const float z = expf(y);
      g = y > 20.0f ? dy : dy * z / (z + 1.0f);
    }
    dy1[t] = g;
  } else {
    dy1[t] = dx2 ? head_gelu_grad(y, dx2[b * (unsigned)(F + E) + (j - 1)]) : 0.f;
  }
}

}  // namespace

// rows per launch: element indices stay below 2^31
static long long head_chunk_rows(int width) { return ((1ll << 31) - 256) / width; }

extern "C" int vsa_field_head_fwd(const float* y1, const float* dirs_enc, long long nr_points,
                                  int nr_feat, int nr_dir, float* x2, float* density, void* stream) {
  if (nr_points < 0 || nr_feat < 1 || nr_dir < 0) return VSA_ERR_ARG;
  if (nr_points == 0) return VSA_OK;
  if (!y1 || !x2 || !density || (nr_dir > 0 && !dirs_enc)) return VSA_ERR_ARG;
  const int W = nr_feat + nr_dir;
  const long long step = head_chunk_rows(W);
  for (long long r0 = 0; r0 < nr_points; r0 += step) {
    const long long n = nr_points - r0 < step ? nr_points - r0 : step;
    hipLaunchKernelGGL(field_head_fwd_kernel, dim3((unsigned)vsa_div_up(n * W, 256)), dim3(256), 0,
                       (hipStream_t)stream, y1 + r0 * (1 + nr_feat), dirs_enc ? dirs_enc + r0 * nr_dir : nullptr,
                       (int)n, nr_feat, nr_dir, x2 + r0 * W, density + r0);
  }
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_field_head_bwd(const float* y1, const float* dx2, const float* d_density,
                                  long long nr_points, int nr_feat, int nr_dir, float* dy1,
                                  void* stream) {
  if (nr_points < 0 || nr_feat < 1 || nr_dir < 0) return VSA_ERR_ARG;
  if (nr_points == 0) return VSA_OK;
  if (!y1 || !dy1) return VSA_ERR_ARG;
  const int W = 1 + nr_feat;
  const long long step = head_chunk_rows(nr_feat + nr_dir);
  for (long long r0 = 0; r0 < nr_points; r0 += step) {
    const long long n = nr_points - r0 < step ? nr_points - r0 : step;
    hipLaunchKernelGGL(field_head_bwd_kernel, dim3((unsigned)vsa_div_up(n * W, 256)), dim3(256), 0,
                       (hipStream_t)stream, y1 + r0 * W, dx2 ? dx2 + r0 * (nr_feat + nr_dir) : nullptr,
                       d_density ? d_density + r0 : nullptr, (int)n, nr_feat, nr_dir, dy1 + r0 * W);
  }
  VSA_RETURN_LAUNCH_STATUS();
}
