// Fused multi-tensor Adam step: the reference trains with
// apex.optimizers.FusedAdam(betas (0.9, 0.99), eps 1e-15, weight_decay 0, amsgrad False) over
// every model's parameters (/root/reference/volsurfs_py/methods/base_method.py:87-94), stepped
// once per iteration (trainer.py:278) — the same update as torch.optim.Adam, which the tests pin
// it against.  One launch covers ALL parameter tensors of a method:
//   m  = m + (1 - beta1) * (g - m)
//   v  = beta2 * v + (1 - beta2) * g * g
//   p -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// fused with what otherwise are separate passes over the same 100+ MB: the refresh of the
// f16 compute copy of a parameter (tiny-cuda-nn keeps the same fp32 master / fp16 pair) and
// the zeroing of the gradient for the next iteration (trainer.py:118 zero_grad).
// HBM-bound: 16 B read + 12 B written per parameter (+2 B f16 copy, +4 B gradient clear).
#include "common.h"

namespace {

constexpr int ADAM_BLOCK = 256;
constexpr int ADAM_VEC = 4;                                  // one dwordx4 per array per lane
constexpr int ADAM_CHUNK = ADAM_BLOCK * ADAM_VEC * 4;        // 4096 elements per workgroup

__global__ __launch_bounds__(ADAM_BLOCK) void adam_kernel(const vsa_adam_tensor* __restrict__ tensors,
                                                          const int2* __restrict__ chunks,
                                                          float step_size, float beta1, float beta2,
                                                          float inv_bc2_sqrt, float eps,
                                                          float grad_scale, int zero_grads) {
  const int2 ck = chunks[blockIdx.x];
  const vsa_adam_tensor t = tensors[ck.x];
  _Float16* const p16 = static_cast<_Float16*>(t.param_f16);
  const long long base = (long long)ck.y * ADAM_CHUNK;
  const float omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long long i = base + ((long long)r * ADAM_BLOCK + threadIdx.x) * ADAM_VEC;
    if (i >= t.n) break;
    float p[ADAM_VEC], g[ADAM_VEC], m[ADAM_VEC], v[ADAM_VEC];
    const bool full = i + ADAM_VEC <= t.n;      // tensors are 16-byte aligned; only the tail is ragged
    if (full) {
      *reinterpret_cast<float4*>(p) = *reinterpret_cast<const float4*>(t.param + i);
      *reinterpret_cast<float4*>(g) = *reinterpret_cast<const float4*>(t.grad + i);
      *reinterpret_cast<float4*>(m) = *reinterpret_cast<const float4*>(t.exp_avg + i);
      *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(t.exp_avg_sq + i);
    } else {
#pragma unroll
      for (int k = 0; k < ADAM_VEC; ++k) {
        const bool in = i + k < t.n;
        p[k] = in ? t.param[i + k] : 0.f;
        g[k] = in ? t.grad[i + k] : 0.f;
        m[k] = in ? t.exp_avg[i + k] : 0.f;
        v[k] = in ? t.exp_avg_sq[i + k] : 0.f;
      }
    }
    // An entry that has never received a gradient (g = m = v = 0: most of a hash table's coarse
    // levels, and every slot no sample has hashed to yet) stays exactly as it is — m' = v' = 0 and
    // p' = p - step * (0 / eps) = p — so none of its five stores is issued: 16 B instead of 34 B
    // of traffic for it.
    bool idle = true;
#pragma unroll
    for (int k = 0; k < ADAM_VEC; ++k) idle = idle && g[k] == 0.f && m[k] == 0.f && v[k] == 0.f;
#pragma unroll
    for (int k = 0; k < ADAM_VEC; ++k) {
      const float gk = g[k] * grad_scale;
      m[k] = m[k] + omb1 * (gk - m[k]);
      v[k] = v[k] * beta2 + (omb2 * gk) * gk;
      const float denom = sqrtf(v[k]) * inv_bc2_sqrt + eps;
      p[k] = p[k] - step_size * (m[k] / denom);
    }
    if (idle) {
      // nothing to write
    } else if (full) {
      *reinterpret_cast<float4*>(t.param + i) = *reinterpret_cast<const float4*>(p);
      *reinterpret_cast<float4*>(t.exp_avg + i) = *reinterpret_cast<const float4*>(m);
      *reinterpret_cast<float4*>(t.exp_avg_sq + i) = *reinterpret_cast<const float4*>(v);
      if (zero_grads) *reinterpret_cast<float4*>(t.grad + i) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p16) {
        typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
        half4_t h = {(_Float16)p[0], (_Float16)p[1], (_Float16)p[2], (_Float16)p[3]};
        *reinterpret_cast<half4_t*>(p16 + i) = h;
      }
    } else {
#pragma unroll
      for (int k = 0; k < ADAM_VEC; ++k) {
        if (i + k >= t.n) break;
        t.param[i + k] = p[k];
        t.exp_avg[i + k] = m[k];
        t.exp_avg_sq[i + k] = v[k];
        if (zero_grads) t.grad[i + k] = 0.f;
        if (p16) p16[i + k] = (_Float16)p[k];
      }
    }
  }
}

}  // namespace

extern "C" int vsa_adam_chunk_elems(void) { return ADAM_CHUNK; }

extern "C" int vsa_adam_step(const vsa_adam_tensor* tensors_dev, const int32_t* chunks_dev,
                             int nr_chunks, float lr, float beta1, float beta2, float eps,
                             int step, float grad_scale, int zero_grads, void* stream) {
  if (nr_chunks < 0 || step < 1 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f))
    return VSA_ERR_ARG;
  if (nr_chunks == 0) return VSA_OK;
  if (!tensors_dev || !chunks_dev) return VSA_ERR_ARG;
  // bias corrections in double, as torch.optim.Adam forms them on the host
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  hipLaunchKernelGGL(adam_kernel, dim3(nr_chunks), dim3(ADAM_BLOCK), 0, (hipStream_t)stream,
                     tensors_dev, reinterpret_cast<const int2*>(chunks_dev), step_size, beta1, beta2,
                     inv_bc2_sqrt, eps, grad_scale, zero_grads);
  VSA_RETURN_LAUNCH_STATUS();
}
