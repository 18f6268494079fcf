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

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef VSA_ADAM_NT
#define VSA_ADAM_NT 1
#endif
#if VSA_ADAM_NT
#define ADAM_LD(ptr) __builtin_nontemporal_load(ptr)
#define ADAM_ST(val, ptr) __builtin_nontemporal_store(val, ptr)
#else
#define ADAM_LD(ptr) (*(ptr))
#define ADAM_ST(val, ptr) (*(ptr) = (val))
#endif
constexpr int ADAM_BLOCK = 256;
constexpr int ADAM_VEC = 4;                                  // one dwordx4 per array per lane
constexpr int ADAM_CHUNK = ADAM_BLOCK * ADAM_VEC * 4;        // 4096 elements per workgroup

struct AdamCoef {
  float step_size, beta1, beta2, inv_bc2_sqrt, eps, grad_scale;
};

// the update of ADAM_VEC entries in registers; returns whether all of them are idle (see below)
__device__ __forceinline__ bool adam_update(float (&p)[ADAM_VEC], const float (&g)[ADAM_VEC], float (&m)[ADAM_VEC],
                                            float (&v)[ADAM_VEC], const AdamCoef& c) {
  // An entry that has never received a gradient (g = m = v = 0: most of a hash table's coarse
  // levels, and every slot no sample has hashed to yet) stays exactly as it is — m' = v' = 0 and
  // p' = p - step * (0 / eps) = p — so none of its five stores is issued: 16 B instead of 34 B
  // of traffic for it.
  bool idle = true;
#pragma unroll
  for (int k = 0; k < ADAM_VEC; ++k) idle = idle && g[k] == 0.f && m[k] == 0.f && v[k] == 0.f;
  const float omb1 = 1.0f - c.beta1, omb2 = 1.0f - c.beta2;
#pragma unroll
  for (int k = 0; k < ADAM_VEC; ++k) {
    const float gk = g[k] * c.grad_scale;
    m[k] = m[k] + omb1 * (gk - m[k]);
    v[k] = v[k] * c.beta2 + (omb2 * gk) * gk;
    const float denom = sqrtf(v[k]) * c.inv_bc2_sqrt + c.eps;
    p[k] = p[k] - c.step_size * (m[k] / denom);
  }
  return idle;
}

// (non-temporal accesses: 28 B per parameter stream through once; marked for early eviction they leave the L2
//  to whatever runs next — the following iteration's traversal lives on its BVH staying cached: the NT training
//  loop went 1 180-1 207 -> 1 259-1 270 it/s with them, same box)
__device__ __forceinline__ void adam_store4(const vsa_adam_tensor& t, long long i, const float (&p)[ADAM_VEC],
                                            const float (&m)[ADAM_VEC], const float (&v)[ADAM_VEC], int zero_grads) {
  ADAM_ST(*reinterpret_cast<const f32x4*>(p), reinterpret_cast<f32x4*>(t.param + i));
  ADAM_ST(*reinterpret_cast<const f32x4*>(m), reinterpret_cast<f32x4*>(t.exp_avg + i));
  ADAM_ST(*reinterpret_cast<const f32x4*>(v), reinterpret_cast<f32x4*>(t.exp_avg_sq + i));
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  if (zero_grads) ADAM_ST(zero, reinterpret_cast<f32x4*>(t.grad + i));
  if (t.param_f16) {
    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
    half4_t h = {(_Float16)p[0], (_Float16)p[1], (_Float16)p[2], (_Float16)p[3]};
    *reinterpret_cast<half4_t*>(static_cast<_Float16*>(t.param_f16) + i) = h;
  }
}

__global__ __launch_bounds__(ADAM_BLOCK) void adam_kernel(const vsa_adam_tensor* __restrict__ tensors,
                                                          const int2* __restrict__ chunks, AdamCoef coef,
                                                          int zero_grads, int nr_chunks,
                                                          const vsa_train_ctl* __restrict__ ctl) {
  if (ctl) {       // the graph-replayed iteration (vsa_adam_step_ctl): lr and step count from the control block
    const int step = ctl->adam_step;
    if (!ctl->adam_pending || step < 1) return;       // nothing pending (the first replay)
    const double bc1 = 1.0 - pow((double)coef.beta1, (double)step);
    const double bc2 = 1.0 - pow((double)coef.beta2, (double)step);
    coef.step_size = (float)((double)ctl->adam_lr / bc1);
    coef.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  }
  // one chunk per workgroup, or (vsa_adam_step_shared) a bounded grid whose workgroups stride over the chunks
  for (int c = blockIdx.x; c < nr_chunks; c += gridDim.x) {
    const int2 ck = chunks[c];
    const vsa_adam_tensor t = tensors[ck.x];
    const long long base = (long long)ck.y * ADAM_CHUNK;
    if (base + ADAM_CHUNK <= t.n) {
      // a whole chunk (all but the last of a tensor): its sixteen 16-byte loads per lane are requested before the
      // first one is used — trip by trip, every trip's loads queued behind the previous trip's five stores
      float p[4][ADAM_VEC], g[4][ADAM_VEC], m[4][ADAM_VEC], v[4][ADAM_VEC];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long i = base + ((long long)r * ADAM_BLOCK + threadIdx.x) * ADAM_VEC;
        *reinterpret_cast<f32x4*>(p[r]) = ADAM_LD(reinterpret_cast<const f32x4*>(t.param + i));
        *reinterpret_cast<f32x4*>(g[r]) = ADAM_LD(reinterpret_cast<const f32x4*>(t.grad + i));
        *reinterpret_cast<f32x4*>(m[r]) = ADAM_LD(reinterpret_cast<const f32x4*>(t.exp_avg + i));
        *reinterpret_cast<f32x4*>(v[r]) = ADAM_LD(reinterpret_cast<const f32x4*>(t.exp_avg_sq + i));
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long i = base + ((long long)r * ADAM_BLOCK + threadIdx.x) * ADAM_VEC;
        if (!adam_update(p[r], g[r], m[r], v[r], coef)) adam_store4(t, i, p[r], m[r], v[r], zero_grads);
      }
      continue;
    }
    _Float16* const p16 = static_cast<_Float16*>(t.param_f16);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long i = base + ((long long)r * ADAM_BLOCK + threadIdx.x) * ADAM_VEC;
      if (i >= t.n) break;
      float p[ADAM_VEC], g[ADAM_VEC], m[ADAM_VEC], v[ADAM_VEC];
      const bool full = i + ADAM_VEC <= t.n;      // tensors are 16-byte aligned; only the tail is ragged
#pragma unroll
      for (int k = 0; k < ADAM_VEC; ++k) {
        const bool in = i + k < t.n;
        p[k] = in ? t.param[i + k] : 0.f;
        g[k] = in ? t.grad[i + k] : 0.f;
        m[k] = in ? t.exp_avg[i + k] : 0.f;
        v[k] = in ? t.exp_avg_sq[i + k] : 0.f;
      }
      if (adam_update(p, g, m, v, coef)) continue;       // nothing to write
      if (full) {
        adam_store4(t, i, p, m, v, zero_grads);
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
  }   // chunks of this workgroup
}

}  // namespace

extern "C" int vsa_adam_chunk_elems(void) { return ADAM_CHUNK; }

extern "C" int vsa_adam_step_shared(const vsa_adam_tensor* tensors_dev, const int32_t* chunks_dev,
                                    int nr_chunks, float lr, float beta1, float beta2, float eps,
                                    int step, float grad_scale, int zero_grads, int max_workgroups,
                                    void* stream) {
  if (max_workgroups < 0) return VSA_ERR_ARG;
  if (nr_chunks < 0 || step < 1 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f))
    return VSA_ERR_ARG;
  if (nr_chunks == 0) return VSA_OK;
  if (!tensors_dev || !chunks_dev) return VSA_ERR_ARG;
  // bias corrections in double, as torch.optim.Adam forms them on the host
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  const int grid = max_workgroups > 0 && max_workgroups < nr_chunks ? max_workgroups : nr_chunks;
  const AdamCoef coef = {step_size, beta1, beta2, inv_bc2_sqrt, eps, grad_scale};
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(ADAM_BLOCK), 0, (hipStream_t)stream,
                     tensors_dev, reinterpret_cast<const int2*>(chunks_dev), coef, zero_grads, nr_chunks,
                     (const vsa_train_ctl*)nullptr);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_adam_step_ctl(const vsa_adam_tensor* tensors_dev, const int32_t* chunks_dev, int nr_chunks,
                                 float beta1, float beta2, float eps, float grad_scale, int zero_grads,
                                 int max_workgroups, const vsa_train_ctl* ctl, void* stream) {
  if (max_workgroups < 0 || !ctl) return VSA_ERR_ARG;
  if (nr_chunks < 0 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) return VSA_ERR_ARG;
  if (nr_chunks == 0) return VSA_OK;
  if (!tensors_dev || !chunks_dev) return VSA_ERR_ARG;
  const int grid = max_workgroups > 0 && max_workgroups < nr_chunks ? max_workgroups : nr_chunks;
  const AdamCoef coef = {0.f, beta1, beta2, 0.f, eps, grad_scale};       // step_size / inv_bc2_sqrt: formed on the device
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(ADAM_BLOCK), 0, (hipStream_t)stream, tensors_dev,
                     reinterpret_cast<const int2*>(chunks_dev), coef, zero_grads, nr_chunks, ctl);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_adam_step(const vsa_adam_tensor* tensors_dev, const int32_t* chunks_dev,
                             int nr_chunks, float lr, float beta1, float beta2, float eps,
                             int step, float grad_scale, int zero_grads, void* stream) {
  return vsa_adam_step_shared(tensors_dev, chunks_dev, nr_chunks, lr, beta1, beta2, eps, step, grad_scale,
                              zero_grads, 0, stream);
}
