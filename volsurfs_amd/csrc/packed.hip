// Packed (ragged) per-ray sample ops of the background path (SURVEY §8a rows A8,
// A9, A11): the subset of the reference's `volsurfs` pybind module that
// VolSurfs.render_rays reaches through render_contracted_bg
// (volsurfs_py/utils/background.py:31-141).
//
// The reference runs ONE THREAD PER RAY with a serial loop over the ray's samples
// (kernels/volsurfs/VolumeRenderingGPU.cuh, RaySamplerGPU.cuh,
// RaySamplesPackedGPU.cuh; lane stride = samples_per_ray * 4 B, i.e. uncoalesced).
// Here a ray is owned by a 32-lane half-wave: lanes = consecutive samples
// (coalesced 128-B rows), segmented scans / reductions via DPP shuffles, chunks of
// 32 samples with a carried running value for longer rays (the bg path has
// exactly 32 samples per ray: background.py:51-58, hyper_params.py:65).
#include "common.h"
#include "pcg32.h"

namespace {

constexpr int PK_BLOCK = 256;
constexpr int SUB = 32;  // lanes per ray

__device__ __forceinline__ float sub_scan_mul(float v, int l) {
#pragma unroll
  for (int off = 1; off < SUB; off <<= 1) {
    const float u = __shfl_up(v, off, SUB);
    if (l >= off) v *= u;
  }
  return v;
}
__device__ __forceinline__ float sub_scan_add(float v, int l) {
#pragma unroll
  for (int off = 1; off < SUB; off <<= 1) {
    const float u = __shfl_up(v, off, SUB);
    if (l >= off) v += u;
  }
  return v;
}
__device__ __forceinline__ float sub_reduce_add(float v) {
#pragma unroll
  for (int off = SUB / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, SUB);
  return v;
}

#define PK_RAY_PROLOGUE()                                                        \
  const int l = threadIdx.x & (SUB - 1);                                         \
  const long long ray = ((long long)blockIdx.x * PK_BLOCK + threadIdx.x) / SUB;  \
  if (ray >= N) return;                                                          \
  const int i0 = start_end[2 * ray], i1 = start_end[2 * ray + 1];                \
  const int n = i1 - i0;

// VolumeRenderingGPU.cuh:28-78: T_i = prod_{j<i} a_j ; bgT = T_{n-1} (the last
// sample's factor is deliberately excluded); rays without samples keep bgT.
__global__ void cumprod_fwd_kernel(const int* __restrict__ start_end, const float* __restrict__ a,
                                   float* __restrict__ T, float* __restrict__ bgT, int N) {
  PK_RAY_PROLOGUE();
  if (n <= 0) return;
  float carry = 1.0f;
  for (int c = 0; c < n; c += SUB) {
    const int i = c + l;
    const float v = i < n ? a[i0 + i] : 1.0f;
    const float incl = sub_scan_mul(v, l);
    float excl = __shfl_up(incl, 1, SUB);
    if (l == 0) excl = 1.0f;
    const float t = carry * excl;
    if (i < n) T[i0 + i] = t;
    if (i == n - 1) bgT[ray] = t;
    carry *= __shfl(incl, SUB - 1, SUB);
  }
}

// VolumeRenderingGPU.cuh:305-361
__global__ void cumsum_kernel(const int* __restrict__ start_end, const float* __restrict__ v,
                              int inverse, float* __restrict__ out, int N) {
  PK_RAY_PROLOGUE();
  if (n <= 0) return;
  float carry = 0.0f;
  for (int c = 0; c < n; c += SUB) {
    const int i = c + l;
    const int idx = inverse ? i1 - 1 - i : i0 + i;
    const float x = i < n ? v[idx] : 0.0f;
    const float incl = sub_scan_add(x, l);
    if (i < n) out[idx] = carry + incl;
    carry += __shfl(incl, SUB - 1, SUB);
  }
}

// VolumeRenderingGPU.cuh:80-177
template <int D>
__global__ void integrate_fwd_kernel(const int* __restrict__ start_end,
                                     const float* __restrict__ values,
                                     const float* __restrict__ weights, float* __restrict__ out,
                                     int N) {
  PK_RAY_PROLOGUE();
  if (n <= 0) return;
  float acc[D];
#pragma unroll
  for (int d = 0; d < D; ++d) acc[d] = 0.f;
  for (int i = l; i < n; i += SUB) {
    const float w = weights[i0 + i];
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] += w * values[(long long)(i0 + i) * D + d];
  }
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const float s = sub_reduce_add(acc[d]);
    if (l == 0) out[ray * D + d] = s;
  }
}

// VolumeRenderingGPU.cuh:945-1033.  bug_compat reproduces :1021 (the z lane of
// `values` reads column 1); the Python mirror passes it ON by default
// (VolumeRendering.bug_compat = True: compute what the reference computes).
template <int D>
__global__ void integrate_bwd_kernel(const int* __restrict__ start_end,
                                     const float* __restrict__ g_out,
                                     const float* __restrict__ values,
                                     const float* __restrict__ weights,
                                     float* __restrict__ g_values, float* __restrict__ g_weights,
                                     int N, int bug_compat) {
  PK_RAY_PROLOGUE();
  if (n <= 0) return;
  float g[D];
#pragma unroll
  for (int d = 0; d < D; ++d) g[d] = g_out[ray * D + d];
  for (int i = l; i < n; i += SUB) {
    const long long s = i0 + i;
    const float w = weights[s];
    float gw = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      g_values[s * D + d] = g[d] * w;
      const int col = (bug_compat && D == 3 && d == 2) ? 1 : d;
      gw += g[d] * values[s * D + col];
    }
    g_weights[s] = gw;
  }
}

// VolumeRenderingGPU.cuh:896-943
__global__ void cumprod_bwd_kernel(const int* __restrict__ start_end,
                                   const float* __restrict__ g_bgT, const float* __restrict__ a,
                                   const float* __restrict__ bgT,
                                   const float* __restrict__ cumsumLV, float* __restrict__ g_a,
                                   int N) {
  PK_RAY_PROLOGUE();
  if (n <= 0) return;
  const float b = bgT[ray], gb = g_bgT[ray];
  for (int i = l; i < n; i += SUB) {
    float g = 0.f;
    if (i < n - 1) {
      const float d = fmaxf(a[i0 + i], 1e-6f);
      g = cumsumLV[i0 + i + 1] / d;
      g += gb * b / d;
    }
    g_a[i0 + i] = g;
  }
}

// ---------------------------------------------------------------------------------------
// The background composite as ONE launch each way.  render_contracted_bg
// (utils/background.py:93-111) chains  alpha = 1 - exp(-density dt);  T = cumprod_to_transmittance(
// (1 - alpha) + 1e-6);  w = alpha T;  rgb = integrate_with_weights_3d(rgb_s, w)  through four
// elementwise torch kernels and two packed launches (and twice that in backward, with T, w, lv and
// the suffix sums written and re-read).  Here the half-wave that owns a ray keeps all of it in
// registers.  Every fp32 operation is the one the op sequence performs, in its order — the scans
// are sub_scan_mul / sub_scan_add of the stand-alone kernels, the suffix sums run over the reversed
// ray exactly as cumsum_kernel(inverse) does — so results are bit-identical to the chain
// (tests/test_packed.py::test_fused_bg_composite_equals_the_op_sequence).
__global__ void composite_packed_fwd_kernel(const int* __restrict__ start_end,
                                            const float* __restrict__ density,
                                            const float* __restrict__ dt,
                                            const float* __restrict__ rgb, float* __restrict__ pred,
                                            float* __restrict__ weights, int N) {
  PK_RAY_PROLOGUE();
  float acc[3] = {0.f, 0.f, 0.f};
  float carry = 1.0f;
  for (int c = 0; c < n; c += SUB) {
    const int i = c + l;
    const bool in = i < n;
    const long long s = i0 + (in ? i : 0);
    const float e = expf((-density[s]) * dt[s]);
    const float alpha = 1.0f - e;
    const float a1 = (1.0f - alpha) + 1e-6f;
    const float incl = sub_scan_mul(in ? a1 : 1.0f, l);
    float excl = __shfl_up(incl, 1, SUB);
    if (l == 0) excl = 1.0f;
    const float T = carry * excl;
    carry *= __shfl(incl, SUB - 1, SUB);
    if (in) {
      const float w = alpha * T;
      if (weights) weights[s] = w;
#pragma unroll
      for (int d = 0; d < 3; ++d) acc[d] += w * rgb[s * 3 + d];
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float r = sub_reduce_add(acc[d]);
    if (l == 0) pred[ray * 3 + d] = r;        // (a ray without samples: 0, as integrate's zero-initialised output)
  }
}

// scratch: 2 floats per sample (lv = g_T T and g_w T of the forward sweep, read back by the
// reversed sweep; a ray of the background path is one 32-sample chunk each way)
__global__ void composite_packed_bwd_kernel(const int* __restrict__ start_end,
                                            const float* __restrict__ density,
                                            const float* __restrict__ dt,
                                            const float* __restrict__ rgb,
                                            const float* __restrict__ g_pred,
                                            float* __restrict__ g_rgb, float* __restrict__ g_density,
                                            float* __restrict__ scratch, int N, int bug_compat) {
  PK_RAY_PROLOGUE();
  if (n <= 0) return;
  float g[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) g[d] = g_pred[ray * 3 + d];
  float carry = 1.0f;
  for (int c = 0; c < n; c += SUB) {      // forward sweep: T, then everything local to a sample
    const int i = c + l;
    const bool in = i < n;
    const long long s = i0 + (in ? i : 0);
    const float e = expf((-density[s]) * dt[s]);
    const float alpha = 1.0f - e;
    const float a1 = (1.0f - alpha) + 1e-6f;
    const float incl = sub_scan_mul(in ? a1 : 1.0f, l);
    float excl = __shfl_up(incl, 1, SUB);
    if (l == 0) excl = 1.0f;
    const float T = carry * excl;
    carry *= __shfl(incl, SUB - 1, SUB);
    if (in) {
      const float w = alpha * T;
      float gw = 0.f;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        g_rgb[s * 3 + d] = g[d] * w;                                     // integrate_bwd_kernel
        const int col = (bug_compat && d == 2) ? 1 : d;
        gw += g[d] * rgb[s * 3 + col];
      }
      const float gT = gw * alpha;                                       // w = alpha T
      scratch[2 * s] = gT * T;                                           // lv (volume_rendering_funcs.py:152)
      scratch[2 * s + 1] = gw * T;                                       // d w / d alpha
    }
  }
  float csum = 0.0f;
  for (int c = 0; c < n; c += SUB) {      // reversed sweep: suffix sums of lv, cumprod backward, the exp
    const int i = c + l;
    const bool in = i < n;
    const long long s = in ? (long long)i1 - 1 - i : (long long)i0;
    const float incl = sub_scan_add(in ? scratch[2 * s] : 0.0f, l);
    const float cs = csum + incl;                                        // cumsumLV at this sample
    float cs_next = __shfl_up(cs, 1, SUB);                               // ... at the sample after it on the ray
    if (l == 0) cs_next = csum;
    csum += __shfl(incl, SUB - 1, SUB);
    if (in) {
      const float dts = dt[s];
      const float e = expf((-density[s]) * dts);
      const float alpha = 1.0f - e;
      const float a1 = (1.0f - alpha) + 1e-6f;
      float ga1 = 0.f;
      if (i > 0) ga1 = cs_next / fmaxf(a1, 1e-6f);                      // cumprod_bwd_kernel (g_bgT = 0); 0 for the last sample
      const float g_alpha = scratch[2 * s + 1] + (-ga1);
      const float gu = (-g_alpha) * e;                                   // alpha = 1 - e, e = exp(u)
      g_density[s] = -(gu * dts);                                        // u = (-density) dt
    }
  }
}

// VolumeRenderingGPU.cuh:364-409.  fallback_compat reproduces :407
// (samples_z[nr_samples-1] without idx_start); default is the ray's last sample.
__global__ void median_depth_kernel(const int* __restrict__ start_end, const float* __restrict__ z,
                                    const float* __restrict__ w, float thr,
                                    float* __restrict__ out, int N, int fallback_compat) {
  PK_RAY_PROLOGUE();
  if (n <= 0) return;
  float carry = 0.0f;
  for (int c = 0; c < n; c += SUB) {
    const int i = c + l;
    const float incl = carry + sub_scan_add(i < n ? w[i0 + i] : 0.0f, l);
    const unsigned long long m = __ballot(i < n && incl >= thr);
    const unsigned int mine = (unsigned int)(m >> (threadIdx.x & 32));   // this half-wave's 32 bits
    if (mine) {
      const int first = __ffs(mine) - 1;
      if (l == first) out[ray] = z[i0 + i];
      return;
    }
    carry = __shfl(incl, SUB - 1, SUB);
  }
  if (l == 0) out[ray] = fallback_compat ? z[n - 1] : z[i1 - 1];
}

// RaySamplesPackedGPU.cuh:14-88
__global__ void update_dt_kernel(const int* __restrict__ start_end,
                                 const float* __restrict__ ray_max_dt,
                                 const float* __restrict__ ray_exit, const float* __restrict__ z,
                                 int is_background, float* __restrict__ dt, int N) {
  PK_RAY_PROLOGUE();
  if (n <= 0) return;
  const float max_dt = ray_max_dt[ray];
  for (int i = l; i < n; i += SUB) {
    float v;
    if (i < n - 1) {
      v = fminf(fmaxf(z[i0 + i + 1] - z[i0 + i], 0.0f), max_dt);
    } else if (is_background) {
      v = 1e10f;
    } else {
      v = fminf(fmaxf(ray_exit[ray] - z[i0 + i], 0.0f), max_dt);
    }
    dt[i0 + i] = v;
  }
}

// RaySamplerGPU.cuh:39-139.  Without jitter every sample is independent: lanes =
// samples.  With jitter t_i = lerp(t_{i-1}, t_i, u) chains through the ray; the draws of the
// reference's per-thread rng.advance(ray) + next_float() sequence are reproduced lane-parallel
// (skip-ahead), only the lerps run in order (see below).
__global__ void sample_bg_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                 const float* __restrict__ t_start_in, float t_far, int ns,
                                 int jitter, unsigned long long rng_state,
                                 unsigned long long rng_inc, float* __restrict__ ray_max_dt,
                                 float* __restrict__ s3d, float* __restrict__ sdirs,
                                 float* __restrict__ sz, int* __restrict__ start_end, int N) {
  const int l = threadIdx.x & (SUB - 1);
  const long long ray = ((long long)blockIdx.x * PK_BLOCK + threadIdx.x) / SUB;
  if (ray >= N) return;
  const float eps = 1e-6f;
  const float t_start = t_start_in[ray];
  const float ox = rays_o[3 * ray], oy = rays_o[3 * ray + 1], oz = rays_o[3 * ray + 2];
  const float dx = rays_d[3 * ray], dy = rays_d[3 * ray + 1], dz = rays_d[3 * ray + 2];
  // delta_s = 1.0 / (n - 1) is a double expression assigned to float; s -= delta_s in float
  const float delta_s = (float)(1.0 / (double)(ns - 1));
  const long long base = ray * ns;
  if (!jitter) {
    // s_i by the same serial float subtractions as the reference (i is small)
    float max_dt = 0.f;
    for (int c = 0; c < ns; c += SUB) {
      const int i = c + l;
      float s = 1.0f;
      for (int k = 0; k < i && k < ns; ++k) s -= delta_s;
      float t = (float)(1.0 / (double)(s + eps) - 1.0);
      t += t_start;
      t = fminf(fmaxf(t, t_start), t_far);
      float tp = __shfl_up(t, 1, SUB);
      if (l == 0) {
        // previous chunk's last t (or t_start for the first sample)
        float sp = 1.0f;
        for (int k = 0; k < i - 1; ++k) sp -= delta_s;
        float tq = (float)(1.0 / (double)(sp + eps) - 1.0) + t_start;
        tp = i == 0 ? t_start : fminf(fmaxf(tq, t_start), t_far);
      }
      if (i < ns) {
        sz[base + i] = t;
        s3d[3 * (base + i)] = ox + t * dx;
        s3d[3 * (base + i) + 1] = oy + t * dy;
        s3d[3 * (base + i) + 2] = oz + t * dz;
        sdirs[3 * (base + i)] = dx;
        sdirs[3 * (base + i) + 1] = dy;
        sdirs[3 * (base + i) + 2] = dz;
        max_dt = fmaxf(max_dt, t - tp);
      }
    }
#pragma unroll
    for (int off = SUB / 2; off > 0; off >>= 1) max_dt = fmaxf(max_dt, __shfl_xor(max_dt, off, SUB));
    if (l == 0) ray_max_dt[ray] = max_dt;
  } else {
    // Jitter: t_i = lerp(t_{i-1}, t_i, u_i) chains through the ray, but the draws do not: the
    // reference's per-sample rng.advance(ray) + next_float() leaves the generator (i - 1)(ray + 1) +
    // ray steps from its seed before the draw of sample i, so every lane jumps there on its own
    // (O(log) LCG skip-ahead), forms its un-jittered t_i (a double division) in parallel, and only
    // the 32 lerps of the chain run in order, through shuffles.  (One lane doing all of it
    // serially: 0.55 ms for the 65 k rays of a background batch.)
    float t_prec = t_start, max_dt = 0.f;
    for (int c = 0; c < ns; c += SUB) {
      const int i = c + l;
      float s = 1.0f;
      for (int k = 0; k < i && k < ns; ++k) s -= delta_s;
      float t = (float)(1.0 / (double)(s + eps) - 1.0);
      t += t_start;
      t = fminf(fmaxf(t, t_start), t_far);
      const bool jit = i != 0 && i < ns - 1;
      float u = 0.f;
      if (jit) {
        Pcg32 rng{rng_state, rng_inc};
        rng.advance((unsigned long long)(i - 1) * ((unsigned long long)ray + 1ull) + (unsigned long long)ray);
        u = rng.next_float();
      }
      float my_t = t, my_dt = 0.f;
      for (int k = 0; k < SUB && c + k < ns; ++k) {
        const float tk = __shfl(t, k, SUB), uk = __shfl(u, k, SUB);
        const int jk = __shfl((int)jit, k, SUB);
        const float tt = jk ? t_prec + uk * (tk - t_prec) : tk;   // helper_math lerp: a + t*(b-a)
        if (l == k) {
          my_t = tt;
          my_dt = tt - t_prec;
        }
        t_prec = tt;
      }
      if (i < ns) {
        sz[base + i] = my_t;
        s3d[3 * (base + i)] = ox + my_t * dx;
        s3d[3 * (base + i) + 1] = oy + my_t * dy;
        s3d[3 * (base + i) + 2] = oz + my_t * dz;
        sdirs[3 * (base + i)] = dx;
        sdirs[3 * (base + i) + 1] = dy;
        sdirs[3 * (base + i) + 2] = dz;
        max_dt = fmaxf(max_dt, my_dt);
      }
    }
#pragma unroll
    for (int off = SUB / 2; off > 0; off >>= 1) max_dt = fmaxf(max_dt, __shfl_xor(max_dt, off, SUB));
    if (l == 0) ray_max_dt[ray] = max_dt;
  }
  if (l == 0) {
    start_end[2 * ray] = (int)base;
    start_end[2 * ray + 1] = (int)(base + ns);
  }
}

// RaySamplerGPU.cuh:528-592 (scale 2 contraction)
__global__ void contract_kernel(const float* __restrict__ ray_o, const int* __restrict__ start_end,
                                const float* __restrict__ s3d, const float* __restrict__ sz,
                                float* __restrict__ o3d, float* __restrict__ oz, int N) {
  PK_RAY_PROLOGUE();
  if (n <= 0) return;
  const float cx = ray_o[3 * ray], cy = ray_o[3 * ray + 1], cz = ray_o[3 * ray + 2];
  for (int i = l; i < n; i += SUB) {
    const long long s = i0 + i;
    float px = s3d[3 * s], py = s3d[3 * s + 1], pz = s3d[3 * s + 2];
    float z = sz[s];
    const float sx = px * 2.0f, sy = py * 2.0f, sz2 = pz * 2.0f;
    const float norm = sqrtf((sx * sx + sy * sy) + sz2 * sz2);
    if (norm > 1.0f) {
      const float factor = 2.0f - 1.0f / norm;
      px = (factor * px) / norm;
      py = (factor * py) / norm;
      pz = (factor * pz) / norm;
      const float ex = px - cx, ey = py - cy, ez = pz - cz;
      z = sqrtf((ex * ex + ey * ey) + ez * ez);
    }
    o3d[3 * s] = px;
    o3d[3 * s + 1] = py;
    o3d[3 * s + 2] = pz;
    oz[s] = z;
  }
}

inline dim3 pk_grid(int N) { return dim3(vsa_div_up((long long)N * SUB, PK_BLOCK)); }


// ---- ops of the sibling methods: sum_over_rays, sdf2alpha, compute_cdf
// VolumeRenderingGPU.cuh:246-303 / :1036-1077: per-ray sum of D-wide sample values, also
// broadcast back to the ray's samples; backward g_v[i] = g_sum[ray] + g_per_sample[i].
template <int D>
__global__ void sum_over_rays_kernel(const int* __restrict__ start_end, const float* __restrict__ v,
                                     float* __restrict__ sum_ray, float* __restrict__ sum_sample,
                                     int N) {
  PK_RAY_PROLOGUE();
  if (n <= 0) return;
  float acc[D];
#pragma unroll
  for (int d = 0; d < D; ++d) acc[d] = 0.f;
  for (int c = 0; c < n; c += SUB) {
    const int i = c + l;
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] += sub_reduce_add(i < n ? v[(long long)(i0 + i) * D + d] : 0.f);
  }
  if (l == 0) {
#pragma unroll
    for (int d = 0; d < D; ++d) sum_ray[ray * D + d] = acc[d];
  }
  for (int i = l; i < n; i += SUB) {
#pragma unroll
    for (int d = 0; d < D; ++d) sum_sample[(long long)(i0 + i) * D + d] = acc[d];
  }
}

template <int D>
__global__ void sum_over_rays_bwd_kernel(const int* __restrict__ start_end,
                                         const float* __restrict__ g_ray,
                                         const float* __restrict__ g_sample,
                                         float* __restrict__ g_v, int N) {
  PK_RAY_PROLOGUE();
  for (int i = l; i < n; i += SUB) {
#pragma unroll
    for (int d = 0; d < D; ++d)
      g_v[(long long)(i0 + i) * D + d] = g_ray[ray * D + d] + g_sample[(long long)(i0 + i) * D + d];
  }
}

// VolumeRenderingGPU.cuh:185-244 (NeuS alpha from consecutive SDF samples).  The reference
// mixes float variables with double literals, so several intermediate results are formed in
// double and rounded to float on assignment; reproduced operation by operation.
__device__ __forceinline__ float sigmoid_ref(float x) {   // :179-183: float res = 1.0 / (1.0 + exp(-x))
  return (float)(1.0 / (1.0 + (double)expf(-x)));
}

__global__ void sdf2alpha_kernel(const int* __restrict__ start_end, const float* __restrict__ dt,
                                 const float* __restrict__ sdf, const float* __restrict__ beta,
                                 float* __restrict__ alpha, int N) {
  PK_RAY_PROLOGUE();
  for (int i = l; i < n - 1; i += SUB) {
    const float d = dt[i0 + i];
    const float prev = sdf[i0 + i], next = sdf[i0 + i + 1];
    const float mid = (float)((double)(prev + next) * 0.5);
    float cosv = (float)((double)(next - prev) / ((double)d + 1e-6));
    cosv = fminf(fmaxf(cosv, -1e3f), 0.0f);
    const float prev_esti = (float)((double)mid - (double)(cosv * d) * 0.5);
    const float next_esti = (float)((double)mid + (double)(cosv * d) * 0.5);
    const float b = beta[i0 + i];
    const float prev_cdf = sigmoid_ref(prev_esti * b), next_cdf = sigmoid_ref(next_esti * b);
    alpha[i0 + i] = (float)(((double)(prev_cdf - next_cdf) + 1e-6) / ((double)prev_cdf + 1e-6));
  }
}

// VolumeRenderingGPU.cuh:412-460: exclusive running sum of the weights per ray; if the
// weights sum to ~1 the last entry is snapped to 1.  Rays with fewer than 2 samples are left
// at zero (the reference prints and returns).
__global__ void compute_cdf_kernel(const int* __restrict__ start_end, const float* __restrict__ w,
                                   float* __restrict__ cdf, int N) {
  PK_RAY_PROLOGUE();
  if (n < 2) return;
  float carry = 0.0f;
  float last_cdf = 0.0f;
  for (int c = 0; c < n; c += SUB) {
    const int i = c + l;
    const float x = i < n ? w[i0 + i] : 0.0f;
    const float incl = sub_scan_add(x, l);
    const float excl = carry + (incl - x);
    if (i < n) cdf[i0 + i] = excl;
    if (i == n - 1) last_cdf = excl;
    carry += __shfl(incl, SUB - 1, SUB);
  }
  // carry = sum of all weights; the lane that owns the last sample applies the snap
  const int owner = (n - 1) & (SUB - 1);
  if (l == owner && fabs((double)carry - 1.0) < 1e-3 && fabs((double)last_cdf - 1.0) > 1e-3)
    cdf[i1 - 1] = 1.0f;
}


// ---- foreground sampling chain of the sibling methods
// RaySamplerGPU.cuh:141-270 compute_samples_fg: uniform steps of >= min_dist between the
// ray's entry and exit.  One thread per ray: the sample depths come from the serial
// accumulation t += step, which is part of the result; the slot layout is the reference's
// (ray i owns slots [i*max_n, (i+1)*max_n)), compacted afterwards.
__global__ void sample_fg_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                 const float* __restrict__ t_entry, const float* __restrict__ t_exit_in,
                                 float min_dist, int min_n, int max_n, int jitter,
                                 unsigned long long rng_state, unsigned long long rng_inc,
                                 float* __restrict__ ray_max_dt, int* __restrict__ samples_idx,
                                 float* __restrict__ s3d, float* __restrict__ sdirs,
                                 float* __restrict__ sz, int* __restrict__ start_end, int N) {
  const long long ray = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= N) return;
  const float t_start = t_entry[ray], t_exit = t_exit_in[ray];
  const float ox = rays_o[3 * ray], oy = rays_o[3 * ray + 1], oz = rays_o[3 * ray + 2];
  const float dx = rays_d[3 * ray], dy = rays_d[3 * ray + 1], dz = rays_d[3 * ray + 2];
  const float dist = t_exit - t_start;
  int to_create = 0;
  float step = 0.0f;
  if (dist > 0.0f) {
    if (dist > min_dist) {
      to_create = (int)(dist / min_dist);
      to_create = min(max(to_create, 0), max_n);
      step = dist / to_create;
    } else {
      to_create = 1;
      step = dist;
    }
  }
  int created = 0;
  const long long s0 = ray * max_n;
  if (to_create > 0 && to_create >= min_n) {
    float t = t_start;
    if (jitter) {
      Pcg32 rng{rng_state, rng_inc};
      rng.advance((unsigned long long)ray);
      t = t + step * rng.next_float();
    }
    while (t < t_exit) {
      t = fminf(fmaxf(t, t_start), t_exit);
      if (created >= to_create) break;
      const long long o = s0 + created;
      s3d[3 * o] = ox + t * dx;
      s3d[3 * o + 1] = oy + t * dy;
      s3d[3 * o + 2] = oz + t * dz;
      sdirs[3 * o] = dx;
      sdirs[3 * o + 1] = dy;
      sdirs[3 * o + 2] = dz;
      sz[o] = t;
      t += step;
      created += 1;
    }
  }
  if (created < min_n) {
    created = 0;
  } else {
    ray_max_dt[ray] = step;
    start_end[2 * ray] = (int)s0;
    start_end[2 * ray + 1] = (int)s0 + created;
  }
  for (int i = created; i < max_n; ++i) samples_idx[s0 + i] = -1;
}

// RaySamplesPackedGPU.cuh:172-257 compact_to_valid_samples: copy every ray's samples to its
// offset in the compacted pack (out_start = exclusive scan of the per-ray counts, computed
// by the caller).  Half-wave per ray, lanes = samples.
__global__ void pack_compact_kernel(const int* __restrict__ start_end, const int* __restrict__ out_start,
                                    const int* __restrict__ idx, const float* __restrict__ s3d,
                                    const float* __restrict__ sdirs, const float* __restrict__ sz,
                                    const float* __restrict__ sdt, const float* __restrict__ sval,
                                    int V, int* __restrict__ o_idx, float* __restrict__ o_3d,
                                    float* __restrict__ o_dirs, float* __restrict__ o_z,
                                    float* __restrict__ o_dt, float* __restrict__ o_val,
                                    int* __restrict__ o_start_end, int N) {
  PK_RAY_PROLOGUE();
  if (n <= 0) return;
  const int o0 = out_start[ray];
  for (int i = l; i < n; i += SUB) {
    const long long a = i0 + i, b = o0 + i;
    o_idx[b] = idx[a];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      o_3d[3 * b + c] = s3d[3 * a + c];
      o_dirs[3 * b + c] = sdirs[3 * a + c];
    }
    o_z[b] = sz[a];
    o_dt[b] = sdt[a];
    for (int j = 0; j < V; ++j) o_val[b * V + j] = sval[a * V + j];
  }
  if (l == 0) {
    o_start_end[2 * ray] = o0;
    o_start_end[2 * ray + 1] = o0 + n;
  }
}

// VolumeRenderingGPU.cuh:15-21
__device__ __forceinline__ float map_range_val_ref(float v, float i0, float i1, float o0, float o1) {
  const float c = fmaxf(i0, fminf(i1, v));
  if (i0 >= i1) return o1;
  return o0 + ((o1 - o0) / (i1 - i0)) * (c - i0);
}

// VolumeRenderingGPU.cuh:462-505 (binary search in a ray's cdf) and :507-678
// importance_sample: one thread per (ray, new sample); the jittered variant replays the
// reference's per-thread stream (advance(ray) + one draw per sample) by skipping ahead.
__global__ void importance_sample_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                         const int* __restrict__ start_end, const float* __restrict__ sz,
                                         const float* __restrict__ cdf, int n_imp, int jitter,
                                         unsigned long long rng_state, unsigned long long rng_inc,
                                         float* __restrict__ o_3d, float* __restrict__ o_dirs,
                                         float* __restrict__ o_z, int* __restrict__ o_start_end, int N) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long ray = t / n_imp;
  const int i = (int)(t - ray * n_imp);
  if (ray >= N) return;
  const int u0 = start_end[2 * ray], u1 = start_end[2 * ray + 1];
  if (u1 - u0 == 0) return;
  const float dist = (float)(1.0 / (double)(n_imp + 1));
  float ur = dist + i * dist;
  if (jitter) {
    Pcg32 rng{rng_state, rng_inc};
    rng.advance((unsigned long long)(i + 1) * (unsigned long long)ray + (unsigned long long)i);
    const float r = rng.next_float();
    const float mov = (float)((double)dist / 2.0);
    ur += map_range_val_ref(r, 0.0f, 1.0f, -mov, mov);
  }
  ur = fminf(fmaxf(ur, (float)(0.0 + 1e-6)), (float)(1.0 - 1e-6));
  int imin = u0, imax = u1 - 1;
  while (imax >= imin) {   // binary_search (:466-489)
    const int imid = imin + (imax - imin) / 2;
    if (cdf[imid] > ur) imax = imid; else imin = imid;
    if (imax - imin == 1) break;
    if (imax == imin) break;   // single-sample ray: the reference would spin; it never calls this with < 2
  }
  const int hi = imax, lo = max(imax - 1, 0);
  const float z = map_range_val_ref(ur, cdf[lo], cdf[hi], sz[lo], sz[hi]);
  const long long o = ray * n_imp + i;
  const float dx = rays_d[3 * ray], dy = rays_d[3 * ray + 1], dz = rays_d[3 * ray + 2];
  o_3d[3 * o] = rays_o[3 * ray] + z * dx;
  o_3d[3 * o + 1] = rays_o[3 * ray + 1] + z * dy;
  o_3d[3 * o + 2] = rays_o[3 * ray + 2] + z * dz;
  o_dirs[3 * o] = dx;
  o_dirs[3 * o + 1] = dy;
  o_dirs[3 * o + 2] = dz;
  o_z[o] = z;
  if (i == 0) {
    o_start_end[2 * ray] = (int)(ray * n_imp);
    o_start_end[2 * ray + 1] = (int)(ray * n_imp) + n_imp;
  }
}


// RaySamplerGPU.cuh:595-650 uncontract_samples (inverse of the scene contraction, scale 2)
__global__ void uncontract_kernel(const float* __restrict__ ray_o, const int* __restrict__ start_end,
                                  const float* __restrict__ s3d, const float* __restrict__ sz,
                                  float* __restrict__ o3d, float* __restrict__ oz, int N) {
  PK_RAY_PROLOGUE();
  if (n <= 0) return;
  const float cx = ray_o[3 * ray], cy = ray_o[3 * ray + 1], cz = ray_o[3 * ray + 2];
  for (int i = l; i < n; i += SUB) {
    const long long s = i0 + i;
    float px = s3d[3 * s], py = s3d[3 * s + 1], pz = s3d[3 * s + 2];
    float z = sz[s];
    const float sx = px * 2.0f, sy = py * 2.0f, sz2 = pz * 2.0f;
    const float norm = sqrtf((sx * sx + sy * sy) + sz2 * sz2);
    if (norm > 1.0f) {
      const float factor = 1.0f / (2.0f - norm);
      px = (factor * px) / norm;
      py = (factor * py) / norm;
      pz = (factor * pz) / norm;
      const float ex = px - cx, ey = py - cy, ez = pz - cz;
      z = sqrtf((ex * ex + ey * ey) + ez * ez);
    }
    o3d[3 * s] = px;
    o3d[3 * s + 1] = py;
    o3d[3 * s + 2] = pz;
    oz[s] = z;
  }
}


// VolumeRenderingGPU.cuh:680-895 combine_ray_samples_packets: per ray, merge the two packs'
// samples in increasing depth and drop a sample that lies closer than min_dist to the last one
// kept.  One thread per ray: which samples survive depends on the serial scan.  (A pack without
// samples for this ray counts as exhausted from the start; the reference would read one
// element before the ray's range there.)
struct PackView {
  const int* start_end;
  const int* idx;
  const float* s3d;
  const float* sdirs;
  const float* sz;
  const float* sval;
};

__global__ void combine_packs_kernel(PackView a, PackView b, const int* __restrict__ out_start,
                                     float min_dist, int V, int* __restrict__ o_idx,
                                     float* __restrict__ o_3d, float* __restrict__ o_dirs,
                                     float* __restrict__ o_z, float* __restrict__ o_val,
                                     int* __restrict__ o_start_end, int N) {
  const long long ray = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= N) return;
  const int a0 = a.start_end[2 * ray], na = a.start_end[2 * ray + 1] - a0;
  const int b0 = b.start_end[2 * ray], nb = b.start_end[2 * ray + 1] - b0;
  if (na == 0 && nb == 0) return;
  const int o0 = out_start[ray];
  int ca = 0, cb = 0, written = 0;
  bool fa = na == 0, fb = nb == 0;
  float prec_z = 0.0f;
  for (int i = 0; i < na + nb; ++i) {
    if (fa && fb) break;
    const float za = fa ? 1e10f : a.sz[a0 + ca];
    const float zb = fb ? 1e10f : b.sz[b0 + cb];
    const bool take_a = za < zb;
    const PackView& p = take_a ? a : b;
    const long long src = take_a ? a0 + ca : b0 + cb;
    const float z = take_a ? za : zb;
    if (!(z - prec_z < min_dist)) {
      const long long dst = o0 + written;
      o_idx[dst] = p.idx[src];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        o_3d[3 * dst + c] = p.s3d[3 * src + c];
        o_dirs[3 * dst + c] = p.sdirs[3 * src + c];
      }
      o_z[dst] = z;
      prec_z = z;
      for (int v = 0; v < V; ++v) o_val[dst * V + v] = p.sval[src * V + v];
      written += 1;
    }
    if (take_a) {
      if (ca + 1 >= na) fa = true; else ca += 1;
    } else {
      if (cb + 1 >= nb) fb = true; else cb += 1;
    }
  }
  o_start_end[2 * ray] = o0;
  o_start_end[2 * ray + 1] = o0 + written;
}

}  // namespace

#define PK_CHECK(cond) \
  if (!(cond)) return VSA_ERR_ARG
#define PK_LAUNCH(kernel, N, ...)                                                          \
  if ((N) == 0) return VSA_OK;                                                             \
  hipLaunchKernelGGL(kernel, pk_grid(N), dim3(PK_BLOCK), 0, (hipStream_t)stream, __VA_ARGS__); \
  VSA_RETURN_LAUNCH_STATUS()

extern "C" int vsa_packed_cumprod_fwd(const int32_t* start_end, const float* one_minus_alpha,
                                      float* transmittance, float* bg_transmittance, int nr_rays,
                                      void* stream) {
  PK_CHECK(nr_rays >= 0 && start_end && one_minus_alpha && transmittance && bg_transmittance);
  PK_LAUNCH(cumprod_fwd_kernel, nr_rays, start_end, one_minus_alpha, transmittance,
            bg_transmittance, nr_rays);
}
extern "C" int vsa_packed_cumprod_bwd(const int32_t* start_end, const float* g_bg_transmittance,
                                      const float* one_minus_alpha, const float* bg_transmittance,
                                      const float* cumsum_lv, float* g_one_minus_alpha,
                                      int nr_rays, void* stream) {
  PK_CHECK(nr_rays >= 0 && start_end && g_bg_transmittance && one_minus_alpha && bg_transmittance &&
           cumsum_lv && g_one_minus_alpha);
  PK_LAUNCH(cumprod_bwd_kernel, nr_rays, start_end, g_bg_transmittance, one_minus_alpha,
            bg_transmittance, cumsum_lv, g_one_minus_alpha, nr_rays);
}
extern "C" int vsa_packed_cumsum(const int32_t* start_end, const float* values, int inverse,
                                 float* out, int nr_rays, void* stream) {
  PK_CHECK(nr_rays >= 0 && start_end && values && out);
  PK_LAUNCH(cumsum_kernel, nr_rays, start_end, values, inverse, out, nr_rays);
}
extern "C" int vsa_packed_integrate_fwd(const int32_t* start_end, const float* values,
                                        const float* weights, float* out, int nr_rays, int dim,
                                        void* stream) {
  PK_CHECK(nr_rays >= 0 && start_end && values && weights && out && (dim == 1 || dim == 3));
  if (dim == 1) {
    PK_LAUNCH(integrate_fwd_kernel<1>, nr_rays, start_end, values, weights, out, nr_rays);
  }
  PK_LAUNCH(integrate_fwd_kernel<3>, nr_rays, start_end, values, weights, out, nr_rays);
}
extern "C" int vsa_packed_integrate_bwd(const int32_t* start_end, const float* g_out,
                                        const float* values, const float* weights,
                                        float* g_values, float* g_weights, int nr_rays, int dim,
                                        int bug_compat, void* stream) {
  PK_CHECK(nr_rays >= 0 && start_end && g_out && values && weights && g_values && g_weights &&
           (dim == 1 || dim == 3));
  if (dim == 1) {
    PK_LAUNCH(integrate_bwd_kernel<1>, nr_rays, start_end, g_out, values, weights, g_values,
              g_weights, nr_rays, bug_compat);
  }
  PK_LAUNCH(integrate_bwd_kernel<3>, nr_rays, start_end, g_out, values, weights, g_values,
            g_weights, nr_rays, bug_compat);
}
extern "C" int vsa_packed_composite_fwd(const int32_t* start_end, const float* density,
                                        const float* dt, const float* rgb, float* pred_rgb,
                                        float* weights, int nr_rays, void* stream) {
  PK_CHECK(nr_rays >= 0 && start_end && density && dt && rgb && pred_rgb);
  PK_LAUNCH(composite_packed_fwd_kernel, nr_rays, start_end, density, dt, rgb, pred_rgb, weights,
            nr_rays);
}
extern "C" int vsa_packed_composite_bwd(const int32_t* start_end, const float* density,
                                        const float* dt, const float* rgb, const float* g_pred_rgb,
                                        float* g_rgb, float* g_density, float* scratch, int nr_rays,
                                        int bug_compat, void* stream) {
  PK_CHECK(nr_rays >= 0 && start_end && density && dt && rgb && g_pred_rgb && g_rgb && g_density &&
           scratch);
  PK_LAUNCH(composite_packed_bwd_kernel, nr_rays, start_end, density, dt, rgb, g_pred_rgb, g_rgb,
            g_density, scratch, nr_rays, bug_compat);
}
extern "C" int vsa_packed_median_depth(const int32_t* start_end, const float* samples_z,
                                       const float* weights, float threshold, float* out,
                                       int nr_rays, int fallback_compat, void* stream) {
  PK_CHECK(nr_rays >= 0 && start_end && samples_z && weights && out);
  PK_LAUNCH(median_depth_kernel, nr_rays, start_end, samples_z, weights, threshold, out, nr_rays,
            fallback_compat);
}
extern "C" int vsa_packed_update_dt(const int32_t* start_end, const float* ray_max_dt,
                                    const float* ray_exit, const float* samples_z,
                                    int is_background, float* samples_dt, int nr_rays,
                                    void* stream) {
  PK_CHECK(nr_rays >= 0 && start_end && ray_max_dt && ray_exit && samples_z && samples_dt);
  PK_LAUNCH(update_dt_kernel, nr_rays, start_end, ray_max_dt, ray_exit, samples_z, is_background,
            samples_dt, nr_rays);
}
extern "C" int vsa_sample_bg(const float* rays_o, const float* rays_d, const float* ray_t_start,
                             float ray_t_far, int nr_samples_per_ray, int jitter,
                             uint64_t rng_state, uint64_t rng_inc, float* ray_max_dt,
                             float* samples_3d, float* samples_dirs, float* samples_z,
                             int32_t* ray_start_end_idx, int nr_rays, void* stream) {
  PK_CHECK(nr_rays >= 0 && nr_samples_per_ray >= 2 && rays_o && rays_d && ray_t_start &&
           ray_max_dt && samples_3d && samples_dirs && samples_z && ray_start_end_idx);
  PK_LAUNCH(sample_bg_kernel, nr_rays, rays_o, rays_d, ray_t_start, ray_t_far, nr_samples_per_ray,
            jitter, (unsigned long long)rng_state, (unsigned long long)rng_inc, ray_max_dt,
            samples_3d, samples_dirs, samples_z, ray_start_end_idx, nr_rays);
}
extern "C" int vsa_contract_samples(const float* ray_o, const int32_t* start_end,
                                    const float* samples_3d, const float* samples_z,
                                    float* out_samples_3d, float* out_samples_z, int nr_rays,
                                    void* stream) {
  PK_CHECK(nr_rays >= 0 && ray_o && start_end && samples_3d && samples_z && out_samples_3d &&
           out_samples_z);
  PK_LAUNCH(contract_kernel, nr_rays, ray_o, start_end, samples_3d, samples_z, out_samples_3d,
            out_samples_z, nr_rays);
}

extern "C" int vsa_packed_sum_over_rays(const int32_t* start_end, const float* values,
                                        float* sum_per_ray, float* sum_per_sample, int nr_rays,
                                        int dim, void* stream) {
  PK_CHECK(nr_rays >= 0 && start_end && values && sum_per_ray && sum_per_sample &&
           (dim == 1 || dim == 2 || dim == 3 || dim == 32));
  if (dim == 1) {
    PK_LAUNCH(sum_over_rays_kernel<1>, nr_rays, start_end, values, sum_per_ray, sum_per_sample, nr_rays);
  }
  if (dim == 2) {
    PK_LAUNCH(sum_over_rays_kernel<2>, nr_rays, start_end, values, sum_per_ray, sum_per_sample, nr_rays);
  }
  if (dim == 3) {
    PK_LAUNCH(sum_over_rays_kernel<3>, nr_rays, start_end, values, sum_per_ray, sum_per_sample, nr_rays);
  }
  PK_LAUNCH(sum_over_rays_kernel<32>, nr_rays, start_end, values, sum_per_ray, sum_per_sample, nr_rays);
}
extern "C" int vsa_packed_sum_over_rays_bwd(const int32_t* start_end, const float* g_sum_per_ray,
                                            const float* g_sum_per_sample, float* g_values,
                                            int nr_rays, int dim, void* stream) {
  PK_CHECK(nr_rays >= 0 && start_end && g_sum_per_ray && g_sum_per_sample && g_values &&
           (dim == 1 || dim == 2 || dim == 3 || dim == 32));
  if (dim == 1) {
    PK_LAUNCH(sum_over_rays_bwd_kernel<1>, nr_rays, start_end, g_sum_per_ray, g_sum_per_sample, g_values, nr_rays);
  }
  if (dim == 2) {
    PK_LAUNCH(sum_over_rays_bwd_kernel<2>, nr_rays, start_end, g_sum_per_ray, g_sum_per_sample, g_values, nr_rays);
  }
  if (dim == 3) {
    PK_LAUNCH(sum_over_rays_bwd_kernel<3>, nr_rays, start_end, g_sum_per_ray, g_sum_per_sample, g_values, nr_rays);
  }
  PK_LAUNCH(sum_over_rays_bwd_kernel<32>, nr_rays, start_end, g_sum_per_ray, g_sum_per_sample, g_values, nr_rays);
}
extern "C" int vsa_packed_sdf2alpha(const int32_t* start_end, const float* samples_dt,
                                    const float* samples_sdf, const float* logistic_beta,
                                    float* alpha, int nr_rays, void* stream) {
  PK_CHECK(nr_rays >= 0 && start_end && samples_dt && samples_sdf && logistic_beta && alpha);
  PK_LAUNCH(sdf2alpha_kernel, nr_rays, start_end, samples_dt, samples_sdf, logistic_beta, alpha, nr_rays);
}
extern "C" int vsa_packed_compute_cdf(const int32_t* start_end, const float* weights, float* cdf,
                                      int nr_rays, void* stream) {
  PK_CHECK(nr_rays >= 0 && start_end && weights && cdf);
  PK_LAUNCH(compute_cdf_kernel, nr_rays, start_end, weights, cdf, nr_rays);
}

extern "C" int vsa_sample_fg(const float* rays_o, const float* rays_d, const float* ray_t_entry,
                             const float* ray_t_exit, float min_dist_between_samples,
                             int min_nr_samples_per_ray, int max_nr_samples_per_ray, int jitter,
                             uint64_t rng_state, uint64_t rng_inc, float* ray_max_dt,
                             int32_t* samples_idx, float* samples_3d, float* samples_dirs,
                             float* samples_z, int32_t* ray_start_end_idx, int nr_rays, void* stream) {
  PK_CHECK(nr_rays >= 0 && max_nr_samples_per_ray >= 1 && min_dist_between_samples > 0.f && rays_o &&
           rays_d && ray_t_entry && ray_t_exit && ray_max_dt && samples_idx && samples_3d &&
           samples_dirs && samples_z && ray_start_end_idx);
  if (nr_rays == 0) return VSA_OK;
  hipLaunchKernelGGL(sample_fg_kernel, dim3(vsa_div_up(nr_rays, 256)), dim3(256), 0,
                     (hipStream_t)stream, rays_o, rays_d, ray_t_entry, ray_t_exit,
                     min_dist_between_samples, min_nr_samples_per_ray, max_nr_samples_per_ray, jitter,
                     (unsigned long long)rng_state, (unsigned long long)rng_inc, ray_max_dt,
                     samples_idx, samples_3d, samples_dirs, samples_z, ray_start_end_idx, nr_rays);
  VSA_RETURN_LAUNCH_STATUS();
}
extern "C" int vsa_pack_compact(const int32_t* start_end, const int32_t* out_start,
                                const int32_t* samples_idx, const float* samples_3d,
                                const float* samples_dirs, const float* samples_z,
                                const float* samples_dt, const float* samples_values, int values_dim,
                                int32_t* out_idx, float* out_3d, float* out_dirs, float* out_z,
                                float* out_dt, float* out_values, int32_t* out_start_end,
                                int nr_rays, void* stream) {
  PK_CHECK(nr_rays >= 0 && values_dim >= 0 && start_end && out_start && samples_idx && samples_3d &&
           samples_dirs && samples_z && samples_dt && out_idx && out_3d && out_dirs && out_z &&
           out_dt && out_start_end && (values_dim == 0 || (samples_values && out_values)));
  PK_LAUNCH(pack_compact_kernel, nr_rays, start_end, out_start, samples_idx, samples_3d, samples_dirs,
            samples_z, samples_dt, samples_values, values_dim, out_idx, out_3d, out_dirs, out_z,
            out_dt, out_values, out_start_end, nr_rays);
}
extern "C" int vsa_importance_sample(const float* rays_o, const float* rays_d,
                                     const int32_t* start_end, const float* samples_z,
                                     const float* samples_cdf, int nr_importance_samples, int jitter,
                                     uint64_t rng_state, uint64_t rng_inc, float* out_3d,
                                     float* out_dirs, float* out_z, int32_t* out_start_end,
                                     int nr_rays, void* stream) {
  PK_CHECK(nr_rays >= 0 && nr_importance_samples >= 1 && rays_o && rays_d && start_end && samples_z &&
           samples_cdf && out_3d && out_dirs && out_z && out_start_end);
  if (nr_rays == 0) return VSA_OK;
  hipLaunchKernelGGL(importance_sample_kernel,
                     dim3(vsa_div_up((long long)nr_rays * nr_importance_samples, 256)), dim3(256), 0,
                     (hipStream_t)stream, rays_o, rays_d, start_end, samples_z, samples_cdf,
                     nr_importance_samples, jitter, (unsigned long long)rng_state,
                     (unsigned long long)rng_inc, out_3d, out_dirs, out_z, out_start_end, nr_rays);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_uncontract_samples(const float* ray_o, const int32_t* start_end,
                                      const float* samples_3d, const float* samples_z,
                                      float* out_samples_3d, float* out_samples_z, int nr_rays,
                                      void* stream) {
  PK_CHECK(nr_rays >= 0 && ray_o && start_end && samples_3d && samples_z && out_samples_3d &&
           out_samples_z);
  PK_LAUNCH(uncontract_kernel, nr_rays, ray_o, start_end, samples_3d, samples_z, out_samples_3d,
            out_samples_z, nr_rays);
}

extern "C" int vsa_combine_packs(const int32_t* start_end_1, const int32_t* idx_1, const float* s3d_1,
                                 const float* dirs_1, const float* z_1, const float* values_1,
                                 const int32_t* start_end_2, const int32_t* idx_2, const float* s3d_2,
                                 const float* dirs_2, const float* z_2, const float* values_2,
                                 const int32_t* out_start, float min_dist_between_samples,
                                 int values_dim, int32_t* out_idx, float* out_3d, float* out_dirs,
                                 float* out_z, float* out_values, int32_t* out_start_end,
                                 int nr_rays, void* stream) {
  PK_CHECK(nr_rays >= 0 && values_dim >= 0 && start_end_1 && idx_1 && s3d_1 && dirs_1 && z_1 &&
           start_end_2 && idx_2 && s3d_2 && dirs_2 && z_2 && out_start && out_idx && out_3d &&
           out_dirs && out_z && out_start_end &&
           (values_dim == 0 || (values_1 && values_2 && out_values)));
  if (nr_rays == 0) return VSA_OK;
  const PackView a{start_end_1, idx_1, s3d_1, dirs_1, z_1, values_1};
  const PackView b{start_end_2, idx_2, s3d_2, dirs_2, z_2, values_2};
  hipLaunchKernelGGL(combine_packs_kernel, dim3(vsa_div_up(nr_rays, 256)), dim3(256), 0,
                     (hipStream_t)stream, a, b, out_start, min_dist_between_samples, values_dim,
                     out_idx, out_3d, out_dirs, out_z, out_values, out_start_end, nr_rays);
  VSA_RETURN_LAUNCH_STATUS();
}
