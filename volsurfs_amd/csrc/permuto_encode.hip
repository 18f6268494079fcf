// Permutohedral-lattice hash encoding (`PermutoHashEncoder`,
// /root/reference/volsurfs_py/encodings/permutohash.py:10-99, used by models/rgb.py:104-149).
//
// The reference wraps `permutohedral_encoding.PermutoEncoding` (s-esposito fork of the
// PermutoSDF package, an un-vendored and unpinned submodule: .gitmodules:7-9) — source absent,
// PARITY UNPINNED.  This implements the published algorithm (Adams, Baek, Davis: "Fast
// high-dimensional filtering using the permutohedral lattice", 2010, as used for learned
// multi-resolution features by Rosu & Behnke, "PermutoSDF", 2023), restated in
// oracle/permuto.py:
//   per level l (scale sigma_l, random shift s_l):
//     c_i        = (x_i + s_l,i) * 1 / (sigma_l * sqrt((i+1)(i+2)))          i = 0..D-1
//     elevated   = the point lifted onto the hyperplane sum = 0 of R^(D+1)
//     rem0, rank = closest remainder-0 lattice point and the sorting permutation of the
//                  differences -> the enclosing simplex and its D+1 barycentric weights
//     vertex key = rem0 + k, minus (D+1) where rank > D - k                    k = 0..D
//     index      = (((key_0) * 2531011 + key_1) * 2531011 + ...) * 2531011 mod capacity
//     feature    = window_l * sum_k bary_k * values[l][index_k]
// One thread per (sample, level): D+1 float2 gathers from the level's 2 MiB table (24 levels =
// 48 MiB: L2 / MALL resident), fp32 throughout, operations in the oracle's order.
#include "common.h"

namespace {

template <int D>
struct Simplex {
  int rem0[D + 1];
  int rank[D + 1];
  float bary[D + 2];
};

template <int D>
__device__ __forceinline__ Simplex<D> permuto_simplex(const vsa_permuto_plan& p, int l,
                                                      const float* __restrict__ x) {
  Simplex<D> s;
  float el[D + 1];
  float sm = 0.f;
#pragma unroll
  for (int i = D; i > 0; --i) {
    const float cf = (x[i - 1] + p.random_shift[l][i - 1]) * p.scale_factor[l][i - 1];
    el[i] = sm - (float)i * cf;
    sm = sm + cf;
  }
  el[0] = sm;
  int sum = 0;
  constexpr float inv = 1.0f / (D + 1);
#pragma unroll
  for (int i = 0; i <= D; ++i) {
    const float v = el[i] * inv;
    const float up = ceilf(v) * (float)(D + 1);
    const float down = floorf(v) * (float)(D + 1);
    s.rem0[i] = (up - el[i] < el[i] - down) ? (int)up : (int)down;
    sum += s.rem0[i];
    s.rank[i] = 0;
  }
  sum /= (D + 1);
#pragma unroll
  for (int i = 0; i < D; ++i) {
    const float di = el[i] - (float)s.rem0[i];
#pragma unroll
    for (int j = i + 1; j <= D; ++j) {
      if (di < el[j] - (float)s.rem0[j]) s.rank[i]++;
      else s.rank[j]++;
    }
  }
#pragma unroll
  for (int i = 0; i <= D; ++i) {
    s.rank[i] += sum;
    if (s.rank[i] < 0) {
      s.rank[i] += D + 1;
      s.rem0[i] += D + 1;
    } else if (s.rank[i] > D) {
      s.rank[i] -= D + 1;
      s.rem0[i] -= D + 1;
    }
  }
#pragma unroll
  for (int i = 0; i <= D + 1; ++i) s.bary[i] = 0.f;
#pragma unroll
  for (int i = 0; i <= D; ++i) {
    const float delta = (el[i] - (float)s.rem0[i]) * inv;
    // (rank is data dependent: unrolled selects instead of indexed private arrays)
#pragma unroll
    for (int k = 0; k <= D + 1; ++k) {
      if (k == D - s.rank[i]) s.bary[k] = s.bary[k] + delta;
      if (k == D + 1 - s.rank[i]) s.bary[k] = s.bary[k] - delta;
    }
  }
  s.bary[0] = s.bary[0] + (1.0f + s.bary[D + 1]);
  return s;
}

template <int D>
__device__ __forceinline__ unsigned permuto_index(const Simplex<D>& s, int k, unsigned capacity) {
  unsigned h = 0;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    int key = s.rem0[i] + k;
    if (s.rank[i] > D - k) key -= D + 1;
    h += (unsigned)key;
    h *= 2531011u;
  }
  return h % capacity;
}

template <int D>
__global__ __launch_bounds__(256) void permuto_fwd_kernel(vsa_permuto_plan plan,
                                                          const float2* __restrict__ values,
                                                          const float* __restrict__ x,
                                                          const float* __restrict__ window, int B,
                                                          float* __restrict__ out, int out_stride) {
  const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
  const int l = blockIdx.y;
  if (b >= B) return;
  const Simplex<D> s = permuto_simplex<D>(plan, l, x + b * D);
  const float2* tab = values + (long long)l * plan.capacity;
  const float wl = window ? window[l] : 1.0f;
  float f0 = 0.f, f1 = 0.f;
#pragma unroll
  for (int k = 0; k <= D; ++k) {
    const float2 v = tab[permuto_index<D>(s, k, (unsigned)plan.capacity)];
    const float w = s.bary[k] * wl;
    f0 = f0 + v.x * w;
    f1 = f1 + v.y * w;
  }
  *reinterpret_cast<float2*>(out + b * out_stride + 2 * l) = make_float2(f0, f1);
}

// two lanes per (sample, level), one per feature: the two atomics of an entry leave the wave
// as one request (as grid_encode_bwd)
template <int D>
__global__ __launch_bounds__(256) void permuto_bwd_kernel(vsa_permuto_plan plan,
                                                          const float* __restrict__ x,
                                                          const float* __restrict__ window,
                                                          const float* __restrict__ g_out,
                                                          int g_stride, int B,
                                                          float* __restrict__ g_values) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long b = t >> 1;
  const int f = (int)(t & 1);
  const int l = blockIdx.y;
  if (b >= B) return;
  const float wl = window ? window[l] : 1.0f;
  const float go = g_out[b * g_stride + 2 * l + f];
  if (go == 0.f || wl == 0.f) return;
  const Simplex<D> s = permuto_simplex<D>(plan, l, x + b * D);
  float* tab = g_values + 2ll * l * plan.capacity;
#pragma unroll
  for (int k = 0; k <= D; ++k) {
    const float w = s.bary[k] * wl;
    atomicAdd(tab + 2ll * permuto_index<D>(s, k, (unsigned)plan.capacity) + f, go * w);
  }
}

int plan_ok(const vsa_permuto_plan* p) {
  if (!p) return VSA_ERR_ARG;
  if (p->pos_dim < 2 || p->pos_dim > 4) return VSA_ERR_UNSUPPORTED;
  if (p->n_features != 2) return VSA_ERR_UNSUPPORTED;
  if (p->n_levels < 1 || p->n_levels > VSA_GRID_MAX_LEVELS || p->capacity < 1) return VSA_ERR_ARG;
  return VSA_OK;
}

}  // namespace

extern "C" int vsa_permuto_encode_fwd(const vsa_permuto_plan* plan, const float* lattice_values,
                                      const float* x, const float* window, int nr_points,
                                      float* out, int out_stride, void* stream) {
  int rc = plan_ok(plan);
  if (rc) return rc;
  if (nr_points < 0 || out_stride < 2 * plan->n_levels || (out_stride & 1)) return VSA_ERR_ARG;
  if (nr_points == 0) return VSA_OK;
  if (!lattice_values || !x || !out) return VSA_ERR_ARG;
  dim3 grid(vsa_div_up(nr_points, 256), plan->n_levels);
  const float2* v = reinterpret_cast<const float2*>(lattice_values);
#define VSA_PERMUTO_FWD(D)                                                                      \
  hipLaunchKernelGGL(permuto_fwd_kernel<D>, grid, dim3(256), 0, (hipStream_t)stream, *plan, v, x, \
                     window, nr_points, out, out_stride)
  if (plan->pos_dim == 2) VSA_PERMUTO_FWD(2);
  else if (plan->pos_dim == 3) VSA_PERMUTO_FWD(3);
  else VSA_PERMUTO_FWD(4);
#undef VSA_PERMUTO_FWD
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_permuto_encode_bwd(const vsa_permuto_plan* plan, const float* x,
                                      const float* window, const float* g_out, int g_stride,
                                      int nr_points, float* grad_values, void* stream) {
  int rc = plan_ok(plan);
  if (rc) return rc;
  if (nr_points < 0 || g_stride < 2 * plan->n_levels) return VSA_ERR_ARG;
  if (nr_points == 0) return VSA_OK;
  if (!x || !g_out || !grad_values) return VSA_ERR_ARG;
  dim3 grid(vsa_div_up(2ll * nr_points, 256), plan->n_levels);
#define VSA_PERMUTO_BWD(D)                                                                       \
  hipLaunchKernelGGL(permuto_bwd_kernel<D>, grid, dim3(256), 0, (hipStream_t)stream, *plan, x,   \
                     window, g_out, g_stride, nr_points, grad_values)
  if (plan->pos_dim == 2) VSA_PERMUTO_BWD(2);
  else if (plan->pos_dim == 3) VSA_PERMUTO_BWD(3);
  else VSA_PERMUTO_BWD(4);
#undef VSA_PERMUTO_BWD
  VSA_RETURN_LAUNCH_STATUS();
}
