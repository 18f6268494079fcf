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
#include <cstdint>

#ifndef PERMUTO_LDS_LEVELS
#define PERMUTO_LDS_LEVELS 10
#endif
#ifndef PERMUTO_LDS_MIN_POINTS
#define PERMUTO_LDS_MIN_POINTS 512
#endif

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

// One launch covers up to PG_MAX encodings of one geometry (the position encoders of the K per-shell
// models): group = blockIdx.z owns rows [row0, row0 + n) of x / out / g_out and its own lattice
// values.  The per-level shifts may differ between the groups, so the DEV instances read their plan
// from a device array (a plan is 1 KiB: eight do not fit the kernel arguments); the single-encoding
// entry points keep the by-value plan.
constexpr int PG_MAX = 8;
struct PermutoGroups {
  int n[PG_MAX];
  long long row0[PG_MAX];
  const float* values[PG_MAX];
  float* grads[PG_MAX];
};
template <class T, int N>
__device__ __forceinline__ T pg_pick(const T (&a)[N], int g) {      // no run-time index into kernel arguments
  T r = a[0];
#pragma unroll
  for (int i = 1; i < N; ++i)
    if (g == i) r = a[i];
  return r;
}

// LV = 2: two levels per thread and one 16-byte store (as grid_encode_fwd_kernel: one level per thread writes 8 bytes
// per lane a whole row apart)
template <int D, bool DEV, int LV>
__global__ __launch_bounds__(256) void permuto_fwd_kernel(vsa_permuto_plan plan_val,
                                                          const vsa_permuto_plan* __restrict__ plans_dev,
                                                          PermutoGroups gp,
                                                          const float* __restrict__ x,
                                                          const float* __restrict__ window,
                                                          float* __restrict__ out, int out_stride) {
  const int grp = blockIdx.z;
  const vsa_permuto_plan& plan = DEV ? plans_dev[grp] : plan_val;
  const int B = pg_pick(gp.n, grp);
  const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  {
    const long long r0 = pg_pick(gp.row0, grp);
    x += r0 * D;
    out += r0 * out_stride;
  }
  const float2* values = reinterpret_cast<const float2*>(pg_pick(gp.values, grp));
  float res[2 * LV];
#pragma unroll
  for (int lv = 0; lv < LV; ++lv) {
    const int l = blockIdx.y * LV + lv;
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
    res[2 * lv] = f0, res[2 * lv + 1] = f1;
  }
  float* o = out + b * out_stride + 2 * LV * blockIdx.y;
  if constexpr (LV == 2) *reinterpret_cast<float4*>(o) = make_float4(res[0], res[1], res[2], res[3]);
  else *reinterpret_cast<float2*>(o) = make_float2(res[0], res[1]);
}

// two lanes per (sample, level), one per feature: the two atomics of an entry leave the wave
// as one request (as grid_encode_bwd)
template <int D, bool DEV>
__global__ __launch_bounds__(256) void permuto_bwd_kernel(vsa_permuto_plan plan_val,
                                                          const vsa_permuto_plan* __restrict__ plans_dev,
                                                          PermutoGroups gp,
                                                          const float* __restrict__ x,
                                                          const float* __restrict__ window,
                                                          const float* __restrict__ g_out,
                                                          int g_stride, int l0) {
  const int grp = blockIdx.z;
  const vsa_permuto_plan& plan = DEV ? plans_dev[grp] : plan_val;
  const int B = pg_pick(gp.n, grp);
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long b = t >> 1;
  const int f = (int)(t & 1);
  const int l = l0 + blockIdx.y;
  if (b >= B) return;
  {
    const long long r0 = pg_pick(gp.row0, grp);
    x += r0 * D;
    g_out += r0 * g_stride;
  }
  float* g_values = pg_pick(gp.grads, grp);
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

// ---- the coarse levels: accumulate in LDS first.
// The kernel above spends most of its time on the COARSE levels (measured on the 10 k points one
// per-shell model sees in the training loop of BASELINE configs[2]: levels 0-7 take 134 of its
// 172 us, levels 12-23 together 19 us): thousands of points share the few hundred lattice
// vertices a coarse level has on the surface, and device-scope float atomics on one address are
// serialised at the memory side.  Here a workgroup owns (level, chunk of PL_CHUNK points: 1 024 —
// with 4 096 the ten levels of a 10 k-point batch were 30 workgroups, the slowest 51 us): it adds
// the chunk's contributions into an open-addressing table in LDS (keys claimed with ds_cmpst,
// values 64-bit FIXED POINT as in grid_encode.hip: ds_add_u64 is ~16x faster than ds_add_f32 on
// gfx950, and integer sums make the result independent of the order inside the chunk), and then
// flushes each distinct vertex ONCE.  The power-of-two scale comes from the chunk's own max|g|
// (first pass over the chunk's gradient column).  A contribution that finds neither its key nor a
// free slot within PL_PROBES probes (more distinct vertices than slots) goes straight to memory
// like before.
constexpr int PL_THREADS = 512;
constexpr int PL_SLOTS = 4096;            // keys 16 KiB + 2 x 32 KiB of accumulators
constexpr int PL_PROBES = 16;
#ifndef PL_CHUNK_POINTS
#define PL_CHUNK_POINTS 1024
#endif
constexpr int PL_CHUNK = PL_CHUNK_POINTS;  // points per workgroup
constexpr unsigned PL_EMPTY = 0xffffffffu;

__device__ __forceinline__ unsigned long long pl_fixed62(float v) {      // as grid_encode.hip's fixed62
  const float r = rintf(v);
  const float hi = floorf(r * 2.3283064365386963e-10f);
  const float lo = r - hi * 4294967296.0f;
  return ((unsigned long long)(unsigned)(int)hi << 32) + (unsigned long long)(unsigned)lo;
}

template <int D, bool DEV>
__global__ __launch_bounds__(PL_THREADS) void permuto_bwd_lds_kernel(vsa_permuto_plan plan_val,
                                                                     const vsa_permuto_plan* __restrict__ plans_dev,
                                                                     PermutoGroups gp,
                                                                     const float* __restrict__ x,
                                                                     const float* __restrict__ window,
                                                                     const float* __restrict__ g_out,
                                                                     int g_stride) {
  __shared__ unsigned s_key[PL_SLOTS];
  __shared__ unsigned long long s_val[2 * PL_SLOTS];
  __shared__ float s_max[PL_THREADS / 64];
  const int grp = blockIdx.z;
  const vsa_permuto_plan& plan = DEV ? plans_dev[grp] : plan_val;
  const int B = pg_pick(gp.n, grp);
  const int l = blockIdx.y;
  const long long p0 = (long long)blockIdx.x * PL_CHUNK;
  if (p0 >= B) return;                        // uniform: this group has fewer chunks than the largest
  {
    const long long r0 = pg_pick(gp.row0, grp);
    x += r0 * D;
    g_out += r0 * g_stride;
  }
  float* g_values = pg_pick(gp.grads, grp);
  const int np = (int)min((long long)PL_CHUNK, (long long)B - p0);
  const float wl = window ? window[l] : 1.0f;
  if (wl == 0.f) return;
  for (int i = threadIdx.x; i < PL_SLOTS; i += PL_THREADS) {
    s_key[i] = PL_EMPTY;
    s_val[2 * i] = 0ull;
    s_val[2 * i + 1] = 0ull;
  }
  // max |g| of this chunk's column pair
  float m = 0.f;
  for (int t = threadIdx.x; t < 2 * np; t += PL_THREADS) {
    float a = fabsf(g_out[(p0 + (t >> 1)) * g_stride + 2 * l + (t & 1)]);
    if (!(a < INFINITY)) a = 0.f;
    m = fmaxf(m, a);
  }
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
  __syncthreads();
  m = 0.f;
#pragma unroll
  for (int w = 0; w < PL_THREADS / 64; ++w) m = fmaxf(m, s_max[w]);
  if (m == 0.f) return;                       // uniform: nothing to add
  // |contribution| <= m |wl| (barycentric weights are in [0, 1]); a slot receives at most
  // (D + 1) * np of them: 62 - count_bits - exponent bits of headroom
  int e;
  frexpf(m * fabsf(wl), &e);
  int count_bits = 1;
  while ((1 << count_bits) < (D + 1) * PL_CHUNK) ++count_bits;
  const float scale = ldexpf(1.0f, 62 - count_bits - e);
  float* tab = g_values + 2ll * l * plan.capacity;
  for (int t = threadIdx.x; t < 2 * np; t += PL_THREADS) {
    const long long b = p0 + (t >> 1);
    const int f = t & 1;
    const float go = g_out[b * g_stride + 2 * l + f];
    if (go == 0.f || !(fabsf(go) < INFINITY)) {
      if (go != 0.f) {                        // inf / nan: as the plain kernel would propagate it
        const Simplex<D> s = permuto_simplex<D>(plan, l, x + b * D);
#pragma unroll
        for (int k = 0; k <= D; ++k)
          atomicAdd(tab + 2ll * permuto_index<D>(s, k, (unsigned)plan.capacity) + f, go * (s.bary[k] * wl));
      }
      continue;
    }
    const Simplex<D> s = permuto_simplex<D>(plan, l, x + b * D);
#pragma unroll
    for (int k = 0; k <= D; ++k) {
      const float w = s.bary[k] * wl;
      const unsigned idx = permuto_index<D>(s, k, (unsigned)plan.capacity);
      unsigned slot = (idx * 2654435761u) >> 20;          // 12 bits
      int found = -1;
      for (int pr = 0; pr < PL_PROBES; ++pr) {
        const unsigned prev = atomicCAS(&s_key[slot], PL_EMPTY, idx);
        if (prev == PL_EMPTY || prev == idx) {
          found = (int)slot;
          break;
        }
        slot = (slot + 1) & (PL_SLOTS - 1);
      }
      if (found >= 0) atomicAdd(&s_val[2 * found + f], pl_fixed62((go * w) * scale));
      else atomicAdd(tab + 2ll * idx + f, go * w);
    }
  }
  __syncthreads();
  const double inv = 1.0 / (double)scale;
  for (int i = threadIdx.x; i < 2 * PL_SLOTS; i += PL_THREADS) {
    const unsigned key = s_key[i >> 1];
    const long long v = (long long)s_val[i];
    if (key != PL_EMPTY && v != 0) atomicAdd(tab + 2ll * key + (i & 1), (float)((double)v * inv));
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

namespace {

int make_permuto_groups(const vsa_permuto_plan* plan, int nr_groups, const int* nr_points,
                        const float* const* values, float* const* grads, PermutoGroups* gp, int* max_n) {
  int rc = plan_ok(plan);
  if (rc) return rc;
  if (nr_groups < 1 || nr_groups > PG_MAX || !nr_points) return VSA_ERR_ARG;
  long long row = 0;
  int mx = 0;
  for (int g = 0; g < PG_MAX; ++g) {
    const bool in = g < nr_groups;
    if (in && nr_points[g] < 0) return VSA_ERR_ARG;
    gp->n[g] = in ? nr_points[g] : 0;
    gp->row0[g] = row;
    gp->values[g] = in && values ? values[g] : nullptr;
    gp->grads[g] = in && grads ? grads[g] : nullptr;
    if (in) {
      if (nr_points[g] > 0 && ((values && !values[g]) || (grads && !grads[g]))) return VSA_ERR_ARG;
      row += nr_points[g];
      mx = nr_points[g] > mx ? nr_points[g] : mx;
    }
  }
  *max_n = mx;
  return VSA_OK;
}

// plans_dev == nullptr: one group, plan by value
int permuto_fwd_launch(const vsa_permuto_plan* plan, const vsa_permuto_plan* plans_dev,
                       const PermutoGroups& gp, int nr_groups, int max_n, const float* x,
                       const float* window, float* out, int out_stride, hipStream_t st) {
  // two levels per thread where a thread's pair is a 16-byte group of a 16-byte aligned row
  const bool two = plan->n_levels % 2 == 0 && (out_stride & 3) == 0 && ((uintptr_t)out & 15) == 0;
  dim3 grid(vsa_div_up(max_n, 256), two ? plan->n_levels / 2 : plan->n_levels, nr_groups);
#define VSA_PERMUTO_FWD_(D, DEVF, LV)                                                                     \
  hipLaunchKernelGGL((permuto_fwd_kernel<D, DEVF, LV>), grid, dim3(256), 0, st, *plan, plans_dev, gp, x, window, out, \
                     out_stride)
#define VSA_PERMUTO_FWD(D)                                                                          \
  do {                                                                                              \
    if (plans_dev) {                                                                                \
      if (two) VSA_PERMUTO_FWD_(D, true, 2);                                                        \
      else VSA_PERMUTO_FWD_(D, true, 1);                                                            \
    } else {                                                                                        \
      if (two) VSA_PERMUTO_FWD_(D, false, 2);                                                       \
      else VSA_PERMUTO_FWD_(D, false, 1);                                                           \
    }                                                                                               \
  } while (0)
  if (plan->pos_dim == 2) VSA_PERMUTO_FWD(2);
  else if (plan->pos_dim == 3) VSA_PERMUTO_FWD(3);
  else VSA_PERMUTO_FWD(4);
#undef VSA_PERMUTO_FWD
#undef VSA_PERMUTO_FWD_
  VSA_RETURN_LAUNCH_STATUS();
}

int permuto_bwd_launch(const vsa_permuto_plan* plan, const vsa_permuto_plan* plans_dev,
                       const PermutoGroups& gp, int nr_groups, int max_n, const float* x,
                       const float* window, const float* g_out, int g_stride, hipStream_t st) {
  // the first PERMUTO_LDS_LEVELS levels (the coarse ones: levels are ordered coarse to fine) go
  // through the LDS tables, the rest straight to memory
  int n_lds = PERMUTO_LDS_LEVELS < plan->n_levels ? PERMUTO_LDS_LEVELS : plan->n_levels;
  if (max_n < PERMUTO_LDS_MIN_POINTS) n_lds = 0;
#define VSA_PERMUTO_BWD(D, DEV)                                                                     \
  do {                                                                                              \
    if (n_lds > 0)                                                                                  \
      hipLaunchKernelGGL((permuto_bwd_lds_kernel<D, DEV>), dim3(vsa_div_up(max_n, PL_CHUNK), n_lds, nr_groups), \
                         dim3(PL_THREADS), 0, st, *plan, plans_dev, gp, x, window, g_out, g_stride); \
    if (n_lds < plan->n_levels)                                                                     \
      hipLaunchKernelGGL((permuto_bwd_kernel<D, DEV>),                                              \
                         dim3(vsa_div_up(2ll * max_n, 256), plan->n_levels - n_lds, nr_groups), dim3(256), 0, st, \
                         *plan, plans_dev, gp, x, window, g_out, g_stride, n_lds);                  \
  } while (0)
  if (plans_dev) {
    if (plan->pos_dim == 2) VSA_PERMUTO_BWD(2, true);
    else if (plan->pos_dim == 3) VSA_PERMUTO_BWD(3, true);
    else VSA_PERMUTO_BWD(4, true);
  } else {
    if (plan->pos_dim == 2) VSA_PERMUTO_BWD(2, false);
    else if (plan->pos_dim == 3) VSA_PERMUTO_BWD(3, false);
    else VSA_PERMUTO_BWD(4, false);
  }
#undef VSA_PERMUTO_BWD
  VSA_RETURN_LAUNCH_STATUS();
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
  PermutoGroups gp;
  int mx = 0;
  rc = make_permuto_groups(plan, 1, &nr_points, &lattice_values, nullptr, &gp, &mx);
  if (rc) return rc;
  return permuto_fwd_launch(plan, nullptr, gp, 1, mx, x, window, out, out_stride, (hipStream_t)stream);
}

extern "C" int vsa_permuto_encode_bwd(const vsa_permuto_plan* plan, const float* x,
                                      const float* window, const float* g_out, int g_stride,
                                      int nr_points, float* grad_values, void* stream) {
  int rc = plan_ok(plan);
  if (rc) return rc;
  if (nr_points < 0 || g_stride < 2 * plan->n_levels) return VSA_ERR_ARG;
  if (nr_points == 0) return VSA_OK;
  if (!x || !g_out || !grad_values) return VSA_ERR_ARG;
  PermutoGroups gp;
  int mx = 0;
  rc = make_permuto_groups(plan, 1, &nr_points, nullptr, &grad_values, &gp, &mx);
  if (rc) return rc;
  return permuto_bwd_launch(plan, nullptr, gp, 1, mx, x, window, g_out, g_stride, (hipStream_t)stream);
}

// Up to 8 encodings of one geometry in one launch each way.  plan_host: group 0's plan (levels,
// dimension, capacity: shared); plans_dev: the nr_groups plans in DEVICE memory (their shifts may
// differ); lattice_values / grad_values: HOST arrays of nr_groups device pointers; group g owns the
// rows after the first g groups' in x / out / g_out.
extern "C" int vsa_permuto_encode_fwd_grouped(const vsa_permuto_plan* plan_host,
                                              const vsa_permuto_plan* plans_dev,
                                              const float* const* lattice_values, int nr_groups,
                                              const int* nr_points, const float* x,
                                              const float* window, float* out, int out_stride,
                                              void* stream) {
  int rc = plan_ok(plan_host);
  if (rc) return rc;
  if (!plans_dev || !lattice_values || out_stride < 2 * plan_host->n_levels || (out_stride & 1)) return VSA_ERR_ARG;
  PermutoGroups gp;
  int mx = 0;
  rc = make_permuto_groups(plan_host, nr_groups, nr_points, lattice_values, nullptr, &gp, &mx);
  if (rc) return rc;
  if (mx == 0) return VSA_OK;
  if (!x || !out) return VSA_ERR_ARG;
  return permuto_fwd_launch(plan_host, plans_dev, gp, nr_groups, mx, x, window, out, out_stride,
                            (hipStream_t)stream);
}

extern "C" int vsa_permuto_encode_bwd_grouped(const vsa_permuto_plan* plan_host,
                                              const vsa_permuto_plan* plans_dev, int nr_groups,
                                              const int* nr_points, const float* x,
                                              const float* window, const float* g_out, int g_stride,
                                              float* const* grad_values, void* stream) {
  int rc = plan_ok(plan_host);
  if (rc) return rc;
  if (!plans_dev || !grad_values || g_stride < 2 * plan_host->n_levels) return VSA_ERR_ARG;
  PermutoGroups gp;
  int mx = 0;
  rc = make_permuto_groups(plan_host, nr_groups, nr_points, nullptr, grad_values, &gp, &mx);
  if (rc) return rc;
  if (mx == 0) return VSA_OK;
  if (!x || !g_out) return VSA_ERR_ARG;
  return permuto_bwd_launch(plan_host, plans_dev, gp, nr_groups, mx, x, window, g_out, g_stride,
                            (hipStream_t)stream);
}
