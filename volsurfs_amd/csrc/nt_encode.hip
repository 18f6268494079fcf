// Neural-texture step 3 (forward) and its transpose (backward): 2-D
// multiresolution hash-grid encoding of every unique texel, LEVEL-MAJOR.
//
// tiny-cuda-nn's grid kernel (the reference's neural_texture.py:54-63) gathers
// 64 table entries per sample from global memory and scatters 64 atomics per
// sample in backward.  On MI355X a whole level (<= 2^15 entries x 2 features) fits
// one CU's LDS: 128 KiB as f16x2 (forward) / as f32 per feature (backward).  A
// workgroup therefore owns one (texture, level) pair, stages that level's table
// in LDS once, and streams a chunk of slots through it: coalesced 4-B reads of
// the texel ids, 4 LDS gathers, one coalesced 4-B feature write.  Backward is the
// same stream with ds_add_f32 into an LDS-resident gradient table that is
// flushed once with coalesced stores/adds — no scattered HBM atomics.
//
// Arithmetic is the oracle's (oracle/tcnn_like.py hashgrid_forward), fp32, in
// the same order, so features are bit-identical.
#include "nt_common.h"

namespace {

constexpr int ENC_BLOCK = 1024;     // 16 waves: the 128 KiB level pins one workgroup per CU
constexpr int ENC_UNROLL = 4;       // slots in flight per lane
constexpr int ENC_SPAN_FWD = 65536;   // slots per workgroup (forward)
constexpr int ENC_SPAN_BWD = 262144;  // slots per workgroup before a flush (backward)
constexpr unsigned PRIME_Y = 2654435761u;
constexpr int SMALL_LEVEL_ENTRIES = 8192;  // levels up to this size run in the small-LDS launch

struct LevelGeom {
  float scale;
  unsigned res, size, mask;
  bool hashed;
};

__device__ __forceinline__ LevelGeom level_geom(const vsa_nt_plan& p, int l) {
  LevelGeom g;
  g.scale = p.level_scale[l];
  g.res = (unsigned)p.level_res[l];
  g.size = (unsigned)p.level_size[l];
  // tiny-cuda-nn grid_index: dense while the running stride fits the table
  g.hashed = !((long long)g.res <= (long long)g.size && (long long)g.res * g.res <= (long long)g.size);
  g.mask = (g.size & (g.size - 1)) == 0 ? g.size - 1 : 0u;
  return g;
}

__device__ __forceinline__ unsigned level_index(const LevelGeom& g, unsigned cx, unsigned cy) {
  unsigned idx = g.hashed ? (cx ^ (cy * PRIME_Y)) : (cx + cy * g.res);
  if (g.mask) return idx & g.mask;           // power-of-two table: modulo is a mask
  return idx < g.size ? idx : idx % g.size;  // dense level: in range except at the far edge
}

struct CellCorners {
  unsigned idx[4];
  float w[4];
};

// normalised texel centre -> the 4 table entries and bilinear weights at level g
// (oracle/tcnn_like.py hashgrid_forward, same fp32 operations in the same order)
__device__ __forceinline__ CellCorners cell_corners(const LevelGeom& g, float x, float y) {
  const float px = x * g.scale + 0.5f, py = y * g.scale + 0.5f;
  const float flx = floorf(px), fly = floorf(py);
  const float fx = px - flx, fy = py - fly;
  const unsigned cx = (unsigned)(int)flx, cy = (unsigned)(int)fly;
  CellCorners c;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int dx = k & 1, dy = k >> 1;
    const float wx = dx ? fx : 1.0f - fx;
    const float wy = dy ? fy : 1.0f - fy;
    c.w[k] = wx * wy;
    c.idx[k] = level_index(g, cx + dx, cy + dy);
  }
  return c;
}

__device__ __forceinline__ bool tex_active(const vsa_nt_plan& p, int tex) {
  const int deg = tex % VSA_NT_MAX_DEG;
  const int type = (tex / VSA_NT_MAX_DEG) & 1;
  const int shell = tex / (2 * VSA_NT_MAX_DEG);
  if (type == 0) return deg < p.rgb_degrees;
  if (p.inner_solid && shell == 0) return false;
  return deg < p.alpha_degrees;
}

// Work decode.  blockIdx.x enumerates (shell*2+type, degree, group) with a
// per-degree group count derived from the segment's capacity
// min(4*max_rays, (R_d+2)^2), so that small textures do not launch the
// worst-case number of (LDS-hungry) workgroups.
__host__ __device__ inline int groups_of_degree(const vsa_nt_plan& p, int d, int span) {
  const long long T = (long long)(p.tex_res[d] + 2) * (p.tex_res[d] + 2);
  long long cap = 4ll * p.max_rays < T ? 4ll * p.max_rays : T;
  if (cap < 1) cap = 1;
  return (int)((cap + span - 1) / span);
}

struct Work {
  int tex, group;
  int begin, end;  // slot range of the (shell, degree) segment
};

__device__ __forceinline__ bool decode_work(const vsa_nt_plan& p, const int* seg_start, int span,
                                            int bx, Work& w) {
  int per_model = 0;
  for (int d = 0; d < VSA_NT_MAX_DEG; ++d) per_model += groups_of_degree(p, d, span);
  const int model = bx / per_model;  // shell*2 + type
  int r = bx - model * per_model;
  int d = 0;
  for (; d < VSA_NT_MAX_DEG; ++d) {
    const int g = groups_of_degree(p, d, span);
    if (r < g) break;
    r -= g;
  }
  w.tex = model * VSA_NT_MAX_DEG + d;
  w.group = r;
  if (!tex_active(p, w.tex)) return false;
  const int sd = (model >> 1) * VSA_NT_MAX_DEG + d;
  w.begin = seg_start[sd];
  w.end = seg_start[sd + 1];
  return w.begin + (long long)r * span < w.end;
}

__global__ __launch_bounds__(ENC_BLOCK) void nt_encode_fwd_kernel(
    vsa_nt_plan plan, const half2_t* __restrict__ tables, const float2* __restrict__ slot_xy,
    const int* __restrict__ seg_start, half2_t* __restrict__ features, int level0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  half2_t* s_tab = reinterpret_cast<half2_t*>(s_raw);
  const int level = level0 + blockIdx.y;
  Work wk;
  if (!decode_work(plan, seg_start, ENC_SPAN_FWD, blockIdx.x, wk)) return;
  const int first = wk.begin + wk.group * ENC_SPAN_FWD;
  const int last = min(wk.end, first + ENC_SPAN_FWD);
  const LevelGeom g = level_geom(plan, level);
  const long long n_entries = plan.level_offset[plan.n_levels];
  const half2_t* tab = tables + (long long)wk.tex * n_entries + plan.level_offset[level];
  {  // stage the level (size is a multiple of 8 entries = 32 B)
    const uint4* src = reinterpret_cast<const uint4*>(tab);
    uint4* dst = reinterpret_cast<uint4*>(s_tab);
    for (int i = threadIdx.x; i < (int)(g.size / 4); i += ENC_BLOCK) dst[i] = src[i];
  }
  __syncthreads();
  const int type = (wk.tex / VSA_NT_MAX_DEG) & 1;
  half2_t* out = features + ((long long)type * plan.n_levels + level) * plan.slot_capacity;
  for (int base = first + threadIdx.x; base < last; base += ENC_BLOCK * ENC_UNROLL) {
    float2 xy[ENC_UNROLL];
#pragma unroll
    for (int u = 0; u < ENC_UNROLL; ++u) {
      const int slot = base + u * ENC_BLOCK;
      xy[u] = slot_xy[slot < last ? slot : last - 1];
    }
    CellCorners c[ENC_UNROLL];
#pragma unroll
    for (int u = 0; u < ENC_UNROLL; ++u) c[u] = cell_corners(g, xy[u].x, xy[u].y);
    half2_t v[ENC_UNROLL][4];
#pragma unroll
    for (int u = 0; u < ENC_UNROLL; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) v[u][k] = s_tab[c[u].idx[k]];
#pragma unroll
    for (int u = 0; u < ENC_UNROLL; ++u) {
      const int slot = base + u * ENC_BLOCK;
      float f0 = 0.f, f1 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f0 = f0 + c[u].w[k] * (float)v[u][k].x;
        f1 = f1 + c[u].w[k] * (float)v[u][k].y;
      }
      half2_t r;
      r.x = (_Float16)f0;
      r.y = (_Float16)f1;
      if (slot < last) out[slot] = r;
    }
  }
}

// Backward: grad_table[tex][level entries][feature] += w * dF[slot]; one
// workgroup per (texture, level, feature, group) with the feature's gradient
// plane (size entries) in LDS.
//
// gfx950's LDS float atomic (ds_add_f32) retires ~0.4 lanes/clk/CU on random
// addresses, the integer one >= 8x that (tools/ubench/lds_atomics.hip), so the
// plane is accumulated in 32-bit FIXED POINT with a per-workgroup scale that
// makes overflow impossible: S = 2^30 / sum_slots |dF| bounds every entry's
// |sum| (bilinear weights of a slot sum to 1) below 2^30.  Resolution is
// sum|dF| / 2^30, i.e. <= span / 2^30 = 2.4e-4 of the MEAN |dF| per add.
__global__ __launch_bounds__(ENC_BLOCK) void nt_encode_bwd_kernel(
    vsa_nt_plan plan, const half2_t* __restrict__ dfeatures, float dscale_inv,
    const float2* __restrict__ slot_xy, const int* __restrict__ seg_start,
    float* __restrict__ grad_tables, int level0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  int* s_g = reinterpret_cast<int*>(s_raw);
  __shared__ float s_red[ENC_BLOCK / 64];
  const int level = level0 + (blockIdx.y >> 1), feat = blockIdx.y & 1;
  Work wk;
  if (!decode_work(plan, seg_start, ENC_SPAN_BWD, blockIdx.x, wk)) return;
  const int first = wk.begin + wk.group * ENC_SPAN_BWD;
  const int last = min(wk.end, first + ENC_SPAN_BWD);
  const LevelGeom g = level_geom(plan, level);
  for (int i = threadIdx.x; i < (int)g.size; i += ENC_BLOCK) s_g[i] = 0;
  const int type = (wk.tex / VSA_NT_MAX_DEG) & 1;
  const _Float16* dF = reinterpret_cast<const _Float16*>(
                           dfeatures + ((long long)type * plan.n_levels + level) * plan.slot_capacity) + feat;
  // pass 1: sum |dF| over this workgroup's slots -> fixed-point scale
  float asum = 0.f;
  for (int slot = first + threadIdx.x; slot < last; slot += ENC_BLOCK) asum += fabsf((float)dF[2 * (long long)slot]);
  for (int off = 32; off > 0; off >>= 1) asum += __shfl_down(asum, off, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = asum;
  __syncthreads();
  float total = 0.f;
#pragma unroll
  for (int i = 0; i < ENC_BLOCK / 64; ++i) total += s_red[i];
  if (!(total > 0.f)) return;   // nothing to add (uniform across the workgroup)
  // power-of-two scale <= 2^30 / total  (exact scaling, exact un-scaling)
  int e;
  frexpf(total, &e);                       // total = m * 2^e, m in [0.5, 1)
  const float S = ldexpf(1.0f, 30 - e);
  const float S_inv = ldexpf(1.0f, e - 30) * dscale_inv;
  // pass 2: scatter
  for (int base = first + threadIdx.x; base < last; base += ENC_BLOCK * ENC_UNROLL) {
    float2 xy[ENC_UNROLL];
    float gv[ENC_UNROLL];
#pragma unroll
    for (int u = 0; u < ENC_UNROLL; ++u) {
      const int slot = base + u * ENC_BLOCK;
      const int sl = slot < last ? slot : last - 1;
      xy[u] = slot_xy[sl];
      gv[u] = slot < last ? (float)dF[2 * (long long)sl] * S : 0.f;
    }
#pragma unroll
    for (int u = 0; u < ENC_UNROLL; ++u) {
      if (gv[u] != 0.f) {
        const CellCorners c = cell_corners(g, xy[u].x, xy[u].y);
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(&s_g[c.idx[k]], __float2int_rn(c.w[k] * gv[u]));
      }
    }
  }
  __syncthreads();
  const long long n_entries = plan.level_offset[plan.n_levels];
  float* gt = grad_tables + ((long long)wk.tex * n_entries + plan.level_offset[level]) * 2 + feat;
  const bool single = (wk.end - wk.begin) <= ENC_SPAN_BWD;
  for (int i = threadIdx.x; i < (int)g.size; i += ENC_BLOCK) {
    const int vi = s_g[i];
    if (vi == 0) continue;
    const float v = (float)vi * S_inv;
    if (single) {
      gt[2 * (long long)i] += v;  // sole writer of this (texture, level, feature) plane
    } else {
      atomicAdd(&gt[2 * (long long)i], v);
    }
  }
}

}  // namespace

static int enc_grid_x(const vsa_nt_plan* p, int span) {
  int per_model = 0;
  for (int d = 0; d < VSA_NT_MAX_DEG; ++d) per_model += groups_of_degree(*p, d, span);
  return per_model * p->nr_shells * 2;
}

// Levels are launched in two classes so that the many small (dense) levels do
// not reserve the 128 KiB a full 2^15-entry level needs.
static int split_level(const vsa_nt_plan* p) {
  int l = 0;
  while (l < p->n_levels && p->level_size[l] <= SMALL_LEVEL_ENTRIES) ++l;
  return l;
}

static int max_level_size(const vsa_nt_plan* p, int l0, int l1) {
  int m = 0;
  for (int l = l0; l < l1; ++l) m = m > p->level_size[l] ? m : p->level_size[l];
  return m;
}

extern "C" int vsa_nt_encode_fwd(const vsa_nt_plan* plan, const void* tables_h,
                                 const float* slot_xy, const int32_t* seg_start, void* features,
                                 void* stream) {
  if (!plan || !tables_h || !slot_xy || !seg_start || !features) return VSA_ERR_ARG;
  if ((size_t)max_level_size(plan, 0, plan->n_levels) * 4 > 160 * 1024) return VSA_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(nt_encode_fwd_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
    attr_set = true;
  }
  const int ls = split_level(plan);
  const int gx = enc_grid_x(plan, ENC_SPAN_FWD);
  const int ranges[2][2] = {{0, ls}, {ls, plan->n_levels}};
  for (int r = 0; r < 2; ++r) {
    const int l0 = ranges[r][0], l1 = ranges[r][1];
    if (l1 <= l0) continue;
    const size_t lds = (size_t)max_level_size(plan, l0, l1) * 4;
    hipLaunchKernelGGL(nt_encode_fwd_kernel, dim3(gx, l1 - l0), dim3(ENC_BLOCK), lds,
                       (hipStream_t)stream, *plan, reinterpret_cast<const half2_t*>(tables_h),
                       reinterpret_cast<const float2*>(slot_xy), seg_start,
                       reinterpret_cast<half2_t*>(features), l0);
  }
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_nt_encode_bwd(const vsa_nt_plan* plan, const void* dfeatures, float grad_scale,
                                 const float* slot_xy, const int32_t* seg_start,
                                 float* grad_tables, void* stream) {
  if (!plan || !dfeatures || !slot_xy || !seg_start || !grad_tables) return VSA_ERR_ARG;
  if (!(grad_scale > 0.f)) return VSA_ERR_ARG;
  if ((size_t)max_level_size(plan, 0, plan->n_levels) * 4 > 160 * 1024) return VSA_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(nt_encode_bwd_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
    attr_set = true;
  }
  const int ls = split_level(plan);
  const int gx = enc_grid_x(plan, ENC_SPAN_BWD);
  const int ranges[2][2] = {{0, ls}, {ls, plan->n_levels}};
  for (int r = 0; r < 2; ++r) {
    const int l0 = ranges[r][0], l1 = ranges[r][1];
    if (l1 <= l0) continue;
    const size_t lds = (size_t)max_level_size(plan, l0, l1) * 4;
    hipLaunchKernelGGL(nt_encode_bwd_kernel, dim3(gx, 2 * (l1 - l0)), dim3(ENC_BLOCK), lds,
                       (hipStream_t)stream, *plan, reinterpret_cast<const half2_t*>(dfeatures),
                       1.0f / grad_scale, reinterpret_cast<const float2*>(slot_xy), seg_start,
                       grad_tables, l0);
  }
  VSA_RETURN_LAUNCH_STATUS();
}
