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

constexpr int ENC_BLOCK = 512;
constexpr int ENC_CHUNK = 32768;  // slots per workgroup
constexpr unsigned PRIME_Y = 2654435761u;

struct LevelGeom {
  float scale;
  int res, size;
  bool hashed;
};

__device__ __forceinline__ LevelGeom level_geom(const vsa_nt_plan& p, int l) {
  LevelGeom g;
  g.scale = p.level_scale[l];
  g.res = p.level_res[l];
  g.size = p.level_size[l];
  // tiny-cuda-nn grid_index: dense while the running stride fits the table
  g.hashed = !((long long)g.res <= g.size && (long long)g.res * g.res <= g.size);
  return g;
}

__device__ __forceinline__ unsigned level_index(const LevelGeom& g, unsigned cx, unsigned cy) {
  unsigned idx = g.hashed ? (cx ^ (cy * PRIME_Y)) : (cx + cy * (unsigned)g.res);
  return idx % (unsigned)g.size;
}

struct CellCorners {
  unsigned idx[4];
  float w[4];
};

// texel centre (extended-grid ix, iy of a texture of resolution R) -> the 4
// table entries and bilinear weights at level g
__device__ __forceinline__ CellCorners cell_corners(const LevelGeom& g, int ix, int iy, int R) {
  const float Rf = (float)R;
  const float x = ((float)(ix - 1) + 0.5f) / Rf;
  const float y = ((float)(iy - 1) + 0.5f) / Rf;
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

struct SegInfo {
  int begin, end;     // slot range of (shell, degree)
  long long dom_off;  // first domain texel
  int R, W;
};

__device__ __forceinline__ SegInfo seg_info(const vsa_nt_plan& p, const int* seg_start, int tex) {
  const int deg = tex % VSA_NT_MAX_DEG;
  const int shell = tex / (2 * VSA_NT_MAX_DEG);
  const int sd = shell * VSA_NT_MAX_DEG + deg;
  SegInfo s;
  s.begin = seg_start[sd];
  s.end = seg_start[sd + 1];
  s.dom_off = p.dom_off[sd];
  s.R = p.tex_res[deg];
  s.W = s.R + 2;
  return s;
}

__device__ __forceinline__ bool tex_active(const vsa_nt_plan& p, int tex) {
  const int deg = tex % VSA_NT_MAX_DEG;
  const int type = (tex / VSA_NT_MAX_DEG) & 1;
  const int shell = tex / (2 * VSA_NT_MAX_DEG);
  if (type == 0) return deg < p.rgb_degrees;
  if (p.inner_solid && shell == 0) return false;
  return deg < p.alpha_degrees;
}

__global__ __launch_bounds__(ENC_BLOCK) void nt_encode_fwd_kernel(
    vsa_nt_plan plan, const half2_t* __restrict__ tables, const int* __restrict__ texel_of_slot,
    const int* __restrict__ seg_start, half2_t* __restrict__ features) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  half2_t* s_tab = reinterpret_cast<half2_t*>(s_raw);
  const int tex = blockIdx.z, level = blockIdx.y;
  if (!tex_active(plan, tex)) return;
  const SegInfo seg = seg_info(plan, seg_start, tex);
  const int first = seg.begin + blockIdx.x * ENC_CHUNK;
  if (first >= seg.end) return;
  const int last = min(seg.end, first + ENC_CHUNK);
  const LevelGeom g = level_geom(plan, level);
  const long long n_entries = plan.level_offset[plan.n_levels];
  const half2_t* tab = tables + (long long)tex * n_entries + plan.level_offset[level];
  // stage the level (size is a multiple of 8 entries = 32 B)
  {
    const uint4* src = reinterpret_cast<const uint4*>(tab);
    uint4* dst = reinterpret_cast<uint4*>(s_tab);
    for (int i = threadIdx.x; i < g.size / 4; i += ENC_BLOCK) dst[i] = src[i];
  }
  __syncthreads();
  const int type = (tex / VSA_NT_MAX_DEG) & 1;
  half2_t* out = features + ((long long)type * plan.n_levels + level) * plan.slot_capacity;
  for (int slot = first + threadIdx.x; slot < last; slot += ENC_BLOCK) {
    const int local = (int)(texel_of_slot[slot] - seg.dom_off);
    const int iy = local / seg.W, ix = local - iy * seg.W;
    const CellCorners c = cell_corners(g, ix, iy, seg.R);
    float f0 = 0.f, f1 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const half2_t v = s_tab[c.idx[k]];
      f0 = f0 + c.w[k] * (float)v.x;
      f1 = f1 + c.w[k] * (float)v.y;
    }
    half2_t r;
    r.x = (_Float16)f0;
    r.y = (_Float16)f1;
    out[slot] = r;
  }
}

// Backward: grad_table[tex][level entries][feature] += w * dF[slot]; one
// workgroup per (texture, level, feature, chunk-group) with the feature's f32
// gradient plane (size entries) in LDS.
__global__ __launch_bounds__(ENC_BLOCK) void nt_encode_bwd_kernel(
    vsa_nt_plan plan, const half2_t* __restrict__ dfeatures, float dscale_inv,
    const int* __restrict__ texel_of_slot, const int* __restrict__ seg_start,
    float* __restrict__ grad_tables, int chunks_per_group) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  float* s_g = reinterpret_cast<float*>(s_raw);
  const int tex = blockIdx.z, level = blockIdx.y >> 1, feat = blockIdx.y & 1;
  if (!tex_active(plan, tex)) return;
  const SegInfo seg = seg_info(plan, seg_start, tex);
  const long long span = (long long)ENC_CHUNK * chunks_per_group;
  const long long first = seg.begin + blockIdx.x * span;
  if (first >= seg.end) return;
  const int last = (int)min((long long)seg.end, first + span);
  const LevelGeom g = level_geom(plan, level);
  for (int i = threadIdx.x; i < g.size; i += ENC_BLOCK) s_g[i] = 0.f;
  __syncthreads();
  const int type = (tex / VSA_NT_MAX_DEG) & 1;
  const half2_t* dF = dfeatures + ((long long)type * plan.n_levels + level) * plan.slot_capacity;
  for (int slot = (int)first + threadIdx.x; slot < last; slot += ENC_BLOCK) {
    const int local = (int)(texel_of_slot[slot] - seg.dom_off);
    const int iy = local / seg.W, ix = local - iy * seg.W;
    const CellCorners c = cell_corners(g, ix, iy, seg.R);
    const half2_t d = dF[slot];
    const float gv = (float)(feat ? d.y : d.x) * dscale_inv;
    if (gv != 0.f) {
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(&s_g[c.idx[k]], c.w[k] * gv);
    }
  }
  __syncthreads();
  const long long n_entries = plan.level_offset[plan.n_levels];
  float* gt = grad_tables + ((long long)tex * n_entries + plan.level_offset[level]) * 2 + feat;
  const bool single = gridDim.x == 1 || (seg.end - seg.begin) <= span;
  for (int i = threadIdx.x; i < g.size; i += ENC_BLOCK) {
    const float v = s_g[i];
    if (single) {
      gt[2 * (long long)i] += v;   // sole writer of this (texture, level, feature) plane
    } else if (v != 0.f) {
      atomicAdd(&gt[2 * (long long)i], v);
    }
  }
}

}  // namespace

static int enc_chunks(const vsa_nt_plan* p) {
  // worst case: the biggest (shell, degree) segment cannot exceed its domain nor the capacity
  long long worst = 0;
  for (int i = 0; i < p->nr_shells * VSA_NT_MAX_DEG; ++i)
    worst = worst > p->dom_off[i + 1] - p->dom_off[i] ? worst : p->dom_off[i + 1] - p->dom_off[i];
  if (worst > p->slot_capacity) worst = p->slot_capacity;
  int c = (int)((worst + ENC_CHUNK - 1) / ENC_CHUNK);
  return c < 1 ? 1 : c;
}

extern "C" int vsa_nt_encode_fwd(const vsa_nt_plan* plan, const void* tables_h,
                                 const int32_t* texel_of_slot, const int32_t* seg_start,
                                 void* features, void* stream) {
  if (!plan || !tables_h || !texel_of_slot || !seg_start || !features) return VSA_ERR_ARG;
  int max_size = 0;
  for (int l = 0; l < plan->n_levels; ++l) max_size = max_size > plan->level_size[l] ? max_size : plan->level_size[l];
  const size_t lds = (size_t)max_size * 4;
  if (lds > 160 * 1024) return VSA_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(nt_encode_fwd_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  dim3 grid(enc_chunks(plan), plan->n_levels, plan->nr_shells * 2 * VSA_NT_MAX_DEG);
  hipLaunchKernelGGL(nt_encode_fwd_kernel, grid, dim3(ENC_BLOCK), lds, (hipStream_t)stream, *plan,
                     reinterpret_cast<const half2_t*>(tables_h), texel_of_slot, seg_start,
                     reinterpret_cast<half2_t*>(features));
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_nt_encode_bwd(const vsa_nt_plan* plan, const void* dfeatures, float grad_scale,
                                 const int32_t* texel_of_slot, const int32_t* seg_start,
                                 float* grad_tables, void* stream) {
  if (!plan || !dfeatures || !texel_of_slot || !seg_start || !grad_tables) return VSA_ERR_ARG;
  if (!(grad_scale > 0.f)) return VSA_ERR_ARG;
  int max_size = 0;
  for (int l = 0; l < plan->n_levels; ++l) max_size = max_size > plan->level_size[l] ? max_size : plan->level_size[l];
  const size_t lds = (size_t)max_size * 4;
  if (lds > 160 * 1024) return VSA_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(nt_encode_bwd_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int chunks = enc_chunks(plan);
  const int chunks_per_group = 4;  // 131072 slots per workgroup before flushing
  dim3 grid(vsa_div_up(chunks, chunks_per_group), plan->n_levels * 2,
            plan->nr_shells * 2 * VSA_NT_MAX_DEG);
  hipLaunchKernelGGL(nt_encode_bwd_kernel, grid, dim3(ENC_BLOCK), lds, (hipStream_t)stream, *plan,
                     reinterpret_cast<const half2_t*>(dfeatures), 1.0f / grad_scale, texel_of_slot,
                     seg_start, grad_tables, chunks_per_group);
  VSA_RETURN_LAUNCH_STATUS();
}
