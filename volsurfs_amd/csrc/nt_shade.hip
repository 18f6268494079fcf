// Neural-texture step 5: per-hit shading from the quantised texel rows, forward
// and backward.
//
// Forward, per (ray, shell) hit — the reference's op sequence:
//   4 corner rows per degree -> fp16 range expansion (neural_texture.py:177-185,
//   a 256-entry LUT per degree) -> fp32 lerp (:188-191) -> fp16 SH coefficients
//   (sh_neural_textures.py:88) -> SH evaluation with the view direction
//   (encodings/sphericalharmonics.py:155-229) -> sigmoid (:92) -> alpha decay
//   (methods/volsurfs.py:583-594) -> scatter into surfs_rgb / surfs_alpha
//   (:550, :596).
// Backward: the same chain reversed; gradients w.r.t. the quantised texel value
// o = q/255 are accumulated per slot (f32 rows, per-degree width) with wave-cooperative
// atomics: lanes = the 64 (channel, SH coefficient) pairs of ONE hit, so each
// wave instruction adds contiguous row segments instead of 64 scattered dwords.
#include "nt_common.h"

// Launch geometry of the two shading kernels: a workgroup = (one shell, one tile of consecutive rays).
// NT_SHADE_MAP 2 (default): 1-D grid, XCD-aware — workgroups are dealt to the 8 XCDs round-robin by their linear
//   id, so (tile t, shell s) gets id = ((t / 8) * K + s) * 8 + t % 8: the K workgroups of one tile run on ONE XCD,
//   back to back, and their K partial writes into the tile's [N,K,3] / [N,K] lines (12 and 4 bytes at a stride of
//   12 K and 4 K) merge in that XCD's L2 before they leave it.
// 1: grid (shell, tile) — the K workgroups close in time but on K different XCDs (each L2 then writes its partial
//   line back by itself: at 1920x1080, K = 7 the launch wrote 0.98 GB for 0.30 GB of outputs);
// 0: rounds 1-4's grid (tile, shell): every line written K times a whole pass over the frame apart (1.22 GB).
#ifndef NT_SHADE_MAP
#define NT_SHADE_MAP 2
#endif
struct ShadeIdx {
  long long tile;
  int shell;
  bool valid;
};
__device__ __forceinline__ ShadeIdx shade_idx(int K, int tiles) {
#if NT_SHADE_MAP == 2
  const unsigned id = blockIdx.x, x = id & 7u, q = id >> 3;
  const unsigned s = q % (unsigned)K, tg = q / (unsigned)K;
  const long long t = (long long)tg * 8 + x;
  return {t, (int)s, t < tiles};
#elif NT_SHADE_MAP == 1
  return {(long long)blockIdx.y, (int)blockIdx.x, true};
#else
  return {(long long)blockIdx.x, (int)blockIdx.y, true};
#endif
}
static inline dim3 shade_grid(int tiles, int shells) {
#if NT_SHADE_MAP == 2
  return dim3((unsigned)(((tiles + 7) / 8) * 8 * shells));
#elif NT_SHADE_MAP == 1
  return dim3(shells, tiles);
#else
  return dim3(tiles, shells);
#endif
}

namespace {

constexpr int SH_BLOCK = 256;

constexpr float C0 = 0.28209479177387814f;
constexpr float C1 = 0.4886025119029199f;
constexpr float C2_0 = 1.0925484305920792f, C2_1 = -1.0925484305920792f,
                C2_2 = 0.31539156525252005f, C2_3 = -1.0925484305920792f,
                C2_4 = 0.5462742152960396f;
constexpr float C3_0 = -0.5900435899266435f, C3_1 = 2.890611442640554f,
                C3_2 = -0.4570457994644658f, C3_3 = 0.3731763325901154f,
                C3_4 = -0.4570457994644658f, C3_5 = 1.445305721320277f,
                C3_6 = -0.5900435899266435f;

// SH basis factors exactly as SHEncoder.eval forms them (left-to-right fp32
// products; index 0 is applied in fp16 and handled by the caller).
__device__ __forceinline__ void sh_basis(float x, float y, float z, float b[16]) {
  b[0] = C0;
  b[1] = -(C1 * y);
  b[2] = C1 * z;
  b[3] = -(C1 * x);
  const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  b[4] = C2_0 * xy;
  b[5] = C2_1 * yz;
  b[6] = C2_2 * ((2.0f * zz - xx) - yy);
  b[7] = C2_3 * xz;
  b[8] = C2_4 * (xx - yy);
  b[9] = (C3_0 * y) * (3.0f * xx - yy);
  b[10] = (C3_1 * xy) * z;
  b[11] = (C3_2 * y) * ((4.0f * zz - xx) - yy);
  b[12] = (C3_3 * z) * ((2.0f * zz - 3.0f * xx) - 3.0f * yy);
  b[13] = (C3_4 * x) * ((4.0f * zz - xx) - yy);
  b[14] = (C3_5 * z) * (xx - yy);
  b[15] = (C3_6 * x) * (xx - 3.0f * yy);
}

struct HitCtx {
  bool hit;
  float dir[3];
  float decay;        // alpha decay factor (1 when disabled)
  int row[VSA_NT_MAX_DEG][4];   // first quad of each corner's texel / gradient row
  float w[VSA_NT_MAX_DEG][4];
  float fx[VSA_NT_MAX_DEG], fy[VSA_NT_MAX_DEG];   // lerp fractions (w = products of these)
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// Evaluate raw SH sum with the reference's exact order.  y, z, x come in through
// the pre-multiplied basis b[] except for degree 1 where the reference
// multiplies (C1*y)*sh — b[1..3] already carry the sign, so subtraction of
// (C1*y)*sh1 is written as addition of b[1]*sh1 (bit-identical: a - p == a + (-p)).
__device__ __forceinline__ float sh_raw(const float b[16], const float* sh, int degrees) {
  float r = vsa_round_f16(C0 * sh[0]);
  if (degrees > 1) {
    r = r + b[1] * sh[1];
    r = r + b[2] * sh[2];
    r = r + b[3] * sh[3];
    if (degrees > 2) {
#pragma unroll
      for (int m = 4; m < 9; ++m) r = r + b[m] * sh[m];
      if (degrees > 3) {
#pragma unroll
        for (int m = 9; m < 16; ++m) r = r + b[m] * sh[m];
      }
    }
  }
  return r;
}

__device__ __forceinline__ void build_lut(const vsa_nt_plan& plan, float* s_lut) {
  // expand_lut (oracle) : fp16(lo + fp16(span * fp16(q/255)))
  for (int i = threadIdx.x; i < VSA_NT_MAX_DEG * 256; i += blockDim.x) {
    const int d = i >> 8, q = i & 255;
    const float o = vsa_round_f16((float)q / 255.0f);
    const float t = vsa_round_f16(plan.sh_span[d] * o);
    s_lut[i] = vsa_round_f16(plan.sh_lo[d] + t);
  }
}

// FULL4: both texture types carry all four SH bands (the shipped configuration, base_5.cfg:12-20):
// the band counts become compile-time constants, so the per-band `if (d >= D) continue` branches go
// and the compiler issues the slot-id loads of all four bands together — with the run-time counts
// every band was its own load -> wait round (four dependent memory latencies in a kernel that PMC
// shows waiting 74 % of its wave cycles).
template <bool FULL4>
__device__ __forceinline__ int rgb_degs(const vsa_nt_plan& p) { return FULL4 ? VSA_NT_MAX_DEG : p.rgb_degrees; }
template <bool FULL4>
__device__ __forceinline__ int alpha_degs(const vsa_nt_plan& p) { return FULL4 ? VSA_NT_MAX_DEG : p.alpha_degrees; }

template <bool FULL4 = false>
__device__ __forceinline__ bool load_ctx(const vsa_nt_plan& plan, int s, long long n, int N,
                                         const int* hit_slot, const float* tex_uv,
                                         const float* rays_d, const float4* tris,
                                         const int* slot_of, const int* seg_start, HitCtx& c,
                                         float nrm[3]) {
  const long long o = (long long)s * N + n;
  const int tslot = hit_slot[o];
  c.hit = tslot >= 0;
  nrm[0] = nrm[1] = nrm[2] = 0.f;
  if (!c.hit) return false;
  c.dir[0] = rays_d[3 * n];
  c.dir[1] = rays_d[3 * n + 1];
  c.dir[2] = rays_d[3 * n + 2];
  const float4 e1 = tris[3 * (long long)tslot + 1], e2 = tris[3 * (long long)tslot + 2];
  const float cx = e1.y * e2.z - e1.z * e2.y, cy = e1.z * e2.x - e1.x * e2.z,
              cz = e1.x * e2.y - e1.y * e2.x;
  const float len = sqrtf((cx * cx + cy * cy) + cz * cz);
  const float inv = len > 0.f ? 1.0f / len : 0.f;
  nrm[0] = cx * inv;
  nrm[1] = cy * inv;
  nrm[2] = cz * inv;
  c.decay = 1.0f;
  if (plan.with_alpha_decay) {
    float dot = ((-c.dir[0]) * nrm[0] + (-c.dir[1]) * nrm[1]) + (-c.dir[2]) * nrm[2];
    dot = fminf(fmaxf(dot, 0.0f), 1.0f);
    c.decay = sigmoidf_(10.0f * dot) * 2.0f - 1.0f;
  }
  const float u = tex_uv[2 * o], v = tex_uv[2 * o + 1];
  const int D = max(rgb_degs<FULL4>(plan), alpha_degs<FULL4>(plan));
#pragma unroll
  for (int d = 0; d < VSA_NT_MAX_DEG; ++d) {   // static indices: the arrays stay in registers
    if (d >= D) {
#pragma unroll
      for (int k = 0; k < 4; ++k) c.row[d][k] = 0, c.w[d][k] = 0.f;
      c.fx[d] = c.fy[d] = 0.f;
      continue;
    }
    const int R = plan.tex_res[d], W = R + 2;
    const bool anchor = plan.anchor != 0;
    const NtFootprint f = nt_footprint(u, v, R, anchor);
    const long long base = plan.dom_off[s * VSA_NT_MAX_DEG + d] + (long long)(f.j0 + 1) * W + (f.i0 + 1);
    const int sd = s * VSA_NT_MAX_DEG + d;
    const int rb = (int)plan.row_base[sd] - seg_start[sd] * nt_row_quads(d);
    c.row[d][0] = rb + slot_of[base] * nt_row_quads(d);
    // anchor: only that texel is marked (slot_of is valid for marked texels only): the three other
    // "corners" are the same row with weight 0
    c.row[d][1] = anchor ? c.row[d][0] : rb + slot_of[base + 1] * nt_row_quads(d);
    c.row[d][2] = anchor ? c.row[d][0] : rb + slot_of[base + W] * nt_row_quads(d);
    c.row[d][3] = anchor ? c.row[d][0] : rb + slot_of[base + W + 1] * nt_row_quads(d);
    c.fx[d] = f.fx;
    c.fy[d] = f.fy;
    c.w[d][0] = (1.0f - f.fx) * (1.0f - f.fy);
    c.w[d][1] = f.fx * (1.0f - f.fy);
    c.w[d][2] = (1.0f - f.fx) * f.fy;
    c.w[d][3] = f.fx * f.fy;
  }
  return true;
}

// One corner row of band d added into the band's 28 lerp sums (3 n colour + n alpha at 21).  F16ROWS = false: the
// 8-bit quantised row through the expand LUT; true (row_format 1): f16 sigmoid values, expanded as the reference
// does in half arithmetic — fl16(lo + fl16(span * o)) (neural_texture.py:183-187) — a quad being 4 halves (8 bytes).
template <bool F16ROWS>
__device__ __forceinline__ void add_corner_row(const vsa_nt_plan& plan, int d, const unsigned* __restrict__ texels,
                                               int row, float wk, const float* s_lut, float acc[28]) {
  const int n = 2 * d + 1;
  if constexpr (!F16ROWS) {
    unsigned wds[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned* rp = texels + row;
    if (d == 0) {
      const uint2 v = *reinterpret_cast<const uint2*>(rp);
      wds[0] = v.x, wds[1] = v.y;
    } else {
      const uint4 lo = *reinterpret_cast<const uint4*>(rp);
      wds[0] = lo.x, wds[1] = lo.y, wds[2] = lo.z, wds[3] = lo.w;
      if (d >= 2) {
        const uint4 hi = *reinterpret_cast<const uint4*>(rp + 4);
        wds[4] = hi.x, wds[5] = hi.y, wds[6] = hi.z, wds[7] = hi.w;
      }
    }
#pragma unroll
    for (int i = 0; i < 3 * n; ++i) {
      const unsigned q = (wds[i >> 2] >> (8 * (i & 3))) & 255u;
      acc[i] = acc[i] + s_lut[d * 256 + q] * wk;
    }
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const unsigned q = (wds[nt_alpha_quad(d) + (i >> 2)] >> (8 * (i & 3))) & 255u;
      acc[21 + i] = acc[21 + i] + s_lut[d * 256 + q] * wk;
    }
  } else {
    unsigned wds[16];
    const uint4* rp = reinterpret_cast<const uint4*>(texels + 2 * (long long)row);     // quad = 2 dwords
    const int nv = d == 0 ? 1 : d == 1 ? 2 : 4;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const uint4 x = v < nv ? rp[v] : make_uint4(0, 0, 0, 0);
      wds[4 * v] = x.x, wds[4 * v + 1] = x.y, wds[4 * v + 2] = x.z, wds[4 * v + 3] = x.w;
    }
    const float lo = plan.sh_lo[d], span = plan.sh_span[d];
    const bool raw = plan.row_format == 2;     // using_sh_squeezing = 0: the row is the network output, no expansion (:181-187)
    auto expand = [&](int elem) {
      const unsigned short hb = (unsigned short)(wds[elem >> 1] >> (16 * (elem & 1)));
      const float o = (float)__builtin_bit_cast(_Float16, hb);
      return raw ? o : vsa_round_f16(lo + vsa_round_f16(vsa_pin_f32(span * o)));
    };
#pragma unroll
    for (int i = 0; i < 3 * n; ++i) acc[i] = acc[i] + expand(i) * wk;
#pragma unroll
    for (int i = 0; i < n; ++i) acc[21 + i] = acc[21 + i] + expand(4 * nt_alpha_quad(d) + i) * wk;
  }
}

// SH coefficients (fp16-rounded, as floats): sh_rgb[ch][16], sh_a[16]
template <bool F16ROWS = false>
__device__ __forceinline__ void gather_coeffs(const vsa_nt_plan& plan, const HitCtx& c,
                                              const unsigned* __restrict__ texels,
                                              const float* s_lut, bool has_alpha,
                                              float sh_rgb[3][16], float sh_a[16]) {
#pragma unroll
  for (int d = 0; d < VSA_NT_MAX_DEG; ++d) {
    const int n = 2 * d + 1, m0 = d * d;
    const bool do_rgb = d < plan.rgb_degrees, do_a = has_alpha && d < plan.alpha_degrees;
    if (!do_rgb && !do_a) continue;
    float acc[28];
#pragma unroll
    for (int i = 0; i < 28; ++i) acc[i] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) add_corner_row<F16ROWS>(plan, d, texels, c.row[d][k], c.w[d][k], s_lut, acc);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
      for (int i = 0; i < n; ++i) sh_rgb[ch][m0 + i] = do_rgb ? vsa_round_f16(acc[ch * n + i]) : 0.f;
#pragma unroll
    for (int i = 0; i < n; ++i) sh_a[m0 + i] = do_a ? vsa_round_f16(acc[21 + i]) : 0.f;
  }
}

// Forward of one hit, one SH band at a time: the band's 4 corner rows are expanded, lerped and
// rounded to fp16 (gather_coeffs's arithmetic), then added to the raw SH sums in coefficient
// order — the same additions sh_raw performs, so the result is bit-identical, but only one
// band's coefficients are live (the all-bands-first form needed 197 VGPRs = two waves per
// SIMD on a kernel that PMC shows waiting on gathers: 16 % VALU-busy).
template <bool FULL4 = false, bool F16ROWS = false>
__device__ __forceinline__ void shade_hit(const vsa_nt_plan& plan, const HitCtx& c,
                                          const unsigned* __restrict__ texels, const float* s_lut,
                                          bool has_alpha, const float b[16], float raw[4],
                                          float* __restrict__ coeffs_out) {
  raw[0] = raw[1] = raw[2] = raw[3] = 0.f;
#pragma unroll
  for (int d = 0; d < VSA_NT_MAX_DEG; ++d) {
    const int n = 2 * d + 1, m0 = d * d;
    const bool do_rgb = d < rgb_degs<FULL4>(plan), do_a = has_alpha && d < alpha_degs<FULL4>(plan);
    if (!do_rgb && !do_a) {
      if (coeffs_out) {
#pragma unroll
        for (int i = 0; i < n; ++i)
          coeffs_out[m0 + i] = coeffs_out[16 + m0 + i] = coeffs_out[32 + m0 + i] = coeffs_out[48 + m0 + i] = 0.f;
      }
      continue;
    }
    float acc[28];
#pragma unroll
    for (int i = 0; i < 28; ++i) acc[i] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) add_corner_row<F16ROWS>(plan, d, texels, c.row[d][k], c.w[d][k], s_lut, acc);
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      const bool on = ch < 3 ? do_rgb : do_a;
#pragma unroll
      for (int i = 0; i < n; ++i) {
        const int m = m0 + i;
        const float shv = on ? vsa_round_f16(ch < 3 ? acc[ch * n + i] : acc[21 + i]) : 0.f;
        if (coeffs_out) coeffs_out[ch * 16 + m] = shv;
        if (on) raw[ch] = m == 0 ? vsa_round_f16(C0 * shv) : raw[ch] + b[m] * shv;
      }
    }
  }
}

#ifndef NT_SHADE_FWD_OCC
#define NT_SHADE_FWD_OCC 3     /* round 3: 4 waves per SIMD (128 VGPRs) gave 0.163 -> 0.152 ms with run-time band counts; with FULL4 the hoisted slot-id loads spill there (0.19) and 3 waves run 0.139 */
#endif
template <bool FULL4, bool F16ROWS = false>
__global__ __launch_bounds__(SH_BLOCK, NT_SHADE_FWD_OCC) void nt_shade_fwd_kernel(
    vsa_nt_plan plan, const int* __restrict__ hit_slot, const float* __restrict__ tex_uv,
    const float* __restrict__ rays_d, const float4* __restrict__ tris,
    const int* __restrict__ slot_of, const int* __restrict__ seg_start,
    const unsigned* __restrict__ texels, int N, float* __restrict__ surfs_rgb, float* __restrict__ surfs_alpha,
    float* __restrict__ surfs_normals, float* __restrict__ coeffs_out,
    float4* __restrict__ act_out) {
  __shared__ float s_lut[VSA_NT_MAX_DEG * 256];
  build_lut(plan, s_lut);
  __syncthreads();
  const int K = plan.nr_shells;
  const ShadeIdx wi = shade_idx(K, (N + SH_BLOCK - 1) / SH_BLOCK);      // (shell, ray tile), XCD-aware: see NT_SHADE_MAP
  const long long n = wi.tile * SH_BLOCK + threadIdx.x;
  const int s = wi.shell;
  if (!wi.valid) return;
  if (n >= N) return;
  HitCtx c;
  float nrm[3];
  float rgb[3] = {0.f, 0.f, 0.f}, alpha = 0.f;
  float4 act = make_float4(0.f, 0.f, 0.f, 0.f);   // the four sigmoids, for the backward pass
  if (load_ctx<FULL4>(plan, s, n, N, hit_slot, tex_uv, rays_d, tris, slot_of, seg_start, c, nrm)) {
    const bool has_alpha = nt_shell_has_alpha(plan, s);
    float b[16], raw[4];
    sh_basis(c.dir[0], c.dir[1], c.dir[2], b);
    shade_hit<FULL4, F16ROWS>(plan, c, texels, s_lut, has_alpha, b, raw,
              coeffs_out ? coeffs_out + ((long long)s * N + n) * 64 : nullptr);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float sg = sigmoidf_(raw[ch]);
      (ch == 0 ? act.x : ch == 1 ? act.y : act.z) = sg;
      rgb[ch] = rgb_degs<FULL4>(plan) > 1 ? sg : vsa_round_f16(sg);
    }
    if (has_alpha) {
      const float sg = sigmoidf_(raw[3]);
      act.w = sg;
      const float a = alpha_degs<FULL4>(plan) > 1 ? sg : vsa_round_f16(sg);
      alpha = a * c.decay;
    } else {
      alpha = 1.0f;
    }
    // (hits only: the backward reads act_in for hits only, and at 71 % misses the zeros were 165 MB of the
    //  464 MB this launch wrote at 1920x1080, K = 7)
    if (act_out) act_out[(long long)s * N + n] = act;
  } else if (coeffs_out) {
    float* co = coeffs_out + ((long long)s * N + n) * 64;
    for (int i = 0; i < 64; ++i) co[i] = 0.f;
  }
  const long long o = n * K + s;
  surfs_rgb[3 * o] = rgb[0];
  surfs_rgb[3 * o + 1] = rgb[1];
  surfs_rgb[3 * o + 2] = rgb[2];
  surfs_alpha[o] = alpha;
  if (surfs_normals) {
    surfs_normals[3 * o] = nrm[0];
    surfs_normals[3 * o + 1] = nrm[1];
    surfs_normals[3 * o + 2] = nrm[2];
  }
}

// ---------------------------------------------------------------- backward
// Per hit: g_raw[4] (3 rgb + alpha, through the output sigmoid and the decay) and
// the 16 SH basis values; then lanes = 64 (channel, coefficient) pairs of one
// hit at a time add  span_d * w_corner * g_raw[ch] * basis[m]  to the slot rows.
// RECOMPUTE = false: the caller hands back the forward pass's sigmoids (act_in) and phase 1
// is a handful of loads; RECOMPUTE = true re-gathers the texel rows.  The per-hit record
// that phase 2 reads is 39 dwords (4 g_raw, 16 basis, 2 row ids + 2 lerp fractions per band)
// so that a 128-thread workgroup needs 20 KiB of LDS and 16 waves fit a CU: halving the
// occupancy of this atomic-bound kernel cost 27 % (measured), i.e. it is latency-sensitive.
// Diagnostic build only (-DSHADE_SPAN; tools/shade_span.py): begin / end of every workgroup of the
// shading backward, to see whether the launch has idle stretches or a tail.
#ifdef SHADE_SPAN
static __device__ unsigned long long g_sspan[1 << 16][2];
__device__ __forceinline__ unsigned long long shade_now() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#endif
constexpr int SHB_BLOCK = 128;

// no-return packed f16 add at the memory side (global_atomic_pk_add_f16: one dword per lane)
// (uniform base + 32-bit byte offset: the buffer is far below 4 GiB)
__device__ __forceinline__ void atomic_pk_add_f16(const _Float16* base, unsigned byte_off, float a, float b) {
  const half2_t v = {(_Float16)a, (_Float16)b};
  asm volatile("global_atomic_pk_add_f16 %0, %1, %2" ::"v"(byte_off), "v"(__builtin_bit_cast(unsigned, v)), "s"(base) : "memory");
}

#ifndef NT_SHB_PREFETCH
#define NT_SHB_PREFETCH 1
#endif
template <bool RECOMPUTE, bool FULL4, bool F16ROWS = false>
#ifndef NT_SHADE_BWD_OCC
#define NT_SHADE_BWD_OCC 4
#endif
__global__ __launch_bounds__(SHB_BLOCK, RECOMPUTE ? 2 : NT_SHADE_BWD_OCC) void nt_shade_bwd_kernel(
    vsa_nt_plan plan, const int* __restrict__ hit_slot, const float* __restrict__ tex_uv,
    const float* __restrict__ rays_d, const float4* __restrict__ tris,
    const int* __restrict__ slot_of, const int* __restrict__ seg_start,
    const unsigned* __restrict__ texels, int N, const float* __restrict__ g_surfs_rgb,
    const float* __restrict__ g_surfs_alpha, float grad_scale, _Float16* __restrict__ grad_rows,
    const float4* __restrict__ act_in) {
  __shared__ float s_lut[RECOMPUTE ? VSA_NT_MAX_DEG * 256 : 1];
  __shared__ float s_graw[SHB_BLOCK][4];
  __shared__ float s_basis[SHB_BLOCK][17];
  __shared__ int s_row[SHB_BLOCK][9];    // corners (x0,y0) and (x0,y1) per band; x1 = the next slot's row
  __shared__ float s_f[SHB_BLOCK][9];
#ifdef SHADE_SPAN
  const unsigned long long span_t0 = shade_now();
#endif
  if (RECOMPUTE) {
    build_lut(plan, s_lut);
    __syncthreads();
  }
  const int K = plan.nr_shells;
  const ShadeIdx wi = shade_idx(K, (N + SHB_BLOCK - 1) / SHB_BLOCK);    // as in the forward: the reads of g_surfs_* share lines
  const long long n = wi.valid ? wi.tile * SHB_BLOCK + threadIdx.x : (long long)N;
  const int s = wi.shell;
  const int t = threadIdx.x;
  HitCtx c;
  c.hit = false;
  float nrm[3];
  const bool has_alpha = nt_shell_has_alpha(plan, s);
  if (n < N && load_ctx<FULL4>(plan, s, n, N, hit_slot, tex_uv, rays_d, tris, slot_of, seg_start, c, nrm)) {
    float b[16];
    sh_basis(c.dir[0], c.dir[1], c.dir[2], b);
    const long long o = n * K + s;
    float sg4[4];
    if (!RECOMPUTE) {   // the forward pass kept its four sigmoids: no texel gather, no SH sums
      const float4 a = act_in[(long long)s * N + n];
      sg4[0] = a.x, sg4[1] = a.y, sg4[2] = a.z, sg4[3] = a.w;
    } else {
      float sh_rgb[3][16], sh_a[16];
      gather_coeffs<F16ROWS>(plan, c, texels, s_lut, has_alpha, sh_rgb, sh_a);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) sg4[ch] = sigmoidf_(sh_raw(b, sh_rgb[ch], plan.rgb_degrees));
      sg4[3] = has_alpha ? sigmoidf_(sh_raw(b, sh_a, plan.alpha_degrees)) : 0.f;
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float sg = sg4[ch];
      s_graw[t][ch] = g_surfs_rgb[3 * o + ch] * sg * (1.0f - sg) * grad_scale;
    }
    float ga = 0.f;
    if (has_alpha) {
      const float sg = sg4[3];
      ga = g_surfs_alpha[o] * c.decay * sg * (1.0f - sg) * grad_scale;
    }
    s_graw[t][3] = ga;
#pragma unroll
    for (int m = 0; m < 16; ++m) s_basis[t][m] = b[m];
#pragma unroll
    for (int d = 0; d < VSA_NT_MAX_DEG; ++d) {
      // slots are numbered in domain order and all four corners are marked, so the x1
      // corner is always the slot right after the x0 corner
      s_row[t][2 * d] = c.row[d][0];
      s_row[t][2 * d + 1] = c.row[d][2];
      s_f[t][2 * d] = c.fx[d];
      s_f[t][2 * d + 1] = c.fy[d];
    }
  }
  // per-wave cooperative scatter (no workgroup barrier needed: each wave reads
  // only what its own lanes wrote).  The memory-side atomic path is bound by its 64-byte
  // REQUESTS (PMC: 18 G requests/s whatever their fill), so the unit of accumulation is the
  // 64-byte LINE of gradient rows, not the row: a lane owns one PAIR of adjacent f16 elements
  // at a fixed position of a line (one global_atomic_pk_add_f16 per flush) —
  //   degree 0: 16-B rows, 4 slots per line x 3 pairs   (lanes  0..11)
  //   degree 1: 32-B rows, 2 slots per line x 7 pairs   (lanes 12..25)
  //   degree 2: 64-B rows, 11 pairs                      (lanes 26..36)
  //   degree 3: 64-B rows, 15 pairs                      (lanes 37..51)
  // (pairs of a row: ceil(3 nn / 2) rgb + ceil(nn / 2) alpha, nn = 2d + 1; an odd tail pairs
  // with a padding element and adds zero to it.)
  const int lane = t & 63, wbase = t & ~63;
  const unsigned long long hits = __ballot(c.hit);
  const int d = lane < 12 ? 0 : (lane < 26 ? 1 : (lane < 37 ? 2 : 3));
  const int lb = lane - (d == 0 ? 0 : (d == 1 ? 12 : (d == 2 ? 26 : 37)));
  const int nn = 2 * d + 1, n_rgb_pairs = (3 * nn + 1) >> 1, row_pairs = n_rgb_pairs + ((nn + 1) >> 1);
  // (small unsigned divisions by multiply-high: the generic sequence is ~25 instructions each)
  const unsigned magic3 = 0x55555556u, magic5 = 0x33333334u, magic7 = 0x24924925u;   // 2^32 / n + 1
  const unsigned magic_nn = d == 1 ? magic3 : (d == 2 ? magic5 : magic7);
  auto div_nn = [&](int x) { return d == 0 ? x : (int)__umulhi((unsigned)x, magic_nn); };
  const int q = d == 0 ? (int)__umulhi((unsigned)lb, magic3) : (d == 1 ? (int)__umulhi((unsigned)lb, magic7) : 0);
  const int jb = lb - q * row_pairs;                                  // q: slot within the line, jb: pair within the row
  const bool is_alpha = jb >= n_rgb_pairs;
  const int e0 = is_alpha ? 2 * (jb - n_rgb_pairs) : 2 * jb;          // element index within the part
  const int part = is_alpha ? nn : 3 * nn;
  const int qd = nt_row_quads(d), qsh = d == 0 ? 1 : (d == 1 ? 2 : 3);
  // position of the pair in its line, in halfs (even: a 4-byte aligned f16 pair)
  const int lofs = 4 * (q * qd + (is_alpha ? nt_alpha_quad(d) : 0)) + e0;
  const bool band_on = lane < 52 && (is_alpha ? (has_alpha && d < alpha_degs<FULL4>(plan)) : d < rgb_degs<FULL4>(plan));
  const bool on1 = band_on && e0 + 1 < part;
  const int c0 = div_nn(e0), c1 = div_nn(e0 + 1);
  const int ch0 = is_alpha ? 3 : c0, ch1 = is_alpha ? 3 : c1;
  const int m0 = d * d + (is_alpha ? e0 : e0 - c0 * nn), m1 = on1 ? d * d + (is_alpha ? e0 + 1 : e0 + 1 - c1 * nn) : 0;
  float span = plan.row_format == 2 ? 1.0f : plan.sh_span[d];     // (raw rows: d row / d output = 1)
  // `d` differs per lane, so this is a VECTOR load from the kernel-argument segment; its first use
  // is inside the hit loop below, and the compiler put the `s_waitcnt vmcnt(0)` for it THERE — where,
  // on every trip, it also waited for every gradient atomic the wave had in flight (no-return
  // atomics stay counted in vmcnt until the memory side acknowledges them: ~1-3 k cycles).  Using
  // the value once here retires the load in front of the loop.
  asm volatile("" : "+v"(span));
  // Consecutive hits of a wave are neighbouring pixels: their 2x2 footprints fall on the same
  // or on neighbouring lines.  Each lane keeps up to four lines open with a running sum (the
  // lines the previous footprint touched: {x0, x1} x {y0, y1}, fewer when corners share a
  // line); a line of the new footprint that matches ANY open line inherits its sum, and only
  // lines the new footprint no longer touches are flushed — all lanes of a degree take the
  // same decisions, so a flush is ONE request for the whole line.
  // (the pair's two running sums travel as one float2: v_pk_mul / v_pk_fma_f32, and the
  // "inherit if same line" select is an fma with a 1.0 / 0.0 factor — the loop is VALU-bound)
  typedef float float2_t __attribute__((ext_vector_type(2)));
  int cur[4] = {-1, -1, -1, -1};
  float2_t acc[4] = {float2_t(0.f), float2_t(0.f), float2_t(0.f), float2_t(0.f)};
  const unsigned lofs_b = 2u * (unsigned)lofs;
  unsigned long long rem = hits;
#if NT_SHB_PREFETCH
  // the eight LDS values a lane needs of a hit are read one hit ahead of their use
  struct HitIn { float g0, g1, b0, b1, fx, fy; int r0, r1; };
  auto fetch = [&](int ht) {
    HitIn r;
    r.g0 = s_graw[ht][ch0], r.g1 = s_graw[ht][ch1], r.b0 = s_basis[ht][m0], r.b1 = s_basis[ht][m1];
    r.fx = s_f[ht][2 * d], r.fy = s_f[ht][2 * d + 1];
    r.r0 = s_row[ht][2 * d], r.r1 = s_row[ht][2 * d + 1];
    return r;
  };
  HitIn nx = fetch(wbase + (rem ? __ffsll((long long)rem) - 1 : 0));
#endif
  while (rem) {
    const int hl = __ffsll((long long)rem) - 1;
    rem &= rem - 1;
    const int ht = wbase + hl;
#if NT_SHB_PREFETCH
    const HitIn cu = nx;
    nx = fetch(wbase + (rem ? __ffsll((long long)rem) - 1 : hl));
    (void)ht;
#endif
    if (band_on) {
      float2_t gg;
#if NT_SHB_PREFETCH
      gg.x = cu.g0 * cu.b0;
      gg.y = on1 ? cu.g1 * cu.b1 : 0.f;
      const float fx = cu.fx, fy = cu.fy;
#else
      gg.x = s_graw[ht][ch0] * s_basis[ht][m0];
      gg.y = on1 ? s_graw[ht][ch1] * s_basis[ht][m1] : 0.f;
      const float fx = s_f[ht][2 * d], fy = s_f[ht][2 * d + 1];
#endif
      // the lerp weights exactly as load_ctx forms them, then x span, then x g
      const float w[4] = {(1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy};
      int sl[4];
      float wl[4];
#pragma unroll
      for (int y = 0; y < 2; ++y) {
#if NT_SHB_PREFETCH
        const int ra = y ? cu.r1 : cu.r0, rb = ra + qd;              // the x0 and x1 corner rows (quads)
#else
        const int ra = s_row[ht][2 * d + y], rb = ra + qd;           // the x0 and x1 corner rows (quads)
#endif
        const int la = ra >> 3, lbn = rb >> 3;                        // their lines (8 quads of 8 B)
        const bool ma = q == ((ra & 7) >> qsh), mb = q == ((rb & 7) >> qsh), one = la == lbn;
        sl[2 * y] = la;
        wl[2 * y] = ma ? w[2 * y] : ((one && mb) ? w[2 * y + 1] : 0.f);
        sl[2 * y + 1] = one ? -1 : lbn;
        wl[2 * y + 1] = (!one && mb) ? w[2 * y + 1] : 0.f;
      }
      // the y1 corners may sit on a line of the y0 corners (sparsely marked regions): rare,
      // so the fold is behind a wave-uniform test
      const bool dup = sl[2] == sl[0] || sl[2] == sl[1] || ((sl[3] == sl[0] || sl[3] == sl[1]) && sl[3] >= 0);
      if (__builtin_expect(__ballot(dup) != 0ull, 0)) {
#pragma unroll
        for (int k = 2; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            if (sl[k] == sl[i] && sl[k] >= 0) {
              wl[i] += wl[k];
              wl[k] = 0.f;
              sl[k] = -1;
            }
      }
      float2_t v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = float2_t(wl[k] * span) * gg;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bool kept = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool same = cur[j] == sl[k];
          v[k] = __builtin_elementwise_fma(acc[j], float2_t(same ? 1.0f : 0.0f), v[k]);
          kept |= same;
        }
        if (!kept && cur[j] >= 0) atomic_pk_add_f16(grad_rows, (unsigned)cur[j] * 64u + lofs_b, acc[j].x, acc[j].y);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        cur[k] = sl[k];
        acc[k] = v[k];
      }
    }
  }
  if (band_on) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (cur[k] >= 0) atomic_pk_add_f16(grad_rows, (unsigned)cur[k] * 64u + lofs_b, acc[k].x, acc[k].y);
  }
#ifdef SHADE_SPAN
  {
    const int w = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0 && w < (1 << 16)) g_sspan[w][0] = span_t0, g_sspan[w][1] = shade_now();
  }
#endif
}

}  // namespace

extern "C" int vsa_nt_shade_fwd(const vsa_nt_plan* plan, const int32_t* hit_slot,
                                const float* tex_uv, const float* rays_d, const float* tris,
                                const int32_t* slot_of, const int32_t* seg_start,
                                const uint8_t* texels, int nr_rays, float* surfs_rgb,
                                float* surfs_alpha, float* surfs_normals, float* coeffs_out,
                                float* act_out, void* stream) {
  if (!plan || nr_rays < 0) return VSA_ERR_ARG;
  if (nr_rays == 0) return VSA_OK;
  if (!hit_slot || !tex_uv || !rays_d || !tris || !slot_of || !seg_start || !texels || !surfs_rgb ||
      !surfs_alpha)
    return VSA_ERR_ARG;
  if (plan->row_format < 0 || plan->row_format > 2) return VSA_ERR_UNSUPPORTED;
  dim3 grid = shade_grid(vsa_div_up(nr_rays, SH_BLOCK), plan->nr_shells);
  if (plan->row_format != 0)          // f16 rows (using_sh_quantization = 0): the generic-band-count kernel
    hipLaunchKernelGGL((nt_shade_fwd_kernel<false, true>), grid, dim3(SH_BLOCK), 0, (hipStream_t)stream, *plan,
                       hit_slot, tex_uv, rays_d, reinterpret_cast<const float4*>(tris), slot_of,
                       seg_start, reinterpret_cast<const unsigned*>(texels), nr_rays, surfs_rgb,
                       surfs_alpha, surfs_normals, coeffs_out, reinterpret_cast<float4*>(act_out));
  else if (plan->rgb_degrees == VSA_NT_MAX_DEG && plan->alpha_degrees == VSA_NT_MAX_DEG)
    hipLaunchKernelGGL(nt_shade_fwd_kernel<true>, grid, dim3(SH_BLOCK), 0, (hipStream_t)stream, *plan,
                       hit_slot, tex_uv, rays_d, reinterpret_cast<const float4*>(tris), slot_of,
                       seg_start, reinterpret_cast<const unsigned*>(texels), nr_rays, surfs_rgb,
                       surfs_alpha, surfs_normals, coeffs_out, reinterpret_cast<float4*>(act_out));
  else
    hipLaunchKernelGGL(nt_shade_fwd_kernel<false>, grid, dim3(SH_BLOCK), 0, (hipStream_t)stream, *plan,
                       hit_slot, tex_uv, rays_d, reinterpret_cast<const float4*>(tris), slot_of,
                       seg_start, reinterpret_cast<const unsigned*>(texels), nr_rays, surfs_rgb,
                       surfs_alpha, surfs_normals, coeffs_out, reinterpret_cast<float4*>(act_out));
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_nt_shade_bwd(const vsa_nt_plan* plan, const int32_t* hit_slot,
                                const float* tex_uv, const float* rays_d, const float* tris,
                                const int32_t* slot_of, const int32_t* seg_start,
                                const uint8_t* texels, int nr_rays, const float* g_surfs_rgb,
                                const float* g_surfs_alpha, float grad_scale, uint16_t* grad_rows,
                                const float* act_in, void* stream) {
  if (!plan || nr_rays < 0) return VSA_ERR_ARG;
  if (nr_rays == 0) return VSA_OK;
  if (!hit_slot || !tex_uv || !rays_d || !tris || !slot_of || !seg_start || !texels || !g_surfs_rgb ||
      !g_surfs_alpha || !grad_rows)
    return VSA_ERR_ARG;
  if (plan->row_base[VSA_MAX_SHELLS * VSA_NT_MAX_DEG] * 8 >= (1ll << 32)) return VSA_ERR_UNSUPPORTED;   // 32-bit atomic offsets
  if (plan->row_format < 0 || plan->row_format > 2) return VSA_ERR_UNSUPPORTED;
  dim3 grid = shade_grid(vsa_div_up(nr_rays, SHB_BLOCK), plan->nr_shells);
  const bool full4 = plan->rgb_degrees == VSA_NT_MAX_DEG && plan->alpha_degrees == VSA_NT_MAX_DEG;
  if (act_in && full4)
    hipLaunchKernelGGL((nt_shade_bwd_kernel<false, true>), grid, dim3(SHB_BLOCK), 0, (hipStream_t)stream, *plan,
                       hit_slot, tex_uv, rays_d, reinterpret_cast<const float4*>(tris), slot_of,
                       seg_start, reinterpret_cast<const unsigned*>(texels), nr_rays, g_surfs_rgb,
                       g_surfs_alpha, grad_scale, reinterpret_cast<_Float16*>(grad_rows),
                       reinterpret_cast<const float4*>(act_in));
  else if (act_in)
    hipLaunchKernelGGL((nt_shade_bwd_kernel<false, false>), grid, dim3(SHB_BLOCK), 0, (hipStream_t)stream, *plan,
                       hit_slot, tex_uv, rays_d, reinterpret_cast<const float4*>(tris), slot_of,
                       seg_start, reinterpret_cast<const unsigned*>(texels), nr_rays, g_surfs_rgb,
                       g_surfs_alpha, grad_scale, reinterpret_cast<_Float16*>(grad_rows),
                       reinterpret_cast<const float4*>(act_in));
  else if (plan->row_format != 0)     // no kept sigmoids: re-gather the f16 rows
    hipLaunchKernelGGL((nt_shade_bwd_kernel<true, false, true>), grid, dim3(SHB_BLOCK), 0, (hipStream_t)stream, *plan,
                       hit_slot, tex_uv, rays_d, reinterpret_cast<const float4*>(tris), slot_of,
                       seg_start, reinterpret_cast<const unsigned*>(texels), nr_rays, g_surfs_rgb,
                       g_surfs_alpha, grad_scale, reinterpret_cast<_Float16*>(grad_rows), nullptr);
  else
    hipLaunchKernelGGL((nt_shade_bwd_kernel<true, false>), grid, dim3(SHB_BLOCK), 0, (hipStream_t)stream, *plan,
                       hit_slot, tex_uv, rays_d, reinterpret_cast<const float4*>(tris), slot_of,
                       seg_start, reinterpret_cast<const unsigned*>(texels), nr_rays, g_surfs_rgb,
                       g_surfs_alpha, grad_scale, reinterpret_cast<_Float16*>(grad_rows), nullptr);
  VSA_RETURN_LAUNCH_STATUS();
}

#ifdef SHADE_SPAN
extern "C" int vsa_span_read_shade(void* dst) {
  VSA_HIP_TRY(hipDeviceSynchronize());
  VSA_HIP_TRY(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_sspan), sizeof(g_sspan)));
  return 0;
}
#endif
