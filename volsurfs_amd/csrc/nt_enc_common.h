// Device helpers shared by the hash-grid encode kernels (nt_encode.hip) and the fused
// encode + MLP kernel (nt_fused.hip): level geometry, cell arithmetic, corner indices.
// Arithmetic is the oracle's (oracle/tcnn_like.py hashgrid_forward), fp32, same order.
#pragma once
#include "nt_common.h"

namespace {

constexpr unsigned PRIME_Y = 2654435761u;

struct LevelGeom {
  float scale;
  unsigned res, size, mask;
};

__device__ __forceinline__ LevelGeom level_geom(const vsa_nt_plan& p, int l) {
  LevelGeom g;
  g.scale = p.level_scale[l];
  g.res = (unsigned)p.level_res[l];
  g.size = (unsigned)p.level_size[l];
  g.mask = g.size - 1;   // used by hashed levels only (their size is a power of two)
  return g;
}

struct CellCorners {
  unsigned idx[4];
  float w[4];
};

// normalised texel centre -> grid cell and bilinear weights at level g
// (oracle/tcnn_like.py hashgrid_forward: same fp32 operations in the same order).
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct CellRef {
  unsigned cx, cy;
  f32x2 w01, w23;   // corner weights (x0y0, x1y0), (x0y1, x1y1)
};

#ifndef NT_ENC_FRACT
#define NT_ENC_FRACT 1
#endif
// floor and fraction of a positive coordinate in two instructions instead of three:
// v_cvt_flr_i32_f32 = (int)floor(p), v_fract_f32 = p - floor(p) (the subtraction is exact in
// fp32, so this is the same value as the oracle's p - floorf(p))
__device__ __forceinline__ int enc_floor_i(float p) {
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(p));
  return r;
}
__device__ __forceinline__ float enc_fract(float p) {
  float r;
  asm("v_fract_f32 %0, %1" : "=v"(r) : "v"(p));
  return r;
}
// round to the nearest integer, halves upward, in one instruction (v_cvt_rpi_i32_f32 = (int)floor(x + 0.5));
// the backward's fixed-point contributions differ from round-half-to-even only on exact ties
__device__ __forceinline__ int enc_round_i(float x) {
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// Backward form, written on 2-vectors so that the x and y halves (and the weight pairs) go
// through v_pk_mul_f32 / v_pk_add_f32: the kernel is VALU-bound (PMC: ~70 % VALU-busy).
// Every operation is still one IEEE fp32 op with -ffp-contract=off: same bits.
__device__ __forceinline__ CellRef cell_ref(const LevelGeom& g, float x, float y) {
  const f32x2 xy = {x, y};
  const f32x2 p = xy * g.scale + 0.5f;
#if NT_ENC_FRACT
  const f32x2 f = {enc_fract(p.x), enc_fract(p.y)};
  const f32x2 q = 1.0f - f;
  CellRef c;
  c.cx = (unsigned)enc_floor_i(p.x);
  c.cy = (unsigned)enc_floor_i(p.y);
#else
  const f32x2 fl = {floorf(p.x), floorf(p.y)};
  const f32x2 f = p - fl;
  const f32x2 q = 1.0f - f;
  CellRef c;
  c.cx = (unsigned)(int)fl.x;
  c.cy = (unsigned)(int)fl.y;
#endif
  const f32x2 ax = {q.x, f.x};
  c.w01 = ax * q.y;
  c.w23 = ax * f.y;
  return c;
}

// Scalar form for the forward kernel (it keeps 8 cells live per lane; the paired
// registers of the packed form cost it more moves than they save: 0.27 -> 0.30 ms).
struct CellRefS {
  unsigned cx, cy;
  float w[4];
};

__device__ __forceinline__ CellRefS cell_ref_s(const LevelGeom& g, float x, float y) {
  const float px = x * g.scale + 0.5f, py = y * g.scale + 0.5f;
#if NT_ENC_FRACT
  const float fx = enc_fract(px), fy = enc_fract(py);
  const float gx = 1.0f - fx, gy = 1.0f - fy;
  CellRefS c;
  c.cx = (unsigned)enc_floor_i(px);
  c.cy = (unsigned)enc_floor_i(py);
#else
  const float flx = floorf(px), fly = floorf(py);
  const float fx = px - flx, fy = py - fly;
  const float gx = 1.0f - fx, gy = 1.0f - fy;
  CellRefS c;
  c.cx = (unsigned)(int)flx;
  c.cy = (unsigned)(int)fly;
#endif
  c.w[0] = gx * gy;
  c.w[1] = fx * gy;
  c.w[2] = gx * fy;
  c.w[3] = fx * fy;
  return c;
}

#ifndef NT_ENC_MIX
#define NT_ENC_MIX 1
#endif
// fp32 product of one half of a packed f16 pair and an fp32 value (see nt_mlp.hip mul_mix)
template <int HI>
__device__ __forceinline__ float enc_mul_mix(unsigned h2, float f) {
  float r;
  if constexpr (HI)
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(f));
  else
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(f));
  return r;
}

// the 4 table entries of a cell.  HASHED: tiny-cuda-nn's coherent prime hash, table
// size 2^k.  Dense: x + y*res, wrapped only at the far edge / on the apron of tiny
// textures.
template <bool HASHED>
__device__ __forceinline__ void cell_indices(const LevelGeom& g, unsigned cx, unsigned cy,
                                             unsigned idx[4]) {
  if (HASHED) {
    const unsigned h0 = cy * PRIME_Y, h1 = (cy + 1u) * PRIME_Y;
    idx[0] = (cx ^ h0) & g.mask;
    idx[1] = ((cx + 1u) ^ h0) & g.mask;
    idx[2] = (cx ^ h1) & g.mask;
    idx[3] = ((cx + 1u) ^ h1) & g.mask;
  } else {
    const unsigned r0 = cx + cy * g.res;
    idx[0] = r0;
    idx[1] = r0 + 1u;
    idx[2] = r0 + g.res;
    idx[3] = r0 + g.res + 1u;
    if (idx[3] >= g.size || idx[0] > idx[3]) {
#pragma unroll
      for (int k = 0; k < 4; ++k) idx[k] %= g.size;
    }
  }
}

template <bool HASHED>
__device__ __forceinline__ CellCorners cell_corners(const LevelGeom& g, float x, float y) {
  const CellRef r = cell_ref(g, x, y);
  CellCorners c;
  c.w[0] = r.w01.x, c.w[1] = r.w01.y, c.w[2] = r.w23.x, c.w[3] = r.w23.y;
  cell_indices<HASHED>(g, r.cx, r.cy, c.idx);
  return c;
}

// The feature of one level at one texel: bilinear blend of the cell's four table entries
// (packed f16x2 words e[], corner weights w[] in fp32).
// NT_ENC_ACC_F16 = 1 (default): tiny-cuda-nn's published kernel_grid —
// `result = fma((T)weight, grid_val(corner), result)` with T = __half over the corners in index
// order — i.e. four packed half FMAs (v_pk_fma_f16: one rounding each) on the weight rounded to
// half; oracle/tcnn_like.py hashgrid_forward(accumulate="f16").  4 conversions + 4 FMAs.
// NT_ENC_ACC_F16 = 0: the fp32 sum rounded once that rounds 1-2 restated (accumulate="f32"):
// 8 mixed-precision multiplies + 8 adds + 1 pack.
#ifndef NT_ENC_ACC_F16
#define NT_ENC_ACC_F16 1
#endif
#if NT_ENC_ACC_F16
// The blend in two steps, for callers that blend SEVERAL tables with one set of corner weights (the
// colour and the alpha texture of a (shell, degree) pair): the four weights rounded to half once ...
__device__ __forceinline__ void enc_weights_h(const float w[4], half2_t wh[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const _Float16 h = (_Float16)vsa_pin_f32(w[k]);     // (pinned: see enc_blend)
    wh[k] = half2_t{h, h};
  }
}
// ... and the four packed half FMAs per table (same operations, same order as enc_blend)
__device__ __forceinline__ unsigned enc_blend_h(const unsigned e[4], const half2_t wh[4]) {
  half2_t acc = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) acc = __builtin_elementwise_fma(wh[k], __builtin_bit_cast(half2_t, e[k]), acc);
  return __builtin_bit_cast(unsigned, acc);
}
#endif

__device__ __forceinline__ unsigned enc_blend(const unsigned e[4], const float w[4]) {
#if NT_ENC_ACC_F16
  half2_t acc = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // (pinned: hipcc would fold (half)(fx * fy) into v_fma_mixlo_f16, ONE rounding of the exact
    // product; the published kernel rounds the float product first, then casts it to half)
    const _Float16 wh = (_Float16)vsa_pin_f32(w[k]);
    const half2_t w2 = {wh, wh};
    acc = __builtin_elementwise_fma(w2, __builtin_bit_cast(half2_t, e[k]), acc);
  }
  return __builtin_bit_cast(unsigned, acc);
#else
  float f0 = 0.f, f1 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // w * (float)entry in ONE instruction per feature (v_fma_mix_f32 with a zero addend: the
    // f16 -> f32 conversion is exact, so the product is the same single rounding as
    // convert-then-multiply)
    f0 = f0 + enc_mul_mix<0>(e[k], w[k]);
    f1 = f1 + enc_mul_mix<1>(e[k], w[k]);
  }
  half2_t r;
  r.x = (_Float16)f0;
  r.y = (_Float16)f1;
  return __builtin_bit_cast(unsigned, r);
#endif
}

}  // namespace
