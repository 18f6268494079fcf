// Device helpers shared by the neural-texture MLP kernels (nt_mlp.hip) and the fused
// encode + MLP kernel (nt_fused.hip): weight-fragment layout, the forward network of one
// 32-point tile on v_mfma_f32_32x32x16_f16, mixed-precision helpers.
#pragma once
#include "nt_common.h"

namespace {

constexpr int W1_OFF = 0, W2_OFF = 2048, W3_OFF = 6144;

// fragment ids in LDS (each fragment: 64 lanes x 8 halfs)
//   0..3   A1[m][s]   W1 rows 32m+r, cols 16s + 8h + j            (natural k)
//   4..11  A2[m][q]   W2 rows 32m+r, cols 16q + 8(j>>2) + 4h + (j&3)
//   12..15 A3[q]      W3 rows r,     cols 16q + 8(j>>2) + 4h + (j&3)
__device__ __forceinline__ int perm_k(int q, int h, int j) { return 16 * q + 8 * (j >> 2) + 4 * h + (j & 3); }

// INT_RELU: the maximum taken on the f16 BIT patterns as signed 16-bit integers (v_pk_max_i16).
// Same values as the floating-point maximum for every finite input, but a zero always comes out
// as +0 (bits 0), never -0: the backward pass tests "activation > 0" as "bits != 0" without
// stripping the sign first.
template <bool INT_RELU = false>
__device__ __forceinline__ half8_t relu_pack(const float16_t& acc, int s) {
  half8_t b;
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = (_Float16)acc[8 * s + j];
  // ReLU after the fp16 rounding (same values as before it): 4 v_pk_max_f16
  if constexpr (INT_RELU) {
    // (inline asm on the packed words: the vector form on short8 makes the compiler convert the
    // accumulators one by one and pack them with v_perm_b32 instead of v_cvt_pk_f16_f32)
    typedef unsigned uint4v_ __attribute__((ext_vector_type(4)));
    uint4v_ bi = __builtin_bit_cast(uint4v_, b);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned w = bi[i];
      asm("v_pk_max_i16 %0, %0, 0" : "+v"(w));
      bi[i] = w;
    }
    return __builtin_bit_cast(half8_t, bi);
  } else {
    return __builtin_elementwise_max(b, half8_t{0, 0, 0, 0, 0, 0, 0, 0});
  }
}

struct TexInfo {
  int begin, end, type, channels;
  int row_quads;        // quads per slot row of this degree
  long long row_first;  // quad of slot `begin`'s first channel of THIS texture (rgb: 0, alpha: +alpha quad)
  int own_quads;        // quads of a slot row that belong to THIS texture (rgb: up to the alpha quad)
  unsigned own_magic;   // i / own_quads == __umulhi(i, own_magic) for 2 <= own_quads <= 8, i < 2^16
};

__device__ __forceinline__ TexInfo tex_info(const vsa_nt_plan& p, const int* seg_start, int tex) {
  const int deg = tex % VSA_NT_MAX_DEG;
  const int type = (tex / VSA_NT_MAX_DEG) & 1;
  const int shell = tex / (2 * VSA_NT_MAX_DEG);
  TexInfo t;
  t.type = type;
  t.channels = 0;
  if (type == 0) {
    if (deg < p.rgb_degrees) t.channels = 3 * (2 * deg + 1);
  } else if (nt_shell_has_alpha(p, shell) && deg < p.alpha_degrees) {
    t.channels = 2 * deg + 1;
  }
  t.begin = seg_start[shell * VSA_NT_MAX_DEG + deg];
  t.end = seg_start[shell * VSA_NT_MAX_DEG + deg + 1];
  t.row_quads = nt_row_quads(deg);
  t.row_first = p.row_base[shell * VSA_NT_MAX_DEG + deg] + (type ? nt_alpha_quad(deg) : 0);
  t.own_quads = type ? t.row_quads - nt_alpha_quad(deg) : nt_alpha_quad(deg);
  t.own_magic = 0xffffffffu / (unsigned)t.own_quads + 1u;
  return t;
}

// Forward network on one 32-point tile.  Returns acc3 (rows = output channels)
// and, when KEEP, the two hidden accumulators (pre-ReLU) for the backward pass.
// b2 / b3: the ReLU'd hidden activations as f16 B fragments (k-step q, element j
// <-> accumulator register 8(q&1)+j of tile q>>1); the backward pass keeps these
// instead of the fp32 accumulators (ReLU mask = value > 0).
// wf: the 16 forward weight fragments of this lane, register resident (loaded once
// per workgroup task; re-reading them from LDS per MFMA exposes an LDS round trip
// in front of every matrix instruction at one wave per SIMD).
template <bool INT_RELU = false>
__device__ __forceinline__ void mlp_tile_fwd(const half8_t wf[16], const half8_t bx[2],
                                             half8_t b2[4], half8_t b3[4], float16_t& acc3) {
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    float16_t a = {0};
#pragma unroll
    for (int s = 0; s < 2; ++s)
      a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[m * 2 + s], bx[s], a, 0, 0, 0);
    b2[2 * m] = relu_pack<INT_RELU>(a, 0);
    b2[2 * m + 1] = relu_pack<INT_RELU>(a, 1);
  }
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    float16_t a = {0};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[4 + m * 4 + q], b2[q], a, 0, 0, 0);
    b3[2 * m] = relu_pack<INT_RELU>(a, 0);
    b3[2 * m + 1] = relu_pack<INT_RELU>(a, 1);
  }
  float16_t a = {0};
#pragma unroll
  for (int q = 0; q < 4; ++q)
    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[12 + q], b3[q], a, 0, 0, 0);
  acc3 = a;
}

// sigmoid with the hardware exp2 / rcp (1 ulp each): the result is quantised to 8
// bits (forward) or multiplies a gradient (backward), so exact division buys nothing
// fp32 product of one half of a packed f16 pair and an fp32 value: conversion and multiply in
// one instruction (v_fma_mix_f32 with a zero addend; the same product as convert-then-multiply)
template <int HI>
__device__ __forceinline__ float mul_mix(unsigned h2, float f) {
  float r;
  if constexpr (HI)
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(f));
  else
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(f));
  return r;
}

// {f16(lo(h2) * f0), f16(hi(h2) * f1)}: the fp32 products of the two halves of a packed f16 pair,
// each rounded once to f16, packed (v_fma_mixlo_f16 / v_fma_mixhi_f16)
__device__ __forceinline__ unsigned mul_mix_pk(unsigned h2, float f0, float f1) {
  unsigned r;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(f0));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h2), "v"(f1));
  return r;
}

__device__ __forceinline__ float sigmoidf_(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// The 16 forward weight fragments of this lane straight from memory: every fragment is one
// 16-byte run (W1, natural k) or two 8-byte runs (W2 / W3, permuted k) of a weight row, so a
// lane needs 4 + 24 wide loads per run and nothing goes through LDS (the 2-byte gather into a
// shared image that this replaces cost 4 dependent rounds of 8 loads plus two barriers per run).
__device__ __forceinline__ void load_fwd_frags(const _Float16* __restrict__ W, int lane, half8_t wf[16]) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    wf[i] = *reinterpret_cast<const half8_t*>(W + W1_OFF + (32 * (i >> 1) + r) * 32 + 16 * (i & 1) + 8 * h);
#pragma unroll
  for (int i = 4; i < 16; ++i) {
    const _Float16* row = i < 12 ? W + W2_OFF + (32 * ((i - 4) >> 2) + r) * 64 : W + W3_OFF + r * 64;
    const int q = i < 12 ? (i - 4) & 3 : i - 12;
    const half4_t lo = *reinterpret_cast<const half4_t*>(row + 16 * q + 4 * h);
    const half4_t hi = *reinterpret_cast<const half4_t*>(row + 16 * q + 8 + 4 * h);
    wf[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

// Epilogue of the forward network on one 32-point tile: the reference's sigmoid -> x255 -> round
// (models/neural_texture.py:156-169) of the NG 8-channel groups of the output, stored as the
// slot's quantised texel row.  s_qt: the 257 thresholds of nt_quant_table.h in LDS.
// PRE: also write the pre-sigmoid outputs (tests only).
template <int NG, bool PRE>
__device__ __forceinline__ void quant_store_tile(const float16_t& acc3, const TexInfo& ti,
                                                 const unsigned* s_qt, unsigned* __restrict__ texels,
                                                 int slot, bool valid, int h,
                                                 _Float16* __restrict__ pre_out, int pre_base) {
  unsigned* const trow = texels + ti.row_first + (long long)(slot - ti.begin) * ti.row_quads + h;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int row0 = 8 * g + 4 * h;
    unsigned qv[4];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
      qv[2 * i2] = qv[2 * i2 + 1] = 0u;
      // the last group of a narrow texture (band 0: 3 colour / 1 alpha channel, 64 % of the slots):
      // rows 8 g + 2 i2 (h = 0) and up are beyond the texture for every lane -> nothing to quantise
      // (wave-uniform test; the forward is bound by vector issue, profiles/NOTEBOOK.md round 4)
      if (g == NG - 1 && 8 * g + 2 * i2 >= ti.channels) continue;
      const half2_t o_h = {(_Float16)acc3[4 * g + 2 * i2], (_Float16)acc3[4 * g + 2 * i2 + 1]};
      const unsigned ob = __builtin_bit_cast(unsigned, o_h);
      if constexpr (PRE) {
        if (valid && row0 + 2 * i2 < ti.channels) pre_out[(long long)slot * 32 + pre_base + row0 + 2 * i2] = o_h.x;
        if (valid && row0 + 2 * i2 + 1 < ti.channels) pre_out[(long long)slot * 32 + pre_base + row0 + 2 * i2 + 1] = o_h.y;
      }
      // the reference's round(sigmoid(x) * 255) as the exact step function of x: a fast
      // estimate biased low by 0.01 of a step (the hardware exp2 / rcp are good to ~1e-4 of
      // one) is the exact value or one below it; one threshold of the table decides which.
      // The monotone 16-bit keys of both halves at once: negative -> ~bits, else bits | 0x8000.
      unsigned m2 = ob, key2;
      asm("v_pk_ashrrev_i16 %0, %1, %0" : "+v"(m2) : "v"(0x000f000fu));
      key2 = ob ^ (m2 | 0x80008000u);
      const float s0 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(mul_mix<0>(ob, -1.4426950408889634f)));
      unsigned q0 = (unsigned)__builtin_fmaf(s0, 255.0f, 0.49f);
      q0 += (key2 & 0xffffu) >= s_qt[q0 + 1] ? 1u : 0u;
      unsigned q1 = 0u;
      if (!(g == NG - 1 && 8 * g + 2 * i2 + 1 >= ti.channels)) {     // (wave-uniform, as above)
        const float s1 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(mul_mix<1>(ob, -1.4426950408889634f)));
        q1 = (unsigned)__builtin_fmaf(s1, 255.0f, 0.49f);
        q1 += (key2 >> 16) >= s_qt[q1 + 1] ? 1u : 0u;
      }
      if (g == NG - 1) {     // only the last group can hold rows beyond the texture's channels
        q0 = row0 + 2 * i2 < ti.channels ? q0 : 0u;
        q1 = row0 + 2 * i2 + 1 < ti.channels ? q1 : 0u;
      }
      qv[2 * i2] = q0;
      qv[2 * i2 + 1] = q1;
    }
    // (measured: staging the tile's quads through LDS and storing them in memory order,
    // as the backward's clear does, is SLOWER here: 0.28 -> 0.32 ms)
    if (valid && row0 < ti.channels) trow[2 * g] = qv[0] | (qv[1] << 8) | (qv[2] << 16) | (qv[3] << 24);
  }
}

// row_format 1: the same tile stored as f16 rows of sigmoid(x), un-quantised (neural_texture.py:159-164 with
// quantize_output = False): the fp32 sigmoid of the fp16 network output, rounded to half (:177).  A quad = 4 halves.
// raw (row_format 2, squeeze_output = False, :157-169 skipped): the fp16 network output itself.
template <int NG, bool PRE>
__device__ __forceinline__ void half_store_tile(const float16_t& acc3, const TexInfo& ti,
                                                uint2* __restrict__ texels_h, int slot, bool valid, int h,
                                                _Float16* __restrict__ pre_out, int pre_base, bool raw = false) {
  uint2* const trow = texels_h + ti.row_first + (long long)(slot - ti.begin) * ti.row_quads + h;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int row0 = 8 * g + 4 * h;
    unsigned short hv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const _Float16 xh = (_Float16)acc3[4 * g + i];
      if constexpr (PRE) {
        if (valid && row0 + i < ti.channels) pre_out[(long long)slot * 32 + pre_base + row0 + i] = xh;
      }
      const float sg = 1.0f / (1.0f + expf(-(float)xh));          // accurate exp: this path is not the hot one
      const _Float16 sh = raw ? xh : (_Float16)sg;
      hv[i] = row0 + i < ti.channels ? __builtin_bit_cast(unsigned short, sh) : (unsigned short)0;
    }
    if (valid && row0 < ti.channels)
      trow[2 * g] = make_uint2((unsigned)hv[0] | ((unsigned)hv[1] << 16), (unsigned)hv[2] | ((unsigned)hv[3] << 16));
  }
}

}  // namespace
