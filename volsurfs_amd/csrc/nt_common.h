// Shared device helpers for the neural-texture kernels (gfx950).
#pragma once
#include "common.h"

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));

#define NT_DOM_BLOCK 4096  // texel domains are padded to this (one scan block)

__host__ __device__ inline int nt_tex_index(int shell, int type, int deg) {
  return (shell * 2 + type) * VSA_NT_MAX_DEG + deg;
}

// The 2x2 texel footprint of a uv sample in a texture of resolution R
// (models/neural_texture.py:107-138 with align_to_webgl; oracle/neural_texture.py
// texel_corners).  (i0, j0) is the lower corner in texel units, may be -1;
// the footprint lives at extended-grid coordinates (i0+1 .. i0+2, j0+1 .. j0+2).
struct NtFootprint {
  int i0, j0;
  float fx, fy;
};

__device__ __forceinline__ NtFootprint nt_footprint(float u, float v, int R) {
  const float Rf = (float)R;
  const float a = u * Rf, b = v * Rf;  // non_normalize_uv_coord
  const float ap = Rf - b;             // rotate 90 (neural_texture.py:114-121)
  const float bp = a;
  const float fl_x = floorf(ap - 0.5f), fl_y = floorf(bp - 0.5f);
  NtFootprint f;
  f.fx = ap - (fl_x + 0.5f);
  f.fy = bp - (fl_y + 0.5f);
  // uv outside [0,1] is a caller error in the reference (no clamp, :144-145);
  // keep indices inside the one-texel apron so that nothing is written out of bounds
  f.i0 = min(max((int)fl_x, -1), R - 1);
  f.j0 = min(max((int)fl_y, -1), R - 1);
  return f;
}

// uv of a hit from its barycentrics and the face's per-corner uvs (volsurfs.py:511-514)
__device__ __forceinline__ float2 nt_interp_uv(const float* __restrict__ fuv, float bu, float bv) {
  const float b0 = (1.0f - bu) - bv;
  float2 r;
  r.x = (b0 * fuv[0] + bu * fuv[2]) + bv * fuv[4];
  r.y = (b0 * fuv[1] + bu * fuv[3]) + bv * fuv[5];
  return r;
}
