// Neural-texture step 4: the 32->64->64->C' MLP of every unique texel on MFMA,
// fused with the reference's post-processing (sigmoid, x255, round) —
// models/neural_texture.py:65-79 (tcnn FullyFusedMLP: ReLU, no bias, fp16) and
// :156-169.
//
// Orientation: out[neuron][point] = W[neuron][k] * act[k][point], i.e. weights
// are the MFMA A operand and a tile of 32 points (slots) sits on the lanes
// (v_mfma_f32_32x32x16_f16: lane = point, registers = 16 output rows).  The
// 32x32 f32 accumulator of one layer, ReLU'd and converted pairwise to f16,
// IS the B operand of the next layer's MFMA (sum over the accumulator's row
// index) with no lane movement and no LDS round trip; the k order inside a
// k-step is then permuted (row 16q + 8(j>>2) + 4h + (j&3) for element j of lane
// half h), which is absorbed into the order in which the weight fragments are
// staged.  16 MFMAs per 32 points; layer-1 B fragments are 4-byte coalesced
// loads from the level-major feature planes.
#include "nt_common.h"

namespace {

constexpr int MLP_BLOCK = 256;
constexpr int MLP_WAVES = MLP_BLOCK / 64;
constexpr int W1_OFF = 0, W2_OFF = 2048, W3_OFF = 6144;

// fragment ids in LDS (each fragment: 64 lanes x 8 halfs)
//   0..3   A1[m][s]   W1 rows 32m+r, cols 16s + 8h + j            (natural k)
//   4..11  A2[m][q]   W2 rows 32m+r, cols 16q + 8(j>>2) + 4h + (j&3)
//   12..15 A3[q]      W3 rows r,     cols 16q + 8(j>>2) + 4h + (j&3)
__device__ __forceinline__ int perm_k(int q, int h, int j) { return 16 * q + 8 * (j >> 2) + 4 * h + (j & 3); }

__device__ void stage_weights_fwd(const _Float16* __restrict__ W, half8_t* s_frag) {
  for (int idx = threadIdx.x; idx < 16 * 64; idx += MLP_BLOCK) {
    const int frag = idx >> 6, lane = idx & 63, r = lane & 31, h = lane >> 5;
    half8_t v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      _Float16 x;
      if (frag < 4) {
        const int m = frag >> 1, s = frag & 1;
        x = W[W1_OFF + (32 * m + r) * 32 + 16 * s + 8 * h + j];
      } else if (frag < 12) {
        const int f = frag - 4, m = f >> 2, q = f & 3;
        x = W[W2_OFF + (32 * m + r) * 64 + perm_k(q, h, j)];
      } else {
        x = W[W3_OFF + r * 64 + perm_k(frag - 12, h, j)];
      }
      v[j] = x;
    }
    s_frag[idx] = v;
  }
}

__device__ __forceinline__ half8_t relu_pack(const float16_t& acc, int s) {
  half8_t b;
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = (_Float16)fmaxf(acc[8 * s + j], 0.0f);
  return b;
}

struct TexInfo {
  int begin, end, type, channels;
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
  } else if (!(p.inner_solid && shell == 0) && deg < p.alpha_degrees) {
    t.channels = 2 * deg + 1;
  }
  t.begin = seg_start[shell * VSA_NT_MAX_DEG + deg];
  t.end = seg_start[shell * VSA_NT_MAX_DEG + deg + 1];
  return t;
}

// Forward network on one 32-point tile.  Returns acc3 (rows = output channels)
// and, when KEEP, the two hidden accumulators (pre-ReLU) for the backward pass.
template <bool KEEP>
__device__ __forceinline__ void mlp_tile_fwd(const half8_t* s_frag, const half8_t bx[2],
                                             float16_t acc1[2], float16_t acc2[2],
                                             float16_t& acc3) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    float16_t a = {0};
#pragma unroll
    for (int s = 0; s < 2; ++s)
      a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_frag[(m * 2 + s) * 64 + lane], bx[s], a, 0, 0, 0);
    acc1[m] = a;
  }
  half8_t b2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) b2[q] = relu_pack(acc1[q >> 1], q & 1);
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    float16_t a = {0};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_frag[(4 + m * 4 + q) * 64 + lane], b2[q], a, 0, 0, 0);
    acc2[m] = a;
  }
  half8_t b3[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) b3[q] = relu_pack(acc2[q >> 1], q & 1);
  float16_t a = {0};
#pragma unroll
  for (int q = 0; q < 4; ++q)
    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_frag[(12 + q) * 64 + lane], b3[q], a, 0, 0, 0);
  acc3 = a;
}

__device__ __forceinline__ void load_features(const unsigned* __restrict__ F, long long cap,
                                              int type, int n_levels, int slot, int h,
                                              half8_t bx[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[i] = F[((long long)type * n_levels + (8 * s + 4 * h + i)) * cap + slot];
    uint4 u = make_uint4(w[0], w[1], w[2], w[3]);
    bx[s] = __builtin_bit_cast(half8_t, u);
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

__global__ __launch_bounds__(MLP_BLOCK) void nt_mlp_fwd_kernel(
    vsa_nt_plan plan, const _Float16* __restrict__ weights, const unsigned* __restrict__ features,
    const int* __restrict__ seg_start, unsigned* __restrict__ texels, _Float16* __restrict__ pre_out,
    float* __restrict__ grad_rows) {
  __shared__ half8_t s_frag[16 * 64];
  const int tex = blockIdx.y;
  const TexInfo ti = tex_info(plan, seg_start, tex);
  if (ti.channels == 0 || ti.begin >= ti.end) return;
  const int ntiles = (ti.end - ti.begin + 31) >> 5;
  if ((int)blockIdx.x * MLP_WAVES >= ntiles) return;
  stage_weights_fwd(weights + (long long)tex * VSA_NT_WEIGHTS_PER_TEX, s_frag);
  __syncthreads();
  const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
  const int wave = blockIdx.x * MLP_WAVES + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * MLP_WAVES;
  const int dword_base = ti.type == 0 ? 0 : 6;   // rgb bytes 0..23, alpha bytes 24..31
  for (int tile = wave; tile < ntiles; tile += nwaves) {
    const int slot = ti.begin + tile * 32 + p;
    const bool valid = slot < ti.end;
    const int sl = valid ? slot : ti.end - 1;
    half8_t bx[2];
    load_features(features, plan.slot_capacity, ti.type, plan.n_levels, sl, h, bx);
    float16_t acc1[2], acc2[2], acc3;
    mlp_tile_fwd<false>(s_frag, bx, acc1, acc2, acc3);
    if (!valid) continue;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row0 = 8 * g + 4 * h;
      if (row0 >= ti.channels) continue;
      unsigned packed = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const _Float16 o_h = (_Float16)acc3[4 * g + i];
        if (pre_out && row0 + i < ti.channels)
          pre_out[(long long)slot * 32 + 4 * dword_base + row0 + i] = o_h;
        float q = rintf(sigmoidf_((float)o_h) * 255.0f);
        unsigned qb = row0 + i < ti.channels ? (unsigned)q : 0u;
        packed |= qb << (8 * i);
      }
      texels[(long long)slot * 8 + dword_base + (row0 >> 2)] = packed;
      if (grad_rows)
        *reinterpret_cast<float4*>(grad_rows + (long long)slot * 32 + 4 * (dword_base + (row0 >> 2))) =
            make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}


// ------------------------------------------------------------------ backward
// Fragment ids 16..31: the transposed weights for the data-gradient chain
//   16..19 T3[m][s]  elem j = W3[perm_k(s,h,j)][32m + r]
//   20..27 T2[m][q]  elem j = W2[perm_k(q,h,j)][32m + r]
//   28..31 T1[q]     elem j = W1[perm_k(q,h,j)][r]
__device__ void stage_weights_bwd(const _Float16* __restrict__ W, half8_t* s_frag) {
  for (int idx = threadIdx.x; idx < 16 * 64; idx += MLP_BLOCK) {
    const int frag = idx >> 6, lane = idx & 63, r = lane & 31, h = lane >> 5;
    half8_t v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      _Float16 x;
      if (frag < 4) {
        const int m = frag >> 1, s = frag & 1;
        x = W[W3_OFF + perm_k(s, h, j) * 64 + 32 * m + r];
      } else if (frag < 12) {
        const int f = frag - 4, m = f >> 2, q = f & 3;
        x = W[W2_OFF + perm_k(q, h, j) * 64 + 32 * m + r];
      } else {
        x = W[W1_OFF + perm_k(frag - 12, h, j) * 32 + r];
      }
      v[j] = x;
    }
    s_frag[16 * 64 + idx] = v;
  }
}

constexpr int IMG_STRIDE = 40;                 // halfs per image row (32 points + 16-B pad)
constexpr int IMG_ROWS = 320;                  // dOut 32 | H2 64 | dH2 64 | H1 64 | dH1 64 | X 32
constexpr int ROW_DOUT = 0, ROW_H2 = 32, ROW_DH2 = 96, ROW_H1 = 160, ROW_DH1 = 224, ROW_X = 288;

template <bool RELU>
__device__ __forceinline__ void store_image(_Float16* img, int row_base, const float16_t& acc,
                                            int p, int h) {
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
    const float v = RELU ? fmaxf(acc[reg], 0.0f) : acc[reg];
    img[(row_base + row) * IMG_STRIDE + p] = (_Float16)v;
  }
}

__device__ __forceinline__ half8_t read_frag(const _Float16* img, int row, int s, int h) {
  return *reinterpret_cast<const half8_t*>(img + row * IMG_STRIDE + 16 * s + 8 * h);
}

__device__ __forceinline__ half8_t pack8(const float16_t& acc, int s) {
  half8_t b;
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = (_Float16)acc[8 * s + j];
  return b;
}

__global__ __launch_bounds__(MLP_BLOCK, 1) void nt_mlp_bwd_kernel(
    vsa_nt_plan plan, const _Float16* __restrict__ weights, unsigned* __restrict__ features,
    const int* __restrict__ seg_start, const float* __restrict__ grad_rows,
    float* __restrict__ grad_weights) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  half8_t* s_frag = reinterpret_cast<half8_t*>(s_raw);                       // 32 KiB
  _Float16* s_img_all = reinterpret_cast<_Float16*>(s_raw + 32 * 64 * 16);   // 4 x 25.6 KB
  const int tex = blockIdx.y;
  const TexInfo ti = tex_info(plan, seg_start, tex);
  if (ti.channels == 0 || ti.begin >= ti.end) return;
  const int ntiles = (ti.end - ti.begin + 31) >> 5;
  if ((int)blockIdx.x * MLP_WAVES >= ntiles) return;
  const _Float16* W = weights + (long long)tex * VSA_NT_WEIGHTS_PER_TEX;
  stage_weights_fwd(W, s_frag);
  stage_weights_bwd(W, s_frag);
  __syncthreads();
  const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
  const int wave_in_wg = threadIdx.x >> 6;
  _Float16* img = s_img_all + wave_in_wg * IMG_ROWS * IMG_STRIDE;
  const int wave = blockIdx.x * MLP_WAVES + wave_in_wg;
  const int nwaves = gridDim.x * MLP_WAVES;
  const int float_base = ti.type == 0 ? 0 : 24;

  float16_t gW3[2], gW2[2][2], gW1[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    gW3[i] = float16_t{0};
    gW1[i] = float16_t{0};
    gW2[i][0] = float16_t{0};
    gW2[i][1] = float16_t{0};
  }

  for (int tile = wave; tile < ntiles; tile += nwaves) {
    const int slot = ti.begin + tile * 32 + p;
    const bool valid = slot < ti.end;
    const int sl = valid ? slot : ti.end - 1;
    half8_t bx[2];
    load_features(features, plan.slot_capacity, ti.type, plan.n_levels, sl, h, bx);
    float16_t acc1[2], acc2[2], acc3;
    mlp_tile_fwd<true>(s_frag, bx, acc1, acc2, acc3);

    // dL/d(pre-sigmoid output): G * sig * (1 - sig)   (round = STE, x255 /255 cancel)
    float16_t d3;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row0 = 8 * g + 4 * h;
      float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid && row0 < ti.channels)
        gr = *reinterpret_cast<const float4*>(grad_rows + (long long)slot * 32 + float_base + row0);
      const float gv[4] = {gr.x, gr.y, gr.z, gr.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float sg = sigmoidf_((float)(_Float16)acc3[4 * g + i]);
        d3[4 * g + i] = row0 + i < ti.channels ? gv[i] * sg * (1.0f - sg) : 0.0f;
      }
    }
    // images for the weight-gradient products (points along k)
    store_image<false>(img, ROW_DOUT, d3, p, h);
    store_image<true>(img, ROW_H2, acc2[0], p, h);
    store_image<true>(img, ROW_H2 + 32, acc2[1], p, h);
    store_image<true>(img, ROW_H1, acc1[0], p, h);
    store_image<true>(img, ROW_H1 + 32, acc1[1], p, h);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) img[(ROW_X + 16 * s + 8 * h + j) * IMG_STRIDE + p] = bx[s][j];

    // data gradients: dH2 = W3^T dOut ; dH1 = W2^T dH2 ; dX = W1^T dH1
    float16_t dh2[2], dh1[2], dx;
    {
      half8_t b[2] = {pack8(d3, 0), pack8(d3, 1)};
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float16_t a = {0};
#pragma unroll
        for (int s = 0; s < 2; ++s)
          a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_frag[(16 + m * 2 + s) * 64 + lane], b[s], a, 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) a[reg] = acc2[m][reg] > 0.0f ? a[reg] : 0.0f;
        dh2[m] = a;
      }
    }
    store_image<false>(img, ROW_DH2, dh2[0], p, h);
    store_image<false>(img, ROW_DH2 + 32, dh2[1], p, h);
    {
      half8_t b[4] = {pack8(dh2[0], 0), pack8(dh2[0], 1), pack8(dh2[1], 0), pack8(dh2[1], 1)};
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float16_t a = {0};
#pragma unroll
        for (int q = 0; q < 4; ++q)
          a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_frag[(20 + m * 4 + q) * 64 + lane], b[q], a, 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) a[reg] = acc1[m][reg] > 0.0f ? a[reg] : 0.0f;
        dh1[m] = a;
      }
    }
    store_image<false>(img, ROW_DH1, dh1[0], p, h);
    store_image<false>(img, ROW_DH1 + 32, dh1[1], p, h);
    {
      half8_t b[4] = {pack8(dh1[0], 0), pack8(dh1[0], 1), pack8(dh1[1], 0), pack8(dh1[1], 1)};
      float16_t a = {0};
#pragma unroll
      for (int q = 0; q < 4; ++q)
        a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_frag[(28 + q) * 64 + lane], b[q], a, 0, 0, 0);
      dx = a;
    }
    // dF (in place over the features): rows = feature index, pairs -> one level
    if (valid) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int level = 4 * g + 2 * h + i;
          half2_t v;
          v.x = (_Float16)dx[4 * g + 2 * i];
          v.y = (_Float16)dx[4 * g + 2 * i + 1];
          features[((long long)ti.type * plan.n_levels + level) * plan.slot_capacity + slot] =
              __builtin_bit_cast(unsigned, v);
        }
    }
    // weight gradients: sum over the tile's 32 points (2 k-steps of 16)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const half8_t a3 = read_frag(img, ROW_DOUT + p, s, h);
      const half8_t a2[2] = {read_frag(img, ROW_DH2 + p, s, h), read_frag(img, ROW_DH2 + 32 + p, s, h)};
      const half8_t a1[2] = {read_frag(img, ROW_DH1 + p, s, h), read_frag(img, ROW_DH1 + 32 + p, s, h)};
      const half8_t bh2[2] = {read_frag(img, ROW_H2 + p, s, h), read_frag(img, ROW_H2 + 32 + p, s, h)};
      const half8_t bh1[2] = {read_frag(img, ROW_H1 + p, s, h), read_frag(img, ROW_H1 + 32 + p, s, h)};
      const half8_t bxx = read_frag(img, ROW_X + p, s, h);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        gW3[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3, bh2[m], gW3[m], 0, 0, 0);
        gW1[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[m], bxx, gW1[m], 0, 0, 0);
#pragma unroll
        for (int mj = 0; mj < 2; ++mj)
          gW2[m][mj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[m], bh1[mj], gW2[m][mj], 0, 0, 0);
      }
    }
  }

  // flush: accumulator (row = (reg&3)+8(reg>>2)+4h, col = p) -> grad_weights
  float* gw = grad_weights + (long long)tex * VSA_NT_WEIGHTS_PER_TEX;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if (row < ti.channels) atomicAdd(&gw[W3_OFF + row * 64 + 32 * m + p], gW3[m][reg]);
      atomicAdd(&gw[W1_OFF + (32 * m + row) * 32 + p], gW1[m][reg]);
#pragma unroll
      for (int mj = 0; mj < 2; ++mj)
        atomicAdd(&gw[W2_OFF + (32 * m + row) * 64 + 32 * mj + p], gW2[m][mj][reg]);
    }
  }
}

}  // namespace

static int mlp_grid_x(const vsa_nt_plan* p) {
  long long worst = 0;
  for (int i = 0; i < p->nr_shells * VSA_NT_MAX_DEG; ++i) {
    long long d = p->dom_off[i + 1] - p->dom_off[i];
    worst = worst > d ? worst : d;
  }
  if (worst > p->slot_capacity) worst = p->slot_capacity;
  long long tiles = (worst + 31) / 32;
  long long wg = (tiles + MLP_WAVES - 1) / MLP_WAVES;
  if (wg > 64) wg = 64;   // 256 waves per texture at most; tiles are strided over them
  return wg < 1 ? 1 : (int)wg;
}

extern "C" int vsa_nt_mlp_fwd(const vsa_nt_plan* plan, const void* weights_h, const void* features,
                              const int32_t* seg_start, uint8_t* texels, void* pre_out,
                              float* grad_rows, void* stream) {
  if (!plan || !weights_h || !features || !seg_start || !texels) return VSA_ERR_ARG;
  dim3 grid(mlp_grid_x(plan), plan->nr_shells * 2 * VSA_NT_MAX_DEG);
  hipLaunchKernelGGL(nt_mlp_fwd_kernel, grid, dim3(MLP_BLOCK), 0, (hipStream_t)stream, *plan,
                     reinterpret_cast<const _Float16*>(weights_h),
                     reinterpret_cast<const unsigned*>(features), seg_start,
                     reinterpret_cast<unsigned*>(texels), reinterpret_cast<_Float16*>(pre_out),
                     grad_rows);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_nt_mlp_bwd(const vsa_nt_plan* plan, const void* weights_h, void* features,
                              const int32_t* seg_start, const float* grad_rows,
                              float* grad_weights, void* stream) {
  if (!plan || !weights_h || !features || !seg_start || !grad_rows || !grad_weights)
    return VSA_ERR_ARG;
  const size_t lds = 32 * 64 * 16 + (size_t)MLP_WAVES * IMG_ROWS * IMG_STRIDE * 2;
  static bool attr_set = false;
  if (!attr_set) {
    VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(nt_mlp_bwd_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  int gx = mlp_grid_x(plan);
  if (gx > 16) gx = 16;  // fewer, longer-lived waves: one weight-gradient flush per wave
  dim3 grid(gx, plan->nr_shells * 2 * VSA_NT_MAX_DEG);
  hipLaunchKernelGGL(nt_mlp_bwd_kernel, grid, dim3(MLP_BLOCK), lds, (hipStream_t)stream, *plan,
                     reinterpret_cast<const _Float16*>(weights_h),
                     reinterpret_cast<unsigned*>(features), seg_start, grad_rows, grad_weights);
  VSA_RETURN_LAUNCH_STATUS();
}
