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
    const int* __restrict__ seg_start, unsigned* __restrict__ texels, _Float16* __restrict__ pre_out) {
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
                              void* stream) {
  if (!plan || !weights_h || !features || !seg_start || !texels) return VSA_ERR_ARG;
  dim3 grid(mlp_grid_x(plan), plan->nr_shells * 2 * VSA_NT_MAX_DEG);
  hipLaunchKernelGGL(nt_mlp_fwd_kernel, grid, dim3(MLP_BLOCK), 0, (hipStream_t)stream, *plan,
                     reinterpret_cast<const _Float16*>(weights_h),
                     reinterpret_cast<const unsigned*>(features), seg_start,
                     reinterpret_cast<unsigned*>(texels), reinterpret_cast<_Float16*>(pre_out));
  VSA_RETURN_LAUNCH_STATUS();
}
