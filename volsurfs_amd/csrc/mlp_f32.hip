// Fused fp32 MLP (Linear + bias + exact GELU stack) on the fp32-input matrix cores, for the
// legacy appearance models and the background field:
//   MLP      /root/reference/volsurfs_py/models/mlp.py:8-69
//   RGB      models/rgb.py:139 (66 -> 128 -> 128 -> 64 -> C), ColorSH models/color_sh.py
//   NerfHash models/nerfhash.py:58-91 (51 -> 64 -> 64 -> 64 -> 65; 80 -> 64 -> 64 -> 3)
// The reference runs these as torch fp32 GEMMs; parity is 1e-4 against fixtures made from its
// own classes (tests/golden/legacy_models.npz), so the arithmetic stays fp32:
// v_mfma_f32_32x32x2_f32 (exact f32 FMA chains, 157 TFLOP/s peak).
//
// Orientation (as nt_mlp.hip): out[neuron][point] = W[neuron][k] * act[k][point].  A 32-point
// tile sits on the lanes of a wave, weights are the A operand, and the 32x32 accumulator of a
// layer — bias added, GELU applied, all in registers — IS the B operand of the next layer's
// MFMAs: k-step s of a 32-row block consumes accumulator register s of both lane halves, i.e.
// rows rho(s, h) = 8 (s >> 2) + 4 h + (s & 3); the weight fragments are pre-packed in that k
// order (mlp_pack_kernel), one 256-byte conflict-free LDS row per MFMA.  A layer's packed weights
// (<= 64 KiB) are staged in LDS per layer for the 4 tiles of a workgroup.
//
// Backward = two kernels.  mlp_dgrad: the same register chain with transposed weights
// (dA = W^T dZ), dZ = dA * GELU'(z) from the saved pre-activations; it writes dZ and A = GELU(z)
// of every hidden layer [point][width] row-major.  mlp_wgrad: dW = dZ^T A contracts over POINTS,
// and with row-major [point][width] operands the MFMA fragments are plain coalesced dword loads
// (lane (i, kk): dZ[p0 + kk][32 m + i]) — no transposes; persistent workgroups keep their dW
// blocks in registers over their share of the points, partial sums are written per workgroup
// and added up by mlp_reduce (deterministic, no float atomics).
#include "common.h"
#include "gelu_fast.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MLP_MAXB = 4;               // layer widths up to 128 (4 blocks of 32)
constexpr int MLP_BLOCK = 256;            // 4 waves = 4 tiles of 32 points
constexpr int MLP_TILE = 32;
#ifndef MLP_XPF
#define MLP_XPF 1
#endif
#ifndef MLP_ZPF
#define MLP_ZPF 1
#endif

__device__ __forceinline__ int rho(int s, int h) { return 8 * (s >> 2) + 4 * h + (s & 3); }
__host__ __device__ __forceinline__ int blocks_of(int w) { return (w + 31) / 32; }

// offsets (floats) of layer l inside the packed buffers
struct PackOffsets {
  int fwd[VSA_MLP_MAX_LAYERS + 1];
};

__host__ __device__ inline PackOffsets pack_offsets(const vsa_mlp_plan& p) {
  PackOffsets o;
  int acc = 0;
  for (int l = 0; l < p.n_layers; ++l) {
    o.fwd[l] = acc;
    acc += blocks_of(p.dims[l]) * blocks_of(p.dims[l + 1]) * 16 * 64;
  }
  o.fwd[p.n_layers] = acc;
  return o;
}

// Per-layer constants of the plan in LDS.  The kernels walk the layers with a run-time index; on
// the by-value kernel argument that made the compiler copy the whole plan (and the offsets
// derived from it) to scratch memory and fetch dims / offsets / bias pointers from there inside
// the layer loops.  Here the copy is made once with compile-time indices (straight from the
// kernel-argument registers), and the loops read LDS.
struct LayerMeta {
  int dims[VSA_MLP_MAX_LAYERS + 1];
  int fwd[VSA_MLP_MAX_LAYERS + 1];       // PackOffsets::fwd
  const float* b[VSA_MLP_MAX_LAYERS];
};

// Every launch covers up to MLP_MAX_GROUPS networks of ONE architecture (the K per-shell models of
// the legacy appearance branch, volsurfs.py:402-470: each applied to its own shell's hits — ~80
// workgroups per network on a 256-CU part when launched one by one, and five launches where one
// does): `plan` carries the shared layer widths, this descriptor what differs per group — weight /
// bias pointers, the number of rows and where they start.  Group = blockIdx.y (pack / reduce: z).
// A single network is the one-group case.  Rows of group g start at row0[g] in x / y / dy / dx and
// at row0[g] * (sum of hidden widths) floats in the z / dz / a workspaces; packed weights and
// weight-gradient partials are per group (stride given at launch).
constexpr int MLP_MAX_GROUPS = 8;
struct MlpGroups {
  int M[MLP_MAX_GROUPS];
  long long row0[MLP_MAX_GROUPS];
  const float* w[MLP_MAX_GROUPS][VSA_MLP_MAX_LAYERS];
  const float* b[MLP_MAX_GROUPS][VSA_MLP_MAX_LAYERS];
};
struct MlpGroupGrads {
  float* dw[MLP_MAX_GROUPS][VSA_MLP_MAX_LAYERS];
  float* db[MLP_MAX_GROUPS][VSA_MLP_MAX_LAYERS];
  int accumulate;
};
// element g of a kernel-argument array WITHOUT indexing it at run time (that would move the whole
// argument to scratch memory, as the layer loops did with the plan): a chain of selects over
// compile-time indices; g is uniform, so they are scalar
template <class T, int N>
__device__ __forceinline__ T pick(const T (&a)[N], int g) {
  T r = a[0];
#pragma unroll
  for (int i = 1; i < N; ++i)
    if (g == i) r = a[i];
  return r;
}
template <class T, int N, int K>
__device__ __forceinline__ T pick2(const T (&a)[N][K], int g, int k /* compile-time at the call sites that matter */) {
  T r = a[0][k];
#pragma unroll
  for (int i = 1; i < N; ++i)
    if (g == i) r = a[i][k];
  return r;
}

__device__ __forceinline__ void load_layer_meta(const vsa_mlp_plan& plan, const MlpGroups& gp, int g,
                                                LayerMeta& m) {
  if (threadIdx.x == 0) {
    int acc = 0;
#pragma unroll
    for (int l = 0; l < VSA_MLP_MAX_LAYERS; ++l) {
      m.dims[l] = plan.dims[l];
      m.b[l] = pick2(gp.b, g, l);
      m.fwd[l] = acc;
      if (l < plan.n_layers) acc += blocks_of(plan.dims[l]) * blocks_of(plan.dims[l + 1]) * 16 * 64;
    }
    m.dims[VSA_MLP_MAX_LAYERS] = plan.dims[VSA_MLP_MAX_LAYERS];
    m.fwd[VSA_MLP_MAX_LAYERS] = acc;
  }
  __syncthreads();
}
__device__ __forceinline__ int meta_dim(const LayerMeta& m, int l) { return __builtin_amdgcn_readfirstlane(m.dims[l]); }
__device__ __forceinline__ int meta_fwd(const LayerMeta& m, int l) { return __builtin_amdgcn_readfirstlane(m.fwd[l]); }
__device__ __forceinline__ const float* meta_bias(const LayerMeta& m, int l) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(m.b[l]);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}

// packed_fwd[l][(m * inb + b) * 1024 + mlp_frag_pos(s, lane)] = W_l[32 m + (lane & 31)][32 b + rho(s, lane >> 5)]
// packed_bwd[l][(b * outb + m) * 1024 + mlp_frag_pos(s, lane)] = W_l[32 m + rho(s, lane >> 5)][32 b + (lane & 31)]
// mlp_frag_pos: with MLP_FRAG_B128 = 1 a lane's four consecutive k-steps are one 16-byte run
// ([s / 4][lane][s % 4]), so ONE ds_read_b128 (lanes 16 B apart: conflict-free) feeds four MFMAs instead
// of a 4-byte ds_read per v_mfma_f32_32x32x2_f32 (1199 -> 271 narrow reads in the ISA).  Measured on
// tools/bench_bg.py (one box, two runs each): fwd 881 / 873 us, dgrad 1189-1212 / 1186-1216 us - no
// difference, so the issue stalls of profiles/NOTEBOOK.md A9.5 are not the fragment reads; the k-step-major layout
// of rounds 1-2 stays the default.
#ifndef MLP_FRAG_B128
#define MLP_FRAG_B128 0
#endif
__host__ __device__ inline int mlp_frag_pos(int s, int lane) {
  return MLP_FRAG_B128 ? ((s >> 2) * 64 + lane) * 4 + (s & 3) : s * 64 + lane;
}
// the 16 fragments of one 32x32 block pair for this lane
__device__ __forceinline__ void mlp_load_frags(const float* blk, int lane, float w[16]) {
#if MLP_FRAG_B128
  const float4* f4 = reinterpret_cast<const float4*>(blk) + lane;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = f4[q * 64];
    w[4 * q] = v.x, w[4 * q + 1] = v.y, w[4 * q + 2] = v.z, w[4 * q + 3] = v.w;
  }
#else
#pragma unroll
  for (int s = 0; s < 16; ++s) w[s] = blk[s * 64 + lane];
#endif
}
__global__ void mlp_pack_kernel(vsa_mlp_plan plan, MlpGroups gp, long long packed_stride,
                                float* __restrict__ packed_fwd, float* __restrict__ packed_bwd) {
  const PackOffsets off = pack_offsets(plan);
  const int l = blockIdx.y, g = blockIdx.z;
  const int in = plan.dims[l], out = plan.dims[l + 1];
  const int inb = blocks_of(in), outb = blocks_of(out);
  const int n = inb * outb * 16 * 64;
  const float* W = nullptr;
#pragma unroll
  for (int k = 0; k < VSA_MLP_MAX_LAYERS; ++k)
    if (l == k) W = pick2(gp.w, g, k);
  if (packed_fwd) packed_fwd += g * packed_stride;
  if (packed_bwd) packed_bwd += g * packed_stride;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63, s = (idx >> 6) & 15, blk = idx >> 10;
    const int i = lane & 31, h = lane >> 5;
    const long long dst = off.fwd[l] + (long long)blk * 1024 + mlp_frag_pos(s, lane);
    if (packed_fwd) {
      const int m = blk / inb, b = blk - m * inb;
      const int row = 32 * m + i, col = 32 * b + rho(s, h);
      packed_fwd[dst] = (row < out && col < in) ? W[(long long)row * in + col] : 0.f;
    }
    if (packed_bwd) {
      const int b = blk / outb, m = blk - b * outb;
      const int row = 32 * m + rho(s, h), col = 32 * b + i;
      packed_bwd[dst] = (row < out && col < in) ? W[(long long)row * in + col] : 0.f;
    }
  }
}

// exact GELU (torch.nn.GELU default) = z Phi(z); Phi to 6.6e-8 absolute in ~13 instructions (gelu_fast.h; the device
// library's erff: ~45 — the forward spent as long in its GELUs as in its matrix instructions)
#ifndef MLP_GELU_ERFF
#define MLP_GELU_ERFF 0
#endif
__device__ __forceinline__ float gelu_f(float z) {
#if MLP_GELU_ERFF
  return 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f));
#else
  return gelu_fast(z);
#endif
}

// per-lane bias / saved-activation helpers: lane (p, h) owns rows 32 m + 8 g + 4 h + i of block m
__device__ __forceinline__ void stage_layer(const float* __restrict__ src, float* s_w, int n) {
  const float4* s4 = reinterpret_cast<const float4*>(src);
  float4* d4 = reinterpret_cast<float4*>(s_w);
  for (int i = threadIdx.x; i < n / 4; i += MLP_BLOCK) d4[i] = s4[i];
}

// ------------------------------------------------------------------ forward
// RESIDENT: the packed weights of ALL layers fit the workgroup's LDS (networks up to 64 wide:
// NerfHash's two MLPs are 72 and 48 KiB): they are staged once per workgroup and the layer loop has
// no barrier at all — the four waves run their tiles independently.  Otherwise (128-wide RGB /
// ColorSH: 152 KiB) one layer at a time is staged, two barriers per layer.
template <bool RESIDENT, int NB>
__global__ __launch_bounds__(MLP_BLOCK, 2) void mlp_fwd_kernel(
    vsa_mlp_plan plan, MlpGroups gp, long long packed_stride, long long hidden,
    const float* __restrict__ packed, const float* __restrict__ x, int x_stride,
    float* __restrict__ y, int y_stride, float* __restrict__ z_ws, float* __restrict__ a_ws, int io_aligned) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];
  __shared__ LayerMeta s_meta;
  const int grp = blockIdx.y;
  load_layer_meta(plan, gp, grp, s_meta);
  const int L = plan.n_layers;
  const int M = pick(gp.M, grp);
  {
    const long long r0 = pick(gp.row0, grp);
    packed += grp * packed_stride;
    x += r0 * x_stride;
    y += r0 * y_stride;
    if (z_ws) z_ws += r0 * hidden;
    if (a_ws) a_ws += r0 * hidden;
  }
  if (RESIDENT) {
    stage_layer(packed, s_w, meta_fwd(s_meta, L));
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 31, h = lane >> 5;
  const int ntiles = (M + MLP_TILE - 1) / MLP_TILE;
  const int per_round = gridDim.x * 4;
  const int rounds = (ntiles + per_round - 1) / per_round;
  // the input rows of round rd + 1 are requested before round rd's layers (up to 96 wide; at 128
  // the second set of 64 registers does not fit)
  constexpr bool XPF = MLP_XPF && NB <= 3;
  float xn[XPF ? NB : 1][16];
  auto load_x = [&](int rd, float dst[][16]) {
    const int tile = rd * per_round + blockIdx.x * 4 + wave;
    const long long pt = (long long)tile * MLP_TILE + p;
    const bool valid = tile < ntiles && pt < M;
    const int in = plan.dims[0];
    const float* row = x + (valid ? pt : 0) * (long long)x_stride;
    if (io_aligned & 1) {
      // rows of a stride that is a multiple of 4 floats: 16-byte loads of columns 32 b + 8 g + 4 h .. + 3, a group
      // beyond the row read at a clamped address and zeroed by a select — no per-element exec-mask branch (the
      // element-wise form below costs ~8 instructions per element: the kernel was bound by them, not by its MFMAs)
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int k0 = 32 * b + 8 * g + 4 * h;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (32 * b < in) {                // uniform
            const int kc = k0 < x_stride ? k0 : x_stride - 4;
            v = *reinterpret_cast<const float4*>(row + kc);
          }
          dst[b][4 * g] = (valid && k0 < in) ? v.x : 0.f;
          dst[b][4 * g + 1] = (valid && k0 + 1 < in) ? v.y : 0.f;
          dst[b][4 * g + 2] = (valid && k0 + 2 < in) ? v.z : 0.f;
          dst[b][4 * g + 3] = (valid && k0 + 3 < in) ? v.w : 0.f;
        }
      return;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = 32 * b + rho(r, h);
        dst[b][r] = (valid && k < in) ? row[k] : 0.f;
      }
  };
  if (XPF && rounds > 0) load_x(0, xn);
  for (int rd = 0; rd < rounds; ++rd) {
    const int tile = rd * per_round + blockIdx.x * 4 + wave;
    const long long pt = (long long)tile * MLP_TILE + p;
    const bool valid = tile < ntiles && pt < M;
    float act[NB][16];
    if (XPF) {
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) act[b][r] = xn[XPF ? b : 0][r];
      if (rd + 1 < rounds) load_x(rd + 1, xn);
    } else {
      load_x(rd, act);
    }
    long long z_off = 0;
    for (int l = 0; l < L; ++l) {
      const int in = meta_dim(s_meta, l), out = meta_dim(s_meta, l + 1);
      const int inb = blocks_of(in), outb = blocks_of(out);
      if (!RESIDENT) {
        __syncthreads();      // the previous layer's fragments have been read by every wave
        stage_layer(packed + meta_fwd(s_meta, l), s_w, inb * outb * 1024);
        __syncthreads();
      }
      const float* s_l = RESIDENT ? s_w + meta_fwd(s_meta, l) : s_w;
      f32x16 acc[NB];
#pragma unroll
      for (int m = 0; m < NB; ++m) {
        acc[m] = f32x16{0};
        if (m < outb) {
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            if (b < inb) {
              float wv[16];
              mlp_load_frags(s_l + ((m * inb + b) * 16) * 64, lane, wv);
#pragma unroll
              for (int s = 0; s < 16; ++s)
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[s], act[b][s], acc[m], 0, 0, 0);
            }
          }
        }
      }
      const bool last = l + 1 == L;
      const float* bias = meta_bias(s_meta, l);
#pragma unroll
      for (int m = 0; m < NB; ++m) {
        if (m < outb) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n0 = 32 * m + 8 * g + 4 * h;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
              v[i] = acc[m][4 * g + i] + ((bias && n0 + i < out) ? bias[n0 + i] : 0.f);
            if (last) {
              if (valid) {
                if (io_aligned & 2) {       // 16-byte rows: padding columns receive the zeros of the zero weight rows
                  if (n0 < y_stride) *reinterpret_cast<float4*>(y + pt * y_stride + n0) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                  for (int i = 0; i < 4; ++i)
                    if (n0 + i < out) y[pt * y_stride + n0 + i] = v[i];
                }
              }
            } else {
              // hidden widths are multiples of 32: aligned 16-byte rows
              if (z_ws && valid)
                *reinterpret_cast<float4*>(z_ws + z_off + pt * out + n0) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
              for (int i = 0; i < 4; ++i) act[m][4 * g + i] = gelu_f(v[i]);
              // the activations the weight gradients will contract with: stored here, where the
              // kernel has time to spare (it is bound by its MFMAs and the GELU), instead of by
              // mlp_dgrad, which is bound by its traffic (1.50 -> 1.21 ms without these stores)
              if (a_ws && valid)
                *reinterpret_cast<float4*>(a_ws + z_off + pt * out + n0) =
                    make_float4(act[m][4 * g], act[m][4 * g + 1], act[m][4 * g + 2], act[m][4 * g + 3]);
            }
          }
        } else if (!last) {
#pragma unroll
          for (int r = 0; r < 16; ++r) act[m][r] = 0.f;
        }
      }
      if (!last) z_off += (long long)M * out;
    }
  }
}

// ------------------------------------------------------------------ backward 1: data gradients
// dZ_l of every hidden layer is written to dz_ws ([point][width], same offsets as z_ws); dX
// [point][dims[0]] optional.  (A_l = GELU(z_l), the other operand of the weight gradients, was
// stored by the forward pass.)
template <bool RESIDENT, int NB>
__global__ __launch_bounds__(MLP_BLOCK, 2) void mlp_dgrad_kernel(
    vsa_mlp_plan plan, MlpGroups gp, long long packed_stride, long long hidden,
    const float* __restrict__ packed_t, const float* __restrict__ dy,
    int dy_stride, const float* __restrict__ z_ws, float* __restrict__ dz_ws,
    float* __restrict__ dx, int dx_stride) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];
  __shared__ LayerMeta s_meta;
  const int grp = blockIdx.y;
  load_layer_meta(plan, gp, grp, s_meta);
  const int L = plan.n_layers;
  const int M = pick(gp.M, grp);
  {
    const long long r0 = pick(gp.row0, grp);
    packed_t += grp * packed_stride;
    dy += r0 * dy_stride;
    z_ws += r0 * hidden;
    dz_ws += r0 * hidden;
    if (dx) dx += r0 * dx_stride;
  }
  if (RESIDENT) {
    stage_layer(packed_t, s_w, meta_fwd(s_meta, L));
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 31, h = lane >> 5;
  const int ntiles = (M + MLP_TILE - 1) / MLP_TILE;
  const int per_round = gridDim.x * 4;
  const int rounds = (ntiles + per_round - 1) / per_round;
  long long z_end = 0;                      // offset just past the last hidden layer's block
  for (int l = 0; l + 1 < L; ++l) z_end += (long long)M * meta_dim(s_meta, l + 1);
  for (int rd = 0; rd < rounds; ++rd) {
    const int tile = rd * per_round + blockIdx.x * 4 + wave;
    const long long pt = (long long)tile * MLP_TILE + p;
    const bool valid = tile < ntiles && pt < M;
    float dz[NB][16];                 // dZ of the layer being processed (rows of its OUTPUT)
    {
      const int out = meta_dim(s_meta, L);
      const float* row = dy + (valid ? pt : 0) * (long long)dy_stride;
#pragma unroll
      for (int m = 0; m < NB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = 32 * m + rho(r, h);
          dz[m][r] = (valid && n < out) ? row[n] : 0.f;
        }
    }
    long long z_off = z_end;
    for (int l = L - 1; l >= 0; --l) {
      const int in = meta_dim(s_meta, l), out = meta_dim(s_meta, l + 1);
      const int inb = blocks_of(in), outb = blocks_of(out);
      const int w_off = meta_fwd(s_meta, l);
      if (!RESIDENT) {
        __syncthreads();
        stage_layer(packed_t + w_off, s_w, inb * outb * 1024);
        __syncthreads();
      }
      // the saved pre-activations this layer's epilogue needs are requested BEFORE its MFMAs: the
      // loads do not depend on them, and with 8 waves per CU nothing else hides their latency
      // (not at 128 wide: dz, da and a third 64-register block do not fit the 256 registers)
      constexpr bool ZPF = MLP_ZPF && NB <= 3;
      float4 zq[ZPF ? NB : 1][4];
      if (l > 0) z_off -= (long long)M * in;
      if (ZPF && l > 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            zq[ZPF ? b : 0][g] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b < inb && valid)
              zq[ZPF ? b : 0][g] = *reinterpret_cast<const float4*>(z_ws + z_off + pt * in + 32 * b + 8 * g + 4 * h);
          }
      }
      f32x16 da[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        da[b] = f32x16{0};
        if (b < inb && (l > 0 || dx)) {
#pragma unroll
          for (int m = 0; m < NB; ++m) {
            if (m < outb) {
              float wv[16];
              mlp_load_frags(s_w + (RESIDENT ? w_off : 0) + ((b * outb + m) * 16) * 64, lane, wv);
#pragma unroll
              for (int s = 0; s < 16; ++s)
                da[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[s], dz[m][s], da[b], 0, 0, 0);
            }
          }
        }
      }
      if (l == 0) {
        if (dx && valid) {
#pragma unroll
          for (int b = 0; b < NB; ++b)
            if (b < inb)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int k = 32 * b + rho(r, h);
                if (k < in) dx[pt * dx_stride + k] = da[b][r];
              }
        }
      } else {
        // layer l-1's output (width `in`, a multiple of 32): dZ = dA * GELU'(z), A = GELU(z)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (b < inb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int k0 = 32 * b + 8 * g + 4 * h;
              const long long o = z_off + pt * in + k0;
              float4 z4 = zq[ZPF ? b : 0][g];
              if (!ZPF) {
                z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (valid) z4 = *reinterpret_cast<const float4*>(z_ws + o);
              }
              const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
              float d4[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
#if MLP_GELU_ERFF
                const float cdf = 0.5f * (1.0f + erff(zz[i] * 0.70710678118654752440f));
                const float pdf = 0.39894228040143267794f * __expf(-0.5f * zz[i] * zz[i]);
#else
                float cdf, pdf;
                gelu_cdf_pdf(zz[i], cdf, pdf);
#endif
                d4[i] = da[b][4 * g + i] * (cdf + zz[i] * pdf);
                dz[b][4 * g + i] = d4[i];
              }
              if (valid) *reinterpret_cast<float4*>(dz_ws + o) = make_float4(d4[0], d4[1], d4[2], d4[3]);
            }
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) dz[b][r] = 0.f;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------ backward 2: weight gradients
// One workgroup = one layer x one share of the points.  dW_l[n][k] = sum_p dZ_l[p][n] * A_{l-1}[p][k]:
// A fragment lane (i, kk) = dZ[p0 + kk][32 m + i], B fragment lane (j, kk) = A[p0 + kk][32 b + j].
// The (m, b) block pairs of a layer are dealt to the 4 waves (<= 4 pairs each at 128 x 128).
struct WgradLayers {
  int wg_begin[VSA_MLP_MAX_LAYERS + 1];     // workgroups [wg_begin[l], wg_begin[l+1]) serve layer l
  long long part_off[VSA_MLP_MAX_LAYERS + 1];   // offset (floats) of layer l's partial blocks
};

// Operands go through LDS in 32-point tiles: the workgroup loads the tile's dZ rows and A rows
// once (coalesced, 16 bytes per lane where the row stride allows) and its four waves read their
// MFMA fragments from there — lane (i, kk): tile[2 s + kk][32 m + i], 32 consecutive floats per
// half-wave, the row stride = 32 mod 64 floats so that the two halves use different banks.  The
// next tile is fetched into registers while the current one is multiplied.  (First version: every
// wave loaded its own fragments straight from L2, one dword per lane and MFMA: 11.6 ms for the
// 2.1 M samples of a background batch, 2.9 ms with an 8-deep unroll; the matrix time is ~0.5 ms.)
constexpr int WG_TP = 32;                               // points per tile
__host__ __device__ constexpr int wg_stride(int width_pad) { return width_pad + ((width_pad & 63) ? 0 : 32); }

// NB = widest layer of the network in 32-blocks (tile loader registers), Q = block pairs per
// wave (accumulators), PF = tiles requested ahead.  The loop is bound by the latency of the tile
// loads, not by the 16 MFMAs per pair between them: what counts is the number of bytes a CU keeps
// in flight, i.e. co-resident workgroups x PF.
template <int NB, int Q, int PF, int WGS>
__global__ __launch_bounds__(MLP_BLOCK, WGS) void mlp_wgrad_kernel(
    vsa_mlp_plan plan, WgradLayers wl, MlpGroups gp, long long hidden, long long partial_stride,
    const float* __restrict__ x, int x_stride,
    const float* __restrict__ dy, int dy_stride, const float* __restrict__ dz_ws,
    const float* __restrict__ a_ws, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float s_t[];
  const int grp = blockIdx.y;
  const int M = pick(gp.M, grp);
  {
    const long long r0 = pick(gp.row0, grp);
    x += r0 * x_stride;
    dy += r0 * dy_stride;
    dz_ws += r0 * hidden;
    a_ws += r0 * hidden;
    partial += grp * partial_stride;
  }
  int l = 0;
  while (l + 1 < plan.n_layers && (int)blockIdx.x >= wl.wg_begin[l + 1]) ++l;
  const int L = plan.n_layers;
  const int in = plan.dims[l], out = plan.dims[l + 1];
  const int inb = blocks_of(in), outb = blocks_of(out);
  const int nwg = wl.wg_begin[l + 1] - wl.wg_begin[l], wg = blockIdx.x - wl.wg_begin[l];
  long long zo = 0;
  for (int j = 0; j + 1 < l; ++j) zo += (long long)M * plan.dims[j + 1];     // A_{l-1} block (l >= 1)
  const float* aop = l == 0 ? x : a_ws + zo;
  const int a_stride = l == 0 ? x_stride : in;
  long long dzo = 0;
  for (int j = 0; j < l; ++j) dzo += (long long)M * plan.dims[j + 1];         // dZ_l block (l < L-1)
  const float* dop = l == L - 1 ? dy : dz_ws + dzo;
  const int d_stride = l == L - 1 ? dy_stride : out;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, kk = lane >> 5;
  const int npairs = inb * outb;
  // LDS tiles are laid out for the widest layer of the network (compile-time row stride: every
  // fragment read is one base register + an immediate offset, and the loader needs no division)
  constexpr int W = 32 * NB, S = wg_stride(W);
  float* s_a = s_t;                       // [WG_TP][S]  dZ rows
  float* s_b = s_t + WG_TP * S;           // [WG_TP][S]  A rows
  // this workgroup's points: a contiguous range of whole tiles
  const long long ntiles = (M + WG_TP - 1) / WG_TP;
  const long long t_per = (ntiles + nwg - 1) / nwg;
  const long long t_begin = wg * t_per, t_end = min(ntiles, t_begin + t_per);
  f32x16 acc[Q];
  float bsum[Q];
  int pm[Q], pb[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    acc[q] = f32x16{0};
    bsum[q] = 0.f;
    const int pair = wave + 4 * q;
    pm[q] = pair < npairs ? pair / inb : -1;
    pb[q] = pair < npairs ? pair % inb : 0;
  }
  // tile loader: 4 consecutive columns of the [WG_TP][W] tile per lane and request
  constexpr int LD_MAX = (WG_TP * W) / (4 * MLP_BLOCK);      // float4 per thread and operand
  float4 ra[PF][LD_MAX], rb[PF][LD_MAX];
  auto fetch = [&](long long tile, const float* src, int stride, int width, float4 r[LD_MAX]) {
    const long long p0 = tile * WG_TP;
    const bool vec_ok = ((stride & 3) == 0) && ((reinterpret_cast<size_t>(src) & 15) == 0);
#pragma unroll
    for (int k = 0; k < LD_MAX; ++k) {
      const int v = threadIdx.x + k * MLP_BLOCK;
      const int row = (v * 4) / W, c = (v * 4) % W;
      const long long pt = p0 + row;
      r[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pt < M && c < width) {
        const float* g = src + pt * stride + c;
        if (vec_ok && c + 3 < width) {
          r[k] = *reinterpret_cast<const float4*>(g);
        } else {
          r[k].x = g[0];
          if (c + 1 < width) r[k].y = g[1];
          if (c + 2 < width) r[k].z = g[2];
          if (c + 3 < width) r[k].w = g[3];
        }
      }
    }
  };
  auto stash = [&](float* dst, const float4 r[LD_MAX]) {
#pragma unroll
    for (int k = 0; k < LD_MAX; ++k) {
      const int v = threadIdx.x + k * MLP_BLOCK;
      const int row = (v * 4) / W, c = (v * 4) % W;
      *reinterpret_cast<float4*>(dst + row * S + c) = r[k];
    }
  };
#pragma unroll
  for (int f = 0; f < PF; ++f)
    if (t_begin + f < t_end) {
      fetch(t_begin + f, dop, d_stride, out, ra[f]);
      fetch(t_begin + f, aop, a_stride, in, rb[f]);
    }
  for (long long t0 = t_begin; t0 < t_end; t0 += PF) {
#pragma unroll
    for (int f = 0; f < PF; ++f) {
      const long long t = t0 + f;
      if (t < t_end) {                       // uniform over the workgroup
        __syncthreads();                     // the previous tile's fragments have been read
        stash(s_a, ra[f]);
        stash(s_b, rb[f]);
        __syncthreads();
        if (t + PF < t_end) {
          fetch(t + PF, dop, d_stride, out, ra[f]);
          fetch(t + PF, aop, a_stride, in, rb[f]);
        }
#pragma unroll
        for (int st = 0; st < WG_TP / 2; ++st) {
          const float* ar = s_a + (2 * st + kk) * S + i;
          const float* br = s_b + (2 * st + kk) * S + i;
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            if (pm[q] >= 0) {
              const float av = ar[32 * pm[q]];
              acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, br[32 * pb[q]], acc[q], 0, 0, 0);
              bsum[q] += av;
            }
          }
        }
      }
    }
  }
  // partial[l][wg][out_pad][in_pad] then [out_pad] bias sums
  const int in_pad = 32 * inb, out_pad = 32 * outb;
  float* part = partial + wl.part_off[l] + (long long)wg * (out_pad * in_pad + out_pad);
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    if (pm[q] >= 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * pm[q] + (r & 3) + 8 * (r >> 2) + 4 * kk, col = 32 * pb[q] + i;
        part[row * in_pad + col] = acc[q][r];
      }
      if (pb[q] == 0) {     // the pair with b == 0 of every m also owns that block's bias sums
        const float other = __shfl_xor(bsum[q], 32, 64);
        if (kk == 0) part[out_pad * in_pad + 32 * pm[q] + i] = bsum[q] + other;
      }
    }
  }
}

// dW_l / db_l = sum over the layer's workgroups of their partial blocks
__global__ void mlp_reduce_kernel(vsa_mlp_plan plan, WgradLayers wl, long long partial_stride,
                                  const float* __restrict__ partial, MlpGroupGrads grads) {
  const int l = blockIdx.y, g = blockIdx.z;
  partial += g * partial_stride;
  const int in = plan.dims[l], out = plan.dims[l + 1];
  const int in_pad = 32 * blocks_of(in), out_pad = 32 * blocks_of(out);
  const int nwg = wl.wg_begin[l + 1] - wl.wg_begin[l];
  const int per = out_pad * in_pad + out_pad;
  float* dw = nullptr;
  float* db = nullptr;
#pragma unroll
  for (int k = 0; k < VSA_MLP_MAX_LAYERS; ++k)
    if (l == k) {
      dw = pick2(grads.dw, g, k);
      db = pick2(grads.db, g, k);
    }
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < per; idx += gridDim.x * blockDim.x) {
    // four independent running sums: the loads of a thread's ~250 partials are then in flight four
    // at a time instead of one dependent add after the other (fixed order: still deterministic)
    const float* pp = partial + wl.part_off[l] + idx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < nwg; k += 4) {
      s0 += pp[(long long)k * per];
      s1 += pp[(long long)(k + 1) * per];
      s2 += pp[(long long)(k + 2) * per];
      s3 += pp[(long long)(k + 3) * per];
    }
    for (; k < nwg; ++k) s0 += pp[(long long)k * per];
    const float s = (s0 + s1) + (s2 + s3);
    if (idx < out_pad * in_pad) {
      const int row = idx / in_pad, col = idx - row * in_pad;
      if (row < out && col < in && dw) {
        float* d = dw + (long long)row * in + col;
        *d = grads.accumulate ? *d + s : s;
      }
    } else {
      const int n = idx - out_pad * in_pad;
      if (n < out && db) db[n] = grads.accumulate ? db[n] + s : s;
    }
  }
}

#include "mlp_f32_fused.h"

int plan_ok(const vsa_mlp_plan* p) {
  if (!p) return VSA_ERR_ARG;
  if (p->n_layers < 1 || p->n_layers > VSA_MLP_MAX_LAYERS) return VSA_ERR_ARG;
  for (int l = 0; l <= p->n_layers; ++l)
    if (p->dims[l] < 1 || p->dims[l] > 32 * MLP_MAXB) return VSA_ERR_UNSUPPORTED;
  for (int l = 1; l < p->n_layers; ++l)
    if (p->dims[l] % 32) return VSA_ERR_UNSUPPORTED;     // hidden widths: multiples of 32
  for (int l = 0; l < p->n_layers; ++l)
    if (!p->w[l]) return VSA_ERR_ARG;
  return VSA_OK;
}

size_t max_layer_bytes(const vsa_mlp_plan& p) {
  size_t mx = 0;
  for (int l = 0; l < p.n_layers; ++l)
  {
    const size_t b = (size_t)blocks_of(p.dims[l]) * blocks_of(p.dims[l + 1]) * 1024 * sizeof(float);
    if (b > mx) mx = b;
  }
  return mx;
}

// Instance of the weight-gradient kernel for a network: NB from its widest layer, Q from the layer
// with the most 32 x 32 block pairs.  Networks up to 96 wide (NerfHash) take the small instances:
// 32 / 48 accumulator and loader registers instead of 64 / 64 leave room for a second tile in
// flight and a fourth co-resident workgroup.
#ifndef MLP_WG_SMALL_WGS
#define MLP_WG_SMALL_WGS 4
#endif
#ifndef MLP_WG_SMALL_PF
#define MLP_WG_SMALL_PF 1
#endif
#ifndef MLP_WG_COST
#define MLP_WG_COST 1
#endif
struct WgradShape {
  int nb, q, wgs;
};
WgradShape wgrad_shape(const vsa_mlp_plan& p) {
  int wmax = 1, pmax = 1;
  for (int l = 0; l <= p.n_layers; ++l) wmax = blocks_of(p.dims[l]) > wmax ? blocks_of(p.dims[l]) : wmax;
  for (int l = 0; l < p.n_layers; ++l) {
    const int pr = blocks_of(p.dims[l]) * blocks_of(p.dims[l + 1]);
    pmax = pr > pmax ? pr : pmax;
  }
  if (wmax <= 2) return {2, 1, MLP_WG_SMALL_WGS};
  if (wmax == 3 && pmax <= 8) return {3, 2, MLP_WG_SMALL_WGS};
  return {4, 4, 3};
}

// workgroups of the weight-gradient kernel: as many as are co-resident for large batches, fewer
// for small ones (every workgroup writes a full set of partial blocks that mlp_reduce then has to
// read: 512 workgroups on a 10 k-point batch made the reduction the most expensive kernel of the
// step)
int wgrad_total_wgs(const vsa_mlp_plan& p, long long nr_points, int nr_cus, int nr_groups = 1) {
  long long n = nr_points / 128;
  long long cap = (long long)wgrad_shape(p).wgs * nr_cus / nr_groups;
  if (n > cap) n = cap;
  if (n < p.n_layers) n = p.n_layers;
  return (int)n;
}

constexpr size_t MLP_RESIDENT_BYTES = 78 * 1024;     // two workgroups per CU

int max_blocks(const vsa_mlp_plan& p) {
  int wmax = 2;
  for (int l = 0; l <= p.n_layers; ++l) wmax = blocks_of(p.dims[l]) > wmax ? blocks_of(p.dims[l]) : wmax;
  return wmax;
}

template <bool RESIDENT, int NB>
int set_lds_attr_of() {
  const int bytes = RESIDENT ? (int)MLP_RESIDENT_BYTES : 64 * 1024;
  VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwd_kernel<RESIDENT, NB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_dgrad_kernel<RESIDENT, NB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return VSA_OK;
}

int set_lds_attrs() {
  static bool done = false;
  if (!done) {
    int rc = set_lds_attr_of<true, 2>();
    if (!rc) rc = set_lds_attr_of<true, 3>();
    if (!rc) rc = set_lds_attr_of<true, 4>();
    if (!rc) rc = set_lds_attr_of<false, 2>();
    if (!rc) rc = set_lds_attr_of<false, 3>();
    if (!rc) rc = set_lds_attr_of<false, 4>();
    if (rc) return rc;
    done = true;
  }
  return VSA_OK;
}

// kernel instance by (weights resident in LDS, widest layer in 32-blocks)
#define VSA_MLP_DISPATCH(KERNEL, RES, NB, ...)                                            \
  do {                                                                                    \
    if (RES) {                                                                            \
      if ((NB) == 2) hipLaunchKernelGGL((KERNEL<true, 2>), __VA_ARGS__);                  \
      else if ((NB) == 3) hipLaunchKernelGGL((KERNEL<true, 3>), __VA_ARGS__);             \
      else hipLaunchKernelGGL((KERNEL<true, 4>), __VA_ARGS__);                            \
    } else {                                                                              \
      if ((NB) == 2) hipLaunchKernelGGL((KERNEL<false, 2>), __VA_ARGS__);                 \
      else if ((NB) == 3) hipLaunchKernelGGL((KERNEL<false, 3>), __VA_ARGS__);            \
      else hipLaunchKernelGGL((KERNEL<false, 4>), __VA_ARGS__);                           \
    }                                                                                     \
  } while (0)

WgradLayers wgrad_layers(const vsa_mlp_plan& p, int total_wgs) {
  // workgroups per layer in proportion to the layer's MFMA count (>= 1 each)
  WgradLayers wl;
  int cost[VSA_MLP_MAX_LAYERS], sum = 0;
  for (int l = 0; l < p.n_layers; ++l) {
#if MLP_WG_COST == 0
    // a wave issues ceil(pairs / 4) MFMAs per point pair
    cost[l] = (blocks_of(p.dims[l]) * blocks_of(p.dims[l + 1]) + 3) / 4;
#else
    // the loop waits for its tile loads, not for its MFMAs: bytes per point
    cost[l] = blocks_of(p.dims[l]) + blocks_of(p.dims[l + 1]);
#endif
    sum += cost[l];
  }
  int begin = 0;
  long long off = 0;
  for (int l = 0; l < p.n_layers; ++l) {
    int n = (int)((long long)total_wgs * cost[l] / sum);
    if (n < 1) n = 1;
    wl.wg_begin[l] = begin;
    wl.part_off[l] = off;
    begin += n;
    const long long ip = 32 * blocks_of(p.dims[l]), op = 32 * blocks_of(p.dims[l + 1]);
    off += (long long)n * (op * ip + op);
  }
  wl.wg_begin[p.n_layers] = begin;
  wl.part_off[p.n_layers] = off;
  return wl;
}

// The fused backward (mlp_f32_fused.h) serves a network when its transposed weights stay resident in LDS beside the
// staging rows (<= 160 KiB), its layers are at most 96 wide and its block pairs fit five accumulators per wave.
// MLP_BWD_FUSED=0 in the environment: the two-kernel backward of rounds 1-5 (A/B switch).
constexpr size_t MLP_FUSED_LDS_MAX = 160 * 1024 - 512;
// a row-major matrix whose rows can be read / written as 16-byte groups
bool rows_aligned(const void* ptr, int stride) { return stride % 4 == 0 && (reinterpret_cast<size_t>(ptr) & 15) == 0; }

bool bwd_is_fused(const vsa_mlp_plan& p) {
  static const bool off = [] { const char* e = getenv("MLP_BWD_FUSED"); return e && e[0] == '0'; }();
  if (off || p.n_layers < 2) return false;
  const size_t all = (size_t)pack_offsets(p).fwd[p.n_layers] * sizeof(float);
  for (int l = 1; l < p.n_layers; ++l)
    if (p.dims[l] > 64) return false;          // hidden layers: two 32-blocks (the z registers of the data waves)
  return all <= MLP_RESIDENT_BYTES && max_blocks(p) <= 3 && fb_pairs(p) <= 20 && fb_lds_bytes(p) <= MLP_FUSED_LDS_MAX;
}

// every workgroup holds partial blocks of every layer: layer l's nwg = total (only differences of wg_begin are read)
WgradLayers fused_layers(const vsa_mlp_plan& p, int total_wgs) {
  WgradLayers wl;
  long long off = 0;
  for (int l = 0; l < p.n_layers; ++l) {
    wl.wg_begin[l] = l * total_wgs;
    wl.part_off[l] = off;
    const long long ip = 32 * blocks_of(p.dims[l]), op = 32 * blocks_of(p.dims[l + 1]);
    off += (long long)total_wgs * (op * ip + op);
  }
  wl.wg_begin[p.n_layers] = p.n_layers * total_wgs;
  wl.part_off[p.n_layers] = off;
  return wl;
}

int set_fused_lds_attrs() {
  static bool done = false;
  if (!done) {
    const int bytes = (int)MLP_FUSED_LDS_MAX;
    VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_fused_kernel<2, 2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_fused_kernel<2, 2, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_fused_kernel<3, 2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_fused_kernel<3, 2, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done = true;
  }
  return VSA_OK;
}

}  // namespace

extern "C" int vsa_mlp_bwd_needs_act(const vsa_mlp_plan* plan, int x_stride, int dx_stride) {
  const int rc = plan_ok(plan);
  if (rc) return rc < 0 ? rc : -rc;
  return bwd_is_fused(*plan) && x_stride % 4 == 0 && (dx_stride == 0 || dx_stride % 4 == 0) ? 0 : 1;
}

extern "C" int vsa_mlp_workspace(const vsa_mlp_plan* plan, long long nr_points,
                                 long long* packed_floats, long long* act_floats,
                                 long long* partial_floats) {
  int rc = plan_ok(plan);
  if (rc) return rc;
  if (nr_points < 0) return VSA_ERR_ARG;
  const PackOffsets off = pack_offsets(*plan);
  if (packed_floats) *packed_floats = off.fwd[plan->n_layers];
  long long hidden = 0;
  for (int l = 1; l < plan->n_layers; ++l) hidden += plan->dims[l];
  if (act_floats) *act_floats = hidden * nr_points;
  int nr_cus = 0;
  rc = vsa_cu_count(&nr_cus);
  if (rc) return rc;
  if (partial_floats) {
    long long pf = wgrad_layers(*plan, wgrad_total_wgs(*plan, nr_points, nr_cus)).part_off[plan->n_layers];
    if (bwd_is_fused(*plan)) {
      const long long ff = fused_layers(*plan, nr_cus).part_off[plan->n_layers];
      pf = ff > pf ? ff : pf;
    }
    *partial_floats = pf;
  }
  return VSA_OK;
}

namespace {

// shared-architecture check + the group descriptor of a launch
int make_groups(const vsa_mlp_plan* plans, int nr_groups, const int* nr_points, MlpGroups* gp,
                long long* total_rows, int* max_rows) {
  if (!plans || !nr_points || nr_groups < 1 || nr_groups > MLP_MAX_GROUPS) return VSA_ERR_ARG;
  int rc = plan_ok(&plans[0]);
  if (rc) return rc;
  long long row = 0;
  int mx = 0;
  for (int g = 0; g < MLP_MAX_GROUPS; ++g) {
    const int src = g < nr_groups ? g : 0;
    if (g < nr_groups) {
      rc = plan_ok(&plans[g]);
      if (rc) return rc;
      if (plans[g].n_layers != plans[0].n_layers) return VSA_ERR_ARG;
      for (int l = 0; l <= plans[0].n_layers; ++l)
        if (plans[g].dims[l] != plans[0].dims[l]) return VSA_ERR_ARG;
      if (nr_points[g] < 0) return VSA_ERR_ARG;
    }
    gp->M[g] = g < nr_groups ? nr_points[g] : 0;
    gp->row0[g] = row;
    for (int l = 0; l < VSA_MLP_MAX_LAYERS; ++l) {
      gp->w[g][l] = l < plans[src].n_layers ? plans[src].w[l] : nullptr;
      gp->b[g][l] = l < plans[src].n_layers ? plans[src].b[l] : nullptr;
    }
    if (g < nr_groups) {
      row += nr_points[g];
      mx = nr_points[g] > mx ? nr_points[g] : mx;
    }
  }
  *total_rows = row;
  *max_rows = mx;
  return VSA_OK;
}

long long hidden_width(const vsa_mlp_plan& p) {
  long long h = 0;
  for (int l = 1; l < p.n_layers; ++l) h += p.dims[l];
  return h;
}

}  // namespace

extern "C" int vsa_mlp_fwd_grouped(const vsa_mlp_plan* plans, int nr_groups, const int* nr_points,
                                   const float* x, int x_stride, float* y, int y_stride, float* z_ws,
                                   float* a_ws, float* packed_ws, float* packed_bwd_ws, void* stream) {
  MlpGroups gp;
  long long rows = 0;
  int mx = 0;
  int rc = make_groups(plans, nr_groups, nr_points, &gp, &rows, &mx);
  if (rc) return rc;
  const vsa_mlp_plan* plan = &plans[0];
  if (x_stride < plan->dims[0] || y_stride < plan->dims[plan->n_layers]) return VSA_ERR_ARG;
  if (rows == 0) return VSA_OK;
  if (!x || !y || !packed_ws || (a_ws && !z_ws)) return VSA_ERR_ARG;      // (a_ws: only the two-kernel backward reads it)
  hipStream_t st = (hipStream_t)stream;
  const long long packed_stride = pack_offsets(*plan).fwd[plan->n_layers];
  // (packed_bwd_ws: the transposed fragment order the backward pass needs, written by the same
  //  launch when the caller will run one — vsa_mlp_bwd_grouped(packed_ready = 1) then skips its own)
  hipLaunchKernelGGL(mlp_pack_kernel, dim3(16, plan->n_layers, nr_groups), dim3(256), 0, st, *plan, gp,
                     packed_stride, packed_ws, packed_bwd_ws);
  int nr_cus = 0;
  rc = vsa_cu_count(&nr_cus);
  if (rc) return rc;
  rc = set_lds_attrs();
  if (rc) return rc;
  const size_t all = (size_t)packed_stride * sizeof(float);
  const bool resident = all <= MLP_RESIDENT_BYTES;
  const size_t lds = resident ? all : max_layer_bytes(*plan);
  const int ntiles = vsa_div_up(mx, MLP_TILE);
  int grid = vsa_div_up(ntiles, 4);
  const int cap = vsa_div_up(2 * nr_cus, nr_groups);          // two workgroups per CU over all groups
  if (grid > cap) grid = cap;
  VSA_MLP_DISPATCH(mlp_fwd_kernel, resident, max_blocks(*plan), dim3(grid, nr_groups), dim3(MLP_BLOCK), lds, st,
                   *plan, gp, packed_stride, hidden_width(*plan), packed_ws, x, x_stride, y, y_stride, z_ws, a_ws,
                   (rows_aligned(x, x_stride) ? 1 : 0) | (rows_aligned(y, y_stride) ? 2 : 0));
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_mlp_fwd(const vsa_mlp_plan* plan, const float* x, int x_stride, int nr_points,
                           float* y, int y_stride, float* z_ws, float* a_ws, float* packed_ws, void* stream) {
  return vsa_mlp_fwd_grouped(plan, 1, &nr_points, x, x_stride, y, y_stride, z_ws, a_ws, packed_ws, nullptr, stream);
}

extern "C" int vsa_mlp_bwd_grouped(const vsa_mlp_plan* plans, int nr_groups, const int* nr_points,
                                   const float* x, int x_stride, const float* dy, int dy_stride,
                                   const float* z_ws, float* dz_ws, const float* a_ws, float* packed_ws,
                                   int packed_ready, float* partial_ws, float* dx, int dx_stride,
                                   const vsa_mlp_grads* grads, void* stream) {
  MlpGroups gp;
  long long rows = 0;
  int mx = 0;
  int rc = make_groups(plans, nr_groups, nr_points, &gp, &rows, &mx);
  if (rc) return rc;
  const vsa_mlp_plan* plan = &plans[0];
  const int L = plan->n_layers;
  if (x_stride < plan->dims[0] || dy_stride < plan->dims[L] || (dx && dx_stride < plan->dims[0]))
    return VSA_ERR_ARG;
  if (!grads) return VSA_ERR_ARG;
  if (rows == 0) return VSA_OK;
  const bool fused = bwd_is_fused(*plan) && rows_aligned(x, x_stride) && rows_aligned(dy, dy_stride) && (!dx || rows_aligned(dx, dx_stride));
  if (!x || !dy || !packed_ws || !partial_ws || (L > 1 && (!z_ws || (!fused && (!dz_ws || !a_ws)))))
    return VSA_ERR_ARG;
  MlpGroupGrads gg;
  gg.accumulate = grads[0].accumulate;
  for (int g = 0; g < MLP_MAX_GROUPS; ++g)
    for (int l = 0; l < VSA_MLP_MAX_LAYERS; ++l) {
      gg.dw[g][l] = g < nr_groups ? grads[g].dw[l] : nullptr;
      gg.db[g][l] = g < nr_groups ? grads[g].db[l] : nullptr;
    }
  for (int g = 1; g < nr_groups; ++g)
    if ((grads[g].accumulate != 0) != (gg.accumulate != 0)) return VSA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long long packed_stride = pack_offsets(*plan).fwd[L];
  const long long hidden = hidden_width(*plan);
  if (!packed_ready)
    hipLaunchKernelGGL(mlp_pack_kernel, dim3(16, L, nr_groups), dim3(256), 0, st, *plan, gp, packed_stride,
                       (float*)nullptr, packed_ws);
  int nr_cus = 0;
  rc = vsa_cu_count(&nr_cus);
  if (rc) return rc;
  rc = set_lds_attrs();
  if (rc) return rc;
  const size_t all = (size_t)packed_stride * sizeof(float);
  const bool resident = all <= MLP_RESIDENT_BYTES;
  const size_t lds = resident ? all : max_layer_bytes(*plan);
  const int ntiles = vsa_div_up(mx, MLP_TILE);
  int grid = vsa_div_up(ntiles, 4);
  if (fused) {
    // ONE persistent launch: data gradients, weight gradients and bias sums (mlp_f32_fused.h); one workgroup per CU
    const int fcap = nr_cus / nr_groups > 0 ? nr_cus / nr_groups : 1;
    if (grid > fcap) grid = fcap;
    const WgradLayers wl = fused_layers(*plan, grid);
    const long long partial_stride = wl.part_off[L];
    const size_t flds = fb_lds_bytes(*plan);
    rc = set_fused_lds_attrs();
    if (rc) return rc;
    const int nb = max_blocks(*plan), q = (fb_pairs(*plan) + 3) / 4;
#define VSA_FUSED_LAUNCH(NB_, NH_, Q_)                                                                                  \
  hipLaunchKernelGGL((mlp_bwd_fused_kernel<NB_, NH_, Q_>), dim3(grid, nr_groups), dim3(FB_BLOCK), flds, st, *plan, wl, gp, \
                     packed_stride, hidden, partial_stride, fb_rows_dz(*plan), packed_ws, x, x_stride, dy, dy_stride, \
                     z_ws, dx, dx_stride, partial_ws)
    if (nb == 2 && q <= 3) VSA_FUSED_LAUNCH(2, 2, 3);
    else if (nb == 2) VSA_FUSED_LAUNCH(2, 2, 5);
    else if (q <= 3) VSA_FUSED_LAUNCH(3, 2, 3);
    else VSA_FUSED_LAUNCH(3, 2, 5);
#undef VSA_FUSED_LAUNCH
    hipLaunchKernelGGL(mlp_reduce_kernel, dim3(68, L, nr_groups), dim3(256), 0, st, *plan, wl, partial_stride,
                       partial_ws, gg);
    VSA_RETURN_LAUNCH_STATUS();
  }
  const int cap = vsa_div_up(2 * nr_cus, nr_groups);
  if (grid > cap) grid = cap;
  if (L > 1 || dx)
    VSA_MLP_DISPATCH(mlp_dgrad_kernel, resident, max_blocks(*plan), dim3(grid, nr_groups), dim3(MLP_BLOCK), lds, st,
                     *plan, gp, packed_stride, hidden, packed_ws, dy, dy_stride, z_ws, dz_ws, dx, dx_stride);
  // the weight-gradient workgroups of ONE group (every group gets the same split; the co-resident
  // budget is shared between the groups)
  const WgradLayers wl = wgrad_layers(*plan, wgrad_total_wgs(*plan, mx, nr_cus, nr_groups));
  const long long partial_stride = wl.part_off[L];
  const WgradShape ws = wgrad_shape(*plan);
  const size_t wg_lds = (size_t)WG_TP * (wg_stride(32 * ws.nb) * 2) * sizeof(float);   // 24 KiB (64 / 96 wide) .. 40 KiB (128)
#define VSA_WGRAD_LAUNCH(NB_, Q_, PF_, WGS_)                                                              \
  hipLaunchKernelGGL((mlp_wgrad_kernel<NB_, Q_, PF_, WGS_>), dim3(wl.wg_begin[L], nr_groups), dim3(MLP_BLOCK), \
                     wg_lds, st, *plan, wl, gp, hidden, partial_stride, x, x_stride, dy, dy_stride, dz_ws, a_ws, \
                     partial_ws)
  if (ws.nb == 2) VSA_WGRAD_LAUNCH(2, 1, MLP_WG_SMALL_PF, MLP_WG_SMALL_WGS);
  else if (ws.nb == 3) VSA_WGRAD_LAUNCH(3, 2, MLP_WG_SMALL_PF, MLP_WG_SMALL_WGS);
  else VSA_WGRAD_LAUNCH(4, 4, 1, 3);
#undef VSA_WGRAD_LAUNCH
  hipLaunchKernelGGL(mlp_reduce_kernel, dim3(68, L, nr_groups), dim3(256), 0, st, *plan, wl, partial_stride,
                     partial_ws, gg);   // 68 x 256 >= one thread per element of a 128 x 128 (+bias) layer
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_mlp_bwd(const vsa_mlp_plan* plan, const float* x, int x_stride, int nr_points,
                           const float* dy, int dy_stride, const float* z_ws, float* dz_ws,
                           const float* a_ws, float* packed_ws, float* partial_ws, float* dx,
                           int dx_stride, const vsa_mlp_grads* grads, void* stream) {
  return vsa_mlp_bwd_grouped(plan, 1, &nr_points, x, x_stride, dy, dy_stride, z_ws, dz_ws, a_ws, packed_ws, 0,
                             partial_ws, dx, dx_stride, grads, stream);
}
