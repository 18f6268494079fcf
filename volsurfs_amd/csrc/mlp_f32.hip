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

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MLP_MAXB = 4;               // layer widths up to 128 (4 blocks of 32)
constexpr int MLP_BLOCK = 256;            // 4 waves = 4 tiles of 32 points
constexpr int MLP_TILE = 32;

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

// packed_fwd[l][((m * inb + b) * 16 + s) * 64 + lane] = W_l[32 m + (lane & 31)][32 b + rho(s, lane >> 5)]
// packed_bwd[l][((b * outb + m) * 16 + s) * 64 + lane] = W_l[32 m + rho(s, lane >> 5)][32 b + (lane & 31)]
__global__ void mlp_pack_kernel(vsa_mlp_plan plan, float* __restrict__ packed_fwd,
                                float* __restrict__ packed_bwd) {
  const PackOffsets off = pack_offsets(plan);
  const int l = blockIdx.y;
  const int in = plan.dims[l], out = plan.dims[l + 1];
  const int inb = blocks_of(in), outb = blocks_of(out);
  const int n = inb * outb * 16 * 64;
  const float* W = plan.w[l];
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63, s = (idx >> 6) & 15, blk = idx >> 10;
    const int i = lane & 31, h = lane >> 5;
    if (packed_fwd) {
      const int m = blk / inb, b = blk - m * inb;
      const int row = 32 * m + i, col = 32 * b + rho(s, h);
      packed_fwd[off.fwd[l] + idx] = (row < out && col < in) ? W[(long long)row * in + col] : 0.f;
    }
    if (packed_bwd) {
      const int b = blk / outb, m = blk - b * outb;
      const int row = 32 * m + rho(s, h), col = 32 * b + i;
      packed_bwd[off.fwd[l] + idx] = (row < out && col < in) ? W[(long long)row * in + col] : 0.f;
    }
  }
}

__device__ __forceinline__ float gelu_f(float z) {          // exact GELU (torch.nn.GELU default)
  return 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f));
}

// per-lane bias / saved-activation helpers: lane (p, h) owns rows 32 m + 8 g + 4 h + i of block m
__device__ __forceinline__ void stage_layer(const float* __restrict__ src, float* s_w, int n) {
  const float4* s4 = reinterpret_cast<const float4*>(src);
  float4* d4 = reinterpret_cast<float4*>(s_w);
  for (int i = threadIdx.x; i < n / 4; i += MLP_BLOCK) d4[i] = s4[i];
}

// ------------------------------------------------------------------ forward
__global__ __launch_bounds__(MLP_BLOCK, 2) void mlp_fwd_kernel(
    vsa_mlp_plan plan, const float* __restrict__ packed, const float* __restrict__ x, int x_stride,
    int M, float* __restrict__ y, int y_stride, float* __restrict__ z_ws) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];
  const PackOffsets off = pack_offsets(plan);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 31, h = lane >> 5;
  const int ntiles = (M + MLP_TILE - 1) / MLP_TILE;
  const int per_round = gridDim.x * 4;
  const int rounds = (ntiles + per_round - 1) / per_round;
  for (int rd = 0; rd < rounds; ++rd) {
    const int tile = rd * per_round + blockIdx.x * 4 + wave;
    const long long pt = (long long)tile * MLP_TILE + p;
    const bool valid = tile < ntiles && pt < M;
    float act[MLP_MAXB][16];
    {
      const int in = plan.dims[0];
      const float* row = x + (valid ? pt : 0) * (long long)x_stride;
#pragma unroll
      for (int b = 0; b < MLP_MAXB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = 32 * b + rho(r, h);
          act[b][r] = (valid && k < in) ? row[k] : 0.f;
        }
    }
    long long z_off = 0;
    for (int l = 0; l < plan.n_layers; ++l) {
      const int in = plan.dims[l], out = plan.dims[l + 1];
      const int inb = blocks_of(in), outb = blocks_of(out);
      __syncthreads();      // the previous layer's fragments have been read by every wave
      stage_layer(packed + off.fwd[l], s_w, inb * outb * 1024);
      __syncthreads();
      f32x16 acc[MLP_MAXB];
#pragma unroll
      for (int m = 0; m < MLP_MAXB; ++m) {
        acc[m] = f32x16{0};
        if (m < outb) {
#pragma unroll
          for (int b = 0; b < MLP_MAXB; ++b) {
            if (b < inb) {
              const float* frag = s_w + ((m * inb + b) * 16) * 64 + lane;
#pragma unroll
              for (int s = 0; s < 16; ++s)
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(frag[s * 64], act[b][s], acc[m], 0, 0, 0);
            }
          }
        }
      }
      const bool last = l + 1 == plan.n_layers;
      const float* bias = plan.b[l];
#pragma unroll
      for (int m = 0; m < MLP_MAXB; ++m) {
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
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  if (n0 + i < out) y[pt * y_stride + n0 + i] = v[i];
              }
            } else {
              // hidden widths are multiples of 32: aligned 16-byte rows
              if (z_ws && valid)
                *reinterpret_cast<float4*>(z_ws + z_off + pt * out + n0) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
              for (int i = 0; i < 4; ++i) act[m][4 * g + i] = gelu_f(v[i]);
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
// dZ_l and A_l = GELU(z_l) of every hidden layer are written to dz_ws / a_ws ([point][width],
// same offsets as z_ws); dX [point][dims[0]] optional.
__global__ __launch_bounds__(MLP_BLOCK, 2) void mlp_dgrad_kernel(
    vsa_mlp_plan plan, const float* __restrict__ packed_t, const float* __restrict__ dy,
    int dy_stride, int M, const float* __restrict__ z_ws, float* __restrict__ dz_ws,
    float* __restrict__ a_ws, float* __restrict__ dx, int dx_stride) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];
  const PackOffsets off = pack_offsets(plan);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 31, h = lane >> 5;
  const int ntiles = (M + MLP_TILE - 1) / MLP_TILE;
  const int per_round = gridDim.x * 4;
  const int rounds = (ntiles + per_round - 1) / per_round;
  const int L = plan.n_layers;
  long long z_end = 0;                      // offset just past the last hidden layer's block
  for (int l = 0; l + 1 < L; ++l) z_end += (long long)M * plan.dims[l + 1];
  for (int rd = 0; rd < rounds; ++rd) {
    const int tile = rd * per_round + blockIdx.x * 4 + wave;
    const long long pt = (long long)tile * MLP_TILE + p;
    const bool valid = tile < ntiles && pt < M;
    float dz[MLP_MAXB][16];                 // dZ of the layer being processed (rows of its OUTPUT)
    {
      const int out = plan.dims[L];
      const float* row = dy + (valid ? pt : 0) * (long long)dy_stride;
#pragma unroll
      for (int m = 0; m < MLP_MAXB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = 32 * m + rho(r, h);
          dz[m][r] = (valid && n < out) ? row[n] : 0.f;
        }
    }
    long long z_off = z_end;
    for (int l = L - 1; l >= 0; --l) {
      const int in = plan.dims[l], out = plan.dims[l + 1];
      const int inb = blocks_of(in), outb = blocks_of(out);
      __syncthreads();
      stage_layer(packed_t + off.fwd[l], s_w, inb * outb * 1024);
      __syncthreads();
      f32x16 da[MLP_MAXB];
#pragma unroll
      for (int b = 0; b < MLP_MAXB; ++b) {
        da[b] = f32x16{0};
        if (b < inb && (l > 0 || dx)) {
#pragma unroll
          for (int m = 0; m < MLP_MAXB; ++m) {
            if (m < outb) {
              const float* frag = s_w + ((b * outb + m) * 16) * 64 + lane;
#pragma unroll
              for (int s = 0; s < 16; ++s)
                da[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(frag[s * 64], dz[m][s], da[b], 0, 0, 0);
            }
          }
        }
      }
      if (l == 0) {
        if (dx && valid) {
#pragma unroll
          for (int b = 0; b < MLP_MAXB; ++b)
            if (b < inb)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int k = 32 * b + rho(r, h);
                if (k < in) dx[pt * dx_stride + k] = da[b][r];
              }
        }
      } else {
        // layer l-1's output (width `in`, a multiple of 32): dZ = dA * GELU'(z), A = GELU(z)
        z_off -= (long long)M * in;
#pragma unroll
        for (int b = 0; b < MLP_MAXB; ++b) {
          if (b < inb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int k0 = 32 * b + 8 * g + 4 * h;
              const long long o = z_off + pt * in + k0;
              float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
              if (valid) z4 = *reinterpret_cast<const float4*>(z_ws + o);
              const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
              float a4[4], d4[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float cdf = 0.5f * (1.0f + erff(zz[i] * 0.70710678118654752440f));
                const float pdf = 0.39894228040143267794f * __expf(-0.5f * zz[i] * zz[i]);
                a4[i] = zz[i] * cdf;
                d4[i] = da[b][4 * g + i] * (cdf + zz[i] * pdf);
                dz[b][4 * g + i] = d4[i];
              }
              if (valid) {
                *reinterpret_cast<float4*>(dz_ws + o) = make_float4(d4[0], d4[1], d4[2], d4[3]);
                *reinterpret_cast<float4*>(a_ws + o) = make_float4(a4[0], a4[1], a4[2], a4[3]);
              }
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

__global__ __launch_bounds__(MLP_BLOCK, 2) void mlp_wgrad_kernel(
    vsa_mlp_plan plan, WgradLayers wl, const float* __restrict__ x, int x_stride,
    const float* __restrict__ dy, int dy_stride, int M, const float* __restrict__ dz_ws,
    const float* __restrict__ a_ws, float* __restrict__ partial) {
  int l = 0;
  while (l + 1 < plan.n_layers && (int)blockIdx.x >= wl.wg_begin[l + 1]) ++l;
  const int L = plan.n_layers;
  const int in = plan.dims[l], out = plan.dims[l + 1];
  const int inb = blocks_of(in), outb = blocks_of(out);
  const int nwg = wl.wg_begin[l + 1] - wl.wg_begin[l], wg = blockIdx.x - wl.wg_begin[l];
  // operand arrays of this layer
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
  // this workgroup's points: contiguous range, even length (two points per MFMA)
  const long long per = ((M + nwg - 1) / nwg + 1) & ~1ll;
  const long long p_begin = wg * per, p_end = min((long long)M, p_begin + per);
  f32x16 acc[MLP_MAXB];
  float bsum[MLP_MAXB];
  int pm[MLP_MAXB], pb[MLP_MAXB];
#pragma unroll
  for (int q = 0; q < MLP_MAXB; ++q) {
    acc[q] = f32x16{0};
    bsum[q] = 0.f;
    const int pair = wave + 4 * q;
    pm[q] = pair < npairs ? pair / inb : -1;
    pb[q] = pair < npairs ? pair % inb : 0;
  }
  // WG_UNROLL point pairs per trip: all operand loads of a trip are issued before its first MFMA
  // (one load -> one MFMA per trip left every wave waiting out a full memory latency: 11.6 ms
  // for the 2.1 M samples of a background batch, 20x the MFMA time)
  constexpr int WG_UNROLL = 8;
  // a wave's pairs are wave, wave + 4, ...: for 1, 2 or 4 input blocks they all share ONE input
  // block (4 is a multiple of inb), so the B operand is loaded once per point pair, not per pair
  const bool shared_b = (4 % inb) == 0;
  for (long long p0 = p_begin; p0 < p_end; p0 += 2 * WG_UNROLL) {
    float av[WG_UNROLL][MLP_MAXB], bv[WG_UNROLL][MLP_MAXB];
#pragma unroll
    for (int u = 0; u < WG_UNROLL; ++u) {
      const long long pt = p0 + 2 * u + kk;
      const bool ok = pt < p_end;
#pragma unroll
      for (int q = 0; q < MLP_MAXB; ++q) {
        av[u][q] = bv[u][q] = 0.f;
        if (pm[q] >= 0) {
          const int n = 32 * pm[q] + i, k = 32 * pb[q] + i;
          if (ok && n < out) av[u][q] = dop[pt * d_stride + n];
          if (ok && k < in && (q == 0 || !shared_b)) bv[u][q] = aop[pt * a_stride + k];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < WG_UNROLL; ++u) {
#pragma unroll
      for (int q = 0; q < MLP_MAXB; ++q) {
        if (pm[q] >= 0) {
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][q], shared_b ? bv[u][0] : bv[u][q], acc[q], 0, 0, 0);
          bsum[q] += av[u][q];
        }
      }
    }
  }
  // partial[l][wg][out_pad][in_pad] then [out_pad] bias sums
  const int in_pad = 32 * inb, out_pad = 32 * outb;
  float* part = partial + wl.part_off[l] + (long long)wg * (out_pad * in_pad + out_pad);
#pragma unroll
  for (int q = 0; q < MLP_MAXB; ++q) {
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
__global__ void mlp_reduce_kernel(vsa_mlp_plan plan, WgradLayers wl, const float* __restrict__ partial,
                                  vsa_mlp_grads grads) {
  const int l = blockIdx.y;
  const int in = plan.dims[l], out = plan.dims[l + 1];
  const int in_pad = 32 * blocks_of(in), out_pad = 32 * blocks_of(out);
  const int nwg = wl.wg_begin[l + 1] - wl.wg_begin[l];
  const int per = out_pad * in_pad + out_pad;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < per; idx += gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int g = 0; g < nwg; ++g) s += partial[wl.part_off[l] + (long long)g * per + idx];
    if (idx < out_pad * in_pad) {
      const int row = idx / in_pad, col = idx - row * in_pad;
      if (row < out && col < in && grads.dw[l]) grads.dw[l][(long long)row * in + col] = s;
    } else {
      const int n = idx - out_pad * in_pad;
      if (n < out && grads.db[l]) grads.db[l][n] = s;
    }
  }
}

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

// workgroups of the weight-gradient kernel: two per CU for large batches, fewer for small ones
// (every workgroup writes a full set of partial blocks that mlp_reduce then has to read: 512
// workgroups on a 10 k-point batch made the reduction the most expensive kernel of the step)
int wgrad_total_wgs(const vsa_mlp_plan& p, long long nr_points, int nr_cus) {
  long long n = nr_points / 512;
  if (n < p.n_layers) n = p.n_layers;
  if (n > 2ll * nr_cus) n = 2ll * nr_cus;
  return (int)n;
}

WgradLayers wgrad_layers(const vsa_mlp_plan& p, int total_wgs) {
  // workgroups per layer in proportion to the layer's MFMA count (>= 1 each)
  WgradLayers wl;
  int cost[VSA_MLP_MAX_LAYERS], sum = 0;
  for (int l = 0; l < p.n_layers; ++l) {
    // a wave issues ceil(pairs / 4) MFMAs per point pair
    cost[l] = (blocks_of(p.dims[l]) * blocks_of(p.dims[l + 1]) + 3) / 4;
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

}  // namespace

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
  if (partial_floats)
    *partial_floats = wgrad_layers(*plan, wgrad_total_wgs(*plan, nr_points, nr_cus)).part_off[plan->n_layers];
  return VSA_OK;
}

extern "C" int vsa_mlp_fwd(const vsa_mlp_plan* plan, const float* x, int x_stride, int nr_points,
                           float* y, int y_stride, float* z_ws, float* packed_ws, void* stream) {
  int rc = plan_ok(plan);
  if (rc) return rc;
  if (nr_points < 0 || x_stride < plan->dims[0] || y_stride < plan->dims[plan->n_layers])
    return VSA_ERR_ARG;
  if (nr_points == 0) return VSA_OK;
  if (!x || !y || !packed_ws) return VSA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mlp_pack_kernel, dim3(16, plan->n_layers), dim3(256), 0, st, *plan, packed_ws,
                     (float*)nullptr);
  int nr_cus = 0;
  rc = vsa_cu_count(&nr_cus);
  if (rc) return rc;
  const size_t lds = max_layer_bytes(*plan);
  static bool attr_set = false;
  if (!attr_set) {
    VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwd_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_dgrad_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    attr_set = true;
  }
  const int ntiles = vsa_div_up(nr_points, MLP_TILE);
  int grid = vsa_div_up(ntiles, 4);
  if (grid > 2 * nr_cus) grid = 2 * nr_cus;
  hipLaunchKernelGGL(mlp_fwd_kernel, dim3(grid), dim3(MLP_BLOCK), lds, st, *plan, packed_ws, x, x_stride,
                     nr_points, y, y_stride, z_ws);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_mlp_bwd(const vsa_mlp_plan* plan, const float* x, int x_stride, int nr_points,
                           const float* dy, int dy_stride, const float* z_ws, float* dz_ws,
                           float* a_ws, float* packed_ws, float* partial_ws, float* dx,
                           int dx_stride, const vsa_mlp_grads* grads, void* stream) {
  int rc = plan_ok(plan);
  if (rc) return rc;
  const int L = plan->n_layers;
  if (nr_points < 0 || x_stride < plan->dims[0] || dy_stride < plan->dims[L] ||
      (dx && dx_stride < plan->dims[0]))
    return VSA_ERR_ARG;
  if (!grads) return VSA_ERR_ARG;
  if (nr_points == 0) return VSA_OK;
  if (!x || !dy || !packed_ws || !partial_ws || (L > 1 && (!z_ws || !dz_ws || !a_ws)))
    return VSA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mlp_pack_kernel, dim3(16, L), dim3(256), 0, st, *plan, (float*)nullptr, packed_ws);
  int nr_cus = 0;
  rc = vsa_cu_count(&nr_cus);
  if (rc) return rc;
  static bool attr_set = false;
  if (!attr_set) {
    VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_dgrad_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    attr_set = true;
  }
  const int ntiles = vsa_div_up(nr_points, MLP_TILE);
  int grid = vsa_div_up(ntiles, 4);
  if (grid > 2 * nr_cus) grid = 2 * nr_cus;
  if (L > 1 || dx)
    hipLaunchKernelGGL(mlp_dgrad_kernel, dim3(grid), dim3(MLP_BLOCK), max_layer_bytes(*plan), st, *plan,
                       packed_ws, dy, dy_stride, nr_points, z_ws, dz_ws, a_ws, dx, dx_stride);
  const WgradLayers wl = wgrad_layers(*plan, wgrad_total_wgs(*plan, nr_points, nr_cus));
  hipLaunchKernelGGL(mlp_wgrad_kernel, dim3(wl.wg_begin[L]), dim3(MLP_BLOCK), 0, st, *plan, wl, x,
                     x_stride, dy, dy_stride, nr_points, dz_ws, a_ws, partial_ws);
  hipLaunchKernelGGL(mlp_reduce_kernel, dim3(68, L), dim3(256), 0, st, *plan, wl, partial_ws, *grads);   // 68 x 256 >= one thread per element of a 128 x 128 (+bias) layer
  VSA_RETURN_LAUNCH_STATUS();
}
