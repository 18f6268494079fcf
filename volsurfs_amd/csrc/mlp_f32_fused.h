// Fused backward of the fp32 MLP for networks whose weights stay resident in LDS (NerfHash's two networks,
// models/nerfhash.py:58-91: 51-64-64-64-65 and 80-64-64-3; models/mlp.py:8-69).  Included by mlp_f32.hip inside
// its anonymous namespace.
//
// Rounds 1-5 ran the backward as two kernels: mlp_dgrad wrote dZ of every hidden layer to HBM and mlp_wgrad read
// it back together with the activations A = GELU(z) the forward had stored — per sample of the background field
// 2 560 B written by the forward, 2 560 B read + 1 280 B written by dgrad, 2 560 B read by wgrad: 16 GB per
// 65 536-ray batch for 0.3 TFLOP, the two launches at 0.27 of the fp32 matrix peak (profiles/r05/bench_dtu.json).
// Here one persistent workgroup does both and the only per-sample traffic is z (the forward's pre-activations:
// 1 280 B in, nothing out) plus the layer-0 input and the output gradient:
//
//   workgroup = 8 waves, two per SIMD: a DATA wave and a WEIGHT wave (the two halves of the matrix work are equal, and
//   what one of them waits for — memory, the GELU arithmetic, LDS — the other covers with its MFMAs; a first version
//   with one wave per SIMD doing both measured 3.2 ms where its matrix instructions alone take 1.0: profiles/r06/mlp_diag.txt).
//   per round = 4 data waves x one 32-point tile each, per layer l = L-1 .. 0:
//     data wave:   A_{l-1} = GELU(z_{l-1}) and GELU'(z_{l-1}) from ONE evaluation of Phi / phi (gelu_fast.h) [x for l = 0]
//                  (barrier B: the weight waves are done with the staging rows)
//                  stage dZ_l and A_{l-1} of its tile into LDS, n-major: s_dz[n][point], s_a[k][point]
//                  (barrier A: staged)
//                  dA_{l-1} = W_l^T dZ_l from registers (transposed fragments resident in LDS); requests for the next
//                  layer's z (or x, or the next round's dy); dZ_{l-1} = dA_{l-1} * GELU'(z_{l-1})
//     weight wave: (barrier B) (barrier A) dW_l[n][k] += sum_p dZ_l[n][p] A_{l-1}[k][p] over the round's 128 staged points
//                  for ITS 32 x 32 block pairs (all layers' pairs are dealt round-robin to the four weight waves:
//                  <= 5 accumulators each for NerfHash); fragments are 16-byte LDS reads of 4 consecutive points
//   at the end every workgroup writes its dW / db partial blocks; mlp_reduce_kernel adds them up (deterministic).
//
// MFMA work per round and SIMD: ~290 (data) + ~290 (weight) v_mfma_f32_32x32x2_f32 of 64 cycles each.
#pragma once

// FB_DIAG (timing-only builds, results WRONG; tools/diag_mlp.sh): 1 no weight-gradient MFMAs, 2 no data-gradient
// MFMAs, 4 identity instead of the GELU evaluation, 8 no staging stores, 16 no barriers
#ifndef FB_DIAG
#define FB_DIAG 0
#endif
constexpr int FB_BLOCK = 512;               // 4 data waves + 4 weight waves
constexpr int FB_POINTS = 4 * MLP_TILE;      // points of a round (4 waves x 32)
constexpr int FB_SP = FB_POINTS + 4;         // staging row stride in floats: 16-byte aligned rows, 4-bank skew

// LDS floats of the fused backward: transposed packed weights + staging rows for the widest dZ and the widest A
__host__ __device__ inline int fb_rows_dz(const vsa_mlp_plan& p) {
  int w = 1;
  for (int l = 1; l <= p.n_layers; ++l) w = blocks_of(p.dims[l]) > w ? blocks_of(p.dims[l]) : w;
  return 32 * w;
}
__host__ __device__ inline int fb_rows_a(const vsa_mlp_plan& p) {
  int w = 1;
  for (int l = 0; l < p.n_layers; ++l) w = blocks_of(p.dims[l]) > w ? blocks_of(p.dims[l]) : w;
  return 32 * w;
}
inline int fb_pairs(const vsa_mlp_plan& p) {
  int n = 0;
  for (int l = 0; l < p.n_layers; ++l) n += blocks_of(p.dims[l]) * blocks_of(p.dims[l + 1]);
  return n;
}
inline size_t fb_lds_bytes(const vsa_mlp_plan& p) {
  return ((size_t)pack_offsets(p).fwd[p.n_layers] + (size_t)(fb_rows_dz(p) + fb_rows_a(p)) * FB_SP) * sizeof(float);
}

// NB: widest layer in 32-blocks (dy / x rows), NH: widest HIDDEN layer (z rows), Q: block pairs per weight wave
template <int NB, int NH, int Q>
__global__ __launch_bounds__(FB_BLOCK, 1) void mlp_bwd_fused_kernel(
    vsa_mlp_plan plan, WgradLayers wl, MlpGroups gp, long long packed_stride, long long hidden, long long partial_stride,
    int rows_dz, const float* __restrict__ packed_t, const float* __restrict__ x, int x_stride,
    const float* __restrict__ dy, int dy_stride, const float* __restrict__ z_ws, float* __restrict__ dx,
    int dx_stride, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];
  __shared__ LayerMeta s_meta;
  const int grp = blockIdx.y;
  load_layer_meta(plan, gp, grp, s_meta);
  const int L = plan.n_layers;
  const int M = pick(gp.M, grp);
  {
    const long long r0 = pick(gp.row0, grp);
    packed_t += grp * packed_stride;
    x += r0 * x_stride;
    dy += r0 * dy_stride;
    if (z_ws) z_ws += r0 * hidden;
    if (dx) dx += r0 * dx_stride;
    partial += grp * partial_stride;
  }
  const int w_floats = meta_fwd(s_meta, L);
  float* const s_dz = s_w + w_floats;
  float* const s_a = s_dz + rows_dz * FB_SP;
  {
    const float4* s4 = reinterpret_cast<const float4*>(packed_t);
    float4* d4 = reinterpret_cast<float4*>(s_w);
    for (int i = threadIdx.x; i < w_floats / 4; i += FB_BLOCK) d4[i] = s4[i];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 31, h = lane >> 5;
  const bool data_role = wave < 4;
  const int tw = wave & 3;                  // tile slot of a data wave / pair class of a weight wave
  const int ntiles = (M + MLP_TILE - 1) / MLP_TILE;
  const int per_round = gridDim.x * 4;
  const int rounds = (ntiles + per_round - 1) / per_round;
  long long z_end = 0;                      // offset just past the last hidden layer's block
  for (int l = 0; l + 1 < L; ++l) z_end += (long long)M * meta_dim(s_meta, l + 1);

  if (!data_role) {
    // ================================================================== weight waves
    // this wave's block pairs of all layers: global pair number g -> weight wave g % 4, accumulator g / 4
    int pl[Q], pm[Q], pb[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) pl[q] = -1, pm[q] = 0, pb[q] = 0;
    {
      int g = 0;
      for (int l = 0; l < L; ++l) {
        const int inb = blocks_of(meta_dim(s_meta, l)), outb = blocks_of(meta_dim(s_meta, l + 1));
        for (int m = 0; m < outb; ++m)
          for (int b = 0; b < inb; ++b, ++g)
            if ((g & 3) == tw) {
#pragma unroll
              for (int q = 0; q < Q; ++q)
                if (q == (g >> 2)) pl[q] = l, pm[q] = m, pb[q] = b;
            }
      }
    }
    f32x16 accw[Q];
    float bsum[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) accw[q] = f32x16{0}, bsum[q] = 0.f;
    for (int rd = 0; rd < rounds; ++rd) {
      for (int l = L - 1; l >= 0; --l) {
        if (!(FB_DIAG & 16) || (rd == 0 && l == L - 1)) __syncthreads();     // B
        if (!(FB_DIAG & 16)) __syncthreads();                                // A: layer l is staged
        // A operand lane (i, kk): dZ[32 m + i][point], B operand lane (j, kk): A[32 b + j][point]; the k-slot -> point
        // map is free: half kk takes points 64 kk + 4 t .. + 3 of step group t (one 16-byte read = four MFMAs)
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          if (pl[q] == l && !(FB_DIAG & 1)) {
            const float4* ar = reinterpret_cast<const float4*>(s_dz + (32 * pm[q] + p) * FB_SP + 64 * h);
            const float4* br = reinterpret_cast<const float4*>(s_a + (32 * pb[q] + p) * FB_SP + 64 * h);
            float bs = 0.f;
#pragma unroll 4      /* (fully unrolled, the 32 fragment reads were hoisted into 128 registers: spills) */
            for (int t = 0; t < 16; ++t) {
              const float4 av = ar[t], bv = br[t];
              accw[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, accw[q], 0, 0, 0);
              accw[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, accw[q], 0, 0, 0);
              accw[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, accw[q], 0, 0, 0);
              accw[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, accw[q], 0, 0, 0);
              bs += (av.x + av.y) + (av.z + av.w);
            }
            bsum[q] += bs;
          }
        }
      }
    }
    // this workgroup's partial blocks: partial[l][wg][out_pad][in_pad] then [out_pad] bias sums (mlp_reduce_kernel)
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (pl[q] >= 0) {
        const int l = pl[q];
        const int in_pad = 32 * blocks_of(meta_dim(s_meta, l)), out_pad = 32 * blocks_of(meta_dim(s_meta, l + 1));
        long long off = 0;
#pragma unroll
        for (int k = 0; k < VSA_MLP_MAX_LAYERS; ++k)
          if (k == l) off = wl.part_off[k];
        float* part = partial + off + (long long)blockIdx.x * (out_pad * in_pad + out_pad);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * pm[q] + (r & 3) + 8 * (r >> 2) + 4 * h, c = 32 * pb[q] + p;
          part[row * in_pad + c] = accw[q][r];
        }
        if (pb[q] == 0) {     // the pair with b == 0 of every m also owns that block's bias sums
          const float other = __shfl_xor(bsum[q], 32, 64);
          if (h == 0) part[out_pad * in_pad + 32 * pm[q] + p] = bsum[q] + other;
        }
      }
    }
    return;
  }

  // ==================================================================== data waves
  const int col = tw * MLP_TILE + p;        // this lane's point among the round's 128
  // Loads run one step ahead of their use: `zq` / `xq` hold the NEXT layer's input — the z of the hidden layer below,
  // or x for layer 0 — requested behind the current layer's MFMAs; the next round's dy and top-layer z behind layer 0's.
  auto tile_of = [&](int rd) { return rd * per_round + (int)blockIdx.x * 4 + tw; };
  // Rows whose stride is a multiple of 4 floats (x and dx always here: the host takes the two-kernel backward otherwise;
  // dy when the caller could arrange it) move as 16-byte accesses in the layout the MFMAs want — lane (p, h) owns
  // columns 32 b + 8 g + 4 h .. + 3 of its point's row — with no per-element branch: a column group beyond the row is
  // read at a clamped address and zeroed by a select.  (The first version read every element behind its own predicate:
  // ~150 exec-mask branch regions with 64-bit address arithmetic per round, 8 000 instructions for 290 MFMAs.)
  auto load_rows4 = [&](const float* __restrict__ T, int st, int w, int rd, float dst[NB][16]) __attribute__((always_inline)) {
    const int tile = tile_of(rd);
    const long long pt = (long long)tile * MLP_TILE + p;
    const bool valid = tile < ntiles && pt < M;
    const float* row = T + (valid ? pt : 0) * (long long)st;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k0 = 32 * b + 8 * g + 4 * h;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (32 * b < w) {                   // uniform: the block exists
          const int kc = k0 < st ? k0 : st - 4;
          v = *reinterpret_cast<const float4*>(row + kc);
        }
        dst[b][4 * g] = (valid && k0 < w) ? v.x : 0.f;
        dst[b][4 * g + 1] = (valid && k0 + 1 < w) ? v.y : 0.f;
        dst[b][4 * g + 2] = (valid && k0 + 2 < w) ? v.z : 0.f;
        dst[b][4 * g + 3] = (valid && k0 + 3 < w) ? v.w : 0.f;
      }
  };
  auto load_dy = [&](int rd, float dst[NB][16]) __attribute__((always_inline)) { load_rows4(dy, dy_stride, meta_dim(s_meta, L), rd, dst); };
  auto load_z = [&](int rd, int l, long long z_off, float dst[NB][16]) {       // z_{l-1}, the input of layer l >= 1
    const int tile = tile_of(rd);
    const long long pt = (long long)tile * MLP_TILE + p;
    const bool valid = tile < ntiles && pt < M;
    const int in = meta_dim(s_meta, l);
    const int inb = blocks_of(in);
#pragma unroll
    for (int b = 0; b < NH; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < inb && valid) z4 = *reinterpret_cast<const float4*>(z_ws + z_off + pt * in + 32 * b + 8 * g + 4 * h);
        dst[b][4 * g] = z4.x, dst[b][4 * g + 1] = z4.y, dst[b][4 * g + 2] = z4.z, dst[b][4 * g + 3] = z4.w;
      }
  };
  auto load_x = [&](int rd, float dst[NB][16]) __attribute__((always_inline)) { load_rows4(x, x_stride, meta_dim(s_meta, 0), rd, dst); };
  const long long z_top = L > 1 ? z_end - (long long)M * meta_dim(s_meta, L - 1) : 0;   // offset of z_{L-2}: the top layer's input
  float dz[NB][16];                   // dZ of the layer being processed (rows rho(r, h) of its OUTPUT blocks)
  float nq[NB][16];                   // the next layer's input, in flight: z (its first NH blocks) or, for layer 0, x
  if (rounds > 0) {
    load_dy(0, dz);
    if (L > 1) load_z(0, L - 1, z_top, nq);
    else load_x(0, nq);
  }
  // One layer of one round.  TOP (l = L-1): dZ is dy, up to NB blocks wide; every other layer's dZ is a hidden layer's (NH
  // blocks).  BOT (l = 0): the input is x (NB blocks, no GELU); every other layer's input is a hidden layer's z.  Compiled
  // per (TOP, BOT) so that the third block of dz / da / nq is live only where a layer has one (two waves per SIMD: 256
  // registers; with run-time block counts the allocator kept all three blocks of everything and spilled).
  auto layer = [&](auto top_tag, auto bot_tag, int rd, int l, long long z_off, bool valid, long long pt) {
    constexpr bool TOP = decltype(top_tag)::value, BOT = decltype(bot_tag)::value;
    constexpr int OB = TOP ? NB : NH, IB = BOT ? NB : NH;
    const int in = meta_dim(s_meta, l), out = meta_dim(s_meta, l + 1);
    const int inb = blocks_of(in), outb = blocks_of(out);
    const int w_off = meta_fwd(s_meta, l);
    // ---- the layer's input A_{l-1} and GELU' at it (hidden layers), rows rho(r, h) of the input blocks
    float gd[BOT ? 1 : NH][16], a[BOT ? 1 : NH][16];
    if constexpr (!BOT) {
#pragma unroll
      for (int b = 0; b < NH; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float zz = nq[b][r];
          float cdf, pdf;
          if (FB_DIAG & 4) cdf = 0.5f, pdf = zz; else
          gelu_cdf_pdf(zz, cdf, pdf);
          gd[b][r] = __builtin_fmaf(zz, pdf, cdf);
          a[b][r] = zz * cdf;          // (rows of invalid points / absent blocks: z = 0 -> 0)
        }
    }
    if (!(FB_DIAG & 16) || (rd == 0 && TOP)) __syncthreads();     // B: the weight waves are done with the rows
    if (!(FB_DIAG & 8)) {
#pragma unroll
      for (int b = 0; b < IB; ++b)
        if (b < inb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v;
            if constexpr (BOT) v = nq[b][r]; else v = a[b][r];
            s_a[(32 * b + rho(r, h)) * FB_SP + col] = v;
          }
        }
#pragma unroll
      for (int m = 0; m < OB; ++m)
        if (m < outb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s_dz[(32 * m + rho(r, h)) * FB_SP + col] = dz[m][r];
        }
    }
    if (!(FB_DIAG & 16)) __syncthreads();                                // A: staged
    // ---- data gradient of the own tile: dA[k][p] = sum_n W[n][k] dZ[n][p]
    f32x16 da[IB];
#pragma unroll
    for (int b = 0; b < IB; ++b) {
      da[b] = f32x16{0};
      if (b < inb && (!BOT || dx != nullptr) && !(FB_DIAG & 2)) {
#pragma unroll
        for (int m = 0; m < OB; ++m) {
          if (m < outb) {
            float wv[16];
            mlp_load_frags(s_w + w_off + ((b * outb + m) * 16) * 64, lane, wv);
#pragma unroll
            for (int s = 0; s < 16; ++s)
              da[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[s], dz[m][s], da[b], 0, 0, 0);
          }
        }
      }
    }
    // ---- requests for what comes next, behind the MFMAs that read dz
    if (l > 1) load_z(rd, l - 1, z_off - (long long)M * meta_dim(s_meta, l - 1), nq);
    else if (l == 1) load_x(rd, nq);
    else if (rd + 1 < rounds) {
      if (L > 1) load_z(rd + 1, L - 1, z_top, nq);
      else load_x(rd + 1, nq);
    }
    // ---- epilogue: dX (and the next round's dy), or dZ_{l-1} = dA * GELU'(z_{l-1})
    if constexpr (BOT) {
      if (dx && valid) {      // (16-byte stores; the row's padding columns get the zeros the zero weight columns produce)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          if (b < inb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int k0 = 32 * b + 8 * g + 4 * h;
              if (k0 < dx_stride)
                *reinterpret_cast<float4*>(dx + pt * dx_stride + k0) =
                    make_float4(da[b][4 * g], da[b][4 * g + 1], da[b][4 * g + 2], da[b][4 * g + 3]);
            }
          }
      }
      if (rd + 1 < rounds) load_dy(rd + 1, dz);
    } else {
#pragma unroll
      for (int b = 0; b < NH; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) dz[b][r] = b < inb ? da[b][r] * gd[b][r] : 0.f;
    }
  };
  typedef std::true_type T_;
  typedef std::false_type F_;
  for (int rd = 0; rd < rounds; ++rd) {
    const int tile = tile_of(rd);
    const long long pt = (long long)tile * MLP_TILE + p;
    const bool valid = tile < ntiles && pt < M;
    if (L == 1) {
      layer(T_{}, T_{}, rd, 0, 0, valid, pt);
      continue;
    }
    long long z_off = z_top;                 // z_{l-1} of the layer being processed
    layer(T_{}, F_{}, rd, L - 1, z_off, valid, pt);
    for (int l = L - 2; l >= 1; --l) {
      z_off -= (long long)M * meta_dim(s_meta, l);
      layer(F_{}, F_{}, rd, l, z_off, valid, pt);
    }
    layer(F_{}, T_{}, rd, 0, 0, valid, pt);
  }
}
