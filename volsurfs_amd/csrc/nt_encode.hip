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
#include <type_traits>

#include "nt_common.h"
#include "nt_enc_common.h"

namespace {

// per-piece overheads of the cost axis (units of 256 slots): forward hashed / dense, backward hashed / dense
// (re-swept after the per-lane work split: 60/20/120/45 and 110/35/180/80 are within 2 % of these either way)
#ifndef NT_ENC_OVH_FH
#define NT_ENC_OVH_FH 87
#endif
#ifndef NT_ENC_OVH_FD
#define NT_ENC_OVH_FD 27
#endif
#ifndef NT_ENC_OVH_BH
#define NT_ENC_OVH_BH 145
#endif
#ifndef NT_ENC_OVH_BD
#define NT_ENC_OVH_BD 64
#endif
#ifndef NT_ENC_PAIR_W
#define NT_ENC_PAIR_W 22       /* cost of a (colour, alpha) pair unit in 1/16 of a single texture's (forward, dense) */
#endif
constexpr int ENC_BLOCK = 1024;     // 16 waves: a 128 KiB level pins one workgroup per CU
constexpr int ENC_UNROLL = 8;       // slots in flight per lane (backward)
#ifndef ENC_UNROLL_FWD_N
#define ENC_UNROLL_FWD_N 4
#endif
constexpr int ENC_UNROLL_FWD = ENC_UNROLL_FWD_N;   // forward
constexpr int ENC_UNIT = 256;       // slot granularity of the persistent work split (nt_common.h)
constexpr int LDS_ENTRIES = 32768;    // 4-byte entries of LDS a workgroup may use (128 KiB)


#ifndef NT_ENC_DIAG_FWD
#define NT_ENC_DIAG_FWD 0
#endif
#ifndef NT_ENC_FWD_NOREUSE
#define NT_ENC_FWD_NOREUSE 0
#endif
#ifndef NT_ENC_PREFETCH_FWD
#define NT_ENC_PREFETCH_FWD 0     /* stretches of texel centres in flight per lane (0: loaded at their use) */
#endif
#ifndef NT_ENC_ONE_LAUNCH
#define NT_ENC_ONE_LAUNCH 1      /* dense + hashed levels of a direction in one launch */
#endif
#ifndef NT_ENC_PAIR_DENSE
#define NT_ENC_PAIR_DENSE 1      /* dense levels: colour + alpha texture of a (shell, degree) in ONE piece (shared cell / weights) */
#endif
constexpr int ENC_PAIR_OFF = 15360;   // entries: the second table of a pair sits at a FIXED LDS offset (61 440 B: an
                                      // immediate of the gathers); the largest dense level has 15 136 entries
template <bool HASHED>
__device__ __forceinline__ void nt_encode_fwd_body(
    const vsa_nt_plan& plan, int level0, int n_levels, const half2_t* __restrict__ tables,
    const float2* __restrict__ slot_xy, const int* __restrict__ seg_start,
    half2_t* __restrict__ features, unsigned char* s_raw) {
  half2_t* s_tab = reinterpret_cast<half2_t*>(s_raw);
  const long long n_entries = plan.level_offset[plan.n_levels];
  const int nl = plan.n_levels;
  NT_SPAN_MARK(HASHED ? 1 : 0, 0);
  NT_BAL_BEGIN();
  // dense levels: pieces are (shell, degree) pairs (nt_for_each_piece, paired) — both tables in LDS
  const bool pair_mode = !HASHED && NT_ENC_PAIR_DENSE && NT_ENC_ACC_F16 && !NT_ENC_PREFETCH_FWD &&
                         plan.nr_shells * 2 * VSA_NT_MAX_DEG <= 64;
  // per-piece overheads and unit weights: fitted from per-workgroup timings (tools/fit_cost.py):
  // a piece costs 87 (hashed: a 128 KiB table to stage) / 27 (dense) units of 256 slots, and a
  // unit of a level finer than the texture (no reuse of the previous slot's cell) 0.92 of one
  // that goes through the reuse bookkeeping; a pair costs NT_ENC_PAIR_W / 16 of one texture
  auto unit_weight = [&](int pl, int deg, int wtype) {
    const int w = !HASHED || plan.level_scale[level0 + pl] < (float)plan.tex_res[deg] ? 16 : 15;
    return wtype == 2 ? (w * NT_ENC_PAIR_W) >> 4 : w;
  };
  nt_for_each_piece<ENC_UNIT>(plan, seg_start, n_levels, HASHED ? NT_ENC_OVH_FH : NT_ENC_OVH_FD,
                              [&](int pl, int tex, int first, int last, int, int) {
    const int level = level0 + pl;
    const LevelGeom g = level_geom(plan, level);
    // (a pair stages its second table at the fixed offset ENC_PAIR_OFF: a dense level of another grid
    //  geometry that is larger than that — up to LDS_ENTRIES is accepted — is walked once per texture instead)
    const int partner = pair_mode ? nt_pair_partner(plan, tex) : -1;
    const bool pair_fits = (int)g.size <= ENC_PAIR_OFF && ENC_PAIR_OFF + (int)g.size <= LDS_ENTRIES;
    const int tex2 = pair_fits ? partner : -1;
    for (int pass = 0; pass < (partner >= 0 && !pair_fits ? 2 : 1); ++pass) {
    const int tex_p = pass ? partner : tex;
    __syncthreads();   // the previous piece is done with the table
    auto stage = [&](int t, half2_t* dst_h) {  // stage the level (size is a multiple of 8 entries = 32 B)
      // all of a thread's loads in flight together (a 2^15-entry level is 8 x 16 B per thread; one
      // load, wait, store per trip made staging eight memory latencies long).  Native vectors and
      // an unconditional (clamped) fill: HIP's uint4 struct, or a predicated fill, sends r[] to scratch.
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const u32x4* src = reinterpret_cast<const u32x4*>(tables + (long long)nt_param_tex(plan, t) * n_entries + plan.level_offset[level]);
      u32x4* dst = reinterpret_cast<u32x4*>(dst_h);
      constexpr int SB = 8;
      const int nvec = (int)(g.size / 4);
      for (int i0 = threadIdx.x; i0 < nvec; i0 += ENC_BLOCK * SB) {
        u32x4 r[SB];
#pragma unroll
        for (int k = 0; k < SB; ++k) {
          const int i = i0 + k * ENC_BLOCK;
          r[k] = src[i < nvec ? i : nvec - 1];
        }
#pragma unroll
        for (int k = 0; k < SB; ++k)
          if (i0 + k * ENC_BLOCK < nvec) dst[i0 + k * ENC_BLOCK] = r[k];
      }
    };
    stage(tex_p, s_tab);
    if (tex2 >= 0) stage(tex2, s_tab + ENC_PAIR_OFF);
    __syncthreads();
    const int type = (tex_p / VSA_NT_MAX_DEG) & 1;
    half2_t* out = features + nt_feat_plane_base(plan, type, level);
    half2_t* out2 = features + nt_feat_plane_base(plan, 1, level);      // the alpha plane of a pair
    // A lane owns ENC_UNROLL_FWD consecutive slots (neighbouring texels of one texture
    // row): 32 B of texel centres in, 16 B of features out per lane as dwordx4 accesses.
    // REUSE (levels coarser than the texture): the 4 LDS gathers are skipped while
    // consecutive slots stay in one grid cell; at finer levels that never happens and the
    // bookkeeping (11 moves + 3 branches per slot in the ISA) is left out.
    // PAIR: cell, corner indices and the half-rounded corner weights once, gathers + blend per table.
    const int a_first = first & ~(ENC_UNROLL_FWD - 1);
    auto run = [&](auto reuse_tag, auto pair_tag) {
      constexpr bool REUSE = decltype(reuse_tag)::value;
      constexpr bool PAIR = decltype(pair_tag)::value;
#if NT_ENC_PREFETCH_FWD
      // NT_ENC_PREFETCH_FWD stretches of texel centres are in flight per lane, requested AFTER the
      // stores of the stretch that is being finished: the wait at the head of a trip then never
      // includes those stores (past the end: the last stretch again, unused)
      constexpr int PD = NT_ENC_PREFETCH_FWD;
      const int s_max = (last - 1) & ~(ENC_UNROLL_FWD - 1);
      float4 xyn[PD][ENC_UNROLL_FWD / 2];
      auto request = [&](int s, float4 (&dst)[ENC_UNROLL_FWD / 2]) {
        const float4* xp = reinterpret_cast<const float4*>(slot_xy + (s < s_max ? s : s_max));
#pragma unroll
        for (int i = 0; i < ENC_UNROLL_FWD / 2; ++i) dst[i] = xp[i];
      };
#pragma unroll
      for (int d = 0; d < PD; ++d) request(a_first + threadIdx.x * ENC_UNROLL_FWD + d * ENC_BLOCK * ENC_UNROLL_FWD, xyn[d]);
#endif
      for (int s0 = a_first + threadIdx.x * ENC_UNROLL_FWD; s0 < last; s0 += ENC_BLOCK * ENC_UNROLL_FWD) {
        float4 xyv[ENC_UNROLL_FWD / 2];
#if NT_ENC_PREFETCH_FWD
#pragma unroll
        for (int i = 0; i < ENC_UNROLL_FWD / 2; ++i) xyv[i] = xyn[0][i];
#pragma unroll
        for (int d = 0; d + 1 < PD; ++d)
#pragma unroll
          for (int i = 0; i < ENC_UNROLL_FWD / 2; ++i) xyn[d][i] = xyn[d + 1][i];
#else
#if NT_ENC_DIAG_FWD & 1   /* timing-only: texel centres from a 64 KiB window (L2-resident) */
        const float4* xp = reinterpret_cast<const float4*>(slot_xy + (s0 & 0x1fff));
#else
        const float4* xp = reinterpret_cast<const float4*>(slot_xy + s0);
#endif
#pragma unroll
        for (int i = 0; i < ENC_UNROLL_FWD / 2; ++i) xyv[i] = xp[i];
#endif
        CellRefS cr[ENC_UNROLL_FWD];
        bool fresh[ENC_UNROLL_FWD];
#pragma unroll
        for (int u = 0; u < ENC_UNROLL_FWD; ++u) {
          const float x = (u & 1) ? xyv[u >> 1].z : xyv[u >> 1].x;
          const float y = (u & 1) ? xyv[u >> 1].w : xyv[u >> 1].y;
          cr[u] = cell_ref_s(g, x, y);
          fresh[u] = !REUSE || u == 0 || cr[u].cx != cr[u - 1].cx || cr[u].cy != cr[u - 1].cy;
        }
        half2_t v[ENC_UNROLL_FWD][4], v2[PAIR ? ENC_UNROLL_FWD : 1][4];
#pragma unroll
        for (int u = 0; u < ENC_UNROLL_FWD; ++u) {
          if (fresh[u]) {
            unsigned idx[4];
            cell_indices<HASHED>(g, cr[u].cx, cr[u].cy, idx);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[u][k] = s_tab[idx[k]];
            if constexpr (PAIR) {
#pragma unroll
              for (int k = 0; k < 4; ++k) v2[u][k] = s_tab[idx[k] + ENC_PAIR_OFF];
            }
          }
        }
        unsigned outw[ENC_UNROLL_FWD], outw2[PAIR ? ENC_UNROLL_FWD : 1];
#pragma unroll
        for (int u = 0; u < ENC_UNROLL_FWD; ++u) {
          if (REUSE && u > 0 && !fresh[u]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[u][k] = v[u - 1][k];
            if constexpr (PAIR) {
#pragma unroll
              for (int k = 0; k < 4; ++k) v2[u][k] = v2[u - 1][k];
            }
          }
          unsigned ew[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) ew[k] = __builtin_bit_cast(unsigned, v[u][k]);
#if NT_ENC_ACC_F16
          if constexpr (PAIR) {
            half2_t wh[4];
            enc_weights_h(cr[u].w, wh);
            outw[u] = enc_blend_h(ew, wh);
#pragma unroll
            for (int k = 0; k < 4; ++k) ew[k] = __builtin_bit_cast(unsigned, v2[u][k]);
            outw2[u] = enc_blend_h(ew, wh);
          } else
#endif
          {
            outw[u] = enc_blend(ew, cr[u].w);
          }
        }
#if NT_ENC_DIAG_FWD & 2   /* timing-only: no feature stores */
#pragma unroll
        for (int u = 0; u < ENC_UNROLL_FWD; ++u) asm volatile("" ::"v"(outw[u]));
        continue;
#endif
#if NT_ENC_DIAG_FWD & 4   /* timing-only: feature stores into a 64 KiB window */
        const long long o_in = nt_feat_in_plane(nl, s0 & 0x3fff);
#else
        const long long o_in = nt_feat_in_plane(nl, s0);
#endif
        unsigned* op = reinterpret_cast<unsigned*>(out) + o_in;
        unsigned* op2 = reinterpret_cast<unsigned*>(out2) + o_in;
        if (s0 >= first && s0 + ENC_UNROLL_FWD <= last) {
          uint4* o4 = reinterpret_cast<uint4*>(op);
#pragma unroll
          for (int i = 0; i < ENC_UNROLL_FWD / 4; ++i)
            o4[i] = make_uint4(outw[4 * i], outw[4 * i + 1], outw[4 * i + 2], outw[4 * i + 3]);
          if constexpr (PAIR) {
            uint4* p4 = reinterpret_cast<uint4*>(op2);
#pragma unroll
            for (int i = 0; i < ENC_UNROLL_FWD / 4; ++i)
              p4[i] = make_uint4(outw2[4 * i], outw2[4 * i + 1], outw2[4 * i + 2], outw2[4 * i + 3]);
          }
        } else {
#pragma unroll
          for (int u = 0; u < ENC_UNROLL_FWD; ++u)
            if (s0 + u >= first && s0 + u < last) {
              op[u] = outw[u];
              if constexpr (PAIR) op2[u] = outw2[u];
            }
        }
#if NT_ENC_PREFETCH_FWD
        request(s0 + PD * ENC_BLOCK * ENC_UNROLL_FWD, xyn[PD - 1]);
#endif
      }
    };
    const bool reuse = !NT_ENC_FWD_NOREUSE && g.scale < (float)plan.tex_res[tex % VSA_NT_MAX_DEG];
    if (!HASHED && tex2 >= 0) {
      if (reuse) run(std::true_type{}, std::true_type{});
      else run(std::false_type{}, std::true_type{});
    } else {
      if (reuse) run(std::true_type{}, std::false_type{});
      else run(std::false_type{}, std::false_type{});
    }
    }   // pass
  }, 0, 1 << 30, unit_weight, HASHED ? NT_BAL_ENC_FWD_H : NT_BAL_ENC_FWD_D, pair_mode);
  __syncthreads();
  NT_SPAN_MARK(HASHED ? 1 : 0, 1);
  NT_BAL_END(HASHED ? NT_BAL_ENC_FWD_H : NT_BAL_ENC_FWD_D);
}

template <bool HASHED>
__global__ __launch_bounds__(ENC_BLOCK) void nt_encode_fwd_kernel(
    vsa_nt_plan plan, int level0, int n_levels, const half2_t* __restrict__ tables,
    const float2* __restrict__ slot_xy, const int* __restrict__ seg_start,
    half2_t* __restrict__ features) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  nt_encode_fwd_body<HASHED>(plan, level0, n_levels, tables, slot_xy, seg_start, features, s_raw);
}

// Dense levels [0, lh) and hashed levels [lh, n) in ONE launch: a workgroup walks its share of the
// dense cost axis, then its share of the hashed one (the two classes touch different planes: no
// ordering between workgroups).  As two launches the dense one (0.1 ms) ended behind its slowest
// workgroups (span efficiency 0.76-0.80: stragglers, not a slope of the axis) with the chip idle.
__global__ __launch_bounds__(ENC_BLOCK) void nt_encode_fwd_both_kernel(
    vsa_nt_plan plan, int lh, const half2_t* __restrict__ tables,
    const float2* __restrict__ slot_xy, const int* __restrict__ seg_start,
    half2_t* __restrict__ features) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  nt_encode_fwd_body<false>(plan, 0, lh, tables, slot_xy, seg_start, features, s_raw);
  __syncthreads();
  nt_encode_fwd_body<true>(plan, lh, plan.n_levels - lh, tables, slot_xy, seg_start, features, s_raw);
}

// Backward: grad_table[tex][level entries][feature] += w * dF[slot]; one
// workgroup per (texture, level, feature, group) with the feature's gradient
// plane (size entries) in LDS.
//
// gfx950's LDS float atomic (ds_add_f32) retires ~0.4 lanes/clk/CU on random
// addresses, the integer one ~8.5 (tools/ubench/lds_atomics.hip), so the plane
// is accumulated in 32-bit FIXED POINT with a per-workgroup scale that makes
// overflow impossible: S = 2^30 / sum_slots |dF| bounds every entry's |sum|
// (bilinear weights of a slot sum to 1) below 2^30.  Resolution is
// sum|dF| / 2^30, i.e. <= span / 2^30 = 2.4e-4 of the MEAN |dF| per add.
// Same-address adds serialise (16 lanes on one entry: 5x slower), and at the
// coarse dense levels neighbouring texels share their cell; those small planes
// are therefore replicated `copies` times in LDS (lane & (copies-1) picks one)
// and summed at the flush.
// One piece of the backward: slots [first, last) of `tex` scattered into the LDS plane(s)
// of `level`.  NF = 1: feature `feat` only (the plane fills the LDS); NF = 2: both
// features in one pass over the slots (two planes; levels of <= 16384 entries), which
// shares the loads, the cell arithmetic and the indices between the features.
#ifndef NT_ENC_DIAG
#define NT_ENC_DIAG 0
#endif
#ifndef NT_ENC_FLUSH_ATOMIC
#define NT_ENC_FLUSH_ATOMIC 1     /* every flush through no-return float atomics: no read-modify-write round trip (r4: bwd 0.626 -> 0.607 ms) */
#endif
#ifndef NT_ENC_FLUSH_BATCH
#define NT_ENC_FLUSH_BATCH 8     /* table entries per thread whose read-modify-write is in flight together (0: one at a time) */
#endif
#ifndef NT_ENC_PREFETCH
#define NT_ENC_PREFETCH 1
#endif
#if NT_ENC_DIAG & 64   /* timing-only: the scatter's LDS atomics compiled out (values kept alive) */
#define ENC_LDS_ADD(ptr, val) asm volatile("" ::"v"(ptr), "v"(val))
#else
#define ENC_LDS_ADD(ptr, val) atomicAdd(ptr, val)
#endif
template <bool HASHED, int NF, bool MERGE>
__device__ __forceinline__ void enc_bwd_piece(
    const vsa_nt_plan& plan, int* s_g, int level, int feat, int tex, int first, int last,
    bool single, bool store, const half2_t* __restrict__ dfeatures, const float* __restrict__ dfeat_abs_sum,
    float dscale_inv, const float2* __restrict__ slot_xy, float* __restrict__ grad_tables) {
  const int nl = plan.n_levels;
  const long long n_entries = plan.level_offset[plan.n_levels];
  const LevelGeom g = level_geom(plan, level);
  // fixed-point scale from sum |dF| over the WHOLE (texture, level, feature) plane,
  // accumulated by the MLP backward kernel while it wrote dF (an upper bound for
  // this piece's share; saves a second pass over dF)
  float S[NF], S_inv[NF];
  bool any = false;
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const float total = dfeat_abs_sum[tex * 32 + 2 * level + feat + f] * 1.001f;
    S[f] = 0.f, S_inv[f] = 0.f;
    if (total > 0.f) {
      int e;
      frexpf(total, &e);                    // total = m * 2^e, m in [0.5, 1)
      S[f] = ldexpf(1.0f, 30 - e);          // power of two: exact scaling and un-scaling
      S_inv[f] = ldexpf(1.0f, e - 30) * dscale_inv;
      any = true;
    }
  }
  if (!any) return;   // nothing to add (uniform across the workgroup)
#if NT_ENC_DIAG & 8
  return;
#endif
  int copies = 1;
  if (!HASHED) {
    while (copies < 32 && (long long)g.size * copies * 2 * NF <= LDS_ENTRIES) copies *= 2;
  }
  const int plane = (int)g.size * copies;   // LDS entries per feature
  __syncthreads();   // the previous piece's flush has read the planes
#if !(NT_ENC_DIAG & 2)
  for (int i = threadIdx.x; i < plane * NF; i += ENC_BLOCK) s_g[i] = 0;
#endif
  int* my_g = s_g + (threadIdx.x & (copies - 1)) * g.size;   // consecutive lanes -> different copies
  const int type = (tex / VSA_NT_MAX_DEG) & 1;
  const unsigned* dFw = reinterpret_cast<const unsigned*>(dfeatures + nt_feat_plane_base(plan, type, level));
  __syncthreads();
  // scatter.  A lane owns ENC_UNROLL = 8 CONSECUTIVE slots (= neighbouring texels of one
  // texture row): 64 B of texel centres and 32 B of dF per lane come in as dwordx4
  // loads; the 64 lanes of a wave are 8 texels apart, so they share a grid cell only at
  // the coarsest levels (same-address LDS atomics serialise; those small planes are
  // replicated, see `copies`); and while consecutive slots of a lane stay in one cell
  // their four corner contributions are summed in registers and added once.  The sums
  // are integers, so the result is independent of this grouping.
  const int a_first = first & ~(ENC_UNROLL - 1);
#if NT_ENC_PREFETCH
  // NT_ENC_PREFETCH stretches of texel centres and gradients are in flight per lane (past the
  // end: the last stretch's again, unused); the loop has no other vector-memory operation, so
  // the wait at the head of a trip is for exactly the oldest request
  constexpr int PD = NT_ENC_PREFETCH;
  const int s_max = (last - 1) & ~(ENC_UNROLL - 1);
  float4 xyn[PD][ENC_UNROLL / 2];
  uint4 dn[PD][ENC_UNROLL / 4];
  auto request = [&](int s, float4 (&xd)[ENC_UNROLL / 2], uint4 (&dd)[ENC_UNROLL / 4]) {
#if NT_ENC_DIAG & 16   /* timing-only: texel centres and gradients from a small window (L2-resident) */
    const int sc = (s < s_max ? s : s_max) & 0x1fff;
#else
    const int sc = s < s_max ? s : s_max;
#endif
#if NT_ENC_DIAG & 32   /* timing-only: only the texel centres from a window */
    const float4* xp = reinterpret_cast<const float4*>(slot_xy + (sc & 0x1fff));
#else
    const float4* xp = reinterpret_cast<const float4*>(slot_xy + sc);
#endif
    const uint4* dp = reinterpret_cast<const uint4*>(dFw + nt_feat_in_plane(nl, sc));
#pragma unroll
    for (int i = 0; i < ENC_UNROLL / 2; ++i) xd[i] = xp[i];
#pragma unroll
    for (int i = 0; i < ENC_UNROLL / 4; ++i) dd[i] = dp[i];
  };
#pragma unroll
  for (int d = 0; d < PD; ++d)
    request(a_first + threadIdx.x * ENC_UNROLL + d * ENC_BLOCK * ENC_UNROLL, xyn[d], dn[d]);
#endif
  for (int s0 = a_first + threadIdx.x * ENC_UNROLL; s0 < last; s0 += ENC_BLOCK * ENC_UNROLL) {
    float4 xyv[ENC_UNROLL / 2];
    uint4 dv[ENC_UNROLL / 4];
#if NT_ENC_PREFETCH
#pragma unroll
    for (int i = 0; i < ENC_UNROLL / 2; ++i) xyv[i] = xyn[0][i];
#pragma unroll
    for (int i = 0; i < ENC_UNROLL / 4; ++i) dv[i] = dn[0][i];
#pragma unroll
    for (int d = 0; d + 1 < PD; ++d) {
#pragma unroll
      for (int i = 0; i < ENC_UNROLL / 2; ++i) xyn[d][i] = xyn[d + 1][i];
#pragma unroll
      for (int i = 0; i < ENC_UNROLL / 4; ++i) dn[d][i] = dn[d + 1][i];
    }
    request(s0 + PD * ENC_BLOCK * ENC_UNROLL, xyn[PD - 1], dn[PD - 1]);
#else
    const float4* xp = reinterpret_cast<const float4*>(slot_xy + s0);
    const uint4* dp = reinterpret_cast<const uint4*>(dFw + nt_feat_in_plane(nl, s0));
#pragma unroll
    for (int i = 0; i < ENC_UNROLL / 2; ++i) xyv[i] = xp[i];
#pragma unroll
    for (int i = 0; i < ENC_UNROLL / 4; ++i) dv[i] = dp[i];
#endif
    const unsigned dws[ENC_UNROLL] = {dv[0].x, dv[0].y, dv[0].z, dv[0].w,
                                      dv[1].x, dv[1].y, dv[1].z, dv[1].w};
    unsigned cur_idx[4] = {0, 0, 0, 0};
    int acc[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f][0] = acc[f][1] = acc[f][2] = acc[f][3] = 0;
    unsigned cur_cx = 0xffffffffu, cur_cy = 0xffffffffu;
#pragma unroll
    for (int u = 0; u < ENC_UNROLL; ++u) {
      const int slot = s0 + u;
      const float x = (u & 1) ? xyv[u >> 1].z : xyv[u >> 1].x;
      const float y = (u & 1) ? xyv[u >> 1].w : xyv[u >> 1].y;
      float gv[NF];
      bool nz = false;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
#if NT_ENC_MIX
        // conversion and scaling in one instruction (enc_mul_mix); NF = 2: the two halves of the
        // word, NF = 1: the word shifted so that feature `feat` is the low half
        gv[f] = NF == 2 ? (f ? enc_mul_mix<1>(dws[u], S[f]) : enc_mul_mix<0>(dws[u], S[f]))
                        : enc_mul_mix<0>(dws[u] >> (16 * feat), S[f]);
#else
        const unsigned short hb = (unsigned short)(dws[u] >> (16 * (feat + f)));
        gv[f] = (float)__builtin_bit_cast(_Float16, hb) * S[f];
#endif
        nz |= gv[f] != 0.f;
      }
      if (slot >= first && slot < last && nz) {
        const CellRef cr = cell_ref(g, x, y);
        int v[NF][4];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const f32x2 p01 = cr.w01 * gv[f], p23 = cr.w23 * gv[f];
#if NT_ENC_FRACT
          v[f][0] = enc_round_i(p01.x), v[f][1] = enc_round_i(p01.y);
          v[f][2] = enc_round_i(p23.x), v[f][3] = enc_round_i(p23.y);
#else
          v[f][0] = __float2int_rn(p01.x), v[f][1] = __float2int_rn(p01.y);
          v[f][2] = __float2int_rn(p23.x), v[f][3] = __float2int_rn(p23.y);
#endif
        }
        if (!MERGE) {   // cells are finer than texels: every slot has its own cell
          unsigned idx[4];
          cell_indices<HASHED>(g, cr.cx, cr.cy, idx);
#pragma unroll
          for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int k = 0; k < 4; ++k) ENC_LDS_ADD(&my_g[f * plane + idx[k]], v[f][k]);
        } else if (cr.cx == cur_cx && cr.cy == cur_cy) {
#pragma unroll
          for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[f][k] += v[f][k];
        } else {
          if (cur_cx != 0xffffffffu) {
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
              for (int k = 0; k < 4; ++k) ENC_LDS_ADD(&my_g[f * plane + cur_idx[k]], acc[f][k]);
          }
          cur_cx = cr.cx, cur_cy = cr.cy;
          cell_indices<HASHED>(g, cr.cx, cr.cy, cur_idx);
#pragma unroll
          for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[f][k] = v[f][k];
        }
      }
    }
    if (MERGE && cur_cx != 0xffffffffu) {
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int k = 0; k < 4; ++k) ENC_LDS_ADD(&my_g[f * plane + cur_idx[k]], acc[f][k]);
    }
  }
#if NT_ENC_PREFETCH
  // retire the last (unused) request here: left pending, its landing registers made the
  // compiler put an s_waitcnt vmcnt(0) in front of every register it reuses in the flush
  // below — i.e. in front of every flush atomic, which then waited for the previous one
#pragma unroll
  for (int d = 0; d < PD; ++d) {
#pragma unroll
    for (int i = 0; i < ENC_UNROLL / 2; ++i)
      asm volatile("" ::"v"(xyn[d][i].x), "v"(xyn[d][i].y), "v"(xyn[d][i].z), "v"(xyn[d][i].w));
#pragma unroll
    for (int i = 0; i < ENC_UNROLL / 4; ++i)
      asm volatile("" ::"v"(dn[d][i].x), "v"(dn[d][i].y), "v"(dn[d][i].z), "v"(dn[d][i].w));
  }
#endif
  __syncthreads();
#if NT_ENC_DIAG & 1
  return;
#endif
  float* gt = grad_tables + ((long long)nt_param_tex(plan, tex) * n_entries + plan.level_offset[level]) * 2 + feat;
#if NT_ENC_FLUSH_BATCH
  // FB entries per thread and trip: their table values are all requested before the first one is
  // waited for.  (One entry at a time — LDS read, test, global load, add, store, with a branch in
  // between — is a serial global round trip per entry: 32 of them per thread and 128 KiB plane,
  // i.e. most of the 17-40 us a piece costs before it has touched a slot.)
  constexpr int FB = NT_ENC_FLUSH_BATCH;
  auto plane_sums = [&](int i0, int (&vi)[FB][NF]) {
#pragma unroll
    for (int b = 0; b < FB; ++b) {
      const int i = i0 + b * ENC_BLOCK;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        int acc = 0;
        if (i < (int)g.size)
          for (int cpy = 0; cpy < copies; ++cpy) acc += s_g[f * plane + cpy * g.size + i];
        vi[b][f] = acc;
      }
    }
  };
  // two SEPARATE loops: with both modes in one loop body the compiler's wait-count bookkeeping
  // carried the other mode's pending loads around the back edge and put an s_waitcnt vmcnt(0)
  // in front of every atomic — each one then waited for the previous one's acknowledgement
  if (store) {
    // sole writer of this (texture, level, feature) plane AND the caller vouches that grad_tables was zero on
    // entry (plan.grads_zeroed): plain stores.  A float atomic retires in the L2 at about one LANE per clock
    // and channel — a 32 768-entry plane is 32 K of them, and at a training batch (49 k hits: 35 k slots per
    // plane) the flushes were 0.136 of the launch's 0.231 ms (timing-only build, profiles/NOTEBOOK.md r5);
    // ~700 of the ~1 215 pieces of a launch are sole writers whatever the batch (960 planes, 255 cuts).
    for (int i0 = threadIdx.x; i0 < (int)g.size; i0 += ENC_BLOCK * FB) {
      int vi[FB][NF];
      plane_sums(i0, vi);
#pragma unroll
      for (int b = 0; b < FB; ++b) {
        const long long i = i0 + b * ENC_BLOCK;
        if constexpr (NF == 2) {      // both features of an entry: one 8-byte store (feat = 0: 8-byte aligned)
          if ((vi[b][0] | vi[b][1]) != 0)
            *reinterpret_cast<float2*>(&gt[2 * i]) = make_float2((float)vi[b][0] * S_inv[0], (float)vi[b][1] * S_inv[1]);
        } else {
          if (vi[b][0] != 0) gt[2 * i] = (float)vi[b][0] * S_inv[0];
        }
      }
    }
  } else if (single) {        // sole writer of this (texture, level, feature) plane: plain read-modify-write
    for (int i0 = threadIdx.x; i0 < (int)g.size; i0 += ENC_BLOCK * FB) {
      int vi[FB][NF];
      float old[FB][NF];
      plane_sums(i0, vi);
#pragma unroll
      for (int b = 0; b < FB; ++b)
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          old[b][f] = 0.f;
          if (vi[b][f] != 0) old[b][f] = gt[2 * (long long)(i0 + b * ENC_BLOCK) + f];
        }
#pragma unroll
      for (int b = 0; b < FB; ++b)
#pragma unroll
        for (int f = 0; f < NF; ++f)
          if (vi[b][f] != 0) gt[2 * (long long)(i0 + b * ENC_BLOCK) + f] = old[b][f] + (float)vi[b][f] * S_inv[f];
    }
  } else {
    for (int i0 = threadIdx.x; i0 < (int)g.size; i0 += ENC_BLOCK * FB) {
      int vi[FB][NF];
      plane_sums(i0, vi);
#pragma unroll
      for (int b = 0; b < FB; ++b)
#pragma unroll
        for (int f = 0; f < NF; ++f)
          if (vi[b][f] != 0) atomicAdd(&gt[2 * (long long)(i0 + b * ENC_BLOCK) + f], (float)vi[b][f] * S_inv[f]);
    }
  }
#else
  for (int i = threadIdx.x; i < (int)g.size; i += ENC_BLOCK) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      int vi = 0;
      for (int cpy = 0; cpy < copies; ++cpy) vi += s_g[f * plane + cpy * g.size + i];
      if (vi == 0) continue;
      const float v = (float)vi * S_inv[f];
      if (store) {
        gt[2 * (long long)i + f] = v;
      } else if (single) {
        gt[2 * (long long)i + f] += v;  // sole writer of this (texture, level, feature) plane
      } else {
        atomicAdd(&gt[2 * (long long)i + f], v);
      }
    }
  }
#endif
}

// Planes of a launch over levels [level0, level0 + n_levels): a dense level whose two
// feature planes fit the LDS together is ONE plane (both features per pass), any other
// level is two (one per feature).
__device__ __host__ inline bool enc_both_features(const vsa_nt_plan& p, int level, bool hashed) {
  return !hashed && (long long)p.level_size[level] * 2 <= LDS_ENTRIES;
}

template <bool HASHED>
__device__ __forceinline__ void nt_encode_bwd_body(
    const vsa_nt_plan& plan, int level0, int n_levels, int n_planes,
    const half2_t* __restrict__ dfeatures, const float* __restrict__ dfeat_abs_sum,
    float dscale_inv, const float2* __restrict__ slot_xy, const int* __restrict__ seg_start,
    float* __restrict__ grad_tables, int tex_begin, int tex_end, unsigned char* s_raw) {
  int* s_g = reinterpret_cast<int*>(s_raw);
  NT_SPAN_MARK(HASHED ? 3 : 2, 0);
  NT_BAL_BEGIN();
  const bool full_range = tex_begin <= 0 && tex_end >= plan.nr_shells * 2 * VSA_NT_MAX_DEG;
  // fitted (tools/fit_cost.py): 145 (hashed) / 64 (dense) units per piece (zeroing + flushing the
  // LDS plane), and a unit on the no-merge path (level finer than the texture) costs 0.875
  auto unit_weight = [&](int pl, int deg, int) {
    return !HASHED || plan.level_scale[level0 + (pl >> 1)] < (float)plan.tex_res[deg] ? 16 : 14;
  };
  nt_for_each_piece<ENC_UNIT>(plan, seg_start, n_planes, HASHED ? NT_ENC_OVH_BH : NT_ENC_OVH_BD,
                              [&](int pl, int tex, int first, int last, int seg_begin, int seg_end) {
    int level = level0, r = pl;
    bool both = enc_both_features(plan, level, HASHED);
    while (r >= (both ? 1 : 2)) {
      r -= both ? 1 : 2;
      ++level;
      both = enc_both_features(plan, level, HASHED);
    }
#if NT_ENC_FLUSH_ATOMIC   /* every flush through no-return float atomics: fire and forget, where the sole-writer form
                            waits for a batch of table entries to come back before it can add and store them.
                            A sole writer adds exactly once per entry, so its result is the same old + v either way */
    const bool single = false;
#else
    const bool single = first == seg_begin && last == seg_end &&
                        !((((tex / VSA_NT_MAX_DEG) & 1) ? plan.shared_alpha : plan.shared_rgb) != 0);
#endif
    // ... unless the caller vouches for a zero gradient buffer: then a sole writer's plane is simply stored
    // (a plane that K shells share — plan.shared_rgb / shared_alpha — has K writers whoever walks the segment)
    const bool shared_plane = (((tex / VSA_NT_MAX_DEG) & 1) ? plan.shared_alpha : plan.shared_rgb) != 0;
    const bool store = plan.grads_zeroed != 0 && first == seg_begin && last == seg_end && !shared_plane;
    // neighbouring texels are scale / R cells apart: from one cell per texel on, the
    // in-register merging of same-cell slots cannot fire and its bookkeeping is skipped
    const bool merge = plan.level_scale[level] < (float)plan.tex_res[tex % VSA_NT_MAX_DEG];
    if (both)
      enc_bwd_piece<HASHED, 2, true>(plan, s_g, level, 0, tex, first, last, single, store, dfeatures,
                                     dfeat_abs_sum, dscale_inv, slot_xy, grad_tables);
    else if (merge)
      enc_bwd_piece<HASHED, 1, true>(plan, s_g, level, r, tex, first, last, single, store, dfeatures,
                                     dfeat_abs_sum, dscale_inv, slot_xy, grad_tables);
    else
      enc_bwd_piece<HASHED, 1, false>(plan, s_g, level, r, tex, first, last, single, store, dfeatures,
                                      dfeat_abs_sum, dscale_inv, slot_xy, grad_tables);
  }, tex_begin, tex_end, unit_weight,
     // a launch over a sub-range of the textures (the sliced backward of parallel.py) keeps equal shares
     full_range ? (HASHED ? NT_BAL_ENC_BWD_H : NT_BAL_ENC_BWD_D) : -1);
  __syncthreads();
  NT_SPAN_MARK(HASHED ? 3 : 2, 1);
  if (full_range) { NT_BAL_END(HASHED ? NT_BAL_ENC_BWD_H : NT_BAL_ENC_BWD_D); }
}

template <bool HASHED>
__global__ __launch_bounds__(ENC_BLOCK) void nt_encode_bwd_kernel(
    vsa_nt_plan plan, int level0, int n_levels, int n_planes,
    const half2_t* __restrict__ dfeatures, const float* __restrict__ dfeat_abs_sum,
    float dscale_inv, const float2* __restrict__ slot_xy, const int* __restrict__ seg_start,
    float* __restrict__ grad_tables, int tex_begin, int tex_end) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  nt_encode_bwd_body<HASHED>(plan, level0, n_levels, n_planes, dfeatures, dfeat_abs_sum, dscale_inv, slot_xy,
                             seg_start, grad_tables, tex_begin, tex_end, s_raw);
}

// dense planes, then hashed planes, in one launch (as nt_encode_fwd_both_kernel)
__global__ __launch_bounds__(ENC_BLOCK) void nt_encode_bwd_both_kernel(
    vsa_nt_plan plan, int lh, int n_planes_dense,
    const half2_t* __restrict__ dfeatures, const float* __restrict__ dfeat_abs_sum,
    float dscale_inv, const float2* __restrict__ slot_xy, const int* __restrict__ seg_start,
    float* __restrict__ grad_tables, int tex_begin, int tex_end) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  nt_encode_bwd_body<false>(plan, 0, lh, n_planes_dense, dfeatures, dfeat_abs_sum, dscale_inv, slot_xy,
                            seg_start, grad_tables, tex_begin, tex_end, s_raw);
  __syncthreads();
  nt_encode_bwd_body<true>(plan, lh, plan.n_levels - lh, 2 * (plan.n_levels - lh), dfeatures, dfeat_abs_sum,
                           dscale_inv, slot_xy, seg_start, grad_tables, tex_begin, tex_end, s_raw);
}

// The data-parallel form (vsa_nt_encode_bwd_phased): every workgroup first walks its share of the dense
// planes of ALL shells (a quarter of the work in many small pieces; un-phased), then the shells' hashed
// planes PHASE BY PHASE — its share of phase 0, then of phase 1, ... — so the table gradients of a phase's
// shells are final roughly (1 + 3 (p + 1) / n_phases) / 4 into the ONE launch instead of all at its end.  A
// workgroup that has finished its share of a phase makes its table atomics visible (agent-scope release)
// and counts itself in; the last one publishes flags[phase] = *epoch at system scope, which a wait on
// another stream (vsa_dp_stream_wait) releases that phase's all-reduce on.  No workgroup ever waits for
// another: a phase boundary is not a barrier.  Price on one MI355X (800x800, K = 5; the walk re-splits
// every phase over all 256 workgroups, i.e. 160 + 256 instead of 160 pieces per shell, each with its own
// LDS plane to zero and to flush): 1 / 2 / 3 / 5 phases = 0.60 / 0.64 / 0.68 / 0.75 ms.
struct EncPhases {
  int n;
  int shell_end[VSA_MAX_SHELLS];
  unsigned* flag[VSA_MAX_SHELLS];      // one word per phase (each its own signal allocation: vsa_dp_flags)
};

__global__ __launch_bounds__(ENC_BLOCK) void nt_encode_bwd_phased_kernel(
    vsa_nt_plan plan, int lh, int n_planes_dense,
    const half2_t* __restrict__ dfeatures, const float* __restrict__ dfeat_abs_sum,
    float dscale_inv, const float2* __restrict__ slot_xy, const int* __restrict__ seg_start,
    float* __restrict__ grad_tables, EncPhases ph,
    unsigned* __restrict__ counters, const unsigned* __restrict__ epoch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  // the dense (coarse) planes of ALL shells first, un-phased: they are a quarter of the work in many small
  // pieces, and a phase is final once its hashed planes are — the dense ones are long done by then
  if (lh > 0)
    nt_encode_bwd_body<false>(plan, 0, lh, n_planes_dense, dfeatures, dfeat_abs_sum, dscale_inv, slot_xy,
                              seg_start, grad_tables, 0, 1 << 30, s_raw);
  __syncthreads();
  int shell0 = 0;
  for (int p = 0; p < ph.n; ++p) {
    const int tex_begin = shell0 * 2 * VSA_NT_MAX_DEG, tex_end = ph.shell_end[p] * 2 * VSA_NT_MAX_DEG;
    shell0 = ph.shell_end[p];
    if (lh < plan.n_levels)
      nt_encode_bwd_body<true>(plan, lh, plan.n_levels - lh, 2 * (plan.n_levels - lh), dfeatures, dfeat_abs_sum,
                               dscale_inv, slot_xy, seg_start, grad_tables, tex_begin, tex_end, s_raw);
    // every wave's flush atomics have been acknowledged (vmcnt 0) before the workgroup counts itself in; ONE
    // wave then issues the agent-scope release (as a fence in all 16 waves it cost 0.16 ms per phase:
    // 4 096 L2 write-back requests per phase chip-wide, profiles/r05/dp_schedule_one_gpu.txt)
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      const unsigned done = __hip_atomic_fetch_add(&counters[p], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (done == gridDim.x - 1) {
        __hip_atomic_store(&counters[p], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        __hip_atomic_store(ph.flag[p], *epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// ---- the small stream-side pieces of the same protocol
__global__ void dp_signal_kernel(unsigned* __restrict__ flag, unsigned* __restrict__ epoch, int advance) {
  unsigned e = *epoch;
  if (advance) *epoch = ++e;
  __hip_atomic_store(flag, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void dp_spin_wait_kernel(const unsigned* __restrict__ flag, unsigned value) {
  // (one lane, no LDS: fits beside the persistent kernels, which leave half of a CU's wave slots free)
  // RELAXED polls (a cache-bypassing load each), ONE acquire when the value is there: an acquire per poll
  // invalidates the caches of the XCD the lane sits on every couple of microseconds, under the kernels it waits for
  while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0)
    __builtin_amdgcn_s_sleep(64);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

}  // namespace

static bool level_hashed(const vsa_nt_plan* p, int l) {
  const long long res = p->level_res[l], size = p->level_size[l];
  return !(res <= size && res * res <= size);  // tiny-cuda-nn grid_index
}

// Dense (coarse) levels first, hashed levels after: the multiresolution grid is
// monotone, which the two-launch split relies on.
static int first_hashed_level(const vsa_nt_plan* p) {
  int l = 0;
  while (l < p->n_levels && !level_hashed(p, l)) ++l;
  return l;
}

static int check_levels(const vsa_nt_plan* p) {
  const int lh = first_hashed_level(p);
  for (int l = lh; l < p->n_levels; ++l) {
    if (!level_hashed(p, l)) return VSA_ERR_UNSUPPORTED;
    const int sz = p->level_size[l];
    if ((sz & (sz - 1)) != 0 || sz > LDS_ENTRIES) return VSA_ERR_UNSUPPORTED;
  }
  for (int l = 0; l < lh; ++l)
    if (p->level_size[l] > LDS_ENTRIES) return VSA_ERR_UNSUPPORTED;
  return VSA_OK;
}

static int max_level_size(const vsa_nt_plan* p, int l0, int l1) {
  int m = 0;
  for (int l = l0; l < l1; ++l) m = m > p->level_size[l] ? m : p->level_size[l];
  return m;
}

template <typename K>
static int set_lds_attr(K kernel) {
  VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
  return VSA_OK;
}

#ifdef NT_SPAN
extern "C" int vsa_span_read_encode(void* dst) {
  VSA_HIP_TRY(hipDeviceSynchronize());
  VSA_HIP_TRY(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_span), sizeof(g_span)));
  return 0;
}
#endif

extern "C" int vsa_nt_encode_fwd(const vsa_nt_plan* plan, const void* tables_h,
                                 const float* slot_xy, const int32_t* seg_start, void* features,
                                 void* stream) {
  if (!plan || !tables_h || !slot_xy || !seg_start || !features) return VSA_ERR_ARG;
  int rc = check_levels(plan);
  if (rc) return rc;
  static bool attr_set = false;
  if (!attr_set) {
    if ((rc = set_lds_attr(nt_encode_fwd_kernel<false>))) return rc;
    if ((rc = set_lds_attr(nt_encode_fwd_kernel<true>))) return rc;
    if ((rc = set_lds_attr(nt_encode_fwd_both_kernel))) return rc;
    attr_set = true;
  }
  const int lh = first_hashed_level(plan);
  const half2_t* tab = reinterpret_cast<const half2_t*>(tables_h);
  const float2* xy = reinterpret_cast<const float2*>(slot_xy);
  half2_t* out = reinterpret_cast<half2_t*>(features);
  int nr_cus = 0;
  if ((rc = vsa_cu_count(&nr_cus))) return rc;
  if (NT_ENC_ONE_LAUNCH && lh > 0 && lh < plan->n_levels) {
    hipLaunchKernelGGL(nt_encode_fwd_both_kernel, dim3(nr_cus), dim3(ENC_BLOCK), (size_t)LDS_ENTRIES * 4,
                       (hipStream_t)stream, *plan, lh, tab, xy, seg_start, out);
    VSA_RETURN_LAUNCH_STATUS();
  }
  if (lh > 0)
    hipLaunchKernelGGL(nt_encode_fwd_kernel<false>, dim3(nr_cus), dim3(ENC_BLOCK),
                       (size_t)LDS_ENTRIES * 4, (hipStream_t)stream, *plan, 0, lh, tab, xy,
                       seg_start, out);
  if (lh < plan->n_levels)
    hipLaunchKernelGGL(nt_encode_fwd_kernel<true>, dim3(nr_cus), dim3(ENC_BLOCK),
                       (size_t)LDS_ENTRIES * 4, (hipStream_t)stream, *plan, lh, plan->n_levels - lh,
                       tab, xy, seg_start, out);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_nt_encode_bwd(const vsa_nt_plan* plan, const void* dfeatures,
                                 const float* dfeat_abs_sum, float grad_scale,
                                 const float* slot_xy, const int32_t* seg_start,
                                 float* grad_tables, void* stream) {
  if (!plan) return VSA_ERR_ARG;
  return vsa_nt_encode_bwd_range(plan, dfeatures, dfeat_abs_sum, grad_scale, slot_xy, seg_start,
                                 grad_tables, 0, plan->nr_shells, stream);
}

extern "C" int vsa_nt_encode_bwd_range(const vsa_nt_plan* plan, const void* dfeatures,
                                       const float* dfeat_abs_sum, float grad_scale,
                                       const float* slot_xy, const int32_t* seg_start,
                                       float* grad_tables, int shell_begin, int shell_end,
                                       void* stream) {
  if (!plan || !dfeatures || !dfeat_abs_sum || !slot_xy || !seg_start || !grad_tables)
    return VSA_ERR_ARG;
  if (shell_begin < 0 || shell_end > plan->nr_shells || shell_begin > shell_end) return VSA_ERR_ARG;
  if (shell_begin == shell_end) return VSA_OK;
  // shared models: every shell writes shell 0's planes, so "the slice of these shells is final" holds for the whole range only
  if ((plan->shared_rgb || plan->shared_alpha) && !(shell_begin == 0 && shell_end == plan->nr_shells)) return VSA_ERR_UNSUPPORTED;
  const int tex_begin = shell_begin * 2 * VSA_NT_MAX_DEG, tex_end = shell_end * 2 * VSA_NT_MAX_DEG;
  if (!(grad_scale > 0.f)) return VSA_ERR_ARG;
  int rc = check_levels(plan);
  if (rc) return rc;
  static bool attr_set = false;
  if (!attr_set) {
    if ((rc = set_lds_attr(nt_encode_bwd_kernel<false>))) return rc;
    if ((rc = set_lds_attr(nt_encode_bwd_kernel<true>))) return rc;
    if ((rc = set_lds_attr(nt_encode_bwd_both_kernel))) return rc;
    attr_set = true;
  }
  const int lh = first_hashed_level(plan);
  const half2_t* dF = reinterpret_cast<const half2_t*>(dfeatures);
  const float2* xy = reinterpret_cast<const float2*>(slot_xy);
  int nr_cus = 0;
  if ((rc = vsa_cu_count(&nr_cus))) return rc;
  if (NT_ENC_ONE_LAUNCH && lh > 0 && lh < plan->n_levels) {
    int n_planes = 0;
    for (int l = 0; l < lh; ++l) n_planes += enc_both_features(*plan, l, false) ? 1 : 2;
    hipLaunchKernelGGL(nt_encode_bwd_both_kernel, dim3(nr_cus), dim3(ENC_BLOCK), (size_t)LDS_ENTRIES * 4,
                       (hipStream_t)stream, *plan, lh, n_planes, dF, dfeat_abs_sum, 1.0f / grad_scale, xy,
                       seg_start, grad_tables, tex_begin, tex_end);
    VSA_RETURN_LAUNCH_STATUS();
  }
  if (lh > 0) {
    int n_planes = 0;
    for (int l = 0; l < lh; ++l) n_planes += enc_both_features(*plan, l, false) ? 1 : 2;
    hipLaunchKernelGGL(nt_encode_bwd_kernel<false>, dim3(nr_cus), dim3(ENC_BLOCK),
                       (size_t)LDS_ENTRIES * 4, (hipStream_t)stream, *plan, 0, lh, n_planes, dF,
                       dfeat_abs_sum, 1.0f / grad_scale, xy, seg_start, grad_tables, tex_begin,
                       tex_end);
  }
  if (lh < plan->n_levels)
    hipLaunchKernelGGL(nt_encode_bwd_kernel<true>, dim3(nr_cus), dim3(ENC_BLOCK),
                       (size_t)LDS_ENTRIES * 4, (hipStream_t)stream, *plan, lh, plan->n_levels - lh,
                       2 * (plan->n_levels - lh), dF, dfeat_abs_sum, 1.0f / grad_scale, xy,
                       seg_start, grad_tables, tex_begin, tex_end);
  VSA_RETURN_LAUNCH_STATUS();
}

// Completion flags of the data-parallel step.  A stream that waits on a flag with a POLLING KERNEL parks a
// wave on some CU — and nt_mlp_fwd / nt_mlp_bwd take every vector register of every SIMD, so that CU could
// not host its persistent workgroup any more: the launch ran two rounds instead of one, +0.5-0.7 ms per step
// (profiles/r05/dp_schedule_one_gpu.txt; HIP's own hipStreamWaitValue32 on plain device memory is such a
// kernel too).  Words allocated with hipMallocSignalMemory are waited for by the command processor itself
// (a barrier-value packet): no wave, no register, no LDS.
struct vsa_dp_flags {
  int n = 0;
  bool signal_memory = false;
  unsigned* word[VSA_MAX_SHELLS + 1] = {};
};

extern "C" int vsa_dp_flags_create(int n, vsa_dp_flags** out) {
  if (!out || n < 1 || n > VSA_MAX_SHELLS + 1) return VSA_ERR_ARG;
  vsa_dp_flags* f = new vsa_dp_flags();
  f->n = n;
  f->signal_memory = true;
  for (int i = 0; i < n; ++i) {
    void* p = nullptr;
    if (f->signal_memory && hipExtMallocWithFlags(&p, 8, hipMallocSignalMemory) != hipSuccess) {
      (void)hipGetLastError();
      f->signal_memory = false;            // (then every word is plain memory: one kind of wait for all)
      for (int j = 0; j < i; ++j) (void)hipFree(f->word[j]);
      i = -1;
      continue;
    }
    if (!f->signal_memory) VSA_HIP_TRY(hipMalloc(&p, 8));
    VSA_HIP_TRY(hipMemset(p, 0, 8));
    f->word[i] = static_cast<unsigned*>(p);
  }
  *out = f;
  return VSA_OK;
}

extern "C" int vsa_dp_flags_destroy(vsa_dp_flags* f) {
  if (!f) return VSA_OK;
  for (int i = 0; i < f->n; ++i) (void)hipFree(f->word[i]);
  delete f;
  return VSA_OK;
}

extern "C" int vsa_dp_flags_read(const vsa_dp_flags* f, uint32_t* host_out) {
  if (!f || !host_out) return VSA_ERR_ARG;
  VSA_HIP_TRY(hipDeviceSynchronize());
  for (int i = 0; i < f->n; ++i) VSA_HIP_TRY(hipMemcpy(host_out + i, f->word[i], 4, hipMemcpyDeviceToHost));
  return VSA_OK;
}

extern "C" int vsa_nt_encode_bwd_phased(const vsa_nt_plan* plan, const void* dfeatures,
                                        const float* dfeat_abs_sum, float grad_scale,
                                        const float* slot_xy, const int32_t* seg_start,
                                        float* grad_tables, int n_phases, const int32_t* phase_shell_end,
                                        vsa_dp_flags* flags, uint32_t* counters, const uint32_t* epoch,
                                        int reserve_cus, void* stream) {
  if (!plan || !dfeatures || !dfeat_abs_sum || !slot_xy || !seg_start || !grad_tables || !phase_shell_end ||
      !flags || !counters || !epoch)
    return VSA_ERR_ARG;
  if (n_phases < 1 || n_phases > VSA_MAX_SHELLS || flags->n < n_phases || !(grad_scale > 0.f)) return VSA_ERR_ARG;
  if ((plan->shared_rgb || plan->shared_alpha) && n_phases != 1) return VSA_ERR_UNSUPPORTED;   // (see vsa_nt_encode_bwd_range)
  EncPhases ph;
  int prev = 0;
  for (int p = 0; p < n_phases; ++p) {       // strictly increasing, the last phase ends at the last shell
    if (phase_shell_end[p] <= prev || phase_shell_end[p] > plan->nr_shells) return VSA_ERR_ARG;
    ph.shell_end[p] = prev = phase_shell_end[p];
    ph.flag[p] = flags->word[p];
  }
  if (prev != plan->nr_shells) return VSA_ERR_ARG;
  ph.n = n_phases;
  int rc = check_levels(plan);
  if (rc) return rc;
  static bool attr_set = false;
  if (!attr_set) {
    if ((rc = set_lds_attr(nt_encode_bwd_phased_kernel))) return rc;
    attr_set = true;
  }
  const int lh = first_hashed_level(plan);
  int n_planes = 0;
  for (int l = 0; l < lh; ++l) n_planes += enc_both_features(*plan, l, false) ? 1 : 2;
  int nr_cus = 0;
  if ((rc = vsa_cu_count(&nr_cus))) return rc;
  if (reserve_cus < 0 || reserve_cus >= nr_cus) return VSA_ERR_ARG;
  nr_cus -= reserve_cus;       // workgroups = compute units left to this launch (the rest: the collectives' kernels)
  hipLaunchKernelGGL(nt_encode_bwd_phased_kernel, dim3(nr_cus), dim3(ENC_BLOCK), (size_t)LDS_ENTRIES * 4,
                     (hipStream_t)stream, *plan, lh, n_planes, reinterpret_cast<const half2_t*>(dfeatures),
                     dfeat_abs_sum, 1.0f / grad_scale, reinterpret_cast<const float2*>(slot_xy), seg_start,
                     grad_tables, ph, counters, epoch);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_dp_signal(vsa_dp_flags* flags, int index, uint32_t* epoch, int advance_epoch, void* stream) {
  if (!flags || !epoch || index < 0 || index >= flags->n) return VSA_ERR_ARG;
  hipLaunchKernelGGL(dp_signal_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flags->word[index], epoch,
                     advance_epoch);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_dp_stream_wait(vsa_dp_flags* flags, int index, uint32_t value, int mode, void* stream) {
  if (!flags || index < 0 || index >= flags->n || mode < 0 || mode > 2) return VSA_ERR_ARG;
  if (mode == 0) {      // auto: the command processor's wait where the words are signal memory and the device offers it
    int dev = 0, can = 0;
    VSA_HIP_TRY(hipGetDevice(&dev));
    VSA_HIP_TRY(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev));
    mode = can && flags->signal_memory ? 1 : 2;
  }
  if (mode == 1) {
    VSA_HIP_TRY(hipStreamWaitValue32((hipStream_t)stream, flags->word[index], value, hipStreamWaitValueGte, 0xffffffffu));
    return VSA_OK;
  }
  hipLaunchKernelGGL(dp_spin_wait_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flags->word[index], value);
  VSA_RETURN_LAUNCH_STATUS();
}
