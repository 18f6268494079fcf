// Neural-texture steps 3 + 4 in ONE launch: 2-D multiresolution hash-grid encoding of every
// unique texel, fed straight into the 32->64->64->C' MLP on MFMA, fused with the reference's
// post-processing (sigmoid, x255, round).
//
// The reference evaluates `Sequential(encoding, network)` as one tiny-cuda-nn call per corner
// batch (models/neural_texture.py:63-79, 153).  nt_encode.hip + nt_mlp.hip split that into a
// VALU-bound kernel (matrix pipe idle) and an MFMA kernel (VALU 39 % busy) joined by a 64 B per
// (slot, texture) feature buffer in HBM.  Here a wave takes 64 consecutive slots of one texture:
//
//   encode   every lane owns ONE slot and evaluates all 16 levels for it: the level constants are
//            wave-uniform (scalar registers), the 64 lanes are 64 neighbouring texels of a texture
//            row, so the four corner gathers of a level fall into a few cache lines of the
//            texture's f16 table (the hash leaves x un-multiplied: cx ^ cy*PRIME, so a run of
//            texels maps to a permuted run of entries).  Tables are gathered from L2, not LDS: a
//            slot needs all 16 levels at once (1.4 MB per texture), so the blockIdx -> work map
//            keeps every XCD on its own eighth of the cost axis and lets all waves of an XCD
//            sweep it together: about one texture's tables are live in an XCD's 4 MiB L2.
//   hand-off the lane of slot i holds its 16 level features; the MFMA B operand of a 32-slot
//            tile wants lane (p, h) to hold levels 8s + 4h + i of slot p.  One
//            v_permlane32_swap per register pair re-deals the two 32-slot tiles of the wave
//            (lower half keeps levels l, sends l + 4; upper half the other way round): 8 swaps
//            per 64 slots, no LDS, no select.
//   MLP      mlp_tile_fwd on tile A then tile B (weights register-resident), epilogue
//            quant_store_tile: exactly the code of nt_mlp_fwd.
//
// Arithmetic is nt_encode_fwd's and nt_mlp_fwd's, operation for operation: features and texels
// are bit-identical to the two-kernel path (tests/test_nt_fused.py).
// FEAT: also store the feature planes (the training step's backward recomputes the forward from
// them); without it nothing but the 8-byte texel centre is read and the texel row written.
#include <type_traits>

#include "nt_common.h"
#include "nt_enc_common.h"
#include "nt_mlp_common.h"
#include "nt_quant_table.h"

namespace {

constexpr int FU_BLOCK = 256;
constexpr int FU_WAVES = FU_BLOCK / 64;
#ifndef NT_FU_WGS_PER_CU
#define NT_FU_WGS_PER_CU 2
#endif
constexpr int FU_WGS_PER_CU = NT_FU_WGS_PER_CU;
#ifndef NT_FU_BATCH
#define NT_FU_BATCH 4            /* levels whose 4 x NT_FU_BATCH gathers are in flight together */
#endif
constexpr int FU_BATCH = NT_FU_BATCH;
#ifndef NT_FU_W_PER_GROUP
#define NT_FU_W_PER_GROUP 1      /* 1/16 unit per further 8-channel output group */
#endif
#ifndef NT_FU_DIAG
#define NT_FU_DIAG 0             /* diagnostic builds (WRONG results): 1 no table gathers, 2 no MLP, 4 no XCD grouping */
#endif

// Row-run staging (NT_FU_STAGE=1; built, bit-exact, measured SLOWER — off by default).
// A 64-lane dword gather costs the vector cache at least 16 tag look-ups (it works a quad of lanes
// at a time: PMC TCP_TOTAL_CACHE_ACCESSES = 25 per gather instruction here, 299 M per launch),
// however few lines the lanes share.  But the wave's 64 slots are neighbouring texels of ONE
// texture row: at a level they touch the cells cx0 .. cx63 + 1 of the grid rows cy and cy + 1, and
// the hash leaves x un-multiplied, so cells 4B .. 4B + 3 of a row are the aligned 16-byte block
// ((4B ^ h) & mask & ~3) of the table, permuted by t ^ (h & 3).  So lane j < nb fetches block j of
// each of the two rows with one 16-byte LDS-DMA (nb = blocks covering the run: 2 .. 57 at 2048^2)
// and every lane then picks its four corners out of the wave's LDS scratch.  Taken when the wave's
// slots lie in one texel row in ascending x (checked per unit) and the run fits 64 blocks (per
// level); the per-lane gathers remain for the rest (row changes, the finest levels of the small
// textures, dense rows that wrap at the table's end).
// Measured (profiles/r03/fused_forward.md): look-ups 299 M -> 186 M (an LDS-DMA still costs its
// quads whatever its active lanes), VALU instructions 188 M -> 210 M, issue cycles +37 %, two
// dependent waits per batch instead of one: 0.65 -> 0.85 ms.
#ifndef NT_FU_STAGE
#define NT_FU_STAGE 0     /* measured: see the comment above and profiles/NOTEBOOK.md A9.1a */
#endif
typedef __attribute__((address_space(3))) void* fu_lds_vp;
constexpr int FU_ROW_DWORDS = 256;                         // 64 blocks x 4 entries per staged row
[[maybe_unused]] constexpr int FU_SCR_DWORDS = FU_BATCH * 2 * FU_ROW_DWORDS; // one wave's scratch (FU_BATCH levels x 2 rows)

// levels [L0, L0 + FU_BATCH) of the lane's slot: all fetches first, then the blends
template <int LH, int L0>
__device__ __forceinline__ void fu_encode_batch(const vsa_nt_plan& plan, int lh,
                                                const unsigned* __restrict__ tab, float x, float y,
                                                bool run_ok, unsigned* scr, int lane,
                                                unsigned F[16]) {
  CellRefS c[FU_BATCH];
  unsigned e[FU_BATCH][4];
  bool fast[FU_BATCH];          // wave-uniform
#if NT_FU_STAGE
  int base0[FU_BATCH], base1[FU_BATCH];    // cell coordinate of a staged row's first entry (scalars)
  unsigned sw0[FU_BATCH], sw1[FU_BATCH];   // the rows' position swizzles (h & 3), scalars
  // the previous batch's corner reads have left the scratch before it is written again
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#pragma unroll
  for (int b = 0; b < FU_BATCH; ++b) {
    const int l = L0 + b;
    const LevelGeom g = level_geom(plan, l);
    c[b] = cell_ref_s(g, x, y);
    const bool hashed = LH >= 0 ? l >= LH : l >= lh;   // wave-uniform
    const unsigned* tl = tab + plan.level_offset[l];
    fast[b] = false;
#if NT_FU_STAGE
    if (run_ok) {
      const int cx0 = __builtin_amdgcn_readlane((int)c[b].cx, 0), cx63 = __builtin_amdgcn_readlane((int)c[b].cx, 63);
      const int cy0 = __builtin_amdgcn_readfirstlane((int)c[b].cy);
      int b0, b1, nb;
      unsigned a0, a1;          // this lane's block of row 0 / row 1 (entry index)
      bool ok;
      if (hashed) {
        const unsigned h0 = (unsigned)cy0 * PRIME_Y, h1 = h0 + PRIME_Y;
        b0 = b1 = cx0 & ~3;
        nb = ((cx63 + 1 - b0) >> 2) + 1;
        const unsigned cell4 = (unsigned)(b0 + 4 * lane);
        a0 = (cell4 ^ h0) & g.mask & ~3u;
        a1 = (cell4 ^ h1) & g.mask & ~3u;
        sw0[b] = h0 & 3u, sw1[b] = h1 & 3u;
        ok = nb <= 64;
      } else {
        // dense: entry = cx + cy * res; each row is staged from its own 16-byte aligned start
        const int r0 = cx0 + cy0 * (int)g.res, r1 = r0 + (int)g.res;
        const int st0 = r0 & ~3, st1 = r1 & ~3;
        b0 = cx0 - (r0 - st0), b1 = cx0 - (r1 - st1);
        const int n0 = ((cx63 + 1 - b0) >> 2) + 1, n1 = ((cx63 + 1 - b1) >> 2) + 1;
        nb = n0 > n1 ? n0 : n1;
        a0 = (unsigned)(st0 + 4 * lane), a1 = (unsigned)(st1 + 4 * lane);
        sw0[b] = sw1[b] = 0u;
        // no wrap at the table's end (grid_index's % size: the per-lane path handles it), no negative cell
        ok = nb <= 64 && cx0 >= 0 && cy0 >= 0 && (long long)r1 + (cx63 - cx0) + 1 < (long long)g.size &&
             (long long)st1 + 4 * nb <= (long long)g.size;
      }
      base0[b] = b0, base1[b] = b1;
      if (ok) {
        fast[b] = true;
        if (lane < nb) {
          __builtin_amdgcn_global_load_lds((const void*)(tl + a0), (fu_lds_vp)(scr + (2 * b) * FU_ROW_DWORDS), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((const void*)(tl + a1), (fu_lds_vp)(scr + (2 * b + 1) * FU_ROW_DWORDS), 16, 0, 0);
        }
      }
    }
#endif
    if (!fast[b]) {
      unsigned idx[4];
      if (hashed)
        cell_indices<true>(g, c[b].cx, c[b].cy, idx);
      else
        cell_indices<false>(g, c[b].cx, c[b].cy, idx);
#pragma unroll
      for (int k = 0; k < 4; ++k) e[b][k] = (NT_FU_DIAG & 1) ? (idx[k] & 0x03ff03ffu) | ((unsigned)(size_t)tl & 1u) : tl[idx[k]];
    }
  }
#if NT_FU_STAGE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the staged rows have landed (LDS-DMA is ordered by vmcnt only)
#pragma unroll
  for (int b = 0; b < FU_BATCH; ++b) {
    if (fast[b]) {
      const unsigned* r0 = scr + (2 * b) * FU_ROW_DWORDS;
      const unsigned* r1 = r0 + FU_ROW_DWORDS;
      const unsigned p0 = (unsigned)((int)c[b].cx - base0[b]), p1 = (unsigned)((int)c[b].cx - base1[b]);
      e[b][0] = r0[p0 ^ sw0[b]];
      e[b][1] = r0[(p0 + 1u) ^ sw0[b]];
      e[b][2] = r1[p1 ^ sw1[b]];
      e[b][3] = r1[(p1 + 1u) ^ sw1[b]];
    }
  }
#endif
#pragma unroll
  for (int b = 0; b < FU_BATCH; ++b) F[L0 + b] = enc_blend(e[b], c[b].w);
}

template <int LH, int L0 = 0>
__device__ __forceinline__ void fu_encode(const vsa_nt_plan& plan, int lh,
                                          const unsigned* __restrict__ tab, float x, float y,
                                          bool run_ok, unsigned* scr, int lane, unsigned F[16]) {
  static_assert(16 % FU_BATCH == 0, "batch");
  fu_encode_batch<LH, L0>(plan, lh, tab, x, y, run_ok, scr, lane, F);
  if constexpr (L0 + FU_BATCH < 16) fu_encode<LH, L0 + FU_BATCH>(plan, lh, tab, x, y, run_ok, scr, lane, F);
}

// lanes p (lower half) and p + 32 (upper half) exchange: afterwards `a` holds, on BOTH halves, what
// tile A's lane needs (lower: its own a, upper: the lower lane's b) and `b` what tile B's lane needs
// (the builtin, not inline asm: v_permlane32_swap needs wait states after a VALU write of its
// operands and before a VALU read of its results that only the compiler's hazard pass inserts —
// the asm form produced wrong tiles at one register allocation and right ones at another)
__device__ __forceinline__ void fu_swap(unsigned& a, unsigned& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}

// Work split.  The launch's work = every active texture's slots in 64-slot units, laid on one
// cost axis in texture order (a unit of a texture with NG output groups weighs
// 16 + NT_FU_W_PER_GROUP (NG - 1)).  XCD x (= blockIdx.x % 8: the workgroups that share an L2)
// owns the stretch [x C / 8, (x + 1) C / 8); the 4 J waves of its J workgroups take the units of
// that stretch round-robin, so at any time they gather from about one texture's tables.
// body(tex, begin, end, u_first, u_step, u_end): units u_first, u_first + u_step, ... < u_end of
// the texture's segment [begin, end) belong to this wave (unit u = slots begin + 64 u ...).
template <typename Body>
__device__ __forceinline__ void fu_for_each_piece(const vsa_nt_plan& plan,
                                                  const int* __restrict__ seg_start, Body&& body) {
  const int n_all = plan.nr_shells * 2 * VSA_NT_MAX_DEG;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_x = (gridDim.x % 8 == 0 && !(NT_FU_DIAG & 4)) ? 8 : 1;   // groups of workgroups that share an L2
  const int x = blockIdx.x % n_x, j = blockIdx.x / n_x, J = gridDim.x / n_x;
  const int gw = j * FU_WAVES + wave, GW = J * FU_WAVES;  // this wave among its group's waves
  const int rgb_deg = plan.rgb_degrees, alpha_deg = plan.alpha_degrees;
  auto lane_tex = [&](int tex, int& begin, int& end, int& w) {
    const int deg = tex % VSA_NT_MAX_DEG, type = (tex / VSA_NT_MAX_DEG) & 1, shell = tex / (2 * VSA_NT_MAX_DEG);
    bool act = tex < n_all && (type == 0 ? deg < rgb_deg : (nt_shell_has_alpha(plan, shell) && deg < alpha_deg));
    begin = end = 0;
    if (act) {
      const int sd = shell * VSA_NT_MAX_DEG + deg;
      begin = seg_start[sd];
      end = seg_start[sd + 1];
    }
    const int channels = type == 0 ? 3 * (2 * deg + 1) : 2 * deg + 1;
    w = 16 + NT_FU_W_PER_GROUP * ((channels + 7) / 8 - 1);
    return act && end > begin ? (end - begin + 63) >> 6 : 0;   // units
  };
  // total cost (1/16 units; < 2^31: <= 128 textures x (a few million slots / 64) x ~20)
  long long total = 0;
  for (int t0 = 0; t0 < n_all; t0 += 64) {
    int b_, e_, w_;
    const int units = lane_tex(t0 + lane, b_, e_, w_);
    total += (long long)(unsigned)__builtin_amdgcn_readlane(nt_wave_incl_scan(units * w_), 63);
  }
  if (total == 0) return;
  const long long lo = total * x / n_x, hi = total * (x + 1) / n_x;
  long long c0 = 0;        // cost before the current batch of 64 textures
  long long seq = 0;       // units of this group's stretch before the current piece
  for (int t0 = 0; t0 < n_all && c0 < hi; t0 += 64) {
    int begin, end, w;
    const int units = lane_tex(t0 + lane, begin, end, w);
    const int cl = units * w;
    const int incl = nt_wave_incl_scan(cl);
    const long long t0l = c0 + (long long)(incl - cl), span = cl;
    bool hit = units > 0 && t0l < hi && t0l + span > lo;
    int ua = 0, ub = 0;
    if (hit) {
      const long long a = lo > t0l ? lo - t0l : 0, b = hi - t0l < span ? hi - t0l : span;
      ua = (int)(((unsigned)a + (unsigned)w - 1u) / (unsigned)w);      // units whose start lies in [a, b)
      ub = (int)(((unsigned)b + (unsigned)w - 1u) / (unsigned)w);
      hit = ub > ua;
    }
    unsigned long long todo = __ballot(hit);
    while (todo) {
      const int k = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int jb = __builtin_amdgcn_readlane(begin, k), je = __builtin_amdgcn_readlane(end, k);
      const int ja = __builtin_amdgcn_readlane(ua, k), jub = __builtin_amdgcn_readlane(ub, k);
      // unit u of the piece is unit seq + (u - ja) of the stretch; wave gw takes those = gw mod GW
      const int r = (int)((seq + GW - gw) % GW);          // (seq + i) % GW == gw  <=>  i % GW == (gw - seq) % GW
      const int first = ja + (r == 0 ? 0 : GW - r);
      body(t0 + k, jb, je, first, GW, jub);
      seq += jub - ja;
    }
    c0 += (long long)(unsigned)__builtin_amdgcn_readlane(incl, 63);
  }
}

template <int LH, bool FEAT, bool PRE>
__global__ __launch_bounds__(FU_BLOCK, FU_WGS_PER_CU) void nt_encmlp_fwd_kernel(
    vsa_nt_plan plan, int lh, const _Float16* __restrict__ weights,
    const unsigned* __restrict__ tables, const float2* __restrict__ slot_xy,
    const int* __restrict__ seg_start, unsigned* __restrict__ features,
    unsigned* __restrict__ texels, _Float16* __restrict__ pre_out) {
  __shared__ unsigned s_qt[257];      // thresholds of the 8-bit quantisation (nt_quant_table.h)
#if NT_FU_STAGE
  __shared__ __attribute__((aligned(16))) unsigned s_scr[FU_WAVES * FU_SCR_DWORDS];
  unsigned* scr = s_scr + (threadIdx.x >> 6) * FU_SCR_DWORDS;
#else
  unsigned* scr = nullptr;
#endif
  const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < 257; i += FU_BLOCK) s_qt[i] = NT_QUANT_THR[i];
  __syncthreads();
  const long long n_entries = plan.level_offset[plan.n_levels];
  const int nl = plan.n_levels;
  fu_for_each_piece(plan, seg_start, [&](int tex, int begin, int end, int u_first, int u_step, int u_end) {
    if (u_first >= u_end) return;
    const TexInfo ti = tex_info(plan, seg_start, tex);
    half8_t wf[16];
    const int ptex = nt_param_tex(plan, tex);      // (shared models: shell 0's parameters)
    load_fwd_frags(weights + (long long)ptex * VSA_NT_WEIGHTS_PER_TEX, lane, wf);
    const unsigned* tab = tables + (long long)ptex * n_entries;
    const int pre_base = ti.type == 0 ? 0 : 24;
    unsigned* fplane = nullptr;
    if constexpr (FEAT) fplane = features + nt_feat_plane_base(plan, ti.type, 0);
    auto run = [&](auto ng_tag) {
      constexpr int NG = decltype(ng_tag)::value;
      auto centre = [&](int u) {
        const int s = begin + u * 64 + lane;
        return slot_xy[s < end ? s : end - 1];
      };
      float2 xy_next = centre(u_first);
      for (int u = u_first; u < u_end; u += u_step) {
        const float2 xy = xy_next;
        {   // the next unit's texel centres (past the end: this unit's again, unused)
          const int un = u + u_step;
          xy_next = centre(un < u_end ? un : u);
        }
        const int slot0 = begin + u * 64;
        // one texel row, ascending x (slots are in texel order: vsa_nt_compact): the row-run staging applies
        const float x_lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xy.x), 0));
        const float x_hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xy.x), 63));
        const float y_lo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, xy.y)));
        const bool run_ok = __all(xy.y == y_lo && xy.x >= x_lo && xy.x <= x_hi) != 0;
        unsigned F[16];
        fu_encode<LH>(plan, lh, tab, xy.x, xy.y, run_ok, scr, lane, F);
        if constexpr (FEAT) {
          const int s = slot0 + lane;
          if (s < end) {
            unsigned* o = fplane + nt_feat_in_plane(nl, s);
#pragma unroll
            for (int l = 0; l < 16; ++l) o[l * NT_FBLOCK] = F[l];
          }
        }
        // re-deal: F[l] (tile A) / F[l + 4] (tile B) for l = 8s + i
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < 4; ++i) fu_swap(F[8 * s + i], F[8 * s + 4 + i]);
        half8_t bxa[2], bxb[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bxa[s] = __builtin_bit_cast(half8_t, make_uint4(F[8 * s], F[8 * s + 1], F[8 * s + 2], F[8 * s + 3]));
          bxb[s] = __builtin_bit_cast(half8_t, make_uint4(F[8 * s + 4], F[8 * s + 5], F[8 * s + 6], F[8 * s + 7]));
        }
#if NT_FU_DIAG & 2
        asm volatile("" ::"v"(bxa[0]), "v"(bxa[1]), "v"(bxb[0]), "v"(bxb[1]));
        continue;
#endif
        half8_t b2[4], b3[4];
        float16_t acc3;
        mlp_tile_fwd(wf, bxa, b2, b3, acc3);
        quant_store_tile<NG, PRE>(acc3, ti, s_qt, texels, slot0 + p, slot0 + p < end, h, pre_out, pre_base);
        if (slot0 + 32 < end) {    // wave-uniform
          mlp_tile_fwd(wf, bxb, b2, b3, acc3);
          quant_store_tile<NG, PRE>(acc3, ti, s_qt, texels, slot0 + 32 + p, slot0 + 32 + p < end, h, pre_out, pre_base);
        }
      }
    };
    if (ti.channels <= 8) run(std::integral_constant<int, 1>{});
    else if (ti.channels <= 16) run(std::integral_constant<int, 2>{});
    else if (ti.channels <= 24) run(std::integral_constant<int, 3>{});
    else run(std::integral_constant<int, 4>{});
  });
}

}  // namespace

static bool fu_level_hashed(const vsa_nt_plan* p, int l) {
  const long long res = p->level_res[l], size = p->level_size[l];
  return !(res <= size && res * res <= size);  // tiny-cuda-nn grid_index
}

template <int LH>
static void fu_launch(const vsa_nt_plan* plan, int lh, const void* tables_h, const void* weights_h,
                      const float* slot_xy, const int32_t* seg_start, void* features, uint8_t* texels,
                      void* pre_out, int grid, hipStream_t st) {
  const _Float16* W = reinterpret_cast<const _Float16*>(weights_h);
  const unsigned* T = reinterpret_cast<const unsigned*>(tables_h);
  const float2* xy = reinterpret_cast<const float2*>(slot_xy);
  unsigned* F = reinterpret_cast<unsigned*>(features);
  unsigned* X = reinterpret_cast<unsigned*>(texels);
  _Float16* P = reinterpret_cast<_Float16*>(pre_out);
#define FU_GO(FEAT, PRE)                                                                      \
  hipLaunchKernelGGL((nt_encmlp_fwd_kernel<LH, FEAT, PRE>), dim3(grid), dim3(FU_BLOCK), 0, st, \
                     *plan, lh, W, T, xy, seg_start, F, X, P)
  if (features && pre_out) FU_GO(true, true);
  else if (features) FU_GO(true, false);
  else if (pre_out) FU_GO(false, true);
  else FU_GO(false, false);
#undef FU_GO
}

extern "C" int vsa_nt_encode_mlp_fwd(const vsa_nt_plan* plan, const void* tables_h,
                                     const void* weights_h, const float* slot_xy,
                                     const int32_t* seg_start, void* features, uint8_t* texels,
                                     void* pre_out, void* stream) {
  if (plan && plan->row_format != 0) return VSA_ERR_UNSUPPORTED;     // 8-bit rows only: use vsa_nt_encode_fwd + vsa_nt_mlp_fwd
  if (!plan || !tables_h || !weights_h || !slot_xy || !seg_start || !texels) return VSA_ERR_ARG;
  if (plan->n_levels != 16) return VSA_ERR_UNSUPPORTED;    // the MLP's input width is 16 levels x 2
  int lh = 0;
  while (lh < plan->n_levels && !fu_level_hashed(plan, lh)) ++lh;
  for (int l = lh; l < plan->n_levels; ++l) {
    const int sz = plan->level_size[l];
    if (!fu_level_hashed(plan, l) || (sz & (sz - 1)) != 0) return VSA_ERR_UNSUPPORTED;
  }
  int nr_cus = 0;
  { const int rc = vsa_cu_count(&nr_cus); if (rc) return rc; }
  const int grid = nr_cus * FU_WGS_PER_CU;
  if (lh == 6)
    fu_launch<6>(plan, lh, tables_h, weights_h, slot_xy, seg_start, features, texels, pre_out, grid, (hipStream_t)stream);
  else
    fu_launch<-1>(plan, lh, tables_h, weights_h, slot_xy, seg_start, features, texels, pre_out, grid, (hipStream_t)stream);
  VSA_RETURN_LAUNCH_STATUS();
}
