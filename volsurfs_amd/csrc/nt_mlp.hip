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
#include <stdlib.h>

#include <type_traits>

#include "nt_common.h"
#include "nt_quant_table.h"
#include "nt_mlp_common.h"

namespace {

constexpr int MLP_BLOCK = 256;
constexpr int MLP_WAVES = MLP_BLOCK / 64;




__device__ __forceinline__ void load_features(const unsigned* __restrict__ F,
                                              const vsa_nt_plan& plan, int type, int slot, int h,
                                              half8_t bx[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    unsigned w[4];
    const unsigned* base = F + nt_feat_plane_base(plan, type, 8 * s + 4 * h) +
                           nt_feat_in_plane(plan.n_levels, slot);
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = base[i * NT_FBLOCK];
    uint4 u = make_uint4(w[0], w[1], w[2], w[3]);
    bx[s] = __builtin_bit_cast(half8_t, u);
  }
}


// Persistent: gridDim.x workgroups split the frame's 32-slot tiles evenly
// (nt_for_each_piece: cost axis with the weight staging of a run priced at 8 tiles).
#ifndef NT_FWD_W_PER_GROUP
#define NT_FWD_W_PER_GROUP 2     /* 1/16 tile units per further 8-channel output group */
#endif
#ifndef NT_FWD_RUN_COST
#define NT_FWD_RUN_COST 16
#endif
constexpr int MLP_FWD_RUN_COST = NT_FWD_RUN_COST;   // tiles: fitted 7.6 us per run / 0.48 us per tile (tools/fit_cost.py)
constexpr int MLP_FWD_WGS_PER_CU = 3;   // 148 VGPRs -> 3 waves per SIMD, one per workgroup

// PRE: also write the pre-sigmoid outputs (tests only; the production launch has no such stores).
template <bool PRE, bool F16ROWS = false>
__global__ __launch_bounds__(MLP_BLOCK, MLP_FWD_WGS_PER_CU) void nt_mlp_fwd_kernel(
    vsa_nt_plan plan, const _Float16* __restrict__ weights,
    const unsigned* __restrict__ features, const int* __restrict__ seg_start,
    unsigned* __restrict__ texels, _Float16* __restrict__ pre_out) {
  __shared__ unsigned s_qt[257];      // thresholds of the 8-bit quantisation (nt_quant_table.h)
  const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 257; i += MLP_BLOCK) s_qt[i] = NT_QUANT_THR[i];
  __syncthreads();
  NT_SPAN_MARK(0, 0);
  NT_BAL_BEGIN();
  // a tile's cost grows with the number of 8-channel groups the quantiser has to form and store
  // (tools/wg_span.py: 0.85-0.88 span efficiency with equal weights)
  auto unit_weight = [&](int, int deg, int type) {
    const int channels = type == 0 ? 3 * (2 * deg + 1) : 2 * deg + 1;
    return 16 + NT_FWD_W_PER_GROUP * ((channels + 7) / 8 - 1);
  };
  nt_for_each_piece<32>(plan, seg_start, 1, MLP_FWD_RUN_COST,
                        [&](int, int tex, int first, int last, int, int) {
    const TexInfo ti = tex_info(plan, seg_start, tex);
    half8_t wf[16];
    load_fwd_frags(weights + (long long)nt_param_tex(plan, tex) * VSA_NT_WEIGHTS_PER_TEX, lane, wf);
    const int pre_base = ti.type == 0 ? 0 : 24;    // pre_out keeps the fixed 32-wide test layout
    const int ntiles = (last - first + 31) >> 5;
    // the tile loop, compiled once per number NG of 8-channel groups of the output (as the
    // backward's producer): no test of ti.channels inside
    auto run = [&](auto ng_tag) {
      constexpr int NG = decltype(ng_tag)::value;
      half8_t q[2];     // one tile of features in flight per wave
      if (wave < ntiles) {
        const int s0 = first + wave * 32 + p;
        load_features(features, plan, ti.type, s0 < last ? s0 : last - 1, h, q);
      }
      for (int tile = wave; tile < ntiles; tile += MLP_WAVES) {
        const int slot = first + tile * 32 + p;
        const bool valid = slot < last;
        half8_t bx[2];
        bx[0] = q[0];
        bx[1] = q[1];
        {   // the next tile's features (past the end: the last slot's again)
          const int sn = slot + MLP_WAVES * 32;
          load_features(features, plan, ti.type, sn < last ? sn : last - 1, h, q);
        }
        half8_t b2[4], b3[4];
        float16_t acc3;
        mlp_tile_fwd(wf, bx, b2, b3, acc3);
        if constexpr (F16ROWS)
          half_store_tile<NG, PRE>(acc3, ti, reinterpret_cast<uint2*>(texels), slot, valid, h, pre_out, pre_base, plan.row_format == 2);
        else
          quant_store_tile<NG, PRE>(acc3, ti, s_qt, texels, slot, valid, h, pre_out, pre_base);
      }
    };
    if (ti.channels <= 8) run(std::integral_constant<int, 1>{});
    else if (ti.channels <= 16) run(std::integral_constant<int, 2>{});
    else if (ti.channels <= 24) run(std::integral_constant<int, 3>{});
    else run(std::integral_constant<int, 4>{});
  }, 0, 1 << 30, unit_weight, NT_BAL_MLP_FWD);
  NT_SPAN_MARK(0, 1);
  NT_BAL_END(NT_BAL_MLP_FWD);
}

// ------------------------------------------------------------------ backward
// Fragment ids 16..31: the transposed weights for the data-gradient chain
//   16..19 T3[m][s]  elem j = W3[perm_k(s,h,j)][32m + r]
//   20..27 T2[m][q]  elem j = W2[perm_k(q,h,j)][32m + r]
//   28..31 T1[q]     elem j = W1[perm_k(q,h,j)][r]

typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));

// accumulator -> two f16 fragments, zeroed where the forward activation was <= 0.
// Packed integer form, 3 VALU ops per f16 pair (the element-wise select compiled to ~12):
// the activations are the backward recompute's ReLU outputs (relu_pack<true>: a zero is
// always bits 0), so "> 0" is "bits != 0"; the gradient's f16 bits times min(bits, 1) as
// 16-bit integers are the gradient or 0.
__device__ __forceinline__ unsigned relu_mask_pair(float a0, float a1, unsigned act_bits) {
  const half2_t hp = {(_Float16)a0, (_Float16)a1};
  unsigned r = __builtin_bit_cast(unsigned, hp);
  // (inline asm: the compiler scalarises the vector form into per-half compares and selects)
  asm("v_pk_min_u16 %1, %1, %2\n\tv_pk_mul_lo_u16 %0, %0, %1" : "+v"(r), "+v"(act_bits) : "v"(0x00010001u));
  return r;
}

__device__ __forceinline__ void mask_pack(const float16_t& acc, const half8_t& act0,
                                          const half8_t& act1, half8_t& o0, half8_t& o1) {
  typedef unsigned uint4v __attribute__((ext_vector_type(4)));
  const uint4v a0 = __builtin_bit_cast(uint4v, act0), a1 = __builtin_bit_cast(uint4v, act1);
  uint4v r0, r1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r0[i] = relu_mask_pair(acc[2 * i], acc[2 * i + 1], a0[i]);
    r1[i] = relu_mask_pair(acc[8 + 2 * i], acc[8 + 2 * i + 1], a1[i]);
  }
  o0 = __builtin_bit_cast(half8_t, r0);
  o1 = __builtin_bit_cast(half8_t, r1);
}


__device__ __forceinline__ void prefetch_features(const vsa_nt_plan& plan,
                                                  const unsigned* features, int type, int slot,
                                                  int last, int h, half8_t bx[2]) {
  load_features(features, plan, type, slot < last ? slot : last - 1, h, bx);
}

// ---------------------------------------------------------------- backward v4
// Producer / consumer split.  A single-role kernel (one wave doing everything) needs 128 accumulator
// registers for the weight gradients on top of the recompute + dgrad chain (~410
// VGPRs -> one wave per SIMD, every MFMA result copied AGPR<->VGPR) and is bound by
// the ISSUE rate of that one wave (in-kernel stamps: 7.6k cycles per tile, of which
// 170 wait on memory).  Here the two waves that share a SIMD (w and w+4) split a
// tile stream so that both issue streams are busy:
//   PRODUCER  (t)   gradient rows + features (prefetched one tile ahead), forward
//                   recompute, dOut = G * sigmoid' -> writes the point-major f16 images
//                   {dOut, H2, H1, X} of tile t into LDS buffer t&1 (each as soon as its
//                   fragments exist); dW3 of tile t-1 from buffer (t-1)&1 at the head of the trip
//   CONSUMER  (t-1) reads buffer (t-1)&1: dgrad chain dH2, dH1, dX (B operands by
//                   16-B row reads, ReLU masks from the H images), dW2 and dW1 (operands by
//                   ds_read_b64_tr_b16), dF stores, sum|dF|
// Each role stays below 256 registers with nothing spilled inside the loops (MFMA results in
// plain VGPRs), hand-off is ONE workgroup barrier per tile over a double-buffered image set.
// The tile loops are compiled once per output width (producer: 1..4 groups of 8 channels,
// consumer: 1 or 2 k-steps of dH2); each instance owns its accumulators and its epilogue.
// Round-2 stamps (NT_STAMP, tools/wg_timeline.py; cycles per tile, producer | consumer work):
// 5050 | 3880 before, 2740 | 3460 now — see profiles/NOTEBOOK.md A9.1 for what moved and what it cost.
#ifndef NT_PC_DW3_CONSUMER
#define NT_PC_DW3_CONSUMER 0    /* dW3: 0 producer, 1 consumer (from the finished images), 2 split by 32-column block */
#endif
#ifndef NT_PC_BATCH
#define NT_PC_BATCH 1           /* consumer: operands fetched in batches a stage ahead of their MFMAs */
#endif
#ifndef NT_PC_DW3_LATE
#define NT_PC_DW3_LATE 1        /* the producer's dW3 blocks one trip late, from the set written before the last barrier */
#endif
#ifndef NT_PC_DABS_MOD
#define NT_PC_DABS_MOD 0
#endif
#ifndef NT_PC_PRIO
#define NT_PC_PRIO 2          /* issue priority: 0 consumer raised, 1 producer raised, 2 none, 3 producer at 3 */
#endif
constexpr int PC_BLOCK = 512;
constexpr int PC_PAIRS = 4;
#ifndef NT_PC_S64
#define NT_PC_S64 68
#endif
#ifndef NT_PC_S32
#define NT_PC_S32 40
#endif
constexpr int S64 = NT_PC_S64, S32 = NT_PC_S32;   // row strides (halfs): 136 B (8-B aligned, bank-spread), 80 B (16-B aligned)
constexpr int SET_DOUT = 0, SET_X = 32 * S32, SET_H2 = 2 * 32 * S32, SET_H1 = SET_H2 + 32 * S64;
constexpr int SET_HALFS = SET_H1 + 32 * S64;          // one {dOut, X, H2, H1} set
constexpr int PRIV_HALFS = 32 * S64;                  // consumer-private dH2 / dH1 image
constexpr int PAIR_HALFS = 2 * SET_HALFS + PRIV_HALFS;
constexpr int PC_FRAGS = 20;   // persistent in LDS: ids 16..31 transposed (perm k), 32..35 W3^T natural k
constexpr int PC_FRAG_ID0 = 16;
                               // (ids 0..15, the forward set, are staged through the image area into registers)

template <int STRIDE>
__device__ __forceinline__ void store_frags_s(_Float16* img, int col_base, const half8_t& f0,
                                              const half8_t& f1, int p, int h) {
  _Float16* row = img + p * STRIDE + col_base + 4 * h;
  *reinterpret_cast<half4_t*>(row + 0) = __builtin_shufflevector(f0, f0, 0, 1, 2, 3);
  *reinterpret_cast<half4_t*>(row + 8) = __builtin_shufflevector(f0, f0, 4, 5, 6, 7);
  *reinterpret_cast<half4_t*>(row + 16) = __builtin_shufflevector(f1, f1, 0, 1, 2, 3);
  *reinterpret_cast<half4_t*>(row + 24) = __builtin_shufflevector(f1, f1, 4, 5, 6, 7);
}

// the two fragments (accumulator-register order) of a 32-channel tile of this lane's point
template <int STRIDE>
__device__ __forceinline__ void load_frags_s(const _Float16* img, int col_base, half8_t& f0,
                                             half8_t& f1, int p, int h) {
  const _Float16* row = img + p * STRIDE + col_base + 4 * h;
  const half4_t a = *reinterpret_cast<const half4_t*>(row + 0);
  const half4_t b = *reinterpret_cast<const half4_t*>(row + 8);
  const half4_t c = *reinterpret_cast<const half4_t*>(row + 16);
  const half4_t d = *reinterpret_cast<const half4_t*>(row + 24);
  f0 = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  f1 = __builtin_shufflevector(c, d, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int STRIDE>
__device__ __forceinline__ half8_t read_tr_s(const _Float16* img, int col_base, int s, int lane) {
  const int h = lane >> 5, li = lane & 15, q = li >> 2, pp = li & 3, grp = (lane >> 4) & 1;
  const _Float16* a0 = img + (16 * s + 8 * h + q) * STRIDE + col_base + 16 * grp + 4 * pp;
  typedef __attribute__((address_space(3))) short4v* lds_p;
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0));
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0 + 4 * STRIDE));
  typedef short short8v __attribute__((__vector_size__(8 * sizeof(short))));
  const short8v both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(half8_t, both);
}

#ifdef NT_STAMP
__device__ unsigned long long g_dbg[16384 * 8];   // per-workgroup timeline of the last pc launch
__device__ unsigned long long g_dbg_stage[16384 * 8];   // consumer wave 4 / producer wave 0: cycles per stage (NT_PC_BATCH build)
__device__ unsigned long long g_dbg_role[16384 * 4];   // {producer work, wait, consumer work, wait} cycles of wave 0 / 4
#endif
#ifdef NT_STAMP   // diagnostic build only (tools: make EXTRA=-DNT_STAMP): per-role cycles per tile
#define STAMP(var)                                                               \
  {                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");   \
    __builtin_amdgcn_sched_barrier(0);                                           \
  }
#else
#define STAMP(var)
#endif

// LDS operations of this wave are done, then the workgroup barrier; does NOT drain the
// vector-memory queue (prefetches and stores stay in flight across it)
__device__ __forceinline__ void pc_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One run = a contiguous slot range [wk.first, wk.last) of ONE texture: stage the weight
// fragments, stream the tiles through the producer/consumer pairs, reduce and flush the
// weight gradients.
constexpr int PC_DABS_STRIDE = 66;     // floats per (pair, register) row of the sum|dF| partials: 64 lanes + 2 (bank spread)
constexpr int PC_PART_FLOATS = VSA_NT_WEIGHTS_PER_TEX;   // one wave pair's weight-gradient partials (32 KiB)
static_assert(PC_PAIRS * 16 * PC_DABS_STRIDE * 4 <= PC_FRAGS * 64 * 16, "sum|dF| partials live in the fragment area");

__device__ __forceinline__ void pc_run(
    const vsa_nt_plan& plan, const Work wk, unsigned char* s_raw,
    const _Float16* __restrict__ weights, unsigned* __restrict__ features,
    const int* __restrict__ seg_start, _Float16* __restrict__ grad_rows,
    float* __restrict__ grad_weights, float* __restrict__ dfeat_abs_sum, float gw_scale) {
  half8_t* s_frag = reinterpret_cast<half8_t*>(s_raw) - PC_FRAG_ID0 * 64;   // indexed by fragment id (16|20)..35
  _Float16* s_img_all = reinterpret_cast<_Float16*>(s_raw + PC_FRAGS * 64 * 16);
#ifdef NT_STAMP
  unsigned long long ph0, ph1, ph2, ph3, rt0, rt1;
  STAMP(ph0);
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0)::"memory");
#endif
  const int tex = wk.tex;
  const int ptex = nt_param_tex(plan, tex);     // parameters: shell 0's when the models are shared
  const TexInfo ti = tex_info(plan, seg_start, tex);
  const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
  const int wave = threadIdx.x >> 6;
  const bool producer = wave < PC_PAIRS;
  // forward fragments: straight from memory into the producers' registers (load_fwd_frags)
  half8_t wf[16];
  if (producer) load_fwd_frags(weights + (long long)ptex * VSA_NT_WEIGHTS_PER_TEX, lane, wf);
  {
    // the transposed fragments 16..35 are gathers with a stride: the texture's 8192 weights come
    // in with 16-byte loads into the (still free) image area and the fragments are gathered from
    // that LDS copy (gathering them from memory with 2-byte loads was 36 dependent-latency
    // rounds per thread at the head of every run)
    _Float16* W = s_img_all;
    {
      const half8_t* Wg = reinterpret_cast<const half8_t*>(weights + (long long)ptex * VSA_NT_WEIGHTS_PER_TEX);
      for (int i = threadIdx.x; i < VSA_NT_WEIGHTS_PER_TEX / 8; i += PC_BLOCK)
        reinterpret_cast<half8_t*>(W)[i] = Wg[i];
    }
    __syncthreads();
    for (int idx = 16 * 64 + threadIdx.x; idx < 36 * 64; idx += PC_BLOCK) {
      const int frag = idx >> 6, ln = idx & 63, r = ln & 31, hh = ln >> 5;
      half8_t v;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        _Float16 x;
        if (frag < 20) {                     // W3^T, perm k (unused by v4, kept for id parity)
          const int f_ = frag - 16;
          x = W[W3_OFF + perm_k(f_ & 1, hh, j) * 64 + 32 * (f_ >> 1) + r];
        } else if (frag < 28) {              // W2^T, perm k
          const int f_ = frag - 20;
          x = W[W2_OFF + perm_k(f_ & 3, hh, j) * 64 + 32 * (f_ >> 2) + r];
        } else if (frag < 32) {              // W1^T, perm k
          x = W[W1_OFF + perm_k(frag - 28, hh, j) * 32 + r];
        } else {                             // W3^T, natural k (B operand comes from LDS rows)
          const int f_ = frag - 32;
          x = W[W3_OFF + (16 * (f_ & 1) + 8 * hh + j) * 64 + 32 * (f_ >> 1) + r];
        }
        v[j] = x;
      }
      s_frag[idx] = v;
    }
  }
  __syncthreads();   // the image area may be overwritten from here on
  const int pr = wave & (PC_PAIRS - 1);
  _Float16* pair = s_img_all + pr * PAIR_HALFS;
  _Float16* priv = pair + 2 * SET_HALFS;
  float* s_part = reinterpret_cast<float*>(s_img_all) + pr * PC_PART_FLOATS;   // after the tile loop
  const int ntiles = (wk.last - wk.first + 31) >> 5;
  const int iters = (ntiles + PC_PAIRS - 1) / PC_PAIRS;     // same for every wave: barriers match
#ifdef NT_STAMP
  STAMP(ph1);
#endif

  if (producer) {
#if NT_PC_PRIO == 1
    __builtin_amdgcn_s_setprio(1);
#elif NT_PC_PRIO == 3
    __builtin_amdgcn_s_setprio(3);
#endif
#ifdef NT_STAMP
    unsigned long long tw_ = 0, tb_ = 0, q0, q1, q2;
#endif
    // The tile loop, compiled once per number NG of 8-channel groups the texture's output has
    // (alpha textures and degree 0: 1, degrees 1-2: 2, degree 3: 3): nothing inside tests
    // ti.channels, and channels 16..31 of dOut are neither formed nor stored when NG <= 2 (the
    // consumer then skips their k-step; the transposed reads for dW3 see stale columns there,
    // which only reach accumulator rows >= channels, never flushed).
    auto run_producer = [&](auto ng_tag) {
      constexpr int NG = decltype(ng_tag)::value;
      // (accumulators and epilogue live inside the instance: values merging across the instances
      // made the register allocator copy whole accumulator sets around inside the loops)
      constexpr int PM0 = 0, PM1 = NT_PC_DW3_CONSUMER == 0 ? 2 : NT_PC_DW3_CONSUMER == 2 ? 1 : 0;   // producer's dW3 blocks
      float16_t gW3[2] = {float16_t{0}, float16_t{0}};
      // gradient rows of one slot (raw f16: converting here would wait for the prefetch).  No
      // branches around the quads this lane has no use for — slots past the end, rows beyond the
      // texture's channels: those lanes read a valid quad (the last slot's / quad 0) and dOut
      // is zeroed by a select once it is formed (row_ok / the slot test)
      const half4_t* const grow_base = reinterpret_cast<const half4_t*>(grad_rows) + ti.row_first;
      bool row_ok[NG];
      int quad_of[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        row_ok[g] = 8 * g + 4 * h < ti.channels;
        quad_of[g] = row_ok[g] ? 2 * g + h : 0;
      }
      auto load_grows = [&](int slot, half4_t (&gr)[4]) {
        const half4_t* rp = grow_base + (long long)(min(slot, wk.last - 1) - ti.begin) * ti.row_quads;
#pragma unroll
        for (int g = 0; g < NG; ++g) gr[g] = rp[quad_of[g]];
      };
      // {features, gradient rows} of the next tile are in flight while a tile is processed.  The
      // copy out of the landing registers comes FIRST in a trip, before the next loads are
      // issued: the wait it needs is then for loads a whole trip old, and nothing after the new
      // loads has to wait on the vector-memory counter (with the copy at the end of the trip the
      // compiler put an s_waitcnt vmcnt(0) right behind the prefetch, for the path that enters the
      // loop from its preheader — the full memory latency exposed in every trip)
      half8_t bx[2], bx_next[2];
      half4_t gr[4], gr_next[4];
      {
        const int s0 = wk.first + pr * 32 + p;
        prefetch_features(plan, features, ti.type, s0, wk.last, h, bx_next);
        load_grows(s0, gr_next);
      }
      auto dw3_from = [&](const _Float16* img_dout, const _Float16* img_h2) {
        if constexpr (PM1 > PM0) {
#pragma unroll
          for (int sx = 0; sx < 2; ++sx) {
            const half8_t a3 = read_tr_s<S32>(img_dout, 0, sx, lane);
#pragma unroll
            for (int m = PM0; m < PM1; ++m)
              gW3[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3, read_tr_s<S64>(img_h2, 32 * m, sx, lane), gW3[m], 0, 0, 0);
          }
        }
      };
      auto trip = [&](const int it) {
        STAMP(q0);
        {
#if NT_PC_DW3_LATE
          // dW3 of the PREVIOUS tile, from the set written before the last barrier (the consumer
          // reads the same set meanwhile): no wait on this trip's own LDS stores
          if (it > 0) dw3_from(pair + ((it - 1) & 1) * SET_HALFS + SET_DOUT, pair + ((it - 1) & 1) * SET_HALFS + SET_H2);
#endif
          const int slot = wk.first + (pr + it * PC_PAIRS) * 32 + p;
          _Float16* set = pair + (it & 1) * SET_HALFS;
          _Float16* const img_dout_w = set + SET_DOUT;
          _Float16* const img_h2_w = set + SET_H2;
          bx[0] = bx_next[0];
          bx[1] = bx_next[1];
#pragma unroll
          for (int g = 0; g < NG; ++g) gr[g] = gr_next[g];
          // pin the wait for the landed operands HERE, ahead of the clear stores and the next
          // requests (the compiler otherwise sinks these register copies below the clear loop,
          // where the wait for them also drains the stores just issued): an empty asm that reads
          // every landed register and orders memory operations around it
          {
            typedef unsigned uint4v_ __attribute__((ext_vector_type(4)));
            typedef unsigned uint2v_ __attribute__((ext_vector_type(2)));
            const uint4v_ u0 = __builtin_bit_cast(uint4v_, bx[0]), u1 = __builtin_bit_cast(uint4v_, bx[1]);
            asm volatile("" ::"v"(u0), "v"(u1) : "memory");
#pragma unroll
            for (int g = 0; g < NG; ++g) {
              const uint2v_ ug = __builtin_bit_cast(uint2v_, gr[g]);
              asm volatile("" ::"v"(ug) : "memory");
            }
          }
          // consume-and-clear.  Lanes run over (slot, own quad) pairs in memory order, so a
          // store instruction covers whole stretches of a few lines (one 8-byte store per lane at
          // its own row stride touched 32 lines per instruction and cost 180 us a frame).  The
          // rows' loads have returned (the copy above waited for them).  Program order per trip is
          // {copy, clear stores, next loads}: the only wait on the vector-memory counter is the
          // copy's, for loads that are a whole trip old, and it is exact (nothing newer in flight).
#if !defined(NT_DIAG_NOCLEAR)
          if (ti.channels > 0) {
            const int s0 = slot - p, nq = min(32, wk.last - s0) * ti.own_quads;
            half4_t* rows = reinterpret_cast<half4_t*>(grad_rows) + ti.row_first +
                            (long long)(s0 - ti.begin) * ti.row_quads;
            for (int i = lane; i < nq; i += 64) {
              const int sl = ti.own_quads == 1 ? i : (int)__umulhi((unsigned)i, ti.own_magic);
              rows[sl * ti.row_quads + (i - sl * ti.own_quads)] = half4_t{0, 0, 0, 0};
            }
          }
#endif
          // the next tile's operands: unconditional (past the end the loads clamp to the last slot)
          prefetch_features(plan, features, ti.type, slot + PC_PAIRS * 32, wk.last, h, bx_next);
          load_grows(slot + PC_PAIRS * 32, gr_next);
          // forward recompute (mlp_tile_fwd's chain); every image is stored as soon as its
          // fragments exist, so that they stop occupying registers and the LDS writes overlap
          // the next layer's matrix instructions
          float16_t acc3;
          half8_t b3[4];
          {
            half8_t b2[4];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
              float16_t a = {0};
#pragma unroll
              for (int s_ = 0; s_ < 2; ++s_)
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[m * 2 + s_], bx[s_], a, 0, 0, 0);
              b2[2 * m] = relu_pack<true>(a, 0);
              b2[2 * m + 1] = relu_pack<true>(a, 1);
              store_frags_s<S64>(set + SET_H1, 32 * m, b2[2 * m], b2[2 * m + 1], p, h);
            }
#pragma unroll
            for (int sx = 0; sx < 2; ++sx)
              *reinterpret_cast<half8_t*>(set + SET_X + p * S32 + 16 * sx + 8 * h) = bx[sx];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
              float16_t a = {0};
#pragma unroll
              for (int q = 0; q < 4; ++q)
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[4 + m * 4 + q], b2[q], a, 0, 0, 0);
              b3[2 * m] = relu_pack<true>(a, 0);
              b3[2 * m + 1] = relu_pack<true>(a, 1);
              store_frags_s<S64>(img_h2_w, 32 * m, b3[2 * m], b3[2 * m + 1], p, h);
            }
            float16_t a = {0};
#pragma unroll
            for (int q = 0; q < 4; ++q)
              a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[12 + q], b3[q], a, 0, 0, 0);
            acc3 = a;
          }
          // dL/d(pre-sigmoid output): G * sig' with sig' = sig - sig^2 (round = STE); zero for
          // padding rows and slots past the end.  The f16 operands enter through mixed-precision
          // fused multiply-adds (conversion and fp32 product in one instruction), the last of
          // which rounds straight to the f16 half it is stored in.
          unsigned dq[2 * NG];     // dOut of this lane's point: channel pairs (8g + 4h + {0,1}, {2,3})
          if (__builtin_expect(plan.row_format == 2, 0)) {
            // raw rows (using_sh_squeezing = 0): the row IS the network output, dOut = G (wave-uniform branch)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
              typedef unsigned uint2v_ __attribute__((ext_vector_type(2)));
              const uint2v_ gb = __builtin_bit_cast(uint2v_, gr[g]);
              const bool keep = row_ok[g] && slot < wk.last;
              dq[2 * g] = keep ? gb[0] : 0u;
              dq[2 * g + 1] = keep ? gb[1] : 0u;
            }
          } else
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            typedef unsigned uint2v_ __attribute__((ext_vector_type(2)));
            const uint2v_ gb = __builtin_bit_cast(uint2v_, gr[g]);
            const bool keep = row_ok[g] && slot < wk.last;
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
#ifdef NT_DIAG_NOSIG      /* diagnostic only: prices the sigmoid' of dOut in the producer's stream */
              const unsigned db = mul_mix_pk(gb[i2], 0.25f, 0.25f);
#else
              const half2_t o_h = {(_Float16)acc3[4 * g + 2 * i2], (_Float16)acc3[4 * g + 2 * i2 + 1]};
              const unsigned ob = __builtin_bit_cast(unsigned, o_h);
              const float sg0 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(mul_mix<0>(ob, -1.4426950408889634f)));
              const float sg1 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(mul_mix<1>(ob, -1.4426950408889634f)));
              const unsigned db = mul_mix_pk(gb[i2], __builtin_fmaf(-sg0, sg0, sg0), __builtin_fmaf(-sg1, sg1, sg1));
#endif
              dq[2 * g + i2] = keep ? db : 0u;
            }
          }
          {
            typedef unsigned uint2v_ __attribute__((ext_vector_type(2)));
            _Float16* row = img_dout_w + p * S32 + 4 * h;
#pragma unroll
            for (int g = 0; g < (NG > 2 ? 4 : 2); ++g)
              *reinterpret_cast<uint2v_*>(row + 8 * g) = g < NG ? uint2v_{dq[2 * (g < NG ? g : 0)], dq[2 * (g < NG ? g : 0) + 1]} : uint2v_{0u, 0u};
          }
#if !NT_PC_DW3_LATE
          // ---- dW3 += dOut . H2^T  (transposed reads of this wave's own, just-written images)
          dw3_from(img_dout_w, img_h2_w);
#endif
        }
        STAMP(q1);
        pc_barrier();
        STAMP(q2);
#ifdef NT_STAMP
        tw_ += q1 - q0; tb_ += q2 - q1;
#endif
      };
      for (int it = 0; it < iters; ++it) trip(it);
#if NT_PC_DW3_LATE
      if (iters > 0) dw3_from(pair + ((iters - 1) & 1) * SET_HALFS + SET_DOUT, pair + ((iters - 1) & 1) * SET_HALFS + SET_H2);
#endif
      pc_barrier();     // the consumer's last tile
#ifdef NT_STAMP
      STAMP(ph2);
      if (wave == 0 && lane == 0 && blockIdx.x < 16384) { g_dbg_role[4 * blockIdx.x + 0] = tw_; g_dbg_role[4 * blockIdx.x + 1] = tb_; }
#endif
      // weight-gradient partials of this pair into its own 32 KiB of the (now free) image area
      __syncthreads();
      if constexpr (PM1 > PM0) {
        // one lane-dependent base, the (register, tile) part of the index as an immediate offset
        float* const b3_ = s_part + W3_OFF + 4 * h * 64 + p;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int r0 = (reg & 3) + 8 * (reg >> 2);
#pragma unroll
          for (int m = PM0; m < PM1; ++m) b3_[r0 * 64 + 32 * m] = gW3[m][reg];
        }
      }
    };
    if (ti.channels <= 8) run_producer(std::integral_constant<int, 1>{});
    else if (ti.channels <= 16) run_producer(std::integral_constant<int, 2>{});
    else if (ti.channels <= 24) run_producer(std::integral_constant<int, 3>{});
    else run_producer(std::integral_constant<int, 4>{});
  } else {
    // the consumer is the later-dispatched wave of its SIMD (the arbitration loser at equal
    // priority): raise it once, statically (measured against the other assignments, profiles/NOTEBOOK.md A9.1)
#if NT_PC_PRIO == 0
    __builtin_amdgcn_s_setprio(1);
#endif
#ifdef NT_STAMP
    unsigned long long tw_ = 0, tb_ = 0, q0, q1, q2;
    unsigned long long st_[5] = {0, 0, 0, 0, 0};
#endif
    auto run_consumer = [&](auto ks_tag) {
    constexpr int KS3 = decltype(ks_tag)::value;   // k-steps of dH2 = W3^T dOut
    float16_t gW2[2][2], gW1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      gW1[i] = float16_t{0};
      gW2[i][0] = float16_t{0};
      gW2[i][1] = float16_t{0};
    }
    float16_t dabs = {0};   // per-lane sum |dF| per feature row (hash-grad fixed-point bound)
    constexpr int CM0 = NT_PC_DW3_CONSUMER == 2 ? 1 : 0, CM1 = NT_PC_DW3_CONSUMER ? 2 : 0;   // consumer's dW3 blocks
    float16_t gW3[2] = {float16_t{0}, float16_t{0}};
    for (int it = 0; it <= iters; ++it) {
      STAMP(q0);
      if (it > 0) {
        const int t = it - 1;
        const int slot = wk.first + (pr + t * PC_PAIRS) * 32 + p;
        const bool valid = slot < wk.last;
        const _Float16* set = pair + (t & 1) * SET_HALFS;
#if NT_PC_BATCH
        // Operands are fetched a stage ahead of the matrix instructions that use them, in batches,
        // and the data-gradient chain runs while the private image's store -> transposed-read
        // round trips are in flight (issued in the order written: each MFMA of the plain version
        // waited for an LDS read issued right in front of it, ~28 exposed LDS latencies per tile).
#define NT_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifdef NT_STAMP
        unsigned long long c0_, c1_, c2_, c3_, c4_, c5_;
        STAMP(c0_);
#define CSTAMP(v) STAMP(v)
#else
#define CSTAMP(v)
#endif
        half8_t dh2[4], dh1[4];
        float16_t dx;
        {
          // ---- stage 1: dH2 = W3^T dOut, masked by H2 > 0
          const half8_t d0 = *reinterpret_cast<const half8_t*>(set + SET_DOUT + p * S32 + 8 * h);
          half8_t d1 = {0, 0, 0, 0, 0, 0, 0, 0};     // channels 16..31: only degree-3 colour textures have them
          if constexpr (KS3 == 2) d1 = *reinterpret_cast<const half8_t*>(set + SET_DOUT + p * S32 + 16 + 8 * h);
          half8_t f3[4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (KS3 == 2 || !(i & 1)) f3[i] = s_frag[(32 + i) * 64 + lane];
          half8_t hr[4];
          load_frags_s<S64>(set + SET_H2, 0, hr[0], hr[1], p, h);
          load_frags_s<S64>(set + SET_H2, 32, hr[2], hr[3], p, h);
          NT_FENCE();
          float16_t a0 = {0}, a1 = {0};
          a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f3[0], d0, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f3[2], d0, a1, 0, 0, 0);
          if constexpr (KS3 == 2) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f3[1], d1, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f3[3], d1, a1, 0, 0, 0);
          }
          mask_pack(a0, hr[0], hr[1], dh2[0], dh2[1]);
          mask_pack(a1, hr[2], hr[3], dh2[2], dh2[3]);
        }
        store_frags_s<S64>(priv, 0, dh2[0], dh2[1], p, h);
        store_frags_s<S64>(priv, 32, dh2[2], dh2[3], p, h);
        NT_FENCE();
        CSTAMP(c1_);
        {
          // ---- stage 2: dH1 = W2^T dH2 (from registers) while the dH2 image lands
          half8_t f2[8], hr[4];
#pragma unroll
          for (int i = 0; i < 8; ++i) f2[i] = s_frag[(20 + i) * 64 + lane];
          load_frags_s<S64>(set + SET_H1, 0, hr[0], hr[1], p, h);
          load_frags_s<S64>(set + SET_H1, 32, hr[2], hr[3], p, h);
          NT_FENCE();
          float16_t a0 = {0}, a1 = {0};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f2[q], dh2[q], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f2[4 + q], dh2[q], a1, 0, 0, 0);
          }
          NT_FENCE();
          CSTAMP(c2_);
          // ---- stage 3: operands of dW2 += dH2 . H1^T (transposed reads), then the masks of dH1
          half8_t a2[2][2], b1[2][2];
#pragma unroll
          for (int sx = 0; sx < 2; ++sx)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
              a2[sx][m] = read_tr_s<S64>(priv, 32 * m, sx, lane);
              b1[sx][m] = read_tr_s<S64>(set + SET_H1, 32 * m, sx, lane);
            }
          mask_pack(a0, hr[0], hr[1], dh1[0], dh1[1]);
          mask_pack(a1, hr[2], hr[3], dh1[2], dh1[3]);
          NT_FENCE();
#pragma unroll
          for (int sx = 0; sx < 2; ++sx)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int mj = 0; mj < 2; ++mj)
                gW2[m][mj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[sx][m], b1[sx][mj], gW2[m][mj], 0, 0, 0);
        }
        if constexpr (CM1 > CM0) {
          // ---- dW3 += dOut . H2^T: operands from the finished images, independent of the chain
          half8_t a3[2], b3t[2][2];
#pragma unroll
          for (int sx = 0; sx < 2; ++sx) {
            a3[sx] = read_tr_s<S32>(set + SET_DOUT, 0, sx, lane);
#pragma unroll
            for (int m = CM0; m < CM1; ++m) b3t[sx][m] = read_tr_s<S64>(set + SET_H2, 32 * m, sx, lane);
          }
#pragma unroll
          for (int sx = 0; sx < 2; ++sx)
#pragma unroll
            for (int m = CM0; m < CM1; ++m)
              gW3[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3[sx], b3t[sx][m], gW3[m], 0, 0, 0);
        }
        NT_FENCE();
        CSTAMP(c3_);
        // ---- stage 4: dH1 image (dW2's reads of the private image are done), dX = W1^T dH1 meanwhile
        store_frags_s<S64>(priv, 0, dh1[0], dh1[1], p, h);
        store_frags_s<S64>(priv, 32, dh1[2], dh1[3], p, h);
        {
          half8_t f1[4], bxx[2];
#pragma unroll
          for (int q = 0; q < 4; ++q) f1[q] = s_frag[(28 + q) * 64 + lane];
#pragma unroll
          for (int sx = 0; sx < 2; ++sx) bxx[sx] = read_tr_s<S32>(set + SET_X, 0, sx, lane);
          NT_FENCE();
          dx = float16_t{0};
#pragma unroll
          for (int q = 0; q < 4; ++q) dx = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1[q], dh1[q], dx, 0, 0, 0);
          NT_FENCE();
          CSTAMP(c4_);
          // ---- stage 5: dW1 += dH1 . X^T
          half8_t a1t[2][2];
#pragma unroll
          for (int sx = 0; sx < 2; ++sx)
#pragma unroll
            for (int m = 0; m < 2; ++m) a1t[sx][m] = read_tr_s<S64>(priv, 32 * m, sx, lane);
          NT_FENCE();
#pragma unroll
          for (int sx = 0; sx < 2; ++sx)
#pragma unroll
            for (int m = 0; m < 2; ++m)
              gW1[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1t[sx][m], bxx[sx], gW1[m], 0, 0, 0);
        }
        CSTAMP(c5_);
#ifdef NT_STAMP
        st_[0] += c1_ - c0_; st_[1] += c2_ - c1_; st_[2] += c3_ - c2_; st_[3] += c4_ - c3_; st_[4] += c5_ - c4_;
#endif
#undef CSTAMP
#undef NT_FENCE
#else
        // ---- dH2 = W3^T dOut (B operand: this point's dOut row, natural channel order)
        half8_t dh2[4];
        {
          const half8_t d0 = *reinterpret_cast<const half8_t*>(set + SET_DOUT + p * S32 + 8 * h);
          half8_t d1 = {0, 0, 0, 0, 0, 0, 0, 0};     // channels 16..31: only degree-3 colour textures have them
          if constexpr (KS3 == 2) d1 = *reinterpret_cast<const half8_t*>(set + SET_DOUT + p * S32 + 16 + 8 * h);
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            float16_t a = {0};
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_frag[(32 + m * 2 + 0) * 64 + lane], d0, a, 0, 0, 0);
            if constexpr (KS3 == 2)
              a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_frag[(32 + m * 2 + 1) * 64 + lane], d1, a, 0, 0, 0);
            half8_t h0, h1;
            load_frags_s<S64>(set + SET_H2, 32 * m, h0, h1, p, h);
            mask_pack(a, h0, h1, dh2[2 * m], dh2[2 * m + 1]);
          }
        }
        // ---- dW3 += dOut . H2^T: operands from the finished images, independent of the chain
        if constexpr (CM1 > CM0) {
#pragma unroll
          for (int sx = 0; sx < 2; ++sx) {
            const half8_t a3 = read_tr_s<S32>(set + SET_DOUT, 0, sx, lane);
#pragma unroll
            for (int m = CM0; m < CM1; ++m)
              gW3[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3, read_tr_s<S64>(set + SET_H2, 32 * m, sx, lane), gW3[m], 0, 0, 0);
          }
        }
        // ---- dW2 += dH2 . H1^T
        store_frags_s<S64>(priv, 0, dh2[0], dh2[1], p, h);
        store_frags_s<S64>(priv, 32, dh2[2], dh2[3], p, h);
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
          const half8_t b1[2] = {read_tr_s<S64>(set + SET_H1, 0, sx, lane), read_tr_s<S64>(set + SET_H1, 32, sx, lane)};
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const half8_t a2 = read_tr_s<S64>(priv, 32 * m, sx, lane);
#pragma unroll
            for (int mj = 0; mj < 2; ++mj)
              gW2[m][mj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1[mj], gW2[m][mj], 0, 0, 0);
          }
        }
        // ---- dH1 = W2^T dH2, masked
        half8_t dh1[4];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          float16_t a = {0};
#pragma unroll
          for (int q = 0; q < 4; ++q)
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_frag[(20 + m * 4 + q) * 64 + lane], dh2[q], a, 0, 0, 0);
          half8_t h0, h1;
          load_frags_s<S64>(set + SET_H1, 32 * m, h0, h1, p, h);
          mask_pack(a, h0, h1, dh1[2 * m], dh1[2 * m + 1]);
        }
        // ---- dW1 += dH1 . X^T   (the private image is reused: dW2's reads are done)
        store_frags_s<S64>(priv, 0, dh1[0], dh1[1], p, h);
        store_frags_s<S64>(priv, 32, dh1[2], dh1[3], p, h);
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
          const half8_t bxx = read_tr_s<S32>(set + SET_X, 0, sx, lane);
#pragma unroll
          for (int m = 0; m < 2; ++m)
            gW1[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(read_tr_s<S64>(priv, 32 * m, sx, lane), bxx, gW1[m], 0, 0, 0);
        }
        // ---- dX = W1^T dH1 -> dF, in place over the features
        float16_t dx = {0};
#pragma unroll
        for (int q = 0; q < 4; ++q)
          dx = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_frag[(28 + q) * 64 + lane], dh1[q], dx, 0, 0, 0);
#endif
        if (valid) {
#pragma unroll
#if NT_PC_DABS_MOD   /* |x| as a source modifier of the add: 16 instructions instead of 16 and + 8 packed adds */
          for (int reg = 0; reg < 16; ++reg) {
            float a = dabs[reg];
            asm("v_add_f32 %0, %0, |%1|" : "+v"(a) : "v"(dx[reg]));
            dabs[reg] = a;
          }
#else
          for (int reg = 0; reg < 16; ++reg) dabs[reg] += fabsf(dx[reg]);
#endif
          unsigned* base = features + nt_feat_plane_base(plan, ti.type, 2 * h) +
                           nt_feat_in_plane(plan.n_levels, slot);
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              half2_t v;
              v.x = (_Float16)dx[4 * g + 2 * i];
              v.y = (_Float16)dx[4 * g + 2 * i + 1];
              base[(4 * g + i) * NT_FBLOCK] = __builtin_bit_cast(unsigned, v);
            }
        }
      }
      STAMP(q1);
      pc_barrier();
      STAMP(q2);
#ifdef NT_STAMP
      tw_ += q1 - q0; tb_ += q2 - q1;
#endif
    }
#ifdef NT_STAMP
    STAMP(ph2);
    if (wave == PC_PAIRS && lane == 0 && blockIdx.x < 16384) {
      g_dbg_role[4 * blockIdx.x + 2] = tw_; g_dbg_role[4 * blockIdx.x + 3] = tb_;
      for (int i = 0; i < 5; ++i) g_dbg_stage[8 * blockIdx.x + i] = st_[i];
    }
#endif
    __syncthreads();
    {
      // sum |dF| per feature row: the lanes' partial sums go to the (now free) fragment area and
      // are added up after the barrier below (80 dependent ds_bpermute steps per wave before)
      float* const d_ = reinterpret_cast<float*>(s_raw) + pr * (16 * PC_DABS_STRIDE) + 32 * h + p;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) d_[reg * PC_DABS_STRIDE] = dabs[reg];
      float* const b1_ = s_part + W1_OFF + 4 * h * 32 + p;
      float* const b2_ = s_part + W2_OFF + 4 * h * 64 + p;
      if constexpr (CM1 > CM0) {
        float* const b3_ = s_part + W3_OFF + 4 * h * 64 + p;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
#pragma unroll
          for (int m = CM0; m < CM1; ++m) b3_[((reg & 3) + 8 * (reg >> 2)) * 64 + 32 * m] = gW3[m][reg];
        }
      }
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r0 = (reg & 3) + 8 * (reg >> 2);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          b1_[(32 * m + r0) * 32] = gW1[m][reg];
#pragma unroll
          for (int mj = 0; mj < 2; ++mj) b2_[(32 * m + r0) * 64 + 32 * mj] = gW2[m][mj][reg];
        }
      }
    }
    };
    if (ti.channels <= 16) run_consumer(std::integral_constant<int, 1>{});
    else run_consumer(std::integral_constant<int, 2>{});
  }
  // workgroup reduction of the weight gradients: the four pairs' partials, summed in pair order
  // (all eight waves wrote theirs at once; the earlier scheme — pairs taking turns on one buffer,
  // eight barrier rounds of read-modify-write — was most of a run's 38 us of fixed cost), then one
  // global atomic per non-zero weight
  __syncthreads();
  {
    const float* s_all = reinterpret_cast<const float*>(s_img_all);
    float* gw = grad_weights + (long long)ptex * VSA_NT_WEIGHTS_PER_TEX;   // (atomics: K shells may share it)
    const int w3_end = W3_OFF + ti.channels * 64;
    for (int i = threadIdx.x; i < w3_end; i += PC_BLOCK) {
      float v = s_all[i];
#pragma unroll
      for (int w = 1; w < PC_PAIRS; ++w) v += s_all[w * PC_PART_FLOATS + i];
      if (v != 0.0f) atomicAdd(&gw[i], v * gw_scale);
    }
    if (threadIdx.x < 32 * PC_PAIRS) {
      const int f = threadIdx.x & 31, w = threadIdx.x >> 5;
      const float* d_ = reinterpret_cast<const float*>(s_raw) +
                        (w * 16 + (f & 3) + 4 * (f >> 3)) * PC_DABS_STRIDE + 32 * ((f >> 2) & 1);
      float v = 0.0f;
#pragma unroll 8
      for (int i = 0; i < 32; ++i) v += d_[i];
      if (v != 0.0f) atomicAdd(&dfeat_abs_sum[tex * 32 + f], v);
    }
  }
#ifdef NT_STAMP
  STAMP(ph3);
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
  if (threadIdx.x == 0 && blockIdx.x < 16384) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* d = g_dbg + 8 * blockIdx.x;
    d[0] = rt0; d[1] = rt1; d[2] = ((unsigned long long)xcc << 32) | hwid; d[3] = iters;
    d[4] = ph1 - ph0; d[5] = ph2 - ph1; d[6] = ph3 - ph2; d[7] = 1 + tex;
  }
#endif
}

// (The one-role backward "t" of round 4 — every weight-gradient operand formed transposed by the matrix
//  cores, no LDS hand-off; parity-green, 0.91 / 2.24 ms against 0.60 — left this file in round 5: it lives in
//  the history at e64f229 with its measurements in profiles/r04/mlp_bwd_t_*.txt and DESIGN.md section 5.)

// Persistent launch: gridDim.x workgroups (one per CU; the LDS footprint allows no more)
// split the frame's tiles evenly (nt_for_each_piece, nt_common.h): every active texture
// contributes PC_RUN_COST units of spacing (a run's staging + reduction, measured ~11 loop
// iterations) followed by one unit per 32-slot tile; workgroup w owns the tiles whose
// coordinate falls in [w*C/G, (w+1)*C/G).  Replaces one workgroup per 4096 slots of the
// worst-case capacity (70 % of them empty at the bench frame, each still needing the whole
// CU's LDS to launch and exit; in-kernel timeline: 196 of 256 CUs busy on average, 26 %
// of a workgroup's cycles in staging + reduction): 1.35 -> 0.88 ms.
#ifndef NT_PC_RUN_COST
#define NT_PC_RUN_COST 74
#endif
constexpr int PC_RUN_COST = NT_PC_RUN_COST;   // fitted: 38.6 us per run / 0.52 us per tile (tools/fit_cost.py)

__global__ __launch_bounds__(PC_BLOCK, 2) void nt_mlp_bwd_pc_kernel(
    vsa_nt_plan plan, const _Float16* __restrict__ weights,
    unsigned* __restrict__ features, const int* __restrict__ seg_start,
    _Float16* __restrict__ grad_rows, float* __restrict__ grad_weights,
    float* __restrict__ dfeat_abs_sum, float gw_scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  NT_SPAN_MARK(1, 0);
  NT_BAL_BEGIN();
  nt_for_each_piece<32>(plan, seg_start, 1, PC_RUN_COST,
                        [&](int, int tex, int first, int last, int seg_begin, int seg_end) {
    Work wk;
    wk.tex = tex;
    wk.seg_len = seg_end - seg_begin;
    wk.first = first;
    wk.last = last;
    pc_run(plan, wk, s_raw, weights, features, seg_start, grad_rows, grad_weights, dfeat_abs_sum, gw_scale);
    __syncthreads();   // the next run re-stages the fragments
  }, 0, 1 << 30, NtUnitWeight16(), NT_BAL_MLP_BWD);
  NT_SPAN_MARK(1, 1);
  NT_BAL_END(NT_BAL_MLP_BWD);
}

}  // namespace

#ifdef NT_SPAN
extern "C" int vsa_span_read_mlp(void* dst) {
  VSA_HIP_TRY(hipDeviceSynchronize());
  VSA_HIP_TRY(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_span), sizeof(g_span)));
  return 0;
}
#endif

#ifdef NT_STAMP
extern "C" int vsa_debug_read(void* dst) {
  VSA_HIP_TRY(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * 16384 * 8));
  VSA_HIP_TRY(hipMemcpyFromSymbol(static_cast<unsigned long long*>(dst) + 16384 * 8, HIP_SYMBOL(g_dbg_role),
                                  sizeof(unsigned long long) * 16384 * 4));
  VSA_HIP_TRY(hipMemcpyFromSymbol(static_cast<unsigned long long*>(dst) + 16384 * 12, HIP_SYMBOL(g_dbg_stage),
                                  sizeof(unsigned long long) * 16384 * 8));
  VSA_HIP_TRY(hipMemset(nullptr, 0, 0));
  return 0;
}
#endif

extern "C" int vsa_nt_mlp_fwd(const vsa_nt_plan* plan, const void* weights_h, const void* features,
                              const int32_t* seg_start, uint8_t* texels, void* pre_out,
                              void* stream) {
  if (!plan || !weights_h || !features || !seg_start || !texels) return VSA_ERR_ARG;
  if (plan->row_format < 0 || plan->row_format > 2) return VSA_ERR_UNSUPPORTED;
  int nr_cus = 0;
  { const int rc = vsa_cu_count(&nr_cus); if (rc) return rc; }
  if (plan->row_format != 0) {       // f16 rows: sigmoid(x) un-quantised (using_sh_quantization = 0), or x itself (using_sh_squeezing = 0)
    if (pre_out)
      hipLaunchKernelGGL((nt_mlp_fwd_kernel<true, true>), dim3(nr_cus * MLP_FWD_WGS_PER_CU), dim3(MLP_BLOCK), 0,
                         (hipStream_t)stream, *plan, reinterpret_cast<const _Float16*>(weights_h),
                         reinterpret_cast<const unsigned*>(features), seg_start,
                         reinterpret_cast<unsigned*>(texels), reinterpret_cast<_Float16*>(pre_out));
    else
      hipLaunchKernelGGL((nt_mlp_fwd_kernel<false, true>), dim3(nr_cus * MLP_FWD_WGS_PER_CU), dim3(MLP_BLOCK), 0,
                         (hipStream_t)stream, *plan, reinterpret_cast<const _Float16*>(weights_h),
                         reinterpret_cast<const unsigned*>(features), seg_start,
                         reinterpret_cast<unsigned*>(texels), nullptr);
    VSA_RETURN_LAUNCH_STATUS();
  }
  if (pre_out)
    hipLaunchKernelGGL(nt_mlp_fwd_kernel<true>, dim3(nr_cus * MLP_FWD_WGS_PER_CU), dim3(MLP_BLOCK), 0,
                       (hipStream_t)stream, *plan, reinterpret_cast<const _Float16*>(weights_h),
                       reinterpret_cast<const unsigned*>(features), seg_start,
                       reinterpret_cast<unsigned*>(texels), reinterpret_cast<_Float16*>(pre_out));
  else
    hipLaunchKernelGGL(nt_mlp_fwd_kernel<false>, dim3(nr_cus * MLP_FWD_WGS_PER_CU), dim3(MLP_BLOCK), 0,
                       (hipStream_t)stream, *plan, reinterpret_cast<const _Float16*>(weights_h),
                       reinterpret_cast<const unsigned*>(features), seg_start,
                       reinterpret_cast<unsigned*>(texels), nullptr);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_nt_mlp_bwd(const vsa_nt_plan* plan, const void* weights_h, void* features,
                              const int32_t* seg_start, uint16_t* grad_rows, float* grad_weights,
                              float* dfeat_abs_sum, float weight_grad_scale, void* stream) {
  if (!plan || !weights_h || !features || !seg_start || !grad_rows || !grad_weights ||
      !dfeat_abs_sum)
    return VSA_ERR_ARG;
  static bool attr_set = false;
  if (!attr_set) {
    VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(nt_mlp_bwd_pc_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
    attr_set = true;
  }
  // fragments + the larger of {image sets, the four pairs' weight-gradient partials (4 x 32 KiB)}
  constexpr size_t img_bytes = (size_t)PC_PAIRS * PAIR_HALFS * 2, part_bytes = (size_t)PC_PAIRS * PC_PART_FLOATS * 4;
  const size_t lds = (size_t)PC_FRAGS * 64 * 16 + (img_bytes > part_bytes ? img_bytes : part_bytes);
  int nr_cus = 0;
  { const int rc = vsa_cu_count(&nr_cus); if (rc) return rc; }
  hipLaunchKernelGGL(nt_mlp_bwd_pc_kernel, dim3(nr_cus), dim3(PC_BLOCK), lds, (hipStream_t)stream,
                     *plan, reinterpret_cast<const _Float16*>(weights_h),
                     reinterpret_cast<unsigned*>(features), seg_start,
                     reinterpret_cast<_Float16*>(grad_rows), grad_weights,
                     dfeat_abs_sum, weight_grad_scale);
  VSA_RETURN_LAUNCH_STATUS();
}
