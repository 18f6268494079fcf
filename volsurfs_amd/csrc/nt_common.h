// Shared device helpers for the neural-texture kernels (gfx950).
#pragma once
#include "common.h"

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));

#define NT_DOM_BLOCK 4096  // texel domains are padded to this (one scan block)

// Feature planes are blocked: [type][slot / 256][level][slot % 256] (f16x2 each),
// so that the 16 levels of a 32-slot MLP tile sit within one 16 KiB block (a
// level-major [type][level][slot] layout spreads a tile over 16 pages that are
// ~80 MB apart and the MLP kernels become TLB-miss bound), while the encode
// kernels still write / read 1 KiB contiguous runs per level.
#define NT_FBLOCK 256
__device__ __forceinline__ long long nt_feat_index(const vsa_nt_plan& p, int type, int level,
                                                   int slot) {
  const long long blocks = p.slot_capacity / NT_FBLOCK;
  return (((long long)type * blocks + (slot >> 8)) * p.n_levels + level) * NT_FBLOCK + (slot & 255);
}
// (type, level) fixed: element offset = plane_base + (slot >> 8) * n_levels * 256 + (slot & 255)
__device__ __forceinline__ long long nt_feat_plane_base(const vsa_nt_plan& p, int type, int level) {
  return ((long long)type * (p.slot_capacity / NT_FBLOCK) * p.n_levels + level) * NT_FBLOCK;
}
__device__ __forceinline__ long long nt_feat_in_plane(int n_levels, int slot) {
  return (long long)(slot >> 8) * (n_levels * NT_FBLOCK) + (slot & 255);
}

__host__ __device__ inline int nt_tex_index(int shell, int type, int deg) {
  return (shell * 2 + type) * VSA_NT_MAX_DEG + deg;
}

// Per-degree row layout of texels / grad_rows (include/volsurfs_hip.h: VSA_NT_ROW_QUADS)
__host__ __device__ inline int nt_row_quads(int d) { return VSA_NT_ROW_QUADS(d); }
__host__ __device__ inline int nt_alpha_quad(int d) { return VSA_NT_ALPHA_QUAD(d); }

// The 2x2 texel footprint of a uv sample in a texture of resolution R
// (models/neural_texture.py:107-138 with align_to_webgl; oracle/neural_texture.py
// texel_corners).  (i0, j0) is the lower corner in texel units, may be -1;
// the footprint lives at extended-grid coordinates (i0+1 .. i0+2, j0+1 .. j0+2).
struct NtFootprint {
  int i0, j0;
  float fx, fy;
};

// anchor (models/neural_texture.py:88-104, `anchor=True`): the sample snaps to ONE texel — pixel
// (floor(u R), floor(v R)) clamped to the texture (uv_coords_to_pix), rotated i, j -> (R - 1) - j, i — and
// takes its value unblended: (i0, j0) is that texel, fx = fy = 0 (lerp weights 1, 0, 0, 0).
__device__ __forceinline__ NtFootprint nt_footprint(float u, float v, int R, bool anchor = false) {
  const float Rf = (float)R;
  const float a = u * Rf, b = v * Rf;  // non_normalize_uv_coord
  if (anchor) {
    NtFootprint f;
    f.fx = f.fy = 0.f;
    f.i0 = (R - 1) - min(max((int)floorf(b), 0), R - 1);
    f.j0 = min(max((int)floorf(a), 0), R - 1);
    return f;
  }
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

// Which shells have an alpha model at all (methods/volsurfs.py:167-206): a solid inner mesh has none; with ONE alpha
// model for all shells (shared_alpha) the reference's loop stores models["alpha"] = None at i = 0 and leaves, so a
// solid inner mesh then switches the alpha model off on EVERY shell.
__host__ __device__ inline bool nt_shell_has_alpha(const vsa_nt_plan& p, int shell) {
  return !(p.inner_solid && (shell == 0 || p.shared_alpha));
}

// Parameter texture of logical texture `tex`: itself, or shell 0's texture of the same type and degree when the
// models of that type are shared by all shells (plan.shared_rgb / shared_alpha; volsurfs.py:524-527, 553-556).
// Slots, feature planes, texel rows and dfeat_abs_sum stay indexed by the LOGICAL texture.
__device__ __forceinline__ int nt_param_tex(const vsa_nt_plan& p, int tex) {
  const bool shared = ((tex / VSA_NT_MAX_DEG) & 1) ? p.shared_alpha != 0 : p.shared_rgb != 0;
  return shared ? tex % (2 * VSA_NT_MAX_DEG) : tex;
}

__device__ __forceinline__ bool tex_active(const vsa_nt_plan& p, int tex) {
  const int deg = tex % VSA_NT_MAX_DEG;
  const int type = (tex / VSA_NT_MAX_DEG) & 1;
  const int shell = tex / (2 * VSA_NT_MAX_DEG);
  if (type == 0) return deg < p.rgb_degrees;
  if (!nt_shell_has_alpha(p, shell)) return false;
  return deg < p.alpha_degrees;
}

struct Work {
  int tex, first, last;  // slots [first, last) of texture tex
  int seg_len;           // slots in the whole (shell, degree) segment
};

// Persistent work split (one workgroup per CU for the LDS-hungry kernels).  The launch's
// work = n_planes x (every active texture's slots), laid out on one cost axis: plane-major,
// and within a plane each non-empty texture contributes `ovh` units of spacing (the per-piece
// setup: staging / zeroing / flushing LDS, fitted from per-workgroup timings, tools/fit_cost.py)
// followed by ceil(len / UNIT) units of UNIT slots, each weighted by wt(plane, degree, type) / 16
// (a unit's relative cost where the kernel takes different code paths per level, or per output
// width: type 0 = colour, 1 = alpha texture).  The axis is kept
// in 1/16 units so that weighted lengths stay exact.  Workgroup w owns the stretch
// [w*C/G, (w+1)*C/G) and calls body(plane, tex, first_slot, last_slot, seg_begin, seg_end) once
// per (plane, texture) it overlaps; a unit belongs to the workgroup whose stretch contains its
// start.  All of it is wave-uniform scalar code.
// Replaces grids sized for the worst-case slot capacity, where ~2/3 of the workgroups
// found nothing to do yet each needed a whole CU's LDS to launch and exit.
struct NtUnitWeight16 {
  __device__ int operator()(int, int, int) const { return 16; }
};

// ---- measured-time rebalancing of the work split (plan.balance; vsa_nt_rebalance).  The cost axis is
// a fitted model; what it cannot know (which CU a workgroup shares with whom, table locality, the
// scene) shows up as 4-11 % of span lost to the slowest workgroup (tools/wg_span.py).  Every
// persistent kernel stamps its workgroups' busy times; once per frame nt_rebalance_kernel turns the
// previous frame's times into shares of the axis.  Pieces and arithmetic are unchanged - only the
// [lo, hi) a workgroup takes - so results are bit-identical to the equal split.
constexpr int NT_BAL_KERNELS = 6, NT_BAL_MAX_WG = 1024, NT_BAL_ONE = 1 << 24;
enum { NT_BAL_ENC_FWD_D, NT_BAL_ENC_FWD_H, NT_BAL_ENC_BWD_D, NT_BAL_ENC_BWD_H, NT_BAL_MLP_FWD, NT_BAL_MLP_BWD };
struct NtBalance {
  unsigned frac[NT_BAL_KERNELS][NT_BAL_MAX_WG + 1];   // cumulative share, NT_BAL_ONE = the whole axis
  unsigned ticks[NT_BAL_KERNELS][NT_BAL_MAX_WG];      // busy time per workgroup of the last launch (100 MHz)
  int frac_wgs[NT_BAL_KERNELS];                       // grid size frac[k] was made for (0: equal shares)
  int tick_wgs[NT_BAL_KERNELS];                       // grid size of the launch that wrote ticks[k]
  float ema[NT_BAL_KERNELS][NT_BAL_MAX_WG];           // running mean of the times (nt_rebalance_kernel)
};

__device__ __forceinline__ void nt_split_range(const vsa_nt_plan& plan, int bal_id, long long total,
                                               long long& lo, long long& hi) {
  const NtBalance* b = static_cast<const NtBalance*>(plan.balance);
  if (b && bal_id >= 0 && b->frac_wgs[bal_id] == (int)gridDim.x) {
    lo = (total * b->frac[bal_id][blockIdx.x]) >> 24;        // total < 2^38: no overflow
    hi = (total * b->frac[bal_id][blockIdx.x + 1]) >> 24;
  } else {
    lo = total * blockIdx.x / gridDim.x;
    hi = total * (blockIdx.x + 1) / gridDim.x;
  }
}

__device__ __forceinline__ unsigned long long nt_bal_now() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define NT_BAL_BEGIN() const unsigned long long bal_t0__ = plan.balance ? nt_bal_now() : 0ull
#define NT_BAL_END(id)                                                                   \
  if (plan.balance && threadIdx.x == 0 && gridDim.x <= NT_BAL_MAX_WG) {                  \
    NtBalance* b__ = static_cast<NtBalance*>(plan.balance);                              \
    b__->ticks[id][blockIdx.x] = (unsigned)(nt_bal_now() - bal_t0__);                    \
    if (blockIdx.x == 0) b__->tick_wgs[id] = (int)gridDim.x;                             \
  }

template <int UNIT, typename Body, typename Weight = NtUnitWeight16>
__device__ __forceinline__ void nt_for_each_piece_scalar(const vsa_nt_plan& plan,
                                                  const int* __restrict__ seg_start, int n_planes,
                                                  int ovh, Body&& body, int tex_begin = 0,
                                                  int tex_end = 1 << 30, Weight wt = Weight(), int bal_id = -1) {
  const int n_all = plan.nr_shells * 2 * VSA_NT_MAX_DEG;
  const int n_tex = tex_end < n_all ? tex_end : n_all;   // textures [tex_begin, n_tex) only
  // units per (type, degree) class.  Only ever indexed with compile-time constants (the update
  // below is a fully unrolled compare-and-add): a dynamically indexed private array lands in
  // scratch memory, and the ~40 dependent scratch round trips of this prologue cost every
  // persistent kernel 15-20 us per launch
  long long units_td[2 * VSA_NT_MAX_DEG] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long pieces = 0;
  for (int tex = tex_begin; tex < n_tex; ++tex) {
    if (!tex_active(plan, tex)) continue;
    const int deg = tex % VSA_NT_MAX_DEG;
    const int sd = (tex / (2 * VSA_NT_MAX_DEG)) * VSA_NT_MAX_DEG + deg;
    const int len = seg_start[sd + 1] - seg_start[sd];
    if (len > 0) {
      const int cls = ((tex / VSA_NT_MAX_DEG) & 1) * VSA_NT_MAX_DEG + deg;
      const long long n_units = (len + UNIT - 1) / UNIT;
#pragma unroll
      for (int c = 0; c < 2 * VSA_NT_MAX_DEG; ++c) units_td[c] += c == cls ? n_units : 0;
      pieces += 1;
    }
  }
  if (pieces == 0) return;
  auto plane_cost = [&](int pl) {
    long long c = pieces * ovh * 16;
#pragma unroll
    for (int d = 0; d < VSA_NT_MAX_DEG; ++d) c += units_td[d] * wt(pl, d, 0) + units_td[VSA_NT_MAX_DEG + d] * wt(pl, d, 1);
    return c;
  };
  long long total = 0;
  for (int pl = 0; pl < n_planes; ++pl) total += plane_cost(pl);
  long long lo, hi;
  nt_split_range(plan, bal_id, total, lo, hi);
  if (hi <= lo) return;
  long long c0 = 0;
  for (int pl = 0; pl < n_planes && c0 < hi; ++pl) {
    const long long pc = plane_cost(pl);
    if (c0 + pc <= lo) {
      c0 += pc;
      continue;
    }
    for (int tex = tex_begin; tex < n_tex; ++tex) {
      if (!tex_active(plan, tex)) continue;
      const int deg = tex % VSA_NT_MAX_DEG;
      const int sd = (tex / (2 * VSA_NT_MAX_DEG)) * VSA_NT_MAX_DEG + deg;
      const int begin = seg_start[sd], end = seg_start[sd + 1];
      if (end <= begin) continue;
      const int units = (end - begin + UNIT - 1) / UNIT, w = wt(pl, deg, (tex / VSA_NT_MAX_DEG) & 1);
      const long long t0 = c0 + (long long)ovh * 16;
      c0 = t0 + (long long)units * w;
      if (t0 >= hi) break;
      const long long a = lo > t0 ? lo - t0 : 0, b = hi - t0 < (long long)units * w ? hi - t0 : (long long)units * w;
      if (b <= a) continue;
      const int ua = (int)((a + w - 1) / w), ub = (int)((b + w - 1) / w);   // units whose start lies in [a, b)
      if (ub <= ua) continue;
      const int first = begin + ua * UNIT;
      const long long lastl = (long long)begin + (long long)ub * UNIT;
      body(pl, tex, first, lastl < end ? (int)lastl : end, begin, end);
    }
  }
}

// Inclusive prefix sum over the 64 lanes of a wave (DPP: four shifts within the rows of 16, then
// the row broadcasts 15 and 31); lane 63 holds the total.
__device__ __forceinline__ int nt_wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
  return v;
}

// The split with ONE TEXTURE PER LANE (up to 64 textures = 8 shells; more fall back to the scalar
// walk below).  The scalar version looks the segment table and the plan up ~100 times per
// workgroup with dependent scalar loads: 24 us per workgroup at the head of every persistent
// launch (stamps: 19 k cycles for the class counts, 9 k for the plane costs, 23 k for the scan —
// `-DNT_SPAN`, g_split), whatever the size of the frame.  Here every lane loads its texture's
// segment once (two vector loads per pass), the plane costs are wave sums, a texture's position in
// a plane is a wave prefix sum, and the few textures that overlap the workgroup's stretch are
// picked off a ballot.  Same axis, same integer arithmetic, same pieces.
// paired = true (<= 64 textures only): a piece is a (shell, degree) PAIR — the colour and the alpha
// texture of one shell and degree share their slots, texel centres, grid cells and corner weights
// (models/sh_neural_textures.py:41-60: one resolution per degree for both), so the dense-level encode
// kernels form cell, indices and weights once and gather / blend twice.  The pair is carried by the
// lane of its colour texture (by the alpha texture's where the colour texture is inactive); its
// weight is wt(plane, degree, 2) when both textures are active.  body() receives the carrying
// texture; nt_pair_partner() tells whether (and which) second texture rides along.
__device__ __forceinline__ int nt_pair_partner(const vsa_nt_plan& p, int tex) {
  const int other = tex ^ VSA_NT_MAX_DEG;          // same shell and degree, the other type
  return ((tex / VSA_NT_MAX_DEG) & 1) == 0 && tex_active(p, other) ? other : -1;
}

template <int UNIT, typename Body, typename Weight = NtUnitWeight16>
__device__ __forceinline__ void nt_for_each_piece(const vsa_nt_plan& plan,
                                                  const int* __restrict__ seg_start, int n_planes,
                                                  int ovh, Body&& body, int tex_begin = 0,
                                                  int tex_end = 1 << 30, Weight wt = Weight(), int bal_id = -1,
                                                  bool paired = false) {
  const int n_all = plan.nr_shells * 2 * VSA_NT_MAX_DEG;
  if (n_all > 64) {
    nt_for_each_piece_scalar<UNIT>(plan, seg_start, n_planes, ovh, body, tex_begin, tex_end, wt, bal_id);
    return;
  }
  const int n_tex = tex_end < n_all ? tex_end : n_all;   // textures [tex_begin, n_tex) only
  const int lane = threadIdx.x & 63;
  // this lane's texture (lanes beyond the textures, inactive or empty textures: units = 0)
  const int tex = lane;
  const int deg = tex % VSA_NT_MAX_DEG, type = (tex / VSA_NT_MAX_DEG) & 1, shell = tex / (2 * VSA_NT_MAX_DEG);
  const int rgb_deg = plan.rgb_degrees, alpha_deg = plan.alpha_degrees;
  const bool shell_alpha = nt_shell_has_alpha(plan, shell);
  bool act = tex >= tex_begin && tex < n_tex &&
             (type == 0 ? deg < rgb_deg : (shell_alpha && deg < alpha_deg));
  int wtype = type;
  if (paired) {
    const bool rgb_act = deg < rgb_deg, alpha_act = shell_alpha && deg < alpha_deg;
    if (type == 1 && rgb_act) act = false;            // rides along with its colour texture
    if (type == 0 && act && alpha_act) wtype = 2;     // a pair
  }
  const int sd = shell * VSA_NT_MAX_DEG + deg;
  int begin = 0, end = 0;
  if (act) {
    begin = seg_start[sd];
    end = seg_start[sd + 1];
  }
  act = act && end > begin;
  const int units = act ? (end - begin + UNIT - 1) / UNIT : 0;
  const unsigned long long act_mask = __ballot(act);
  const long long pieces = __popcll(act_mask);
  if (pieces == 0) return;
  // a texture's cost in plane pl (1/16 units; < 2^31: at most 64 textures x (a few million slots / UNIT) x 16)
  auto lane_cost = [&](int pl) { return act ? ovh * 16 + units * wt(pl, deg, wtype) : 0; };
  auto plane_cost = [&](int pl) -> long long {
    return (long long)(unsigned)__builtin_amdgcn_readlane(nt_wave_incl_scan(lane_cost(pl)), 63);
  };
  long long total = 0;
  for (int pl = 0; pl < n_planes; ++pl) total += plane_cost(pl);
  long long lo, hi;
  nt_split_range(plan, bal_id, total, lo, hi);
  if (hi <= lo) return;
  long long c0 = 0;
  for (int pl = 0; pl < n_planes && c0 < hi; ++pl) {
    const int cl = lane_cost(pl);
    const int incl = nt_wave_incl_scan(cl);
    const long long pc = (long long)(unsigned)__builtin_amdgcn_readlane(incl, 63);
    if (c0 + pc <= lo) {
      c0 += pc;
      continue;
    }
    // lane's texture occupies [t0, t0 + ovh*16 + units*w) of the axis; its units start at t0 + ovh*16
    const long long t0l = c0 + (long long)(incl - cl) + (long long)ovh * 16;
    const int w = wt(pl, deg, wtype);
    const long long span = (long long)units * w;
    // the same tests as the scalar walk: t0 < hi, and [max(lo - t0, 0), min(hi - t0, span)) non-empty in whole units
    bool hit = act && t0l < hi;
    int ua = 0, ub = 0;
    if (hit) {
      const long long a = lo > t0l ? lo - t0l : 0, b = hi - t0l < span ? hi - t0l : span;
      hit = b > a;
      if (hit) {
        ua = (int)(((unsigned)a + (unsigned)w - 1u) / (unsigned)w);      // units whose start lies in [a, b); a, b <= span < 2^31
        ub = (int)(((unsigned)b + (unsigned)w - 1u) / (unsigned)w);
        hit = ub > ua;
      }
    }
    unsigned long long todo = __ballot(hit);
    while (todo) {
      const int j = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int jb = __builtin_amdgcn_readlane(begin, j), je = __builtin_amdgcn_readlane(end, j);
      const int ja = __builtin_amdgcn_readlane(ua, j), jub = __builtin_amdgcn_readlane(ub, j);
      const int first = jb + ja * UNIT;
      const long long lastl = (long long)jb + (long long)jub * UNIT;
      body(pl, j, first, lastl < je ? (int)lastl : je, jb, je);
    }
    c0 += pc;
  }
}

// Diagnostic build only (make EXTRA=-DNT_SPAN; tools/wg_span.py): wall-clock begin / end of
// every workgroup of the persistent kernels, to see how evenly the cost axis splits the work.
#ifdef NT_SPAN
static __device__ unsigned long long g_span[4][2048][3];
#define NT_SPAN_MARK(id, which)                                                          \
  if (threadIdx.x == 0 && blockIdx.x < 2048) {                                           \
    unsigned long long t__;                                                              \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");     \
    g_span[id][blockIdx.x][which] = t__;                                                 \
    if (which == 0) {                                                                    \
      unsigned hw__, xcc__;                                                              \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw__));                 \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc__));               \
      g_span[id][blockIdx.x][2] = ((unsigned long long)xcc__ << 32) | hw__;             \
    }                                                                                    \
  }
#else
#define NT_SPAN_MARK(id, which)
#endif

