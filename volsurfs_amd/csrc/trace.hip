// K-shell closest-hit ray / mesh intersection (trace half).
//
// Replaces the K sequential `self.raytracer.trace(rays_o, rays_d, mesh_id=i)`
// calls and their K host syncs on `any_hit`
// (/root/reference/volsurfs_py/methods/volsurfs.py:476-485; raytracelib itself
// is not under /root/reference) with ONE launch over (ray tile, mesh):
// grid.y = mesh, one lane per ray, per-lane traversal stack staged in LDS as
// stack[depth][lane] (bank = lane -> conflict free), 64-byte nodes holding both
// children's boxes (4 x dwordx4 per visit), leaf triangles as contiguous
// (v0,e1,e2) float4 triples; "while-while" control flow (inner-node walk until the
// whole wave holds leaves, then one converged triangle phase with the triangle
// loads issued up front); measured 0.90 ms (leaves tested inline at every node
// visit, 256-thread workgroups) -> 0.45 (while-while) -> 0.34 ms (one-wave workgroups).
//
// Closest hit is defined order-independently (smallest t, ties -> smallest face
// id) with the triangle test evaluated by one fixed fp32 formula, so the result
// is bit-identical to the brute-force oracle (oracle/raytrace_ref.c).
#include "common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int TRACE_BLOCK = 64;   // one wave per workgroup: a finished wave frees its stack at once
constexpr int TRACE_STACK = 48;

struct Roots {
  int root[VSA_MAX_SHELLS];
};

struct Hit {
  float t, u, v;
  int slot;  // index into the leaf-ordered triangle array, -1 = miss
  int id;    // original face id (tie break)
};

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
  return (ax * bx + ay * by) + az * bz;
}

// Moeller-Trumbore, two-sided, fixed evaluation order (mirrored by the oracle).
__device__ __forceinline__ void tri_test(const float4 v0, const float4 e1, const float4 e2,
                                         float ox, float oy, float oz, float dx, float dy,
                                         float dz, float t_min, int slot, Hit& best) {
  float px = dy * e2.z - dz * e2.y;
  float py = dz * e2.x - dx * e2.z;
  float pz = dx * e2.y - dy * e2.x;
  float det = dot3(e1.x, e1.y, e1.z, px, py, pz);
  if (fabsf(det) < 1e-20f) return;
  float inv = 1.0f / det;
  float tx = ox - v0.x, ty = oy - v0.y, tz = oz - v0.z;
  float u = dot3(tx, ty, tz, px, py, pz) * inv;
  if (!(u >= 0.0f && u <= 1.0f)) return;
  float qx = ty * e1.z - tz * e1.y;
  float qy = tz * e1.x - tx * e1.z;
  float qz = tx * e1.y - ty * e1.x;
  float v = dot3(dx, dy, dz, qx, qy, qz) * inv;
  if (!(v >= 0.0f && u + v <= 1.0f)) return;
  float t = dot3(e2.x, e2.y, e2.z, qx, qy, qz) * inv;
  if (!(t > t_min)) return;
  int id = __float_as_int(v0.w);
  if (t < best.t || (t == best.t && id < best.id)) {
    best.t = t;
    best.u = u;
    best.v = v;
    best.slot = slot;
    best.id = id;
  }
}

__device__ __forceinline__ bool box_test(float lx, float ly, float lz, float hx, float hy,
                                         float hz, float ox, float oy, float oz, float ix,
                                         float iy, float iz, float t_min, float t_max,
                                         float& t_near) {
  float a = (lx - ox) * ix, b = (hx - ox) * ix;
  float tn = fminf(a, b), tf = fmaxf(a, b);
  a = (ly - oy) * iy;
  b = (hy - oy) * iy;
  tn = fmaxf(tn, fminf(a, b));
  tf = fminf(tf, fmaxf(a, b));
  a = (lz - oz) * iz;
  b = (hz - oz) * iz;
  tn = fmaxf(tn, fminf(a, b));
  tf = fminf(tf, fmaxf(a, b));
  t_near = tn;
  // widen by a few ulps: the slab arithmetic is not the triangle arithmetic
  return tn <= tf * 1.0000004f + 1e-30f && tf >= t_min && tn <= t_max;
}

// Traversal state per lane: an inner node index (>= 0), a leaf code (< 0:
// ~((first << 4) | count)), or TRACE_EMPTY.  "while-while": the wave first walks
// inner nodes until every lane holds a leaf (or is done), then all lanes test their
// leaf triangles together -- lanes no longer sit idle through other lanes' triangle
// loops at every node visit.
constexpr int TRACE_EMPTY = 0x7fffffff;

// Diagnostic build only (tools/build_variant.sh span "-DTRACE_SPAN"; tools/trace_span.py): wall-clock
// begin / end of every wave of trace_q_kernel and WAVE-level counts (an elected lane adds to LDS; a
// per-lane variable would only count that lane's own trips): trips of the walk loop, lane visits, leaf
// phases.  g_torder, when set, replaces the dispatch order (the launch-order experiments of profiles/NOTEBOOK.md A9.4).
#ifdef TRACE_SPAN
constexpr int TRACE_SPAN_WAVES = 1 << 17;
static __device__ unsigned long long g_tspan[TRACE_SPAN_WAVES][3];
static __device__ int g_torder[TRACE_SPAN_WAVES];
static __device__ int g_torder_on;
__device__ __forceinline__ unsigned long long trace_now() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
enum { SP_ROUNDS, SP_VISITS, SP_OUTER };
#define SPAN_ADD(k, v)                                                                        \
  do {                                                                                        \
    const unsigned v__ = (unsigned)(v);                                                       \
    if ((int)threadIdx.x == __builtin_ctzll(__builtin_amdgcn_read_exec())) s_span[k] += v__;  \
  } while (0)
#endif

template <int STACK>
__global__ __launch_bounds__(TRACE_BLOCK) void trace_ww_kernel(
    const float4* __restrict__ nodes, const float4* __restrict__ tris, Roots roots,
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, int N, float t_min,
    float* __restrict__ hit_t, int* __restrict__ hit_slot, float* __restrict__ hit_uv) {
  __shared__ int s_stack[STACK][TRACE_BLOCK];
  const int lane = threadIdx.x;
  const long long n = (long long)blockIdx.x * TRACE_BLOCK + lane;
  const int mesh = blockIdx.y;
  if (n >= N) return;
  const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
  const float dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
  const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;

  Hit best;
  best.t = INFINITY;
  best.u = best.v = 0.f;
  best.slot = -1;
  best.id = 0x7fffffff;

  int cur = roots.root[mesh];
  int sp = 0;
  while (cur != TRACE_EMPTY) {
    while ((unsigned)cur < (unsigned)TRACE_EMPTY) {
      const float4 q0 = nodes[4 * (long long)cur + 0];
      const float4 q1 = nodes[4 * (long long)cur + 1];
      const float4 q2 = nodes[4 * (long long)cur + 2];
      const float4 q3 = nodes[4 * (long long)cur + 3];
      const int ref0 = __float_as_int(q3.x), ref1 = __float_as_int(q3.y);
      const int cnt0 = __float_as_int(q3.z), cnt1 = __float_as_int(q3.w);
      float tn0, tn1;
      bool h0 = box_test(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, ox, oy, oz, ix, iy, iz, t_min, best.t, tn0);
      bool h1 = box_test(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, ox, oy, oz, ix, iy, iz, t_min, best.t, tn1);
      const int c0 = ref0 < 0 ? ~(((~ref0) << 4) | cnt0) : ref0;
      const int c1 = ref1 < 0 ? ~(((~ref1) << 4) | cnt1) : ref1;
      h0 = h0 && !(ref0 < 0 && cnt0 == 0);
      h1 = h1 && !(ref1 < 0 && cnt1 == 0);
      if (h0 && h1) {
        const bool swap = tn1 < tn0;
        s_stack[sp++][lane] = swap ? c0 : c1;
        cur = swap ? c1 : c0;
      } else if (h0) {
        cur = c0;
      } else if (h1) {
        cur = c1;
      } else {
        cur = sp ? s_stack[--sp][lane] : TRACE_EMPTY;
      }
    }
    if (cur != TRACE_EMPTY) {
      const int code = ~cur;
      const int first = code >> 4, cnt = code & 15;
      // issue the loads of up to 4 triangles before the first test
      for (int i0 = 0; i0 < cnt; i0 += 4) {
        float4 tv[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const long long s = first + min(i0 + i, cnt - 1);
          tv[i][0] = tris[3 * s];
          tv[i][1] = tris[3 * s + 1];
          tv[i][2] = tris[3 * s + 2];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i0 + i < cnt)
            tri_test(tv[i][0], tv[i][1], tv[i][2], ox, oy, oz, dx, dy, dz, t_min, first + i0 + i, best);
      }
      cur = sp ? s_stack[--sp][lane] : TRACE_EMPTY;
    }
  }
  const long long o = (long long)mesh * N + n;
  hit_t[o] = best.slot >= 0 ? best.t : 0.0f;
  hit_slot[o] = best.slot;
  hit_uv[2 * o] = best.u;
  hit_uv[2 * o + 1] = best.v;
}

// ---- quantised nodes (vsa_bvh_export_q): 32 B per node instead of 64.  The inner-node walk
// is bound by the texture-address path (every lane fetches its own node: 64 lanes x 64 B per
// visit), so halving the node halves that traffic.  The ray is moved into each mesh's
// 16-bit grid once (o_g = (o - lo) / step + 1, d_g = d / step: the slab parameter t is
// unchanged), child boxes are tested directly on their u16 coordinates; the boxes were
// rounded outward by more than the fp32 error of that test, and triangles are still tested
// with the original ray, so the closest hit is bit-identical to the fp32-node kernel.
struct Frames {
  float f[VSA_MAX_SHELLS][6];   // lo.xyz, step.xyz
};

typedef float f32x2_t __attribute__((ext_vector_type(2)));

// Slab test on grid coordinates, ~20 VALU ops per box (the fp32-node test is ~40 and PMC
// showed the traversal VALU-bound: 68 % VALU-busy at 37 % lane utilisation): per axis ONE
// packed FMA gives both plane parameters, t = q * (1/d_g) - o_g/d_g.  The different rounding
// (and the NaN an axis-parallel ray produces, which min/max then ignore, i.e. that axis's
// constraint is dropped) can only make the test pass more often; the boxes carry a one-unit
// outward margin, so nothing reachable is pruned.
struct QRay {
  f32x2_t ix, iy, iz;   // (1/d_g, 1/d_g) per axis
  f32x2_t cx, cy, cz;   // (-o_g/d_g, -o_g/d_g)
};

__device__ __forceinline__ bool qbox_test(unsigned w0, unsigned w1, unsigned w2, const QRay& r,
                                          float t_min, float t_max, float& t_near) {
  const f32x2_t qx = {(float)(w0 & 0xffffu), (float)(w1 >> 16)};
  const f32x2_t qy = {(float)(w0 >> 16), (float)(w2 & 0xffffu)};
  const f32x2_t qz = {(float)(w1 & 0xffffu), (float)(w2 >> 16)};
  const f32x2_t tx = __builtin_elementwise_fma(qx, r.ix, r.cx);
  const f32x2_t ty = __builtin_elementwise_fma(qy, r.iy, r.cy);
  const f32x2_t tz = __builtin_elementwise_fma(qz, r.iz, r.cz);
  const float tn = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fminf(tz.x, tz.y));
  const float tf = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fmaxf(tz.x, tz.y));
  t_near = tn;
  return tn <= tf && tf >= t_min && tn <= t_max;
}

template <int STACK>
__global__ __launch_bounds__(TRACE_BLOCK) void trace_q_kernel(
    const uint4* __restrict__ qnodes, const float4* __restrict__ tris, Roots roots, Frames frames,
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, int N, float t_min,
    float* __restrict__ hit_t, int* __restrict__ hit_slot, float* __restrict__ hit_uv, int rpw) {
  __shared__ int s_stack[STACK][TRACE_BLOCK];
  const int lane = threadIdx.x;
#ifndef TRACE_SPAN
  // rpw (rays per wave) < 64: NARROW waves for small batches (vsa_trace_q_narrow) — lanes >= rpw sit out.  A wave walks
  // until its slowest ray is done, and a batch of a few ten thousand incoherent rays (a training batch) is a few
  // waves per SIMD each carrying the maximum of 64 unrelated walks: with 16 rays per wave the chains are shorter and
  // there are four times as many waves to hide their latency behind.  Same rays, same walks, same hits.
  if (lane >= rpw) return;
  const long long n = (long long)blockIdx.x * rpw + lane;
  const int mesh = blockIdx.y;
#else
  int item = blockIdx.y * gridDim.x + blockIdx.x;
  if (g_torder_on) item = g_torder[item];
  const int mesh = item / (int)gridDim.x;
  const long long n = (long long)(item - mesh * (int)gridDim.x) * TRACE_BLOCK + lane;
  const unsigned long long span_t0 = trace_now();
  __shared__ unsigned s_span[8];
  if (threadIdx.x < 8) s_span[threadIdx.x] = 0;
#endif
  if (n >= N) return;
  const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
  const float dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
  const float* fr = frames.f[mesh];
  QRay qr;
  {
    const float gx = (ox - fr[0]) / fr[3] + 1.0f, gy = (oy - fr[1]) / fr[4] + 1.0f,
                gz = (oz - fr[2]) / fr[5] + 1.0f;
    const float ix = 1.0f / (dx / fr[3]), iy = 1.0f / (dy / fr[4]), iz = 1.0f / (dz / fr[5]);
    qr.ix = f32x2_t{ix, ix}, qr.iy = f32x2_t{iy, iy}, qr.iz = f32x2_t{iz, iz};
    qr.cx = f32x2_t{-(gx * ix), -(gx * ix)};
    qr.cy = f32x2_t{-(gy * iy), -(gy * iy)};
    qr.cz = f32x2_t{-(gz * iz), -(gz * iz)};
  }

  Hit best;
  best.t = INFINITY;
  best.u = best.v = 0.f;
  best.slot = -1;
  best.id = 0x7fffffff;

  int cur = roots.root[mesh];
  int sp = 0;
  while (cur != TRACE_EMPTY) {
#ifdef TRACE_SPAN
    SPAN_ADD(SP_OUTER, 1);
#endif
    while ((unsigned)cur < (unsigned)TRACE_EMPTY) {
#ifdef TRACE_SPAN
      SPAN_ADD(SP_ROUNDS, 1);
      SPAN_ADD(SP_VISITS, __builtin_popcountll(__builtin_amdgcn_read_exec()));
#endif
      const uint4 a = qnodes[2 * (long long)cur], b = qnodes[2 * (long long)cur + 1];
      float tn0, tn1;
      const bool h0 = qbox_test(a.x, a.y, a.z, qr, t_min, best.t, tn0);
      const bool h1 = qbox_test(a.w, b.x, b.y, qr, t_min, best.t, tn1);
      const int c0 = (int)b.z, c1 = (int)b.w;
      if (h0 && h1) {
        const bool swap = tn1 < tn0;
        s_stack[sp++][lane] = swap ? c0 : c1;
        cur = swap ? c1 : c0;
      } else if (h0) {
        cur = c0;
      } else if (h1) {
        cur = c1;
      } else {
        cur = sp ? s_stack[--sp][lane] : TRACE_EMPTY;
      }
    }
    if (cur != TRACE_EMPTY) {
      const int code = ~cur;
      const int first = code >> 4, cnt = code & 15;
      for (int i0 = 0; i0 < cnt; i0 += 4) {
        float4 tv[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const long long s = first + min(i0 + i, cnt - 1);
          tv[i][0] = tris[3 * s];
          tv[i][1] = tris[3 * s + 1];
          tv[i][2] = tris[3 * s + 2];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i0 + i < cnt)
            tri_test(tv[i][0], tv[i][1], tv[i][2], ox, oy, oz, dx, dy, dz, t_min, first + i0 + i, best);
      }
      cur = sp ? s_stack[--sp][lane] : TRACE_EMPTY;
    }
  }
#ifdef TRACE_SPAN
  if (lane == __builtin_ctzll(__builtin_amdgcn_read_exec()) && item < TRACE_SPAN_WAVES) {
    g_tspan[item][0] = span_t0;
    g_tspan[item][1] = trace_now();
    g_tspan[item][2] = ((unsigned long long)s_span[SP_ROUNDS] << 48) | ((unsigned long long)(s_span[SP_OUTER] & 0xffff) << 32) | s_span[SP_VISITS];
  }
#endif
  const long long o = (long long)mesh * N + n;
  hit_t[o] = best.slot >= 0 ? best.t : 0.0f;
  hit_slot[o] = best.slot;
  hit_uv[2 * o] = best.u;
  hit_uv[2 * o + 1] = best.v;
}

// The walk of the quantised-node kernels.  BUDGETED: stop after `budget` wave-level trips (trace_qf_kernel
// measures its cost that way; the rejected budgeted three-pass form, the 4-wide nodes and the
// persistent-lane kernel of round 3 — all bit-exact, all slower: profiles/NOTEBOOK.md A9.4 — left the
// library in round 5 and live in the history at e64f229).
template <int STACK, bool BUDGETED, bool COUNT = false>
__device__ __forceinline__ int q_walk(const uint4* __restrict__ qnodes, const float4* __restrict__ tris,
                                       const QRay& qr, float ox, float oy, float oz, float dx, float dy,
                                       float dz, float t_min, int& cur, int& sp, Hit& best,
                                       int (*s_stack)[TRACE_BLOCK], int lane, int budget,
                                       int* lane_visits = nullptr, int* lane_tests = nullptr) {
  // The loops are written on ballots, i.e. as the wave-level loops they are, so that the trip count
  // is a scalar of the WAVE (a per-lane counter would only count that lane's own trips).
  int trips = 0;
  while (__builtin_amdgcn_ballot_w64(cur != TRACE_EMPTY) != 0) {
    if (BUDGETED && trips >= budget) break;
    while (__builtin_amdgcn_ballot_w64((unsigned)cur < (unsigned)TRACE_EMPTY) != 0) {
      ++trips;
      if (!((unsigned)cur < (unsigned)TRACE_EMPTY)) continue;
      if constexpr (COUNT) ++*lane_visits;
      const uint4 a = qnodes[2 * (long long)cur], b = qnodes[2 * (long long)cur + 1];
      float tn0, tn1;
      const bool h0 = qbox_test(a.x, a.y, a.z, qr, t_min, best.t, tn0);
      const bool h1 = qbox_test(a.w, b.x, b.y, qr, t_min, best.t, tn1);
      const int c0 = (int)b.z, c1 = (int)b.w;
      if (h0 && h1) {
        const bool swap = tn1 < tn0;
        s_stack[sp++][lane] = swap ? c0 : c1;
        cur = swap ? c1 : c0;
      } else if (h0) {
        cur = c0;
      } else if (h1) {
        cur = c1;
      } else {
        cur = sp ? s_stack[--sp][lane] : TRACE_EMPTY;
      }
    }
    if (cur != TRACE_EMPTY) {
      const int code = ~cur;
      const int first = code >> 4, cnt = code & 15;
      if constexpr (COUNT) *lane_tests += cnt;
      for (int i0 = 0; i0 < cnt; i0 += 4) {
        float4 tv[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const long long s = first + min(i0 + i, cnt - 1);
          tv[i][0] = tris[3 * s];
          tv[i][1] = tris[3 * s + 1];
          tv[i][2] = tris[3 * s + 2];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i0 + i < cnt)
            tri_test(tv[i][0], tv[i][1], tv[i][2], ox, oy, oz, dx, dy, dz, t_min, first + i0 + i, best);
      }
      cur = sp ? s_stack[--sp][lane] : TRACE_EMPTY;
    }
  }
  return trips;
}

// ---- cooperative finish of a wave's LAST rays.  A wave walks until its slowest ray is done: a grazing ray takes
// 100-360 trips of the loop above against a median of 9, and a launch is as long as its longest wave (8 000 random
// rays take 0.077 ms where 34 000 take 0.110, tools/trace_narrow_ab.py).  Once only a few lanes of a wave are still
// walking (checked every `chunk` trips), they hand their pending subtrees — `cur` and the stack — to the WHOLE wave:
// a queue of (node, ray lane) entries, of which every lane takes one per round, fetches its ray's parameters from the
// owner lane, tests the node's two boxes (or the leaf's triangles) against the ray's current closest hit and appends
// the children that survive.  One ray's remaining walk becomes ~its tree depth in rounds instead of its node count in
// trips.  The closest hit is the minimum of (t, face id) over every triangle whose leaf is reached, and a subtree is
// only dropped when its box starts beyond the closest hit known AT THAT TIME: a stale (larger) bound visits more, never
// less — the hits are the ones of the one-ray-per-lane walk bit for bit (keys: ds_min_u64 on {ordered t, id}).
// The queue lives in the stack's own LDS: the rows above the deepest straggler's stack are free, the entries are
// gathered there and moved down to row 0 (+ 1.8 KB for the entries' ray lanes and the rays' keys).
#ifndef TRACE_COOP_CHUNK
#define TRACE_COOP_CHUNK 16      /* trips between two looks at how many lanes are left */
#endif
#ifndef TRACE_COOP_LANES
#define TRACE_COOP_LANES 24      /* at most this many lanes still walking: the wave finishes them together */
#endif
#ifndef TRACE_COOP_WAVES
#define TRACE_COOP_WAVES 4096    /* launches of at most this many waves: four per SIMD on 256 CUs (the finish's kernel holds no
                                    more; 65 536 rays x 5 shells = 5 120 waves already ran 0.103 -> 0.114 ms with it) */
#endif
constexpr int TRACE_COOP_Q = 1536;      // (the 24-row stack's own 1 536 words hold the queue)
struct TraceCoop {
  unsigned long long key[TRACE_BLOCK];
  int slot[TRACE_BLOCK];
  unsigned char qr[TRACE_COOP_Q];
};
__device__ __forceinline__ unsigned long long coop_key(float t, int id) {
  const unsigned b = __float_as_uint(t);
  const unsigned o = b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);      // order-preserving for all finite t and +inf
  return ((unsigned long long)o << 32) | (unsigned)id;
}
__device__ __forceinline__ float coop_key_t(unsigned long long k) {
  const unsigned o = (unsigned)(k >> 32);
  return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xffffffffu));
}

// Returns the rounds it took, or -1 when the pending entries do not fit the queue (the caller walks on, one ray per
// lane).  Every lane of the wave must call it; `cur` of at least one lane is not TRACE_EMPTY.
template <int STACK>
__device__ __forceinline__ int q_finish_coop(const uint4* __restrict__ qnodes, const float4* __restrict__ tris,
                                          const QRay& qr, float ox, float oy, float oz, float dx, float dy,
                                          float dz, float t_min, int& cur, int& sp, Hit& best,
                                          int (*s_stack)[TRACE_BLOCK], TraceCoop& L, int lane) {
  constexpr int QCAP = STACK * TRACE_BLOCK < TRACE_COOP_Q ? STACK * TRACE_BLOCK : TRACE_COOP_Q;
  int* const qn = &s_stack[0][0];
  const bool strag = cur != TRACE_EMPTY;
  const int mine = strag ? sp + 1 : 0;
  int incl = mine, top = strag ? sp : 0;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
    top = max(top, __shfl_xor(top, off, 64));
  }
  int count = __shfl(incl, 63, 64);
  const int F = top * TRACE_BLOCK;           // rows >= top hold no straggler's entry
  if (count > QCAP / 2 || F + count > STACK * TRACE_BLOCK) return -1;
  if (strag) {
    const int base = incl - mine;
    L.key[lane] = coop_key(best.t, best.id);
    L.slot[lane] = best.slot;
    qn[F + base] = cur;
    L.qr[base] = (unsigned char)lane;
    for (int k = 0; k < sp; ++k) {
      qn[F + base + 1 + k] = s_stack[k][lane];
      L.qr[base + 1 + k] = (unsigned char)lane;
    }
  }
  if (F)
    for (int i = lane; i < count; i += TRACE_BLOCK) {      // down to row 0: a chunk is read whole before it is written
      const int e = qn[F + i];
      qn[i] = e;
    }
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  int rounds = 0;
  while (count > 0) {
    ++rounds;
    // near the cap one entry per round: a depth-first walk of the top entry adds at most the tree depth
    const int P = count > QCAP - 128 - STACK ? 1 : min(64, count);
    const int start = count - P;
    const bool have = lane < P;
    const int e = have ? qn[start + lane] : TRACE_EMPTY;
    const int r = have ? (int)L.qr[start + lane] : lane;
    count = start;
    const unsigned long long k0 = L.key[r];
    bool h0 = false, h1 = false;
    int c0 = 0, c1 = 0;
    {
      QRay q;
      const float ix = __shfl(qr.ix.x, r, 64), iy = __shfl(qr.iy.x, r, 64), iz = __shfl(qr.iz.x, r, 64);
      const float cx = __shfl(qr.cx.x, r, 64), cy = __shfl(qr.cy.x, r, 64), cz = __shfl(qr.cz.x, r, 64);
      q.ix = f32x2_t{ix, ix}, q.iy = f32x2_t{iy, iy}, q.iz = f32x2_t{iz, iz};
      q.cx = f32x2_t{cx, cx}, q.cy = f32x2_t{cy, cy}, q.cz = f32x2_t{cz, cz};
      if (have && e >= 0) {
        const uint4 a = qnodes[2 * (long long)e], b = qnodes[2 * (long long)e + 1];
        float tn0, tn1;
        const float tb = coop_key_t(k0);
        h0 = qbox_test(a.x, a.y, a.z, q, t_min, tb, tn0);
        h1 = qbox_test(a.w, b.x, b.y, q, t_min, tb, tn1);
        c0 = (int)b.z, c1 = (int)b.w;
      }
    }
    // (the two halves of a round keep to themselves: with the ray fetches of both hoisted to the top the finish
    //  took 115 registers where the walk takes 95 — four waves per SIMD instead of five for every launch)
    __builtin_amdgcn_sched_barrier(0);
    Hit hb;
    hb.t = coop_key_t(k0), hb.id = (int)(unsigned)k0, hb.u = hb.v = 0.f, hb.slot = -1;
    {
      const float rox = __shfl(ox, r, 64), roy = __shfl(oy, r, 64), roz = __shfl(oz, r, 64);
      const float rdx = __shfl(dx, r, 64), rdy = __shfl(dy, r, 64), rdz = __shfl(dz, r, 64);
      if (have && e < 0) {
        const int code = ~e;
        const int first = code >> 4, cnt = code & 15;
#pragma nounroll
        for (int i = 0; i < cnt; ++i) {
          const long long s = first + i;
          tri_test(tris[3 * s], tris[3 * s + 1], tris[3 * s + 2], rox, roy, roz, rdx, rdy, rdz, t_min, first + i, hb);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool better = hb.slot >= 0;       // a triangle of this leaf beats the bound the lane started from
    const unsigned long long mykey = coop_key(hb.t, hb.id);
    if (better) atomicMin(&L.key[r], mykey);
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(h0), m1 = __builtin_amdgcn_ballot_w64(h1);
    const int n0 = __builtin_popcountll(m0);
    if (h0) {
      const int p = count + __builtin_popcountll(m0 & lt);
      qn[p] = c0;
      L.qr[p] = (unsigned char)r;
    }
    if (h1) {
      const int p = count + n0 + __builtin_popcountll(m1 & lt);
      qn[p] = c1;
      L.qr[p] = (unsigned char)r;
    }
    count += n0 + __builtin_popcountll(m1);
    // of the lanes that improved ray r this round, the one whose key stands names the triangle
    if (better && L.key[r] == mykey) L.slot[r] = hb.slot;
  }
  if (strag) {
    const unsigned long long k = L.key[lane];
    const int slot = L.slot[lane];
    if (slot != best.slot) {                 // the owner forms u, v (and t again) of the winning triangle itself
      Hit w;
      w.t = INFINITY, w.u = w.v = 0.f, w.slot = -1, w.id = 0x7fffffff;
      tri_test(tris[3 * (long long)slot], tris[3 * (long long)slot + 1], tris[3 * (long long)slot + 2], ox, oy, oz, dx, dy,
               dz, t_min, slot, w);
      best = w;
    }
    (void)k;
  }
  cur = TRACE_EMPTY;
  sp = 0;
  return rounds;
}

__device__ __forceinline__ void write_hit(const Hit& best, long long o, float* __restrict__ hit_t,
                                          int* __restrict__ hit_slot, float* __restrict__ hit_uv) {
  hit_t[o] = best.slot >= 0 ? best.t : 0.0f;
  hit_slot[o] = best.slot;
  hit_uv[2 * o] = best.u;
  hit_uv[2 * o + 1] = best.v;
}

// ---- cost-feedback launch order (vsa_trace_q_fb).  The one-pass kernel's launch is full for its first
// 155 us and then drains for 100 us (tools/trace_span.py, profiles/NOTEBOOK.md A9.4): the waves that hold grazing rays
// walk 100-200 trips against a median of 9, and those that happen to be dispatched late are still walking
// when everything else has finished.  With the heaviest waves dispatched FIRST (sorted by their measured
// trips: the upper bound of any predictor) the same kernel takes 203 us instead of 300 with stamps.  A
// wave's trip count depends on its rays only, not on the order, so the previous launch over the same
// rays is an exact predictor, and a good one for a camera that moves a little: every wave files itself
// into one of three "heavy" lists (by trips) of the NEXT launch's order and sets its flag; the next
// launch runs the lists first (heaviest list first) and then the remaining items in their natural
// order, skipping the flagged ones.  Lists + flags always form a partition of the items, whatever rays
// they were measured on, so the hits never depend on the feedback - only the order of issue does.
struct TraceFeedback {
  const int* prev;                 // header {tag = items of the launch that wrote it, n0, n1, n2}
  const unsigned char* prev_flag;  // [items] 1 = in a list
  const int* prev_lists;           // 3 x cap
  int* next;
  unsigned char* next_flag;
  int* next_lists;
  int cap;
};
// The two halves of a feedback buffer and which one a launch reads: `phase` 0 / 1 from the host, or
// (phase_word != nullptr) a word in the buffer itself that trace_fb_flip_kernel toggles in front of
// every launch — a captured graph then alternates the halves on every REPLAY (a host-side toggle is
// frozen at capture: every replay would read the half the last eager call wrote).
struct TraceFeedbackBuf {
  char* half[2];
  long long flags_bytes;
  const int* phase_word;
  int phase, cap;
};
__device__ __forceinline__ TraceFeedback trace_fb_select(const TraceFeedbackBuf& b) {
  const int ph = (b.phase_word ? __builtin_amdgcn_readfirstlane(*b.phase_word) : b.phase) & 1;
  char* prev = b.half[ph];
  char* next = b.half[ph ^ 1];
  TraceFeedback fb;
  fb.prev = reinterpret_cast<const int*>(prev);
  fb.prev_flag = reinterpret_cast<const unsigned char*>(prev + 16);
  fb.prev_lists = reinterpret_cast<const int*>(prev + 16 + b.flags_bytes);
  fb.next = reinterpret_cast<int*>(next);
  fb.next_flag = reinterpret_cast<unsigned char*>(next + 16);
  fb.next_lists = reinterpret_cast<int*>(next + 16 + b.flags_bytes);
  fb.cap = b.cap;
  return fb;
}
// device-resident phase: flip it and clear the header of the half the coming launch writes
__global__ void trace_fb_flip_kernel(int* phase_word, char* half0, char* half1) {
  const int ph = (*phase_word ^ 1) & 1;
  if (threadIdx.x == 0) *phase_word = ph;
  if (threadIdx.x < 4) reinterpret_cast<int*>(ph ? half0 : half1)[threadIdx.x] = 0;
}
#ifndef TRACE_FB_T
#define TRACE_FB_T 160, 128, 64     /* same-box sweep in profiles/r03/trace_feedback.txt */
#endif
constexpr int TRACE_FB_TS[3] = {TRACE_FB_T};
constexpr int TRACE_FB_T0 = TRACE_FB_TS[0], TRACE_FB_T1 = TRACE_FB_TS[1], TRACE_FB_T2 = TRACE_FB_TS[2];   // trips: list 0 / 1 / 2

// COOP: with the cooperative finish (q_finish_coop).  Its registers come on top of the walk's (95 -> 115 VGPRs: four
// waves per SIMD instead of five, which costs a FRAME's launch of 50 000 waves 0.185 -> 0.209 ms whether the finish is
// ever entered or not), so it is the kernel of launches that cannot fill four waves per SIMD anyway — training
// batches — and the plain walk stays the kernel of the large ones (vsa_trace_q_fb picks by the number of waves).
template <int STACK, bool COOP>
__global__ __launch_bounds__(TRACE_BLOCK) void trace_qf_kernel(
    const uint4* __restrict__ qnodes, const float4* __restrict__ tris, Roots roots, Frames frames,
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, int N, int G, int nr_items, float t_min,
    TraceFeedbackBuf fbb, float* __restrict__ hit_t, int* __restrict__ hit_slot, float* __restrict__ hit_uv,
    int coop_chunk, int coop_lanes) {
  __shared__ int s_stack[STACK][TRACE_BLOCK];
  const int lane = threadIdx.x;
  const TraceFeedback fb = trace_fb_select(fbb);
  // dispatch slot -> item (all wave-uniform)
  int item = blockIdx.x;
  {
    int n0 = 0, n1 = 0, n2 = 0;
    if (fb.prev[0] == nr_items) n0 = min(fb.prev[1], fb.cap), n1 = min(fb.prev[2], fb.cap), n2 = min(fb.prev[3], fb.cap);
    const int heavy = n0 + n1 + n2;
    if (item < heavy) {
      const int l = item < n0 ? 0 : item < n0 + n1 ? 1 : 2;
      item = fb.prev_lists[l * fb.cap + (item - (l == 0 ? 0 : l == 1 ? n0 : n0 + n1))];
    } else {
      item -= heavy;
      if (item >= nr_items) return;
      if (heavy && fb.prev_flag[item]) return;   // ran from a list
    }
  }
  const int mesh = item / G;
  const long long n_raw = (long long)(item - mesh * G) * TRACE_BLOCK + lane;
  const bool alive = n_raw < N;
  const long long n = alive ? n_raw : N - 1;
  const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
  const float dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
  const float* fr = frames.f[mesh];
  QRay qr;
  {
    const float gx = (ox - fr[0]) / fr[3] + 1.0f, gy = (oy - fr[1]) / fr[4] + 1.0f,
                gz = (oz - fr[2]) / fr[5] + 1.0f;
    const float ix = 1.0f / (dx / fr[3]), iy = 1.0f / (dy / fr[4]), iz = 1.0f / (dz / fr[5]);
    qr.ix = f32x2_t{ix, ix}, qr.iy = f32x2_t{iy, iy}, qr.iz = f32x2_t{iz, iz};
    qr.cx = f32x2_t{-(gx * ix), -(gx * ix)};
    qr.cy = f32x2_t{-(gy * iy), -(gy * iy)};
    qr.cz = f32x2_t{-(gz * iz), -(gz * iz)};
  }
  Hit best;
  best.t = INFINITY;
  best.u = best.v = 0.f;
  best.slot = -1;
  best.id = 0x7fffffff;
  int cur = alive ? roots.root[mesh] : TRACE_EMPTY;
  int sp = 0;
  // one ray per lane, `coop_chunk` trips at a time, until at most `coop_lanes` lanes are still walking: the whole wave
  // finishes those together (q_finish_coop)
  int trips = 0;
  if constexpr (COOP) {
    __shared__ TraceCoop s_coop;
    for (int chunk = coop_chunk;;) {
      trips += q_walk<STACK, true>(qnodes, tris, qr, ox, oy, oz, dx, dy, dz, t_min, cur, sp, best, s_stack, lane, chunk);
      const unsigned long long left = __builtin_amdgcn_ballot_w64(cur != TRACE_EMPTY);
      if (left == 0) break;
      if (__builtin_popcountll(left) > coop_lanes) continue;
      const int rounds = q_finish_coop<STACK>(qnodes, tris, qr, ox, oy, oz, dx, dy, dz, t_min, cur, sp, best, s_stack,
                                              s_coop, lane);
      if (rounds >= 0) {
        trips += rounds;
        break;
      }
      // (the pending entries did not fit the queue: another chunk one ray per lane, then another look)
    }
  } else {
    trips = q_walk<STACK, true>(qnodes, tris, qr, ox, oy, oz, dx, dy, dz, t_min, cur, sp, best, s_stack, lane, 0x7fffffff);
  }
  if (alive) write_hit(best, (long long)mesh * N + n, hit_t, hit_slot, hit_uv);
  // file this wave for the next launch
  if (lane == 0) {
    const int l = trips >= TRACE_FB_T0 ? 0 : trips >= TRACE_FB_T1 ? 1 : trips >= TRACE_FB_T2 ? 2 : 3;
    bool listed = false;
    if (l < 3) {
      const int idx = atomicAdd(&fb.next[1 + l], 1);
      listed = idx < fb.cap;                     // a full list: the item stays in the natural order
      if (listed) fb.next_lists[l * fb.cap + idx] = item;
    }
    fb.next_flag[item] = listed ? 1 : 0;
    if (item == 0) fb.next[0] = nr_items;
  }
}

// What a traversal launch DID, for its place against a ceiling (bench.py stage_roofline.trace): the walk of
// trace_q_kernel / trace_qf_kernel (same boxes, same order) with counters — lane-level node visits (one 32-byte
// node fetch each), lane-level triangle tests (48 bytes each), wave-level trips of the walk loop (a trip = one
// dependent node fetch of the whole wave: the unit of the latency chain) and waves.  Not on the product path.
template <int STACK>
__global__ __launch_bounds__(TRACE_BLOCK) void trace_q_count_kernel(
    const uint4* __restrict__ qnodes, const float4* __restrict__ tris, Roots roots, Frames frames,
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, int N, float t_min,
    unsigned long long* __restrict__ stats) {
  __shared__ int s_stack[STACK][TRACE_BLOCK];
  const int lane = threadIdx.x;
  const long long n = (long long)blockIdx.x * TRACE_BLOCK + lane;
  const int mesh = blockIdx.y;
  int visits = 0, tests = 0, trips = 0, max_trips = 0;
  if (n < N) {
    const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
    const float dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
    const float* fr = frames.f[mesh];
    QRay qr;
    const float gx = (ox - fr[0]) / fr[3] + 1.0f, gy = (oy - fr[1]) / fr[4] + 1.0f, gz = (oz - fr[2]) / fr[5] + 1.0f;
    const float ix = 1.0f / (dx / fr[3]), iy = 1.0f / (dy / fr[4]), iz = 1.0f / (dz / fr[5]);
    qr.ix = f32x2_t{ix, ix}, qr.iy = f32x2_t{iy, iy}, qr.iz = f32x2_t{iz, iz};
    qr.cx = f32x2_t{-(gx * ix), -(gx * ix)};
    qr.cy = f32x2_t{-(gy * iy), -(gy * iy)};
    qr.cz = f32x2_t{-(gz * iz), -(gz * iz)};
    Hit best;
    best.t = INFINITY, best.u = best.v = 0.f, best.slot = -1, best.id = 0x7fffffff;
    int cur = roots.root[mesh], sp = 0;
    trips = q_walk<STACK, false, true>(qnodes, tris, qr, ox, oy, oz, dx, dy, dz, t_min, cur, sp, best, s_stack, lane, 0,
                                       &visits, &tests);
  }
  // wave sums -> one atomic per counter and wave
  for (int o = 32; o > 0; o >>= 1) {
    visits += __shfl_down(visits, o);
    tests += __shfl_down(tests, o);
  }
  max_trips = trips;      // (wave-uniform)
  if (lane == 0) {
    atomicAdd(&stats[0], (unsigned long long)visits);
    atomicAdd(&stats[1], (unsigned long long)tests);
    atomicAdd(&stats[2], (unsigned long long)trips);
    atomicAdd(&stats[3], 1ull);
    atomicMax(&stats[4], (unsigned long long)max_trips);
  }
}

extern "C" int vsa_trace_q_stats(const uint32_t* qnodes, const float* tris, const int32_t* mesh_roots,
                                 const float* mesh_frames, int nr_meshes, int max_depth,
                                 const float* rays_o, const float* rays_d, int nr_rays, float t_min,
                                 uint64_t* stats, void* stream) {
  if (nr_meshes < 1 || nr_meshes > VSA_MAX_SHELLS || nr_rays < 1 || !mesh_roots || !mesh_frames) return VSA_ERR_ARG;
  if (max_depth >= TRACE_STACK) return VSA_ERR_UNSUPPORTED;
  if (!qnodes || !tris || !rays_o || !rays_d || !stats) return VSA_ERR_ARG;
  Roots r;
  Frames fr;
  for (int i = 0; i < VSA_MAX_SHELLS; ++i) {
    r.root[i] = i < nr_meshes ? mesh_roots[i] : 0;
    for (int j = 0; j < 6; ++j) fr.f[i][j] = i < nr_meshes ? mesh_frames[6 * i + j] : 1.0f;
  }
  VSA_HIP_TRY(hipMemsetAsync(stats, 0, 5 * sizeof(uint64_t), (hipStream_t)stream));
  dim3 grid(vsa_div_up(nr_rays, TRACE_BLOCK), nr_meshes), block(TRACE_BLOCK);
  if (max_depth < 24)
    hipLaunchKernelGGL(trace_q_count_kernel<24>, grid, block, 0, (hipStream_t)stream,
                       reinterpret_cast<const uint4*>(qnodes), reinterpret_cast<const float4*>(tris), r, fr, rays_o,
                       rays_d, nr_rays, t_min, reinterpret_cast<unsigned long long*>(stats));
  else
    hipLaunchKernelGGL(trace_q_count_kernel<TRACE_STACK>, grid, block, 0, (hipStream_t)stream,
                       reinterpret_cast<const uint4*>(qnodes), reinterpret_cast<const float4*>(tris), r, fr, rays_o,
                       rays_d, nr_rays, t_min, reinterpret_cast<unsigned long long*>(stats));
  VSA_RETURN_LAUNCH_STATUS();
}

// Per-hit attributes in the shape raytracelib returns them
// (volsurfs.py:496-501): positions, face normals, barycentrics, original ids.
__global__ void hit_attributes_kernel(const float4* __restrict__ tris,
                                      const float* __restrict__ rays_o,
                                      const float* __restrict__ rays_d,
                                      const float* __restrict__ hit_t,
                                      const int* __restrict__ hit_slot,
                                      const float* __restrict__ hit_uv, int N,
                                      unsigned char* __restrict__ is_hit,
                                      int* __restrict__ tri_id, float* __restrict__ positions,
                                      float* __restrict__ normals, float* __restrict__ bary) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int slot = hit_slot[n];
  float p[3] = {0, 0, 0}, nn[3] = {0, 0, 0}, b[3] = {0, 0, 0};
  int id = -1;
  if (slot >= 0) {
    const float t = hit_t[n];
    for (int c = 0; c < 3; ++c) p[c] = rays_o[3 * n + c] + t * rays_d[3 * n + c];
    const float4 v0 = tris[3 * (long long)slot], e1 = tris[3 * (long long)slot + 1],
                 e2 = tris[3 * (long long)slot + 2];
    id = __float_as_int(v0.w);
    float cx = e1.y * e2.z - e1.z * e2.y, cy = e1.z * e2.x - e1.x * e2.z,
          cz = e1.x * e2.y - e1.y * e2.x;
    float len = sqrtf(dot3(cx, cy, cz, cx, cy, cz));
    float inv = len > 0.f ? 1.0f / len : 0.f;
    nn[0] = cx * inv; nn[1] = cy * inv; nn[2] = cz * inv;
    const float u = hit_uv[2 * n], v = hit_uv[2 * n + 1];
    b[0] = (1.0f - u) - v; b[1] = u; b[2] = v;
  }
  if (is_hit) is_hit[n] = slot >= 0;
  if (tri_id) tri_id[n] = id;
  for (int c = 0; c < 3; ++c) {
    if (positions) positions[3 * n + c] = p[c];
    if (normals) normals[3 * n + c] = nn[c];
    if (bary) bary[3 * n + c] = b[c];
  }
}

}  // namespace

extern "C" int vsa_trace(const float* nodes, const float* tris, const int32_t* mesh_roots,
                         int nr_meshes, int max_depth, const float* rays_o, const float* rays_d,
                         int nr_rays, float t_min, float* hit_t, int32_t* hit_slot, float* hit_uv,
                         void* stream) {
  if (nr_meshes < 1 || nr_meshes > VSA_MAX_SHELLS || nr_rays < 0 || !mesh_roots) return VSA_ERR_ARG;
  if (max_depth >= TRACE_STACK) return VSA_ERR_UNSUPPORTED;
  if (nr_rays == 0) return VSA_OK;
  if (!nodes || !tris || !rays_o || !rays_d || !hit_t || !hit_slot || !hit_uv) return VSA_ERR_ARG;
  Roots r;
  for (int i = 0; i < VSA_MAX_SHELLS; ++i) r.root[i] = i < nr_meshes ? mesh_roots[i] : 0;
  // the traversal stack never exceeds the tree depth; shallow trees (the usual
  // case: depth 16 for 82k-triangle shells) take a 24-entry stack
  dim3 grid(vsa_div_up(nr_rays, TRACE_BLOCK), nr_meshes), block(TRACE_BLOCK);
  if (max_depth < 24)
    hipLaunchKernelGGL(trace_ww_kernel<24>, grid, block, 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(nodes), reinterpret_cast<const float4*>(tris),
                       r, rays_o, rays_d, nr_rays, t_min, hit_t, hit_slot, hit_uv);
  else
    hipLaunchKernelGGL(trace_ww_kernel<TRACE_STACK>, grid, block, 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(nodes), reinterpret_cast<const float4*>(tris),
                       r, rays_o, rays_d, nr_rays, t_min, hit_t, hit_slot, hit_uv);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_trace_q_narrow(const uint32_t* qnodes, const float* tris, const int32_t* mesh_roots,
                                  const float* mesh_frames, int nr_meshes, int max_depth,
                                  const float* rays_o, const float* rays_d, int nr_rays, float t_min,
                                  float* hit_t, int32_t* hit_slot, float* hit_uv, int rays_per_wave,
                                  void* stream) {
  if (nr_meshes < 1 || nr_meshes > VSA_MAX_SHELLS || nr_rays < 0 || !mesh_roots || !mesh_frames)
    return VSA_ERR_ARG;
  if (rays_per_wave < 1 || rays_per_wave > TRACE_BLOCK) return VSA_ERR_ARG;
  if (max_depth >= TRACE_STACK) return VSA_ERR_UNSUPPORTED;
  if (nr_rays == 0) return VSA_OK;
  if (!qnodes || !tris || !rays_o || !rays_d || !hit_t || !hit_slot || !hit_uv) return VSA_ERR_ARG;
  Roots r;
  Frames fr;
  for (int i = 0; i < VSA_MAX_SHELLS; ++i) {
    r.root[i] = i < nr_meshes ? mesh_roots[i] : 0;
    for (int j = 0; j < 6; ++j) fr.f[i][j] = i < nr_meshes ? mesh_frames[6 * i + j] : 1.0f;
  }
  dim3 grid(vsa_div_up(nr_rays, rays_per_wave), nr_meshes), block(TRACE_BLOCK);
  if (max_depth < 24)
    hipLaunchKernelGGL(trace_q_kernel<24>, grid, block, 0, (hipStream_t)stream,
                       reinterpret_cast<const uint4*>(qnodes), reinterpret_cast<const float4*>(tris),
                       r, fr, rays_o, rays_d, nr_rays, t_min, hit_t, hit_slot, hit_uv, rays_per_wave);
  else
    hipLaunchKernelGGL(trace_q_kernel<TRACE_STACK>, grid, block, 0, (hipStream_t)stream,
                       reinterpret_cast<const uint4*>(qnodes), reinterpret_cast<const float4*>(tris),
                       r, fr, rays_o, rays_d, nr_rays, t_min, hit_t, hit_slot, hit_uv, rays_per_wave);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_trace_q(const uint32_t* qnodes, const float* tris, const int32_t* mesh_roots,
                           const float* mesh_frames, int nr_meshes, int max_depth,
                           const float* rays_o, const float* rays_d, int nr_rays, float t_min,
                           float* hit_t, int32_t* hit_slot, float* hit_uv, void* stream) {
  return vsa_trace_q_narrow(qnodes, tris, mesh_roots, mesh_frames, nr_meshes, max_depth, rays_o, rays_d, nr_rays, t_min,
                            hit_t, hit_slot, hit_uv, TRACE_BLOCK, stream);
}

// Feedback buffer of vsa_trace_q_fb: two halves (written / read alternately) at offsets 0 and H, then
// 256 bytes that hold the device-resident phase word (phase = 2); H = ((bytes - 256) / 2) & ~255.  A half:
// 16 B of header, one flag byte per item (rounded up to 16), three lists of cap ints.  The header
// offsets depend on the buffer only, so a buffer sized for more rays serves fewer (the tag in the header
// tells a half that was written for another item count).
static long long trace_fb_half_bytes(long long items, int* cap_out) {
  const int cap = (int)std::min<long long>(items / 8 + 64, 1 << 27);
  if (cap_out) *cap_out = cap;
  return (16 + (items + 15) / 16 * 16 + 12ll * cap + 255) / 256 * 256;
}

extern "C" long long vsa_trace_feedback_bytes(int nr_rays, int nr_meshes) {
  if (nr_rays < 0 || nr_meshes < 1) return -1;
  return 2 * trace_fb_half_bytes((long long)vsa_div_up(nr_rays, TRACE_BLOCK) * nr_meshes, nullptr) + 256;
}

// process-wide setting of the cooperative finish (vsa_trace_coop_config)
struct TraceCoopConfig {
  int chunk, lanes;
  long long max_waves;
};
static TraceCoopConfig& trace_coop_config() {
  static TraceCoopConfig cfg = [] {
    TraceCoopConfig c{TRACE_COOP_CHUNK, TRACE_COOP_LANES, TRACE_COOP_WAVES};
    if (const char* e = getenv("VSA_TRACE_COOP")) sscanf(e, "%d,%d", &c.chunk, &c.lanes);
    if (const char* e = getenv("VSA_TRACE_COOP_WAVES")) c.max_waves = atoll(e);
    if (c.chunk < 1 || c.lanes < 0 || c.lanes > TRACE_BLOCK || c.max_waves < 0) c = TraceCoopConfig{TRACE_COOP_CHUNK, TRACE_COOP_LANES, TRACE_COOP_WAVES};
    return c;
  }();
  return cfg;
}
extern "C" int vsa_trace_coop_config(int chunk, int lanes, long long max_waves) {
  if (chunk < 1 || lanes < 0 || lanes > TRACE_BLOCK || max_waves < 0) return VSA_ERR_ARG;
  trace_coop_config() = TraceCoopConfig{chunk, lanes, max_waves};
  return VSA_OK;
}

extern "C" int vsa_trace_q_fb(const uint32_t* qnodes, const float* tris, const int32_t* mesh_roots,
                              const float* mesh_frames, int nr_meshes, int max_depth, const float* rays_o,
                              const float* rays_d, int nr_rays, float t_min, float* hit_t, int32_t* hit_slot,
                              float* hit_uv, void* feedback, long long feedback_bytes, int phase, void* stream) {
  if (nr_meshes < 1 || nr_meshes > VSA_MAX_SHELLS || nr_rays < 0 || !mesh_roots || !mesh_frames || phase < 0 || phase > 2)
    return VSA_ERR_ARG;
  if (max_depth >= TRACE_STACK) return VSA_ERR_UNSUPPORTED;
  if (nr_rays == 0) return VSA_OK;
  if (!qnodes || !tris || !rays_o || !rays_d || !hit_t || !hit_slot || !hit_uv || !feedback) return VSA_ERR_ARG;
  const int G = vsa_div_up(nr_rays, TRACE_BLOCK);
  const long long items = (long long)G * nr_meshes;
  int cap = 0;
  if (items > 0x7fffffff / 2 || feedback_bytes < 2 * trace_fb_half_bytes(items, &cap) + 256 || ((uintptr_t)feedback & 15))
    return VSA_ERR_ARG;
  const long long half = ((feedback_bytes - 256) / 2) & ~255ll;
  Roots r;
  Frames fr;
  for (int i = 0; i < VSA_MAX_SHELLS; ++i) {
    r.root[i] = i < nr_meshes ? mesh_roots[i] : 0;
    for (int j = 0; j < 6; ++j) fr.f[i][j] = i < nr_meshes ? mesh_frames[6 * i + j] : 1.0f;
  }
  char* b = static_cast<char*>(feedback);
  TraceFeedbackBuf fb;
  fb.half[0] = b, fb.half[1] = b + half;
  fb.flags_bytes = (items + 15) / 16 * 16;
  fb.phase_word = phase == 2 ? reinterpret_cast<const int*>(b + 2 * half) : nullptr;
  fb.phase = phase & 1;
  fb.cap = cap;
  hipStream_t s = (hipStream_t)stream;
  if (phase == 2)
    hipLaunchKernelGGL(trace_fb_flip_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<int*>(b + 2 * half),
                       fb.half[0], fb.half[1]);
  else
    VSA_HIP_TRY(hipMemsetAsync(fb.half[phase ^ 1], 0, 16, s));
  const uint4* qn = reinterpret_cast<const uint4*>(qnodes);
  const float4* tr = reinterpret_cast<const float4*>(tris);
  // every item once, plus room for the listed ones' second (skipped) appearance
  dim3 grid((unsigned)(items + 3ll * cap)), block(TRACE_BLOCK);
  const TraceCoopConfig& cfg = trace_coop_config();
  const int coop_chunk = cfg.chunk, coop_lanes = cfg.lanes;
  const bool coop = coop_lanes > 0 && items <= cfg.max_waves;
#define TRACE_QF_LAUNCH(ST, CO)                                                                                        \
  hipLaunchKernelGGL((trace_qf_kernel<ST, CO>), grid, block, 0, s, qn, tr, r, fr, rays_o, rays_d, nr_rays, G, (int)items, \
                     t_min, fb, hit_t, hit_slot, hit_uv, coop_chunk, coop_lanes)
  if (max_depth < 24) {
    if (coop) TRACE_QF_LAUNCH(24, true);
    else TRACE_QF_LAUNCH(24, false);
  } else {
    if (coop) TRACE_QF_LAUNCH(TRACE_STACK, true);
    else TRACE_QF_LAUNCH(TRACE_STACK, false);
  }
#undef TRACE_QF_LAUNCH
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_hit_attributes(const float* tris, const float* rays_o, const float* rays_d,
                                  const float* hit_t, const int32_t* hit_slot,
                                  const float* hit_uv, int nr_rays, uint8_t* is_hit,
                                  int32_t* tri_id, float* positions, float* normals,
                                  float* barycentric, void* stream) {
  if (nr_rays < 0) return VSA_ERR_ARG;
  if (nr_rays == 0) return VSA_OK;
  if (!tris || !rays_o || !rays_d || !hit_t || !hit_slot || !hit_uv) return VSA_ERR_ARG;
  hipLaunchKernelGGL(hit_attributes_kernel, dim3(vsa_div_up(nr_rays, 256)), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const float4*>(tris), rays_o, rays_d,
                     hit_t, hit_slot, hit_uv, nr_rays, is_hit, tri_id, positions, normals,
                     barycentric);
  VSA_RETURN_LAUNCH_STATUS();
}

#ifdef TRACE_SPAN
extern "C" int vsa_span_set_order(const void* order, int n) {
  VSA_HIP_TRY(hipDeviceSynchronize());
  const int on = order != nullptr;
  if (on) VSA_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_torder), order, sizeof(int) * n));
  VSA_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_torder_on), &on, sizeof(int)));
  return 0;
}
extern "C" int vsa_span_read_trace(void* dst) {
  VSA_HIP_TRY(hipDeviceSynchronize());
  VSA_HIP_TRY(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tspan), sizeof(g_tspan)));
  return 0;
}
#endif
