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
    float* __restrict__ hit_t, int* __restrict__ hit_slot, float* __restrict__ hit_uv) {
  __shared__ int s_stack[STACK][TRACE_BLOCK];
  const int lane = threadIdx.x;
#ifndef TRACE_SPAN
  const long long n = (long long)blockIdx.x * TRACE_BLOCK + lane;
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

// ---- budgeted walk + continuation (vsa_trace_q_budgeted).  Wave-level stamps of trace_q_kernel at
// 800x800, K = 5 (tools/trace_span.py, profiles/r03/trace_span.txt): the chip is full for the first
// 155 us of the launch and then drains for another 145 us -- 4 % of the waves walk for more than 32
// trips (p99 105, max 210 against a median of 9) because a few of their rays graze a shell (a near
// miss prunes nothing and visits every box along its tangent), at 35 % lane utilisation over the whole
// launch; a trip is ~0.5 us of dependent latency (node fetch from L2 ~470 cycles + tests + stack), so
// no order of issue inside a wave shortens that chain.  Here a wave stops after `round_budget` trips
// of the walk loop: the lanes still holding work hand every pending subtree (the node in hand + every
// stack entry) to a second pass as ITS OWN work item, so a long ray's chain is cut into independent
// pieces that run side by side in dense waves, and a third pass merges the pieces of a ray.  The
// closest hit is order independent (smallest t, ties -> smallest face id) and every item carries the
// ray's best t so far as its bound, so the result is bit-identical to the one-pass walk.
struct TraceWs {
  unsigned* counters;   // [0] items, [1] ray records (zeroed on the stream before pass A)
  int4* ray_rec;        // 4 x int4 per handed-over ray: {n, mesh, first item, nr items}, {t, u, v, slot},
                        // {id, ix, iy, iz}, {cx, cy, cz, -} (the ray on the mesh's 16-bit grid)
  int2* items;          // {ray record, node reference}
  int4* results;        // 2 x int4 per item: {t, u, v, slot}, {id, -, -, -}
  unsigned cap_items, cap_rays;
};

template <int STACK, bool BUDGETED>
__device__ __forceinline__ int q_walk(const uint4* __restrict__ qnodes, const float4* __restrict__ tris,
                                       const QRay& qr, float ox, float oy, float oz, float dx, float dy,
                                       float dz, float t_min, int& cur, int& sp, Hit& best,
                                       int (*s_stack)[TRACE_BLOCK], int lane, int budget) {
  // The loops are written on ballots, i.e. as the wave-level loops they are, so that the trip count
  // is a scalar of the WAVE (a per-lane counter would only count that lane's own trips).
  int trips = 0;
  while (__builtin_amdgcn_ballot_w64(cur != TRACE_EMPTY) != 0) {
    if (BUDGETED && trips >= budget) break;
    while (__builtin_amdgcn_ballot_w64((unsigned)cur < (unsigned)TRACE_EMPTY) != 0) {
      ++trips;
      if (!((unsigned)cur < (unsigned)TRACE_EMPTY)) continue;
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
  return trips;
}

__device__ __forceinline__ void write_hit(const Hit& best, long long o, float* __restrict__ hit_t,
                                          int* __restrict__ hit_slot, float* __restrict__ hit_uv) {
  hit_t[o] = best.slot >= 0 ? best.t : 0.0f;
  hit_slot[o] = best.slot;
  hit_uv[2 * o] = best.u;
  hit_uv[2 * o + 1] = best.v;
}

// Pass A: trace_q_kernel with a trip budget.
template <int STACK>
__global__ __launch_bounds__(TRACE_BLOCK) void trace_qa_kernel(
    const uint4* __restrict__ qnodes, const float4* __restrict__ tris, Roots roots, Frames frames,
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, int N, float t_min, int budget,
    TraceWs ws, float* __restrict__ hit_t, int* __restrict__ hit_slot, float* __restrict__ hit_uv) {
  __shared__ int s_stack[STACK][TRACE_BLOCK];
  const int lane = threadIdx.x;
  const long long n_raw = (long long)blockIdx.x * TRACE_BLOCK + lane;
  const bool alive = n_raw < N;                 // no early return: the hand-over below is a wave operation
  const long long n = alive ? n_raw : N - 1;
  const int mesh = blockIdx.y;
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
  q_walk<STACK, true>(qnodes, tris, qr, ox, oy, oz, dx, dy, dz, t_min, cur, sp, best, s_stack, lane, budget);

  const bool pending = cur != TRACE_EMPTY;
  const unsigned long long pm = __builtin_amdgcn_ballot_w64(pending);
  if (pm != 0) {   // wave-uniform
    // one reservation per wave: items = node in hand + stack entries of every pending lane
    const int mine = pending ? sp + 1 : 0;
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off);
      if (lane >= off) incl += v;
    }
    const int total = __shfl(incl, 63);
    const int nrays = __builtin_popcountll(pm);
    unsigned base = 0, rbase = 0;
    if (lane == 0) {
      base = atomicAdd(&ws.counters[0], (unsigned)total);
      rbase = atomicAdd(&ws.counters[1], (unsigned)nrays);
    }
    base = __builtin_amdgcn_readfirstlane(base);
    rbase = __builtin_amdgcn_readfirstlane(rbase);
    const bool fits = (unsigned long long)base + total <= ws.cap_items && (unsigned long long)rbase + nrays <= ws.cap_rays;
    const unsigned my_first = base + (unsigned)(incl - mine);
    const unsigned my_rec = rbase + (unsigned)__builtin_popcountll(pm & ((1ull << lane) - 1ull));
    if (pending) {
      // (a reservation that does not fit is filled with empty items / records and the wave walks on below)
      if (my_rec < ws.cap_rays) {
        int4* r = ws.ray_rec + 4ll * my_rec;
        r[0] = int4{fits ? (int)n : -1, mesh, (int)my_first, fits ? mine : 0};
        r[1] = int4{__float_as_int(best.t), __float_as_int(best.u), __float_as_int(best.v), best.slot};
        r[2] = int4{best.id, __float_as_int(qr.ix.x), __float_as_int(qr.iy.x), __float_as_int(qr.iz.x)};
        r[3] = int4{__float_as_int(qr.cx.x), __float_as_int(qr.cy.x), __float_as_int(qr.cz.x), 0};
      }
      for (int j = 0; j < mine; ++j) {
        const unsigned it = my_first + (unsigned)j;
        if (it < ws.cap_items)
          ws.items[it] = int2{(int)my_rec, fits ? (j == 0 ? cur : s_stack[j - 1][lane]) : TRACE_EMPTY};
      }
    }
    if (fits) {
      if (alive && !pending) write_hit(best, (long long)mesh * N + n, hit_t, hit_slot, hit_uv);
      return;
    }
    q_walk<STACK, false>(qnodes, tris, qr, ox, oy, oz, dx, dy, dz, t_min, cur, sp, best, s_stack, lane, 0);
  }
  if (alive) write_hit(best, (long long)mesh * N + n, hit_t, hit_slot, hit_uv);
}

// Pass B: one lane per handed-over subtree, 64 of them per trip of a persistent wave.
template <int STACK>
__global__ __launch_bounds__(TRACE_BLOCK) void trace_qb_kernel(
    const uint4* __restrict__ qnodes, const float4* __restrict__ tris, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, float t_min, TraceWs ws) {
  __shared__ int s_stack[STACK][TRACE_BLOCK];
  const int lane = threadIdx.x;
  const unsigned n_items = min(ws.counters[0], ws.cap_items);
  for (unsigned chunk = blockIdx.x; (unsigned long long)chunk * TRACE_BLOCK < n_items; chunk += gridDim.x) {
    const unsigned it = chunk * TRACE_BLOCK + lane;
    const bool valid = it < n_items;
    const int2 item = valid ? ws.items[it] : int2{0, TRACE_EMPTY};
    int cur = item.y;
    const int4* r = ws.ray_rec + 4ll * (cur != TRACE_EMPTY ? item.x : 0);
    const int4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
    const long long n = r0.x < 0 ? 0 : r0.x;
    if (r0.x < 0) cur = TRACE_EMPTY;
    const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
    const float dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
    QRay qr;
    {
      const float ix = __int_as_float(r2.y), iy = __int_as_float(r2.z), iz = __int_as_float(r2.w);
      const float cx = __int_as_float(r3.x), cy = __int_as_float(r3.y), cz = __int_as_float(r3.z);
      qr.ix = f32x2_t{ix, ix}, qr.iy = f32x2_t{iy, iy}, qr.iz = f32x2_t{iz, iz};
      qr.cx = f32x2_t{cx, cx}, qr.cy = f32x2_t{cy, cy}, qr.cz = f32x2_t{cz, cz};
    }
    // the ray's best t so far bounds the subtree; a tie on t is settled by the face id in pass C
    Hit best;
    best.t = __int_as_float(r1.x);
    best.u = best.v = 0.f;
    best.slot = -1;
    best.id = 0x7fffffff;
    int sp = 0;
    const bool had_work = cur != TRACE_EMPTY;
    q_walk<STACK, false>(qnodes, tris, qr, ox, oy, oz, dx, dy, dz, t_min, cur, sp, best, s_stack, lane, 0);
    if (valid && had_work) {
      ws.results[2ll * it] = int4{__float_as_int(best.t), __float_as_int(best.u), __float_as_int(best.v), best.slot};
      ws.results[2ll * it + 1] = int4{best.id, 0, 0, 0};
    }
  }
}

// Pass C: a lane per handed-over ray merges its items' candidates with what pass A had found.
__global__ __launch_bounds__(256) void trace_qc_kernel(TraceWs ws, int N, float* __restrict__ hit_t,
                                                       int* __restrict__ hit_slot, float* __restrict__ hit_uv) {
  const unsigned n_rays = min(ws.counters[1], ws.cap_rays);
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n_rays; i += gridDim.x * 256) {
    const int4* r = ws.ray_rec + 4ll * i;
    const int4 r0 = r[0], r1 = r[1], r2 = r[2];
    if (r0.x < 0) continue;
    Hit best;
    best.t = __int_as_float(r1.x), best.u = __int_as_float(r1.y), best.v = __int_as_float(r1.z);
    best.slot = r1.w, best.id = r2.x;
    for (int j = 0; j < r0.w; ++j) {
      const int4 a = ws.results[2ll * (r0.z + j)], b = ws.results[2ll * (r0.z + j) + 1];
      const float t = __int_as_float(a.x);
      if (a.w >= 0 && (t < best.t || (t == best.t && b.x < best.id))) {
        best.t = t, best.u = __int_as_float(a.y), best.v = __int_as_float(a.z);
        best.slot = a.w, best.id = b.x;
      }
    }
    write_hit(best, (long long)r0.y * N + r0.x, hit_t, hit_slot, hit_uv);
  }
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

template <int STACK>
__global__ __launch_bounds__(TRACE_BLOCK) void trace_qf_kernel(
    const uint4* __restrict__ qnodes, const float4* __restrict__ tris, Roots roots, Frames frames,
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, int N, int G, int nr_items, float t_min,
    TraceFeedbackBuf fbb, float* __restrict__ hit_t, int* __restrict__ hit_slot, float* __restrict__ hit_uv) {
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
  const int trips = q_walk<STACK, true>(qnodes, tris, qr, ox, oy, oz, dx, dy, dz, t_min, cur, sp, best, s_stack,
                                        lane, 0x7fffffff);
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

// ---- 4-wide nodes (vsa_bvh_export_q4).  PMC of trace_q_kernel at 800x800, K=5 (profiles/r03/pmc):
// 53 % of the wave cycles in s_waitcnt, 47 % VALU-busy, the vector cache at 26 % of its look-up
// rate, 10 resident waves per CU — the kernel waits for its chain of dependent node fetches.  A
// node of the collapsed tree holds four child boxes, so the chain is half as long; the four slab
// tests of a visit are the two-by-two tests of two binary visits, the hit children are visited
// nearest first (a 5-comparator sort of (t_near, reference) pairs; misses carry +inf), the other
// hits go on the LDS stack farthest first.  Triangles are tested exactly as before, and the closest
// hit is order independent, so results stay bit-identical to the oracle.
// MEASURED (round 3, same box, profiles/r03/trace_q4.txt): 0.300 -> 0.325 ms at 800x800 K=5, 0.264 ->
// 0.283 on the noisy scene, 0.656 -> 0.745 at 1080p K=7 subdiv 7: the shorter chain does not pay
// for testing all four grandchildren at every visit (the binary walk never loads the children of a
// box it missed) plus the sort and up to three stack pushes.  Not the default (raytrace.py).
__device__ __forceinline__ void q4_cswap(float& ta, int& ra, float& tb, int& rb) {
  const bool sw = tb < ta;
  const float t0 = sw ? tb : ta, t1 = sw ? ta : tb;
  const int r0 = sw ? rb : ra, r1 = sw ? ra : rb;
  ta = t0, tb = t1, ra = r0, rb = r1;
}

template <int STACK>
__global__ __launch_bounds__(TRACE_BLOCK) void trace_q4_kernel(
    const uint4* __restrict__ qnodes, const float4* __restrict__ tris, Roots roots, Frames frames,
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, int N, float t_min,
    float* __restrict__ hit_t, int* __restrict__ hit_slot, float* __restrict__ hit_uv) {
  __shared__ int s_stack[STACK][TRACE_BLOCK];
  const int lane = threadIdx.x;
  const long long n = (long long)blockIdx.x * TRACE_BLOCK + lane;
  const int mesh = blockIdx.y;
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
    while ((unsigned)cur < (unsigned)TRACE_EMPTY) {
      const uint4* np = qnodes + 4 * (long long)cur;
      const uint4 a = np[0], b = np[1], c = np[2], d = np[3];
      float t0, t1, t2, t3;
      const bool h0 = qbox_test(a.x, a.y, a.z, qr, t_min, best.t, t0);
      const bool h1 = qbox_test(a.w, b.x, b.y, qr, t_min, best.t, t1);
      const bool h2 = qbox_test(b.z, b.w, c.x, qr, t_min, best.t, t2);
      const bool h3 = qbox_test(c.y, c.z, c.w, qr, t_min, best.t, t3);
      int r0 = h0 ? (int)d.x : TRACE_EMPTY, r1 = h1 ? (int)d.y : TRACE_EMPTY;
      int r2 = h2 ? (int)d.z : TRACE_EMPTY, r3 = h3 ? (int)d.w : TRACE_EMPTY;
      t0 = h0 ? t0 : INFINITY, t1 = h1 ? t1 : INFINITY, t2 = h2 ? t2 : INFINITY, t3 = h3 ? t3 : INFINITY;
      q4_cswap(t0, r0, t1, r1);
      q4_cswap(t2, r2, t3, r3);
      q4_cswap(t0, r0, t2, r2);
      q4_cswap(t1, r1, t3, r3);
      q4_cswap(t1, r1, t2, r2);
      // (an empty slot's reference is TRACE_EMPTY whether or not its inverted box "hit")
      if (r3 != TRACE_EMPTY) s_stack[sp++][lane] = r3;
      if (r2 != TRACE_EMPTY) s_stack[sp++][lane] = r2;
      if (r1 != TRACE_EMPTY) s_stack[sp++][lane] = r1;
      cur = r0 != TRACE_EMPTY ? r0 : (sp ? s_stack[--sp][lane] : TRACE_EMPTY);
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
  const long long o = (long long)mesh * N + n;
  hit_t[o] = best.slot >= 0 ? best.t : 0.0f;
  hit_slot[o] = best.slot;
  hit_uv[2 * o] = best.u;
  hit_uv[2 * o + 1] = best.v;
}

// ---- persistent lanes.  With one (ray, shell) per lane for the lifetime of a wave, a wave lasts
// as long as its slowest ray: PMC showed 37 % lane utilisation (a tile at the silhouette has a few
// deep traversals and sixty-odd immediate misses; 71 % of the (ray, shell) pairs are misses).  Here
// a wave owns the K x 64 (ray, shell) items of one 64-ray tile and a lane that finishes its item
// takes the next one at the top of the next round (ballot + prefix count, no atomics: the pool is
// the wave's own), so finished lanes go back to work instead of idling.  An item is traversed by
// exactly the code above (same boxes, same triangle arithmetic, same order-independent closest hit),
// so results are bit-identical.  The per-mesh constants (root, quantisation frame) are indexed per
// lane, so they live in LDS.
// MEASURED AND NOT ADOPTED (round 3, same box, profiles/r03/trace_persistent.txt): 0.28 -> 0.43 ms at
// 800x800, K=5 (refilling only once 16 / 32 lanes are idle: 0.42 / 0.39).  A refilled lane starts at the
// root while its neighbours are deep in their trees, so the "walk inner nodes until the whole wave
// holds a leaf" rounds get longer for everybody, and the launch has a fifth of the waves (10 k) to
// balance over the chip.  Kept behind -DTRACE_PERSISTENT=1 with its tests (tests/test_raytrace.py runs
// whichever is built).
#ifndef TRACE_REFILL_MIN
#define TRACE_REFILL_MIN 1      /* idle lanes that trigger a refill round */
#endif
template <int STACK>
__global__ __launch_bounds__(TRACE_BLOCK) void trace_q_persistent_kernel(
    const uint4* __restrict__ qnodes, const float4* __restrict__ tris, Roots roots, Frames frames,
    int K, const float* __restrict__ rays_o, const float* __restrict__ rays_d, int N, float t_min,
    float* __restrict__ hit_t, int* __restrict__ hit_slot, float* __restrict__ hit_uv) {
  __shared__ int s_stack[STACK][TRACE_BLOCK];
  __shared__ float s_frame[VSA_MAX_SHELLS][8];    // lo.xyz, step.xyz, root (as int bits)
  const int lane = threadIdx.x;
  if (lane < K) {
#pragma unroll
    for (int j = 0; j < 6; ++j) s_frame[lane][j] = frames.f[lane][j];
    s_frame[lane][6] = __int_as_float(roots.root[lane]);
  }
  __syncthreads();
  const long long ray0 = (long long)blockIdx.x * TRACE_BLOCK;
  const int rays_here = (int)min((long long)TRACE_BLOCK, (long long)N - ray0);
  const int total = K * TRACE_BLOCK;              // items: shell-major, item = shell * 64 + ray of the tile
  int next_item = 0;
  // lane state
  float ox = 0.f, oy = 0.f, oz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
  QRay qr = {};
  Hit best = {};
  long long out = 0;
  int cur = TRACE_EMPTY, sp = 0;
  bool busy = false;
  while (true) {
    // hand the next items to the idle lanes
    const unsigned long long idle = __ballot(!busy);
    if ((__popcll(idle) >= TRACE_REFILL_MIN || __ballot(busy) == 0ull) && next_item < total) {
      const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
      const int item = next_item + rank;
      if (!busy && item < total) {
        const int mesh = item >> 6, r = item & 63;
        if (r < rays_here) {
          const long long n = ray0 + r;
          ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
          dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
          const float* fr = s_frame[mesh];
          const float gx = (ox - fr[0]) / fr[3] + 1.0f, gy = (oy - fr[1]) / fr[4] + 1.0f,
                      gz = (oz - fr[2]) / fr[5] + 1.0f;
          const float ix = 1.0f / (dx / fr[3]), iy = 1.0f / (dy / fr[4]), iz = 1.0f / (dz / fr[5]);
          qr.ix = f32x2_t{ix, ix}, qr.iy = f32x2_t{iy, iy}, qr.iz = f32x2_t{iz, iz};
          qr.cx = f32x2_t{-(gx * ix), -(gx * ix)};
          qr.cy = f32x2_t{-(gy * iy), -(gy * iy)};
          qr.cz = f32x2_t{-(gz * iz), -(gz * iz)};
          best.t = INFINITY;
          best.u = best.v = 0.f;
          best.slot = -1;
          best.id = 0x7fffffff;
          cur = __float_as_int(fr[6]);
          sp = 0;
          out = (long long)mesh * N + n;
          busy = true;
        }
      }
      next_item += __popcll(idle);
    }
    if (__ballot(busy) == 0ull) {
      if (next_item >= total) break;
      continue;                       // (a tile's tail past N: items without a ray)
    }
    while ((unsigned)cur < (unsigned)TRACE_EMPTY) {
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
    if (busy && cur == TRACE_EMPTY) {   // this lane's item is finished
      hit_t[out] = best.slot >= 0 ? best.t : 0.0f;
      hit_slot[out] = best.slot;
      hit_uv[2 * out] = best.u;
      hit_uv[2 * out + 1] = best.v;
      busy = false;
    }
  }
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

extern "C" int vsa_trace_q(const uint32_t* qnodes, const float* tris, const int32_t* mesh_roots,
                           const float* mesh_frames, int nr_meshes, int max_depth,
                           const float* rays_o, const float* rays_d, int nr_rays, float t_min,
                           float* hit_t, int32_t* hit_slot, float* hit_uv, void* stream) {
  if (nr_meshes < 1 || nr_meshes > VSA_MAX_SHELLS || nr_rays < 0 || !mesh_roots || !mesh_frames)
    return VSA_ERR_ARG;
  if (max_depth >= TRACE_STACK) return VSA_ERR_UNSUPPORTED;
  if (nr_rays == 0) return VSA_OK;
  if (!qnodes || !tris || !rays_o || !rays_d || !hit_t || !hit_slot || !hit_uv) return VSA_ERR_ARG;
  Roots r;
  Frames fr;
  for (int i = 0; i < VSA_MAX_SHELLS; ++i) {
    r.root[i] = i < nr_meshes ? mesh_roots[i] : 0;
    for (int j = 0; j < 6; ++j) fr.f[i][j] = i < nr_meshes ? mesh_frames[6 * i + j] : 1.0f;
  }
#ifndef TRACE_PERSISTENT
#define TRACE_PERSISTENT 0     /* measured slower (0.28 -> 0.43 ms): see the kernel's comment */
#endif
#if TRACE_PERSISTENT
  if (nr_meshes > 1) {     // a wave per 64-ray tile, its lanes shared out over the tile's K x 64 (ray, shell) items
    dim3 grid(vsa_div_up(nr_rays, TRACE_BLOCK)), block(TRACE_BLOCK);
    if (max_depth < 24)
      hipLaunchKernelGGL(trace_q_persistent_kernel<24>, grid, block, 0, (hipStream_t)stream,
                         reinterpret_cast<const uint4*>(qnodes), reinterpret_cast<const float4*>(tris),
                         r, fr, nr_meshes, rays_o, rays_d, nr_rays, t_min, hit_t, hit_slot, hit_uv);
    else
      hipLaunchKernelGGL(trace_q_persistent_kernel<TRACE_STACK>, grid, block, 0, (hipStream_t)stream,
                         reinterpret_cast<const uint4*>(qnodes), reinterpret_cast<const float4*>(tris),
                         r, fr, nr_meshes, rays_o, rays_d, nr_rays, t_min, hit_t, hit_slot, hit_uv);
    VSA_RETURN_LAUNCH_STATUS();
  }
#endif
  dim3 grid(vsa_div_up(nr_rays, TRACE_BLOCK), nr_meshes), block(TRACE_BLOCK);
  if (max_depth < 24)
    hipLaunchKernelGGL(trace_q_kernel<24>, grid, block, 0, (hipStream_t)stream,
                       reinterpret_cast<const uint4*>(qnodes), reinterpret_cast<const float4*>(tris),
                       r, fr, rays_o, rays_d, nr_rays, t_min, hit_t, hit_slot, hit_uv);
  else
    hipLaunchKernelGGL(trace_q_kernel<TRACE_STACK>, grid, block, 0, (hipStream_t)stream,
                       reinterpret_cast<const uint4*>(qnodes), reinterpret_cast<const float4*>(tris),
                       r, fr, rays_o, rays_d, nr_rays, t_min, hit_t, hit_slot, hit_uv);
  VSA_RETURN_LAUNCH_STATUS();
}

// Workspace of vsa_trace_q_budgeted: 256 B of counters, then per ray record 64 B + two items of 8 B +
// their two results of 32 B = 144 B.  The recommended size holds a quarter of the (ray, shell) pairs.
extern "C" long long vsa_trace_q_workspace_bytes(int nr_rays, int nr_meshes) {
  if (nr_rays < 0 || nr_meshes < 1) return -1;
  const long long pairs = (long long)nr_rays * nr_meshes;
  return 256 + 144ll * std::min<long long>(pairs / 4 + 4096, 1ll << 29);
}

extern "C" int vsa_trace_q_budgeted(const uint32_t* qnodes, const float* tris, const int32_t* mesh_roots,
                                    const float* mesh_frames, int nr_meshes, int max_depth,
                                    const float* rays_o, const float* rays_d, int nr_rays, float t_min,
                                    float* hit_t, int32_t* hit_slot, float* hit_uv, int round_budget,
                                    void* workspace, long long workspace_bytes, void* stream) {
  if (nr_meshes < 1 || nr_meshes > VSA_MAX_SHELLS || nr_rays < 0 || !mesh_roots || !mesh_frames || round_budget < 1)
    return VSA_ERR_ARG;
  if (max_depth >= TRACE_STACK) return VSA_ERR_UNSUPPORTED;
  if (nr_rays == 0) return VSA_OK;
  if (!qnodes || !tris || !rays_o || !rays_d || !hit_t || !hit_slot || !hit_uv || !workspace) return VSA_ERR_ARG;
  if (workspace_bytes < 256 + 144 || ((uintptr_t)workspace & 15)) return VSA_ERR_ARG;
  Roots r;
  Frames fr;
  for (int i = 0; i < VSA_MAX_SHELLS; ++i) {
    r.root[i] = i < nr_meshes ? mesh_roots[i] : 0;
    for (int j = 0; j < 6; ++j) fr.f[i][j] = i < nr_meshes ? mesh_frames[6 * i + j] : 1.0f;
  }
  TraceWs ws;
  ws.cap_rays = (unsigned)std::min<long long>((workspace_bytes - 256) / 144, 1ll << 29);   // any size works: what
  ws.cap_items = 2 * ws.cap_rays;                                                            // does not fit is not handed over
  char* w = static_cast<char*>(workspace);
  ws.counters = reinterpret_cast<unsigned*>(w);
  ws.ray_rec = reinterpret_cast<int4*>(w + 256);
  ws.items = reinterpret_cast<int2*>(w + 256 + 64ll * ws.cap_rays);
  ws.results = reinterpret_cast<int4*>(w + 256 + 64ll * ws.cap_rays + 8ll * ws.cap_items);
  hipStream_t s = (hipStream_t)stream;
  VSA_HIP_TRY(hipMemsetAsync(ws.counters, 0, 16, s));
  const uint4* qn = reinterpret_cast<const uint4*>(qnodes);
  const float4* tr = reinterpret_cast<const float4*>(tris);
  dim3 grid(vsa_div_up(nr_rays, TRACE_BLOCK), nr_meshes), block(TRACE_BLOCK);
  // pass B: a fixed number of persistent one-wave workgroups (the item count is only known on the device)
  int cus = 256;
  { const int rc = vsa_cu_count(&cus); if (rc != VSA_OK) return rc; }
  const int nb = 8 * cus;
  if (max_depth < 24) {
    hipLaunchKernelGGL(trace_qa_kernel<24>, grid, block, 0, s, qn, tr, r, fr, rays_o, rays_d, nr_rays, t_min,
                       round_budget, ws, hit_t, hit_slot, hit_uv);
    hipLaunchKernelGGL(trace_qb_kernel<24>, dim3(nb), block, 0, s, qn, tr, rays_o, rays_d, t_min, ws);
  } else {
    hipLaunchKernelGGL(trace_qa_kernel<TRACE_STACK>, grid, block, 0, s, qn, tr, r, fr, rays_o, rays_d, nr_rays,
                       t_min, round_budget, ws, hit_t, hit_slot, hit_uv);
    hipLaunchKernelGGL(trace_qb_kernel<TRACE_STACK>, dim3(nb), block, 0, s, qn, tr, rays_o, rays_d, t_min, ws);
  }
  hipLaunchKernelGGL(trace_qc_kernel, dim3(2 * cus), dim3(256), 0, s, ws, nr_rays, hit_t, hit_slot, hit_uv);
  VSA_RETURN_LAUNCH_STATUS();
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
  if (max_depth < 24)
    hipLaunchKernelGGL(trace_qf_kernel<24>, grid, block, 0, s, qn, tr, r, fr, rays_o, rays_d, nr_rays, G, (int)items,
                       t_min, fb, hit_t, hit_slot, hit_uv);
  else
    hipLaunchKernelGGL(trace_qf_kernel<TRACE_STACK>, grid, block, 0, s, qn, tr, r, fr, rays_o, rays_d, nr_rays, G,
                       (int)items, t_min, fb, hit_t, hit_slot, hit_uv);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_trace_q4(const uint32_t* qnodes4, const float* tris, const int32_t* mesh_roots,
                            const float* mesh_frames, int nr_meshes, int max_depth4,
                            const float* rays_o, const float* rays_d, int nr_rays, float t_min,
                            float* hit_t, int32_t* hit_slot, float* hit_uv, void* stream) {
  if (nr_meshes < 1 || nr_meshes > VSA_MAX_SHELLS || nr_rays < 0 || !mesh_roots || !mesh_frames)
    return VSA_ERR_ARG;
  // a visit pushes at most three references and descends into the fourth: the stack never holds
  // more than 3 x (depth - 1) entries
  const int need = 3 * (max_depth4 > 0 ? max_depth4 - 1 : 0);
  if (need > 96) return VSA_ERR_UNSUPPORTED;
  if (nr_rays == 0) return VSA_OK;
  if (!qnodes4 || !tris || !rays_o || !rays_d || !hit_t || !hit_slot || !hit_uv) return VSA_ERR_ARG;
  Roots r;
  Frames fr;
  for (int i = 0; i < VSA_MAX_SHELLS; ++i) {
    r.root[i] = i < nr_meshes ? mesh_roots[i] : 0;
    for (int j = 0; j < 6; ++j) fr.f[i][j] = i < nr_meshes ? mesh_frames[6 * i + j] : 1.0f;
  }
  dim3 grid(vsa_div_up(nr_rays, TRACE_BLOCK), nr_meshes), block(TRACE_BLOCK);
#define Q4_GO(S)                                                                                      \
  hipLaunchKernelGGL(trace_q4_kernel<S>, grid, block, 0, (hipStream_t)stream,                         \
                     reinterpret_cast<const uint4*>(qnodes4), reinterpret_cast<const float4*>(tris), r, fr, \
                     rays_o, rays_d, nr_rays, t_min, hit_t, hit_slot, hit_uv)
  if (need <= 32) Q4_GO(32);
  else if (need <= 48) Q4_GO(48);
  else Q4_GO(96);
#undef Q4_GO
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
