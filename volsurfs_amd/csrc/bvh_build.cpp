// Host-side BVH builder for the K shell meshes (build half).
//
// Replaces `raytracelib.RayTracer(tensor_meshes)` (call site
// /root/reference/volsurfs_py/methods/volsurfs.py:128); raytracelib's source is
// not under /root/reference, so this is a from-scratch design: a binned-SAH
// binary BVH whose 64-byte nodes carry BOTH children's boxes (one 64-B fetch =
// two slab tests on the GPU), leaves of <= leaf_size triangles stored
// contiguously in traversal order as (v0, e1, e2) float4 triples.
//
// Node (16 x 4 B):  c0.min.xyz c0.max.xyz c1.min.xyz c1.max.xyz ref0 ref1 cnt0 cnt1
//   ref >= 0 : index of an inner node; cnt == 0
//   ref <  0 : leaf, first triangle = ~ref, cnt triangles
// Triangle (12 x 4 B): v0.xyz id | e1.xyz 0 | e2.xyz 0   (id = original face index)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/volsurfs_hip.h"

namespace {

struct V3 {
  float x, y, z;
};
inline V3 vmin(V3 a, V3 b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }
inline float axis(const V3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

struct Box {
  V3 lo{INFINITY, INFINITY, INFINITY}, hi{-INFINITY, -INFINITY, -INFINITY};
  void grow(const V3& p) { lo = vmin(lo, p); hi = vmax(hi, p); }
  void grow(const Box& b) { lo = vmin(lo, b.lo); hi = vmax(hi, b.hi); }
  float area() const {
    float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
    if (dx < 0) return 0.f;
    return 2.f * (dx * dy + dy * dz + dz * dx);
  }
};

struct BuildNode {
  Box box[2];
  int32_t ref[2];
  int32_t cnt[2];
};

}  // namespace

struct vsa_bvh {
  std::vector<BuildNode> nodes;
  std::vector<float> tris;  // 12 floats per triangle, leaf order
  std::vector<int32_t> faces;   // the connectivity it was built on (vsa_bvh_refit)
  int nr_verts = 0;
  int max_depth = 0;
  Box root_box;
};

namespace {

struct Builder {
  const float* verts;
  const int32_t* faces;
  int leaf_size;
  std::vector<Box> tbox;
  std::vector<V3> tcen;
  std::vector<int32_t> order;
  vsa_bvh* out;
  float pad;

  static constexpr int NBINS = 16;

  Box padded(Box b) const {
    // conservative padding so that the fp32 slab test on the GPU never culls a
    // triangle that the (exact-formula) triangle test would accept
    V3 e{pad + 1e-6f * std::max(std::fabs(b.lo.x), std::fabs(b.hi.x)),
         pad + 1e-6f * std::max(std::fabs(b.lo.y), std::fabs(b.hi.y)),
         pad + 1e-6f * std::max(std::fabs(b.lo.z), std::fabs(b.hi.z))};
    b.lo = {b.lo.x - e.x, b.lo.y - e.y, b.lo.z - e.z};
    b.hi = {b.hi.x + e.x, b.hi.y + e.y, b.hi.z + e.z};
    return b;
  }

  // Emits the subtree over order[begin,end); returns (ref, cnt, box).
  void emit_leaf(int begin, int end, int32_t& ref, int32_t& cnt) {
    int first = (int)(out->tris.size() / 12);
    for (int i = begin; i < end; ++i) {
      int f = order[i];
      const float* a = verts + 3 * (size_t)faces[3 * (size_t)f + 0];
      const float* b = verts + 3 * (size_t)faces[3 * (size_t)f + 1];
      const float* c = verts + 3 * (size_t)faces[3 * (size_t)f + 2];
      float idf;
      int32_t id = f;
      std::memcpy(&idf, &id, 4);
      float t[12] = {a[0], a[1], a[2], idf, b[0] - a[0], b[1] - a[1], b[2] - a[2], 0.f,
                     c[0] - a[0], c[1] - a[1], c[2] - a[2], 0.f};
      out->tris.insert(out->tris.end(), t, t + 12);
    }
    ref = ~first;
    cnt = end - begin;
  }

  void build(int begin, int end, int depth, int32_t& ref, int32_t& cnt, Box& box) {
    box = Box();
    Box cbox;
    for (int i = begin; i < end; ++i) {
      box.grow(tbox[order[i]]);
      cbox.grow(tcen[order[i]]);
    }
    out->max_depth = std::max(out->max_depth, depth);
    int n = end - begin;
    if (n <= leaf_size) {
      emit_leaf(begin, end, ref, cnt);
      return;
    }
    // binned SAH over the widest centroid axis first, then the others
    int best_axis = -1, best_bin = -1;
    float best_cost = INFINITY;
    for (int ax = 0; ax < 3; ++ax) {
      float lo = axis(cbox.lo, ax), hi = axis(cbox.hi, ax);
      if (!(hi > lo)) continue;
      Box bb[NBINS];
      int bc[NBINS] = {0};
      float scale = NBINS / (hi - lo);
      for (int i = begin; i < end; ++i) {
        int t = order[i];
        int b = std::min(NBINS - 1, (int)((axis(tcen[t], ax) - lo) * scale));
        bb[b].grow(tbox[t]);
        bc[b]++;
      }
      float ra[NBINS];
      int rc[NBINS];
      Box acc;
      int c = 0;
      for (int b = NBINS - 1; b > 0; --b) {
        acc.grow(bb[b]);
        c += bc[b];
        ra[b] = acc.area();
        rc[b] = c;
      }
      acc = Box();
      c = 0;
      for (int b = 0; b < NBINS - 1; ++b) {
        acc.grow(bb[b]);
        c += bc[b];
        if (c == 0 || rc[b + 1] == 0) continue;
        float cost = acc.area() * c + ra[b + 1] * rc[b + 1];
        if (cost < best_cost) {
          best_cost = cost;
          best_axis = ax;
          best_bin = b;
        }
      }
    }
    int mid;
    if (best_axis < 0 || depth >= 24) {
      // degenerate centroids, or SAH chain deeper than 24: balanced median
      // splits from here on bound the depth by 24 + log2(n / leaf_size), which
      // the traversal kernel's LDS stack (48 entries) relies on
      mid = begin + n / 2;
      int ax = 0;
      float ex = cbox.hi.x - cbox.lo.x, ey = cbox.hi.y - cbox.lo.y, ez = cbox.hi.z - cbox.lo.z;
      if (ey > ex && ey >= ez) ax = 1;
      else if (ez > ex && ez > ey) ax = 2;
      std::nth_element(order.begin() + begin, order.begin() + mid, order.begin() + end,
                       [&](int a, int b) { return axis(tcen[a], ax) < axis(tcen[b], ax); });
    } else {
      float lo = axis(cbox.lo, best_axis), hi = axis(cbox.hi, best_axis);
      float scale = NBINS / (hi - lo);
      auto it = std::partition(order.begin() + begin, order.begin() + end, [&](int t) {
        int b = std::min(NBINS - 1, (int)((axis(tcen[t], best_axis) - lo) * scale));
        return b <= best_bin;
      });
      mid = (int)(it - order.begin());
      if (mid == begin || mid == end) mid = begin + n / 2;
    }
    int me = (int)out->nodes.size();
    out->nodes.emplace_back();
    BuildNode tmp;
    Box b0, b1;
    build(begin, mid, depth + 1, tmp.ref[0], tmp.cnt[0], b0);
    build(mid, end, depth + 1, tmp.ref[1], tmp.cnt[1], b1);
    tmp.box[0] = padded(b0);
    tmp.box[1] = padded(b1);
    out->nodes[me] = tmp;
    ref = me;
    cnt = 0;
  }
};

}  // namespace

extern "C" int vsa_bvh_build(const float* verts, const int32_t* faces, int nr_verts, int nr_faces,
                             int leaf_size, vsa_bvh** out_bvh) {
  if (!verts || !faces || !out_bvh || nr_verts <= 0 || nr_faces <= 0) return VSA_ERR_ARG;
  if (leaf_size < 1) leaf_size = 4;
  if (leaf_size > 8) leaf_size = 8;
  for (size_t i = 0; i < (size_t)nr_faces * 3; ++i)
    if (faces[i] < 0 || faces[i] >= nr_verts) return VSA_ERR_ARG;
  vsa_bvh* bvh = new vsa_bvh();
  bvh->faces.assign(faces, faces + (size_t)nr_faces * 3);
  bvh->nr_verts = nr_verts;
  Builder b;
  b.verts = verts;
  b.faces = faces;
  b.leaf_size = leaf_size;
  b.out = bvh;
  b.tbox.resize(nr_faces);
  b.tcen.resize(nr_faces);
  b.order.resize(nr_faces);
  Box all;
  for (int f = 0; f < nr_faces; ++f) {
    Box bx;
    for (int k = 0; k < 3; ++k) {
      const float* p = verts + 3 * (size_t)faces[3 * (size_t)f + k];
      bx.grow(V3{p[0], p[1], p[2]});
    }
    b.tbox[f] = bx;
    b.tcen[f] = {0.5f * (bx.lo.x + bx.hi.x), 0.5f * (bx.lo.y + bx.hi.y), 0.5f * (bx.lo.z + bx.hi.z)};
    b.order[f] = f;
    all.grow(bx);
  }
  float diag = std::sqrt((all.hi.x - all.lo.x) * (all.hi.x - all.lo.x) +
                         (all.hi.y - all.lo.y) * (all.hi.y - all.lo.y) +
                         (all.hi.z - all.lo.z) * (all.hi.z - all.lo.z));
  b.pad = 1e-6f * diag;
  bvh->root_box = b.padded(all);
  bvh->nodes.reserve((size_t)nr_faces);
  bvh->tris.reserve((size_t)nr_faces * 12);
  int32_t ref, cnt;
  Box box;
  b.build(0, nr_faces, 0, ref, cnt, box);
  if (bvh->nodes.empty()) {
    // a single leaf: wrap it in a root whose second child is empty
    BuildNode root;
    root.box[0] = b.padded(box);
    root.ref[0] = ref;
    root.cnt[0] = cnt;
    root.box[1] = Box();
    root.box[1].lo = {1.f, 1.f, 1.f};
    root.box[1].hi = {-1.f, -1.f, -1.f};  // inverted: never hit
    root.ref[1] = ~0;
    root.cnt[1] = 0;
    bvh->nodes.push_back(root);
  }
  *out_bvh = bvh;
  return VSA_OK;
}

extern "C" int vsa_bvh_sizes(const vsa_bvh* bvh, int* nr_nodes, int* nr_tris, int* max_depth) {
  if (!bvh) return VSA_ERR_ARG;
  if (nr_nodes) *nr_nodes = (int)bvh->nodes.size();
  if (nr_tris) *nr_tris = (int)(bvh->tris.size() / 12);
  if (max_depth) *max_depth = bvh->max_depth;
  return VSA_OK;
}

extern "C" int vsa_bvh_export(const vsa_bvh* bvh, float* nodes_out, float* tris_out,
                              int node_base, int tri_base) {
  if (!bvh || !nodes_out || !tris_out || node_base < 0 || tri_base < 0) return VSA_ERR_ARG;
  for (size_t i = 0; i < bvh->nodes.size(); ++i) {
    const BuildNode& n = bvh->nodes[i];
    float* o = nodes_out + 16 * i;
    o[0] = n.box[0].lo.x; o[1] = n.box[0].lo.y; o[2] = n.box[0].lo.z;
    o[3] = n.box[0].hi.x; o[4] = n.box[0].hi.y; o[5] = n.box[0].hi.z;
    o[6] = n.box[1].lo.x; o[7] = n.box[1].lo.y; o[8] = n.box[1].lo.z;
    o[9] = n.box[1].hi.x; o[10] = n.box[1].hi.y; o[11] = n.box[1].hi.z;
    // rebase mesh-local references into the caller's concatenated arrays
    int32_t tail[4] = {n.ref[0] >= 0 ? n.ref[0] + node_base : ~(~n.ref[0] + tri_base),
                       n.ref[1] >= 0 ? n.ref[1] + node_base : ~(~n.ref[1] + tri_base),
                       n.cnt[0], n.cnt[1]};
    std::memcpy(o + 12, tail, 16);
  }
  std::memcpy(tris_out, bvh->tris.data(), bvh->tris.size() * sizeof(float));
  return VSA_OK;
}

// Quantised export: 32-byte nodes.  Each child box is 6 x u16 on a per-mesh grid
// (frame: lo[3], step[3] with step = extent / 65533), rounded OUTWARD and widened by one
// more grid unit per side, which covers the fp32 error of the traversal's box test in grid
// coordinates; the boxes only prune, so closest hits stay bit-identical.  Dwords 6, 7 are
// the children: inner node index (>= 0), leaf code ~((first_tri << 4) | count), or
// 0x7fffffff for an empty child.
extern "C" int vsa_bvh_export_q(const vsa_bvh* bvh, uint32_t* qnodes_out, float* tris_out,
                                int node_base, int tri_base, float* frame_out) {
  if (!bvh || !qnodes_out || !tris_out || !frame_out || node_base < 0 || tri_base < 0)
    return VSA_ERR_ARG;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (const BuildNode& n : bvh->nodes)
    for (int c = 0; c < 2; ++c) {
      if (n.ref[c] < 0 && n.cnt[c] == 0) continue;
      const float l[3] = {n.box[c].lo.x, n.box[c].lo.y, n.box[c].lo.z};
      const float h[3] = {n.box[c].hi.x, n.box[c].hi.y, n.box[c].hi.z};
      for (int a = 0; a < 3; ++a) {
        lo[a] = std::min(lo[a], l[a]);
        hi[a] = std::max(hi[a], h[a]);
      }
    }
  double step[3];
  for (int a = 0; a < 3; ++a) {
    if (!(hi[a] >= lo[a])) lo[a] = hi[a] = 0.f;   // empty mesh
    step[a] = std::max((double)hi[a] - (double)lo[a], 1e-30) / 65533.0;
    frame_out[a] = lo[a];
    frame_out[3 + a] = (float)step[a];
    step[a] = (double)frame_out[3 + a];            // quantise against the step the device uses
  }
  auto qlo = [&](float x, int a) {
    const double q = std::floor(((double)x - (double)lo[a]) / step[a]) - 1.0 + 1.0;   // grid is offset by +1
    return (uint32_t)std::min(std::max(q, 0.0), 65535.0);
  };
  auto qhi = [&](float x, int a) {
    const double q = std::ceil(((double)x - (double)lo[a]) / step[a]) + 1.0 + 1.0;
    return (uint32_t)std::min(std::max(q, 0.0), 65535.0);
  };
  for (size_t i = 0; i < bvh->nodes.size(); ++i) {
    const BuildNode& n = bvh->nodes[i];
    uint32_t* o = qnodes_out + 8 * i;
    for (int c = 0; c < 2; ++c) {
      uint32_t q[6];
      const bool empty = n.ref[c] < 0 && n.cnt[c] == 0;
      if (empty) {
        q[0] = q[1] = q[2] = 65535u;   // inverted box: never hit
        q[3] = q[4] = q[5] = 0u;
      } else {
        q[0] = qlo(n.box[c].lo.x, 0); q[1] = qlo(n.box[c].lo.y, 1); q[2] = qlo(n.box[c].lo.z, 2);
        q[3] = qhi(n.box[c].hi.x, 0); q[4] = qhi(n.box[c].hi.y, 1); q[5] = qhi(n.box[c].hi.z, 2);
      }
      o[3 * c + 0] = q[0] | (q[1] << 16);
      o[3 * c + 1] = q[2] | (q[3] << 16);
      o[3 * c + 2] = q[4] | (q[5] << 16);
      int32_t ref;
      if (empty) ref = 0x7fffffff;
      else if (n.ref[c] >= 0) ref = n.ref[c] + node_base;
      else ref = ~(((~n.ref[c] + tri_base) << 4) | n.cnt[c]);
      o[6 + c] = (uint32_t)ref;
    }
  }
  std::memcpy(tris_out, bvh->tris.data(), bvh->tris.size() * sizeof(float));
  return VSA_OK;
}

// Refit: the vertices moved, the connectivity did not.  Every triangle record is
// re-formed from the new positions (leaf order, hence every triangle slot and the per-slot uv table, is
// unchanged) and the child boxes are recomputed bottom-up: nodes were emitted parent before children, so
// one pass over the node array from its end sees both children of a node before the node itself.  The
// topology keeps the SAH quality of the positions it was built on; hits stay exact whatever the boxes'
// quality (they only prune), i.e. bit-identical to a rebuild's and to brute force.
extern "C" int vsa_bvh_refit(vsa_bvh* bvh, const float* verts, int nr_verts) {
  if (!bvh || !verts || nr_verts != bvh->nr_verts) return VSA_ERR_ARG;
  const size_t nt = bvh->tris.size() / 12;
  std::vector<Box> tbox(nt);
  Box all;
  for (size_t i = 0; i < nt; ++i) {
    float* t = bvh->tris.data() + 12 * i;
    int32_t id;
    std::memcpy(&id, t + 3, 4);
    const float* a = verts + 3 * (size_t)bvh->faces[3 * (size_t)id + 0];
    const float* b = verts + 3 * (size_t)bvh->faces[3 * (size_t)id + 1];
    const float* c = verts + 3 * (size_t)bvh->faces[3 * (size_t)id + 2];
    t[0] = a[0], t[1] = a[1], t[2] = a[2];
    t[4] = b[0] - a[0], t[5] = b[1] - a[1], t[6] = b[2] - a[2];
    t[8] = c[0] - a[0], t[9] = c[1] - a[1], t[10] = c[2] - a[2];
    Box bx;
    bx.grow(V3{a[0], a[1], a[2]});
    bx.grow(V3{b[0], b[1], b[2]});
    bx.grow(V3{c[0], c[1], c[2]});
    tbox[i] = bx;
    all.grow(bx);
  }
  Builder pd;      // (only for padded(): the same conservative widening as the build)
  const float dx = all.hi.x - all.lo.x, dy = all.hi.y - all.lo.y, dz = all.hi.z - all.lo.z;
  pd.pad = 1e-6f * std::sqrt(dx * dx + dy * dy + dz * dz);
  bvh->root_box = pd.padded(all);
  std::vector<Box> raw(2 * bvh->nodes.size());       // un-padded child boxes
  for (size_t i = bvh->nodes.size(); i-- > 0;) {
    BuildNode& n = bvh->nodes[i];
    for (int c = 0; c < 2; ++c) {
      if (n.ref[c] < 0 && n.cnt[c] == 0) continue;   // the empty child of a one-leaf tree
      Box bx;
      if (n.ref[c] < 0) {
        for (int k = 0; k < n.cnt[c]; ++k) bx.grow(tbox[(size_t)(~n.ref[c]) + k]);
      } else {
        if ((size_t)n.ref[c] <= i) return VSA_ERR_UNSUPPORTED;   // not a pre-order tree: not built by vsa_bvh_build
        bx.grow(raw[2 * (size_t)n.ref[c]]);
        bx.grow(raw[2 * (size_t)n.ref[c] + 1]);
      }
      raw[2 * i + c] = bx;
      n.box[c] = pd.padded(bx);
    }
  }
  return VSA_OK;
}

extern "C" int vsa_bvh_destroy(vsa_bvh* bvh) {
  delete bvh;
  return VSA_OK;
}
