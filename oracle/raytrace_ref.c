/* Oracle: brute-force closest-hit ray / triangle-mesh intersection.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Stands in for `raytracelib.RayTracer.trace` (call sites
 * /root/reference/volsurfs_py/methods/volsurfs.py:128, 476-501 and
 * renderers/mesh_renderer.py:46,131).  raytracelib (s-esposito/raytracelib,
 * branch `volsurfs`, commit unpinned: .gitmodules:14-17) is NOT under
 * /root/reference, so this restates the published Moeller-Trumbore test over
 * every face (O(N*F), no acceleration structure) and anchors parity on the
 * reference's call-site contract: closest hit per mesh; outputs is_hit,
 * triangles_id, depth, positions, normals, barycentric.  PARITY UNPINNED.
 *
 * Closest hit = smallest t > t_min, ties -> smallest face index.  The
 * arithmetic is fp32 in a fixed order (no FMA contraction) so that the HIP
 * traversal kernel can be compared bit for bit.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>

static float dot3(float ax, float ay, float az, float bx, float by, float bz) {
  return (ax * bx + ay * by) + az * bz;
}

/* verts [nv,3], faces [nf,3], rays_o/rays_d [n,3]; out: t [n] (0 on miss),
 * tri [n] (-1 on miss), uv [n,2] (weights of v1, v2). */
void oracle_trace_bruteforce(const float* verts, const int32_t* faces, int nf,
                             const float* rays_o, const float* rays_d, int n, float t_min,
                             float* out_t, int32_t* out_tri, float* out_uv) {
  /* rays are independent: the loop is shared out over the host's cores (same results);
   * bench.py's cpu_baseline reports how many threads ran */
  /* every core, whatever the host program set for its own OpenMP regions (the tests cap torch-CPU at 16 threads:
   * its small ops crawl on a 256-core host; this loop is embarrassingly parallel) */
#pragma omp parallel for schedule(dynamic, 16) num_threads(omp_get_num_procs())
  for (int r = 0; r < n; ++r) {
    const float ox = rays_o[3 * r], oy = rays_o[3 * r + 1], oz = rays_o[3 * r + 2];
    const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz = rays_d[3 * r + 2];
    float bt = INFINITY, bu = 0.f, bv = 0.f;
    int32_t bid = -1;
    for (int f = 0; f < nf; ++f) {
      const float* a = verts + 3 * (long)faces[3 * f];
      const float* b = verts + 3 * (long)faces[3 * f + 1];
      const float* c = verts + 3 * (long)faces[3 * f + 2];
      const float e1x = b[0] - a[0], e1y = b[1] - a[1], e1z = b[2] - a[2];
      const float e2x = c[0] - a[0], e2y = c[1] - a[1], e2z = c[2] - a[2];
      const float px = dy * e2z - dz * e2y;
      const float py = dz * e2x - dx * e2z;
      const float pz = dx * e2y - dy * e2x;
      const float det = dot3(e1x, e1y, e1z, px, py, pz);
      if (fabsf(det) < 1e-20f) continue;
      const float inv = 1.0f / det;
      const float tx = ox - a[0], ty = oy - a[1], tz = oz - a[2];
      const float u = dot3(tx, ty, tz, px, py, pz) * inv;
      if (!(u >= 0.0f && u <= 1.0f)) continue;
      const float qx = ty * e1z - tz * e1y;
      const float qy = tz * e1x - tx * e1z;
      const float qz = tx * e1y - ty * e1x;
      const float v = dot3(dx, dy, dz, qx, qy, qz) * inv;
      if (!(v >= 0.0f && u + v <= 1.0f)) continue;
      const float t = dot3(e2x, e2y, e2z, qx, qy, qz) * inv;
      if (!(t > t_min)) continue;
      if (t < bt) { /* faces visited in increasing index: strict < keeps the smallest id on ties */
        bt = t; bu = u; bv = v; bid = f;
      }
    }
    out_t[r] = bid >= 0 ? bt : 0.0f;
    out_tri[r] = bid;
    out_uv[2 * r] = bu;
    out_uv[2 * r + 1] = bv;
  }
}
