// OccupancyGrid: the 3-D Morton-ordered value / occupancy grid of the
// sibling methods (nerf / surf / offsets_surfs, utils/occupancy_grid.py) and the ray marchers
// that consult it — include/volsurfs/OccupancyGrid.cuh:9-68, src/OccupancyGrid.cu,
// kernels/volsurfs/OccupancyGridGPU.cuh:31-581, kernels/volsurfs/occ_grid_helpers.h:14-180,
// and RaySampler::compute_samples_fg_in_grid_occupied_regions
// (kernels/volsurfs/RaySamplerGPU.cuh:275-457).
//
// The arithmetic is restated operation by operation in fp32 (oracle/occupancy.py is the CPU
// restatement these kernels are bit-compared with), including the reference's conversions:
// float -> unsigned voxel coordinates saturate at 0 for negative positions, the 21-bit Morton
// spread is truncated to 32 bits before the shifts, a voxel index is "out of range" when it is
// >= n^3 or negative as an int.  One thread per voxel / point / ray: every marcher is a serial
// DDA along its ray, the grid (16 MiB of values + 2 x 16 MiB of flags at 256^3) stays L2 / MALL
// resident.  Deviation: every marching loop is capped at OCC_MAX_ITERS steps (the reference's
// `while (t < t_exit)` loops never end once the 1e-6 step falls below the spacing of t).
#include "common.h"
#include "pcg32.h"

namespace {

constexpr int OCC_MAX_ITERS = 1 << 16;

struct Extent {
  float x, y, z;
};

// bit i of the low 21 bits -> bit 3i
__device__ __forceinline__ unsigned long long spread3(unsigned long long w) {
  w &= 0x1fffffull;
  w = (w | (w << 32)) & 0x001f00000000ffffull;
  w = (w | (w << 16)) & 0x001f0000ff0000ffull;
  w = (w | (w << 8)) & 0x100f00f00f00f00full;
  w = (w | (w << 4)) & 0x10c30c30c30c30c3ull;
  w = (w | (w << 2)) & 0x1249249249249249ull;
  return w;
}

__device__ __forceinline__ unsigned morton_encode(unsigned x, unsigned y, unsigned z) {
  const unsigned xx = (unsigned)spread3(x), yy = (unsigned)spread3(y), zz = (unsigned)spread3(z);
  return xx | (yy << 1) | (zz << 2);
}

// bits 0, 3, 6, ... of x gathered into the low bits
__device__ __forceinline__ unsigned gather3(unsigned x) {
  x &= 0x49249249u;
  x = (x | (x >> 2)) & 0xc30c30c3u;
  x = (x | (x >> 4)) & 0x0f00f00fu;
  x = (x | (x >> 8)) & 0xff0000ffu;
  x = (x | (x >> 16)) & 0x0000ffffu;
  return x;
}

// occ_grid_helpers.h:55-78: world position -> Morton voxel index (as an int: may be negative)
__device__ __forceinline__ int voxel_of(float px, float py, float pz, int n, const Extent& e) {
  px = (px / e.x + 0.5f) * (float)n;
  py = (py / e.y + 0.5f) * (float)n;
  pz = (pz / e.z + 0.5f) * (float)n;
  // float -> uint32_t as the reference's implicit conversion compiles: saturating
  return (int)morton_encode((unsigned)fmaxf(px, 0.f), (unsigned)fmaxf(py, 0.f), (unsigned)fmaxf(pz, 0.f));
}

__device__ __forceinline__ bool voxel_ok(int v, int n) { return v >= 0 && v < n * n * n; }

// occ_grid_helpers.h:80-120: Morton index -> position (grid centred on the origin or not,
// lower-left vertex or centre of the voxel)
__device__ __forceinline__ void voxel_pos(unsigned v, int n, const Extent& e, bool centre_grid,
                                          bool centre_of_voxel, float out[3]) {
  float c[3] = {(float)gather3(v), (float)gather3(v >> 1), (float)gather3(v >> 2)};
  const float ext[3] = {e.x, e.y, e.z};
  const float voxel = 1.0f / (float)n, half = voxel / 2.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float x = c[k] / (float)n;
    if (centre_grid) x = x - 0.5f;
    if (centre_of_voxel) x += half;
    out[k] = x * ext[k];
  }
}

__device__ __forceinline__ int sign_of(float x) { return x > 0.f ? 1 : (x < 0.f ? -1 : 0); }

// occ_grid_helpers.h:131-180: distance along the ray to the next voxel boundary (+ 1e-6)
__device__ __forceinline__ float next_voxel_dist(float px, float py, float pz, float dx, float dy,
                                                 float dz, int n, const Extent& e) {
  const float eps = 1e-6f;
  if (fabsf(dx) < eps && fabsf(dy) < eps && fabsf(dz) < eps) return 1e10f;
  const float p[3] = {px / e.x * (float)n, py / e.y * (float)n, pz / e.z * (float)n};
  const float d[3] = {dx, dy, dz}, ext[3] = {e.x, e.y, e.z};
  float t[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    t[k] = 1e10f;
    if (fabsf(d[k]) > eps) {
      const float prime = floorf(p[k] + 1.0f * (float)sign_of(d[k]));
      t[k] = fabsf(prime - p[k]) / (float)n * ext[k];
    }
  }
  return fminf(fminf(t[0], t[1]), t[2]) + eps;
}

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

#define OCC_THREAD(i, n)                                                 \
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; \
  if (i >= (n)) return;

// OccupancyGridGPU.cuh:31-59 (vertices) and :62-119 (samples, optionally jittered)
__global__ void grid_points_kernel(const int* __restrict__ indices, int count, int n, Extent e,
                                   int centre_of_voxel, int jitter, unsigned long long rng_state,
                                   unsigned long long rng_inc, float* __restrict__ out) {
  OCC_THREAD(i, count);
  const unsigned v = indices ? (unsigned)indices[i] : (unsigned)i;
  float p[3];
  voxel_pos(v, n, e, true, centre_of_voxel != 0, p);
  if (jitter) {
    const float ext[3] = {e.x, e.y, e.z};
    Pcg32 rng{rng_state, rng_inc};
    rng.advance((unsigned long long)(long long)((int)i * 3));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float voxel = ext[k] / (float)n, half = voxel / 2.0f;
      p[k] += voxel * rng.next_float() - half;
    }
  }
  out[3 * i] = p[0], out[3 * i + 1] = p[1], out[3 * i + 2] = p[2];
}

// :122-151
__global__ void update_values_kernel(const int* __restrict__ indices, const float* __restrict__ values,
                                     int count, float decay, float* __restrict__ grid) {
  OCC_THREAD(i, count);
  const int v = indices[i];
  grid[v] = fmaxf(values[i], grid[v] * decay);
}

// :153-225
__global__ void occupancy_from_density_kernel(const int* __restrict__ indices, int count, int n,
                                              Extent e, float thresh, int neighbours,
                                              const float* __restrict__ grid,
                                              unsigned char* __restrict__ occ) {
  OCC_THREAD(i, count);
  const int v = indices[i];
  bool empty = true;
  if (neighbours) {
    float p[3];
    voxel_pos((unsigned)v, n, e, false, false, p);   // lower-left vertex, then x n: the voxel coordinates
    p[0] = p[0] * (float)n, p[1] = p[1] * (float)n, p[2] = p[2] * (float)n;
    for (int a = -1; a <= 1; ++a) {
      if (p[0] + a < 0 || p[0] + a > n - 1) continue;
      for (int b = -1; b <= 1; ++b) {
        if (p[1] + b < 0 || p[1] + b > n - 1) continue;
        for (int c = -1; c <= 1; ++c) {
          if (p[2] + c < 0 || p[2] + c > n - 1) continue;
          const unsigned nb = morton_encode((unsigned)(p[0] + a), (unsigned)(p[1] + b), (unsigned)(p[2] + c));
          empty = empty && grid[nb] <= thresh;
        }
      }
    }
  } else {
    empty = grid[v] <= thresh;
  }
  occ[v] = !empty;
}

// :229-315: NeuS logistic density of the smallest |sdf| reachable inside the voxel
__global__ void occupancy_from_sdf_kernel(const int* __restrict__ indices,
                                          const float* __restrict__ beta, int count, int n, Extent e,
                                          float thresh, const float* __restrict__ grid,
                                          unsigned char* __restrict__ occ) {
  OCC_THREAD(i, count);
  const int v = indices[i];
  const float df = fabsf(grid[v]);
  // the largest distance between two vertices of a voxel = its diagonal, found as the
  // reference finds it (max over vertex pairs of sqrt(dx^2 + dy^2 + dz^2))
  const float s[3] = {e.x / (float)n, e.y / (float)n, e.z / (float)n};
  float diag = 0.f;
  for (int a = 0; a < 8; ++a)
    for (int b = 0; b < 8; ++b) {
      const float dx = (float)(b & 1) * s[0] - (float)(a & 1) * s[0];
      const float dy = (float)((b >> 1) & 1) * s[1] - (float)((a >> 1) & 1) * s[1];
      const float dz = (float)((b >> 2) & 1) * s[2] - (float)((a >> 2) & 1) * s[2];
      diag = fmaxf(diag, sqrtf((dx * dx + dy * dy) + dz * dz));
    }
  const float x = clampf(df - diag / 2.0f, 0.0f, 1e10f);
  const float b_ = beta[i];
  const float ex = clampf(expf(-b_ * x), -1e6f, 1e6f);
  const float opx = 1.0f + ex;
  const float w = b_ * ex / (opx * opx);
  occ[v] = w > thresh;
}

// :376-413
__global__ void check_occupancy_kernel(const float* __restrict__ pts, int count, int n, Extent e,
                                       const float* __restrict__ grid,
                                       const unsigned char* __restrict__ occ,
                                       const unsigned char* __restrict__ roi,
                                       unsigned char* __restrict__ out_occ, float* __restrict__ out_val) {
  OCC_THREAD(i, count);
  const int v = voxel_of(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], n, e);
  const bool ok = voxel_ok(v, n);
  out_occ[i] = ok && roi[v] && occ[v];
  out_val[i] = ok ? grid[v] : 0.0f;
}

struct RayIn {
  const float *o, *d, *t0, *t1;
};

// :318-374: where the ray first enters and last leaves occupied voxels of the region of interest
__global__ void rays_near_far_kernel(RayIn r, int count, int n, Extent e,
                                     const unsigned char* __restrict__ occ,
                                     const unsigned char* __restrict__ roi, float* __restrict__ near,
                                     float* __restrict__ far) {
  OCC_THREAD(i, count);
  const float ox = r.o[3 * i], oy = r.o[3 * i + 1], oz = r.o[3 * i + 2];
  const float dx = r.d[3 * i], dy = r.d[3 * i + 1], dz = r.d[3 * i + 2];
  const float t_start = r.t0[i], t_exit = r.t1[i];
  float t = t_start, t_near = t_start, t_far = t_start;
  bool first = true;
  for (int it = 0; t < t_exit && it < OCC_MAX_ITERS; ++it) {
    const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
    const int v = voxel_of(px, py, pz, n, e);
    if (!voxel_ok(v, n)) break;
    const bool in = roi[v] && occ[v];
    if (in && first) {
      t_near = t;
      first = false;
    }
    t += next_voxel_dist(px, py, pz, dx, dy, dz, n, e);
    if (in) t_far = clampf(t, t_start, t_exit);
  }
  near[i] = t_near;
  far[i] = t_far;
}

// :505-581: one sample per ray at the first occupied voxel (sphere tracing start)
__global__ void first_sample_kernel(RayIn r, int count, int n, Extent e,
                                    const unsigned char* __restrict__ occ,
                                    const unsigned char* __restrict__ roi, float* __restrict__ s3d,
                                    float* __restrict__ sdirs, float* __restrict__ sz,
                                    float* __restrict__ sdt, int* __restrict__ start_end) {
  OCC_THREAD(i, count);
  const float ox = r.o[3 * i], oy = r.o[3 * i + 1], oz = r.o[3 * i + 2];
  const float dx = r.d[3 * i], dy = r.d[3 * i + 1], dz = r.d[3 * i + 2];
  const float t_exit = r.t1[i];
  float t = r.t0[i];
  for (int steps = 0; t < t_exit && steps < OCC_MAX_ITERS; ++steps) {   // (the reference's MAX_STEPS() counter is never advanced)
    const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
    const int v = voxel_of(px, py, pz, n, e);
    if (!voxel_ok(v, n)) break;
    t += next_voxel_dist(px, py, pz, dx, dy, dz, n, e);
    t += 1e-6f;
    if (roi[v] && occ[v]) {
      start_end[2 * i] = (int)i, start_end[2 * i + 1] = (int)i + 1;
      s3d[3 * i] = px, s3d[3 * i + 1] = py, s3d[3 * i + 2] = pz;
      sdirs[3 * i] = dx, sdirs[3 * i + 1] = dy, sdirs[3 * i + 2] = dz;
      sz[i] = t;
      sdt[i] = 0.f;
      return;
    }
  }
  start_end[2 * i] = 0, start_end[2 * i + 1] = 0;
}

// :415-503
__global__ void advance_samples_kernel(const float* __restrict__ dirs, const float* __restrict__ pts,
                                       int count, int n, Extent e,
                                       const unsigned char* __restrict__ occ,
                                       const unsigned char* __restrict__ roi, float* __restrict__ out,
                                       unsigned char* __restrict__ within) {
  OCC_THREAD(i, count);
  const float ox = pts[3 * i], oy = pts[3 * i + 1], oz = pts[3 * i + 2];
  const float dx = dirs[3 * i], dy = dirs[3 * i + 1], dz = dirs[3 * i + 2];
  float prec_t = 0.f, t = 0.f;
  bool inside = true;
  for (int it = 0; inside && it < OCC_MAX_ITERS; ++it) {
    const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
    const int v = voxel_of(px, py, pz, n, e);
    if (!voxel_ok(v, n)) {
      inside = false;
      out[3 * i] = ox + prec_t * dx, out[3 * i + 1] = oy + prec_t * dy, out[3 * i + 2] = oz + prec_t * dz;
    } else {
      prec_t = t;
      t += next_voxel_dist(px, py, pz, dx, dy, dz, n, e);
      t += 1e-6f;
      if (roi[v] && occ[v]) {
        out[3 * i] = px, out[3 * i + 1] = py, out[3 * i + 2] = pz;
        break;
      }
    }
  }
  within[i] = inside;
}

// RaySamplerGPU.cuh:275-457: pass 1 measures the length of the ray inside occupied voxels,
// pass 2 drops equidistant samples (in occupied-space arc length) along it.
__global__ void sample_fg_occupied_kernel(RayIn r, int count, float min_dist, int min_nr, int max_nr,
                                          int jitter, unsigned long long rng_state,
                                          unsigned long long rng_inc, int n, Extent e,
                                          const unsigned char* __restrict__ occ,
                                          const unsigned char* __restrict__ roi,
                                          float* __restrict__ ray_max_dt, int* __restrict__ samples_idx,
                                          float* __restrict__ s3d, float* __restrict__ sdirs,
                                          float* __restrict__ sz, int* __restrict__ start_end) {
  OCC_THREAD(i, count);
  const float eps = 1e-6f;
  const float ox = r.o[3 * i], oy = r.o[3 * i + 1], oz = r.o[3 * i + 2];
  const float dx = r.d[3 * i], dy = r.d[3 * i + 1], dz = r.d[3 * i + 2];
  const float t_start = r.t0[i], t_exit = r.t1[i];
  float t = t_start, step = 0.f, occupied = 0.f;
  for (int it = 0; t < t_exit && it < OCC_MAX_ITERS; ++it) {
    const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
    const int v = voxel_of(px, py, pz, n, e);
    if (!voxel_ok(v, n)) break;
    if (roi[v] && occ[v]) occupied += step;      // (the reference adds the PREVIOUS step)
    step = next_voxel_dist(px, py, pz, dx, dy, dz, n, e);
    t += step;
  }
  occupied = clampf(occupied, 0.0f, t_exit - t_start);
  int to_create = 0;
  float spacing = 0.f;
  if (occupied > 0.0f) {
    if (occupied > min_dist) {
      to_create = (int)(occupied / min_dist);
      to_create = min(max(to_create, 0), max_nr);
      spacing = occupied / (float)to_create;
    } else {
      to_create = 1;
      spacing = occupied;
    }
  }
  int created = 0;
  const long long first = i * max_nr;
  if (to_create > 0 && to_create >= min_nr) {
    float to_next = 0.f;
    t = t_start;
    if (jitter) {
      Pcg32 rng{rng_state, rng_inc};
      rng.advance((unsigned long long)i);
      to_next = spacing * rng.next_float();
    }
    for (int it = 0; t < t_exit && it < OCC_MAX_ITERS; ++it) {
      t = clampf(t, t_start, t_exit);
      const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
      if (created >= to_create) break;
      const int v = voxel_of(px, py, pz, n, e);
      if (!voxel_ok(v, n)) break;
      const bool in = roi[v] && occ[v];
      if (in && to_next == 0.0f) {
        const long long s = first + created;
        s3d[3 * s] = px, s3d[3 * s + 1] = py, s3d[3 * s + 2] = pz;
        sdirs[3 * s] = dx, sdirs[3 * s + 1] = dy, sdirs[3 * s + 2] = dz;
        sz[s] = t;
        created += 1;
        to_next = spacing;
      }
      const float to_voxel = next_voxel_dist(px, py, pz, dx, dy, dz, n, e);
      float adv = to_voxel;
      if (in) {
        adv = fminf(to_voxel, to_next);
        to_next -= adv;
        if (to_next <= eps) to_next = 0.0f;
      }
      t += adv;
    }
  }
  if (created < min_nr) {
    created = 0;
  } else {
    ray_max_dt[i] = spacing;
    start_end[2 * i] = (int)first, start_end[2 * i + 1] = (int)first + created;
  }
  for (int k = created; k < max_nr; ++k) samples_idx[first + k] = -1;
}

bool grid_ok(int n) { return n >= 2 && n <= 1024 && (n & (n - 1)) == 0; }

#define OCC_CHECK(cond) \
  if (!(cond)) return VSA_ERR_ARG
#define OCC_LAUNCH(kernel, count, ...)                                                             \
  do {                                                                                             \
    if ((count) == 0) return VSA_OK;                                                               \
    hipLaunchKernelGGL(kernel, dim3(vsa_div_up((count), 256)), dim3(256), 0, (hipStream_t)stream,  \
                       __VA_ARGS__);                                                               \
    VSA_RETURN_LAUNCH_STATUS();                                                                    \
  } while (0)

}  // namespace

extern "C" int vsa_occ_grid_points(const int32_t* point_indices, int nr_points, int nr_voxels_per_dim,
                                   float extent_x, float extent_y, float extent_z, int centre_of_voxel,
                                   int jitter, uint64_t rng_state, uint64_t rng_inc, float* out_points,
                                   void* stream) {
  OCC_CHECK(nr_points >= 0 && grid_ok(nr_voxels_per_dim) && (nr_points == 0 || out_points));
  OCC_LAUNCH(grid_points_kernel, nr_points, point_indices, nr_points, nr_voxels_per_dim,
             Extent{extent_x, extent_y, extent_z}, centre_of_voxel, jitter, (unsigned long long)rng_state,
             (unsigned long long)rng_inc, out_points);
}

extern "C" int vsa_occ_update_values(const int32_t* point_indices, const float* values, int nr_points,
                                     float decay, float* grid_values, void* stream) {
  OCC_CHECK(nr_points >= 0 && decay <= 1.0f && (nr_points == 0 || (point_indices && values && grid_values)));
  OCC_LAUNCH(update_values_kernel, nr_points, point_indices, values, nr_points, decay, grid_values);
}

extern "C" int vsa_occ_update_occupancy_density(const int32_t* point_indices, int nr_points,
                                                int nr_voxels_per_dim, float extent_x, float extent_y,
                                                float extent_z, float occupancy_thresh,
                                                int check_neighbours, const float* grid_values,
                                                uint8_t* grid_occupancy, void* stream) {
  OCC_CHECK(nr_points >= 0 && grid_ok(nr_voxels_per_dim) &&
            (nr_points == 0 || (point_indices && grid_values && grid_occupancy)));
  OCC_LAUNCH(occupancy_from_density_kernel, nr_points, point_indices, nr_points, nr_voxels_per_dim,
             Extent{extent_x, extent_y, extent_z}, occupancy_thresh, check_neighbours, grid_values,
             grid_occupancy);
}

extern "C" int vsa_occ_update_occupancy_sdf(const int32_t* point_indices, const float* logistic_beta,
                                            int nr_points, int nr_voxels_per_dim, float extent_x,
                                            float extent_y, float extent_z, float occupancy_thresh,
                                            const float* grid_values, uint8_t* grid_occupancy,
                                            void* stream) {
  OCC_CHECK(nr_points >= 0 && grid_ok(nr_voxels_per_dim) &&
            (nr_points == 0 || (point_indices && logistic_beta && grid_values && grid_occupancy)));
  OCC_LAUNCH(occupancy_from_sdf_kernel, nr_points, point_indices, logistic_beta, nr_points,
             nr_voxels_per_dim, Extent{extent_x, extent_y, extent_z}, occupancy_thresh, grid_values,
             grid_occupancy);
}

extern "C" int vsa_occ_check(const float* points, int nr_points, int nr_voxels_per_dim, float extent_x,
                             float extent_y, float extent_z, const float* grid_values,
                             const uint8_t* grid_occupancy, const uint8_t* grid_roi,
                             uint8_t* out_occupancy, float* out_values, void* stream) {
  OCC_CHECK(nr_points >= 0 && grid_ok(nr_voxels_per_dim) &&
            (nr_points == 0 || (points && grid_values && grid_occupancy && grid_roi && out_occupancy && out_values)));
  OCC_LAUNCH(check_occupancy_kernel, nr_points, points, nr_points, nr_voxels_per_dim,
             Extent{extent_x, extent_y, extent_z}, grid_values, grid_occupancy, grid_roi, out_occupancy,
             out_values);
}

extern "C" int vsa_occ_rays_t_near_t_far(const float* rays_o, const float* rays_d,
                                         const float* ray_t_entry, const float* ray_t_exit, int nr_rays,
                                         int nr_voxels_per_dim, float extent_x, float extent_y,
                                         float extent_z, const uint8_t* grid_occupancy,
                                         const uint8_t* grid_roi, float* out_t_near, float* out_t_far,
                                         void* stream) {
  OCC_CHECK(nr_rays >= 0 && grid_ok(nr_voxels_per_dim) &&
            (nr_rays == 0 || (rays_o && rays_d && ray_t_entry && ray_t_exit && grid_occupancy && grid_roi &&
                              out_t_near && out_t_far)));
  OCC_LAUNCH(rays_near_far_kernel, nr_rays, RayIn{rays_o, rays_d, ray_t_entry, ray_t_exit}, nr_rays,
             nr_voxels_per_dim, Extent{extent_x, extent_y, extent_z}, grid_occupancy, grid_roi, out_t_near,
             out_t_far);
}

extern "C" int vsa_occ_first_sample(const float* rays_o, const float* rays_d, const float* ray_t_entry,
                                    const float* ray_t_exit, int nr_rays, int nr_voxels_per_dim,
                                    float extent_x, float extent_y, float extent_z,
                                    const uint8_t* grid_occupancy, const uint8_t* grid_roi,
                                    float* samples_3d, float* samples_dirs, float* samples_z,
                                    float* samples_dt, int32_t* ray_start_end_idx, void* stream) {
  OCC_CHECK(nr_rays >= 0 && grid_ok(nr_voxels_per_dim) &&
            (nr_rays == 0 || (rays_o && rays_d && ray_t_entry && ray_t_exit && grid_occupancy && grid_roi &&
                              samples_3d && samples_dirs && samples_z && samples_dt && ray_start_end_idx)));
  OCC_LAUNCH(first_sample_kernel, nr_rays, RayIn{rays_o, rays_d, ray_t_entry, ray_t_exit}, nr_rays,
             nr_voxels_per_dim, Extent{extent_x, extent_y, extent_z}, grid_occupancy, grid_roi, samples_3d,
             samples_dirs, samples_z, samples_dt, ray_start_end_idx);
}

extern "C" int vsa_occ_advance_samples(const float* samples_dirs, const float* samples_3d, int nr_points,
                                       int nr_voxels_per_dim, float extent_x, float extent_y,
                                       float extent_z, const uint8_t* grid_occupancy,
                                       const uint8_t* grid_roi, float* new_samples_3d,
                                       uint8_t* is_within_bounds, void* stream) {
  OCC_CHECK(nr_points >= 0 && grid_ok(nr_voxels_per_dim) &&
            (nr_points == 0 || (samples_dirs && samples_3d && grid_occupancy && grid_roi && new_samples_3d &&
                                is_within_bounds)));
  OCC_LAUNCH(advance_samples_kernel, nr_points, samples_dirs, samples_3d, nr_points, nr_voxels_per_dim,
             Extent{extent_x, extent_y, extent_z}, grid_occupancy, grid_roi, new_samples_3d,
             is_within_bounds);
}

extern "C" int vsa_sample_fg_occupied(const float* rays_o, const float* rays_d, const float* ray_t_entry,
                                      const float* ray_t_exit, float min_dist_between_samples,
                                      int min_nr_samples_per_ray, int max_nr_samples_per_ray,
                                      int jitter_samples, uint64_t rng_state, uint64_t rng_inc,
                                      int nr_voxels_per_dim, float extent_x, float extent_y,
                                      float extent_z, const uint8_t* grid_occupancy,
                                      const uint8_t* grid_roi, float* ray_max_dt, int32_t* samples_idx,
                                      float* samples_3d, float* samples_dirs, float* samples_z,
                                      int32_t* ray_start_end_idx, int nr_rays, void* stream) {
  OCC_CHECK(nr_rays >= 0 && max_nr_samples_per_ray >= 0 && grid_ok(nr_voxels_per_dim) &&
            (nr_rays == 0 || (rays_o && rays_d && ray_t_entry && ray_t_exit && grid_occupancy && grid_roi &&
                              ray_max_dt && samples_idx && samples_3d && samples_dirs && samples_z &&
                              ray_start_end_idx)));
  OCC_LAUNCH(sample_fg_occupied_kernel, nr_rays, RayIn{rays_o, rays_d, ray_t_entry, ray_t_exit}, nr_rays,
             min_dist_between_samples, min_nr_samples_per_ray, max_nr_samples_per_ray, jitter_samples,
             (unsigned long long)rng_state, (unsigned long long)rng_inc, nr_voxels_per_dim,
             Extent{extent_x, extent_y, extent_z}, grid_occupancy, grid_roi, ray_max_dt, samples_idx,
             samples_3d, samples_dirs, samples_z, ray_start_end_idx);
}
