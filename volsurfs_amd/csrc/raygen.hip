// Ray generation and the training-ray sampler: the step before the
// traversal in both entry points — `get_camera_rays(camera, nr_rays_per_pixel, jitter_pixels)`
// for full views (methods/base_method.py:389-394, renderers/base_renderer.py:59) and
// `TensorReel.get_next_rays_batch(batch_size, jitter_pixels, nr_rays_per_pixel)` for training
// batches (trainer.py:176-190).  Both live in mvdatasets, an empty submodule in the reference
// checkout (.gitmodules:10-13): PARITY UNPINNED.  The call shapes are the reference's; the
// arithmetic is this library's own pinhole definition, restated in oracle/raygen.py:
//   point (x, y) in pixels, pixel centre = +0.5 (or + two PCG32 draws when jittered)
//   d_cam = Kinv (x, y, 1);  d = normalise(R d_cam);  o = t           (c2w = [R | t], 3x4)
// Every ray is independent and the camera is wave-uniform (scalar loads): one thread per ray
// (camera rays) or per sampled pixel (reel), writes coalesced, 40 B of output per ray — an
// HBM-write-bound kernel that costs microseconds next to the 3 ms render step.
#include "common.h"
#include "pcg32.h"

namespace {

struct RayOut {
  float ox, oy, oz, dx, dy, dz;
};

__device__ __forceinline__ RayOut pinhole_ray(const float* __restrict__ c2w,
                                              const float* __restrict__ kinv, float x, float y) {
  float dc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) dc[i] = (kinv[3 * i] * x + kinv[3 * i + 1] * y) + kinv[3 * i + 2];
  float d[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    d[i] = (c2w[4 * i] * dc[0] + c2w[4 * i + 1] * dc[1]) + c2w[4 * i + 2] * dc[2];
  const float n = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
  return RayOut{c2w[3], c2w[7], c2w[11], d[0] / n, d[1] / n, d[2] / n};
}

__device__ __forceinline__ void store_ray(const RayOut& r, float x, float y, long long i,
                                          float* __restrict__ rays_o, float* __restrict__ rays_d,
                                          float* __restrict__ points_2d) {
  rays_o[3 * i] = r.ox, rays_o[3 * i + 1] = r.oy, rays_o[3 * i + 2] = r.oz;
  rays_d[3 * i] = r.dx, rays_d[3 * i + 1] = r.dy, rays_d[3 * i + 2] = r.dz;
  if (points_2d) points_2d[2 * i] = x, points_2d[2 * i + 1] = y;
}

// ray i = pixel * R + s, pixel = row * W + col (the order BaseMethod.render averages
// nr_rays_per_pixel consecutive rays in)
__global__ __launch_bounds__(256) void camera_rays_kernel(
    const float* __restrict__ c2w, const float* __restrict__ kinv, int W, int R, int jitter,
    unsigned long long rng_state, unsigned long long rng_inc, long long n_rays,
    float* __restrict__ rays_o, float* __restrict__ rays_d, float* __restrict__ points_2d) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rays) return;
  const long long pixel = i / R;
  const int row = (int)(pixel / W), col = (int)(pixel % W);
  float jx = 0.5f, jy = 0.5f;
  if (jitter) {
    Pcg32 rng{rng_state, rng_inc};
    rng.advance(2ull * (unsigned long long)i);
    jx = rng.next_float();
    jy = rng.next_float();
  }
  const float x = (float)col + jx, y = (float)row + jy;
  store_ray(pinhole_ray(c2w, kinv, x, y), x, y, i, rays_o, rays_d, points_2d);
}

struct ReelDummy {
  float o[3], d[3], rgb[3];
};

// sample b: three draws pick (camera, col, row) uniformly; its R rays follow (jittered: two
// more draws each).  Ground truth is the picked pixel's value whatever the jitter.
__global__ __launch_bounds__(256) void reel_rays_kernel(
    const float* __restrict__ c2w_all, const float* __restrict__ kinv_all,
    const float* __restrict__ rgb_all, const float* __restrict__ mask_all, int C, int H, int W,
    int B, int R, int jitter, unsigned long long rng_state, unsigned long long rng_inc,
    int* __restrict__ camera_idx, float* __restrict__ rays_o, float* __restrict__ rays_d,
    float* __restrict__ gt_rgb, float* __restrict__ gt_mask, float* __restrict__ points_2d,
    const vsa_train_ctl* __restrict__ ctl, ReelDummy dummy) {
  const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
  if (ctl) {      // the graph-replayed iteration (vsa_reel_next_rays_batch_ctl): size and stream from the control block
    rng_state = ctl->rng_state;
    rng_inc = ctl->rng_inc;
    if (b >= ctl->capacity) return;
    if (b >= ctl->nr_rays) {          // a dummy ray: misses every shell, and no loss is taken on it
      camera_idx[b] = 0;
      if (gt_rgb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gt_rgb[3 * b + c] = dummy.rgb[c];
      }
      if (gt_mask) gt_mask[b] = 0.f;
      for (int s = 0; s < R; ++s) {
        const long long i = b * R + s;
#pragma unroll
        for (int c = 0; c < 3; ++c) rays_o[3 * i + c] = dummy.o[c], rays_d[3 * i + c] = dummy.d[c];
        if (points_2d) points_2d[2 * i] = points_2d[2 * i + 1] = 0.f;
      }
      return;
    }
  } else if (b >= B) {
    return;
  }
  Pcg32 rng{rng_state, rng_inc};
  rng.advance((unsigned long long)b * (unsigned long long)(3 + (jitter ? 2 * R : 0)));
  const int cam = min((int)(rng.next_float() * (float)C), C - 1);
  const int col = min((int)(rng.next_float() * (float)W), W - 1);
  const int row = min((int)(rng.next_float() * (float)H), H - 1);
  camera_idx[b] = cam;
  const long long px = ((long long)cam * H + row) * W + col;
  if (gt_rgb) {
#pragma unroll
    for (int c = 0; c < 3; ++c) gt_rgb[3 * b + c] = rgb_all[3 * px + c];
  }
  if (gt_mask) gt_mask[b] = mask_all[px];
  const float* c2w = c2w_all + 12ll * cam;
  const float* kinv = kinv_all + 9ll * cam;
  for (int s = 0; s < R; ++s) {
    float jx = 0.5f, jy = 0.5f;
    if (jitter) {
      jx = rng.next_float();
      jy = rng.next_float();
    }
    const float x = (float)col + jx, y = (float)row + jy;
    store_ray(pinhole_ray(c2w, kinv, x, y), x, y, b * R + s, rays_o, rays_d, points_2d);
  }
}

// Pixel-tile order.  Full-frame rays arrive row-major (a 64-lane wave = a 64x1 strip of
// pixels); in 8x8-tile-major order a wave is a square patch, whose rays visit fewer distinct
// BVH nodes / texel lines (measured on the 800x800 bench frame: trace -10 %, shade_fwd -12 %,
// shade_bwd -7 %).  element i of the tile-major array = pixel (8 ty + j / 8, 8 tx + j % 8),
// t = i / 64 = ty * (W / 8) + tx, j = i % 64.
// Within a tile the pixels run boustrophedon (row 0 left to right, row 1 right to left, ...): consecutive
// elements are always NEIGHBOURING pixels, also across the end of a pixel row — the shading backward keeps the
// gradient lines of the previous hit's texel footprint open, and with raster rows every row end dropped all of them.
// (nt_shade_bwd 0.348 -> 0.312 ms on the K=5 800x800 frame; neutral at 1080p K=7 — profiles/NOTEBOOK.md.)
__device__ __forceinline__ int tile_x_of(int j) { return (j & 8) ? 7 - (j & 7) : (j & 7); }

struct __attribute__((packed, aligned(4))) F3 {      // a [.,3] f32 record: 4-byte aligned, moved as one dwordx3
  float x, y, z;
};

template <int C>
__global__ __launch_bounds__(256) void tile_order_kernel(const float* __restrict__ src,
                                                         float* __restrict__ dst, int W, long long n,
                                                         int inverse) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int tiles_x = W >> 3;
  const long long t = i >> 6;
  const int j = (int)(i & 63);
  const long long row = (t / tiles_x) * 8 + (j >> 3), col = (t % tiles_x) * 8 + tile_x_of(j);
  const long long px = row * W + col;
  const long long from = inverse ? i : px, to = inverse ? px : i;
  if constexpr (C == 3) {     // one 12-byte access each way (global_load / store_dwordx3) instead of three dwords
    *reinterpret_cast<F3*>(dst + to * 3) = *reinterpret_cast<const F3*>(src + from * 3);
  } else {
#pragma unroll
    for (int c = 0; c < C; ++c) dst[to * C + c] = src[from * C + c];
  }
}

// the three per-ray inputs of a training frame in one pass
__global__ __launch_bounds__(256) void tile_order_rays_kernel(
    const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ gt,
    float* __restrict__ o_t, float* __restrict__ d_t, float* __restrict__ gt_t, int W, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int tiles_x = W >> 3;
  const long long t = i >> 6;
  const int j = (int)(i & 63);
  const long long px = ((t / tiles_x) * 8 + (j >> 3)) * W + (t % tiles_x) * 8 + tile_x_of(j);
  const F3 vo = *reinterpret_cast<const F3*>(o + 3 * px), vd = *reinterpret_cast<const F3*>(d + 3 * px);
  F3 vg = {0.f, 0.f, 0.f};
  if (gt) vg = *reinterpret_cast<const F3*>(gt + 3 * px);
  *reinterpret_cast<F3*>(o_t + 3 * i) = vo;
  *reinterpret_cast<F3*>(d_t + 3 * i) = vd;
  if (gt) *reinterpret_cast<F3*>(gt_t + 3 * i) = vg;
}

}  // namespace

extern "C" int vsa_tile_order_rays(const float* rays_o, const float* rays_d, const float* gt_rgb,
                                   float* rays_o_tiled, float* rays_d_tiled, float* gt_rgb_tiled,
                                   int height, int width, void* stream) {
  if (height < 0 || width < 0 || (height & 7) || (width & 7)) return VSA_ERR_ARG;
  const long long n = (long long)height * width;
  if (n == 0) return VSA_OK;
  if (!rays_o || !rays_d || !rays_o_tiled || !rays_d_tiled || (gt_rgb && !gt_rgb_tiled)) return VSA_ERR_ARG;
  hipLaunchKernelGGL(tile_order_rays_kernel, dim3(vsa_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     rays_o, rays_d, gt_rgb, rays_o_tiled, rays_d_tiled, gt_rgb_tiled, width, n);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_tile_order(const float* src, float* dst, int height, int width, int channels,
                              int inverse, void* stream) {
  if (height < 0 || width < 0 || (height & 7) || (width & 7)) return VSA_ERR_ARG;
  const long long n = (long long)height * width;
  if (n == 0) return VSA_OK;
  if (!src || !dst || src == dst) return VSA_ERR_ARG;
  const dim3 grid(vsa_div_up(n, 256)), block(256);
  switch (channels) {
    case 1: hipLaunchKernelGGL(tile_order_kernel<1>, grid, block, 0, (hipStream_t)stream, src, dst, width, n, inverse); break;
    case 2: hipLaunchKernelGGL(tile_order_kernel<2>, grid, block, 0, (hipStream_t)stream, src, dst, width, n, inverse); break;
    case 3: hipLaunchKernelGGL(tile_order_kernel<3>, grid, block, 0, (hipStream_t)stream, src, dst, width, n, inverse); break;
    case 4: hipLaunchKernelGGL(tile_order_kernel<4>, grid, block, 0, (hipStream_t)stream, src, dst, width, n, inverse); break;
    default: return VSA_ERR_UNSUPPORTED;
  }
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_camera_rays(const float* c2w, const float* intrinsics_inv, int height, int width,
                               int nr_rays_per_pixel, int jitter_pixels, uint64_t rng_state,
                               uint64_t rng_inc, float* rays_o, float* rays_d, float* points_2d,
                               void* stream) {
  if (height < 0 || width < 0 || nr_rays_per_pixel < 1) return VSA_ERR_ARG;
  const long long n = (long long)height * width * nr_rays_per_pixel;
  if (n == 0) return VSA_OK;
  if (!c2w || !intrinsics_inv || !rays_o || !rays_d) return VSA_ERR_ARG;
  if (n > 0x7fffffffll * 256) return VSA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(camera_rays_kernel, dim3(vsa_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     c2w, intrinsics_inv, width, nr_rays_per_pixel, jitter_pixels,
                     (unsigned long long)rng_state, (unsigned long long)rng_inc, n, rays_o, rays_d,
                     points_2d);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_reel_next_rays_batch(const float* c2w_all, const float* intrinsics_inv_all,
                                        const float* rgb_all, const float* mask_all, int nr_cameras,
                                        int height, int width, int batch_size, int nr_rays_per_pixel,
                                        int jitter_pixels, uint64_t rng_state, uint64_t rng_inc,
                                        int32_t* camera_idx, float* rays_o, float* rays_d,
                                        float* gt_rgb, float* gt_mask, float* points_2d,
                                        void* stream) {
  if (batch_size < 0 || nr_rays_per_pixel < 1) return VSA_ERR_ARG;
  if (batch_size == 0) return VSA_OK;
  if (nr_cameras < 1 || height < 1 || width < 1) return VSA_ERR_ARG;
  if (!c2w_all || !intrinsics_inv_all || !camera_idx || !rays_o || !rays_d) return VSA_ERR_ARG;
  if ((gt_rgb && !rgb_all) || (gt_mask && !mask_all)) return VSA_ERR_ARG;
  hipLaunchKernelGGL(reel_rays_kernel, dim3(vsa_div_up(batch_size, 256)), dim3(256), 0,
                     (hipStream_t)stream, c2w_all, intrinsics_inv_all, rgb_all, mask_all, nr_cameras,
                     height, width, batch_size, nr_rays_per_pixel, jitter_pixels,
                     (unsigned long long)rng_state, (unsigned long long)rng_inc, camera_idx, rays_o,
                     rays_d, gt_rgb, gt_mask, points_2d, (const vsa_train_ctl*)nullptr, ReelDummy{});
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_reel_next_rays_batch_ctl(const float* c2w_all, const float* intrinsics_inv_all,
                                            const float* rgb_all, const float* mask_all, int nr_cameras, int height,
                                            int width, int capacity, int nr_rays_per_pixel, int jitter_pixels,
                                            const vsa_train_ctl* ctl, const float* dummy_o, const float* dummy_d,
                                            const float* dummy_rgb, int32_t* camera_idx, float* rays_o,
                                            float* rays_d, float* gt_rgb, float* gt_mask, float* points_2d,
                                            void* stream) {
  if (capacity < 1 || nr_rays_per_pixel < 1 || nr_cameras < 1 || height < 1 || width < 1) return VSA_ERR_ARG;
  if (!ctl || !dummy_o || !dummy_d || !dummy_rgb || !c2w_all || !intrinsics_inv_all || !camera_idx || !rays_o || !rays_d)
    return VSA_ERR_ARG;
  if ((gt_rgb && !rgb_all) || (gt_mask && !mask_all)) return VSA_ERR_ARG;
  ReelDummy dm;
  for (int c = 0; c < 3; ++c) dm.o[c] = dummy_o[c], dm.d[c] = dummy_d[c], dm.rgb[c] = dummy_rgb[c];
  // (`capacity` sizes the grid: the host's copy of ctl->capacity — the kernel trusts the device's)
  hipLaunchKernelGGL(reel_rays_kernel, dim3(vsa_div_up(capacity, 256)), dim3(256), 0, (hipStream_t)stream, c2w_all,
                     intrinsics_inv_all, rgb_all, mask_all, nr_cameras, height, width, capacity, nr_rays_per_pixel,
                     jitter_pixels, 0ull, 0ull, camera_idx, rays_o, rays_d, gt_rgb, gt_mask, points_2d, ctl, dm);
  VSA_RETURN_LAUNCH_STATUS();
}

// ---- A1: bounding-primitive intersection (utils/raycasting.py:4-36 -> mvdatasets BoundingBox /
// BoundingSphere .intersect, absent: this library's definition, restated in oracle/raygen.py).
// kind 0: axis-aligned cube of half side `size` centred at the origin (slab test);
// kind 1: sphere of radius `size` centred at the origin.  Misses report t = 0 and the points at
// the ray origin.  Fixed evaluation order (no contraction) so that the oracle matches bit for bit.
namespace {
__global__ void intersect_primitive_kernel(const float* __restrict__ rays_o,
                                           const float* __restrict__ rays_d, int N, int kind,
                                           float size, unsigned char* __restrict__ is_hit,
                                           float* __restrict__ t_near, float* __restrict__ t_far,
                                           float* __restrict__ p_near, float* __restrict__ p_far) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float o[3] = {rays_o[3 * n], rays_o[3 * n + 1], rays_o[3 * n + 2]};
  const float d[3] = {rays_d[3 * n], rays_d[3 * n + 1], rays_d[3 * n + 2]};
  float tn, tf;
  bool hit;
  if (kind == 0) {
    tn = -INFINITY;
    tf = INFINITY;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float inv = 1.0f / d[i];
      const float a = (-size - o[i]) * inv, b = (size - o[i]) * inv;
      tn = fmaxf(tn, fminf(a, b));
      tf = fminf(tf, fmaxf(a, b));
    }
    hit = tn <= tf && tf > 0.0f;
  } else {
    const float a = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
    const float b = 2.0f * ((o[0] * d[0] + o[1] * d[1]) + o[2] * d[2]);
    const float c = ((o[0] * o[0] + o[1] * o[1]) + o[2] * o[2]) - size * size;
    const float disc = b * b - (4.0f * a) * c;
    const float sq = sqrtf(fmaxf(disc, 0.0f));
    tn = (-b - sq) / (2.0f * a);
    tf = (-b + sq) / (2.0f * a);
    hit = disc >= 0.0f && tf > 0.0f;
  }
  tn = hit ? fmaxf(tn, 0.0f) : 0.0f;
  tf = hit ? tf : 0.0f;
  is_hit[n] = hit ? 1 : 0;
  t_near[n] = tn;
  t_far[n] = tf;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (p_near) p_near[3 * n + i] = o[i] + tn * d[i];
    if (p_far) p_far[3 * n + i] = o[i] + tf * d[i];
  }
}
}  // namespace

extern "C" int vsa_intersect_primitive(const float* rays_o, const float* rays_d, int nr_rays,
                                       int kind, float size, uint8_t* is_hit, float* t_near,
                                       float* t_far, float* points_near, float* points_far,
                                       void* stream) {
  if (nr_rays < 0 || (kind != 0 && kind != 1) || !(size > 0.0f)) return VSA_ERR_ARG;
  if (nr_rays == 0) return VSA_OK;
  if (!rays_o || !rays_d || !is_hit || !t_near || !t_far) return VSA_ERR_ARG;
  hipLaunchKernelGGL(intersect_primitive_kernel, dim3(vsa_div_up(nr_rays, 256)), dim3(256), 0,
                     (hipStream_t)stream, rays_o, rays_d, nr_rays, kind, size, is_hit, t_near, t_far,
                     points_near, points_far);
  VSA_RETURN_LAUNCH_STATUS();
}
