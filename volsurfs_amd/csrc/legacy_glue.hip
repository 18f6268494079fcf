// The glue of the legacy appearance branch's training step (BASELINE configs[2]) around its encoders and MLPs,
// /root/reference/volsurfs_py/methods/volsurfs.py:486-599: per hit the shaded point, the view direction and the face
// normal (:504-520), and behind the models the sigmoid, the alpha decay (:583-596) and the dense [N,K] scatter
// (:521-599).  As torch expressions these were 18 + 21 launches of 2-5 us per iteration (and as many autograd nodes) in
// a loop whose GPU sat idle 0.8 of every 3.0 ms waiting for the host (profiles/NOTEBOOK.md, round 5 timeline).
// One launch each way here; same arithmetic, op by op (fp32, no contraction).
#include "common.h"

namespace {

constexpr int LG_BLOCK = 256;

// hit_shell / hit_ray [M] i64 = (shell, ray) of every hit, sorted by shell then ray (the two columns of torch.nonzero,
// which lays its result out column by column)
__global__ __launch_bounds__(LG_BLOCK) void legacy_hit_prep_kernel(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ hit_t,
    const int* __restrict__ hit_slot, const float4* __restrict__ tris, const long long* __restrict__ hit_shell,
    const long long* __restrict__ hit_ray, int M, int N,
    float* __restrict__ pts, float* __restrict__ dirs, float* __restrict__ nrm) {
  const int i = blockIdx.x * LG_BLOCK + threadIdx.x;
  if (i >= M) return;
  const long long s = hit_shell[i], r = hit_ray[i];
  const long long o = s * N + r;
  const int slot = hit_slot[o];
  const float t = hit_t[o];
  const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz = rays_d[3 * r + 2];
  // pts = rays_o + t * d (a product, then a sum: volsurfs.py:507)
  pts[3 * i] = rays_o[3 * r] + t * dx;
  pts[3 * i + 1] = rays_o[3 * r + 1] + t * dy;
  pts[3 * i + 2] = rays_o[3 * r + 2] + t * dz;
  dirs[3 * i] = dx, dirs[3 * i + 1] = dy, dirs[3 * i + 2] = dz;
  // face normal = normalize(cross(e1, e2)) (F.normalize: v / max(|v|, 1e-12))
  const float4 e1 = tris[3 * (long long)slot + 1], e2 = tris[3 * (long long)slot + 2];
  const float cx = e1.y * e2.z - e1.z * e2.y, cy = e1.z * e2.x - e1.x * e2.z, cz = e1.x * e2.y - e1.y * e2.x;
  const float len = fmaxf(sqrtf(cx * cx + cy * cy + cz * cz), 1e-12f);
  nrm[3 * i] = cx / len, nrm[3 * i + 1] = cy / len, nrm[3 * i + 2] = cz / len;
}

__device__ __forceinline__ float lg_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// rows [0, M): colour = sigmoid(y_rgb[i][0..3)); rows [a0, M): alpha = sigmoid(y_alpha[i - a0][0]) (x decay), rows before
// a0 (a solid inner shell) and every row when y_alpha is null: alpha = 1.  Scatter to [N,K,.] at (ray, shell).
__global__ __launch_bounds__(LG_BLOCK) void legacy_shade_out_fwd_kernel(
    const float* __restrict__ y_rgb, int ld_rgb, const float* __restrict__ y_alpha, int ld_alpha, int a0,
    const long long* __restrict__ hit_shell, const long long* __restrict__ hit_ray, const float* __restrict__ dirs,
    const float* __restrict__ nrm, int M, int K,
    int with_decay, float* __restrict__ surfs_rgb, float* __restrict__ surfs_alpha, float* __restrict__ surfs_normals,
    float* __restrict__ sig_rgb, float* __restrict__ sig_alpha, float* __restrict__ decay_out) {
  const int i = blockIdx.x * LG_BLOCK + threadIdx.x;
  if (i >= M) return;
  const long long s = hit_shell[i], r = hit_ray[i];
  const long long e = r * K + s;
  float c[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    c[ch] = lg_sigmoid(y_rgb[(long long)i * ld_rgb + ch]);
    surfs_rgb[3 * e + ch] = c[ch];
    sig_rgb[3 * i + ch] = c[ch];
  }
  const float nx = nrm[3 * i], ny = nrm[3 * i + 1], nz = nrm[3 * i + 2];
  surfs_normals[3 * e] = nx, surfs_normals[3 * e + 1] = ny, surfs_normals[3 * e + 2] = nz;
  float a = 1.0f;
  if (y_alpha && i >= a0) {
    const float sa = lg_sigmoid(y_alpha[(long long)(i - a0) * ld_alpha]);
    float dec = 1.0f;
    if (with_decay) {     // decay = 2 sigmoid(10 clamp(-d.n, 0, 1)) - 1 (volsurfs.py:585-594)
      const float dot = fminf(fmaxf((-dirs[3 * i]) * nx + (-dirs[3 * i + 1]) * ny + (-dirs[3 * i + 2]) * nz, 0.0f), 1.0f);
      dec = lg_sigmoid(10.0f * dot) * 2.0f - 1.0f;
    }
    sig_alpha[i - a0] = sa;
    decay_out[i - a0] = dec;
    a = sa * dec;
  }
  surfs_alpha[e] = a;
}

// dy_rgb[i][ch] = g_surfs_rgb[ray, shell, ch] * s (1 - s) (channels >= 3 of a wider output: 0);
// dy_alpha[i - a0][0] = g_surfs_alpha[ray, shell] * decay * s (1 - s)
__global__ __launch_bounds__(LG_BLOCK) void legacy_shade_out_bwd_kernel(
    const float* __restrict__ g_surfs_rgb, const float* __restrict__ g_surfs_alpha,
    const long long* __restrict__ hit_shell, const long long* __restrict__ hit_ray,
    const float* __restrict__ sig_rgb, const float* __restrict__ sig_alpha, const float* __restrict__ decay, int M, int K,
    int a0, float* __restrict__ dy_rgb, int ld_rgb, float* __restrict__ dy_alpha, int ld_alpha) {
  const int i = blockIdx.x * LG_BLOCK + threadIdx.x;
  if (i >= M) return;
  const long long s = hit_shell[i], r = hit_ray[i];
  const long long e = r * K + s;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float sg = sig_rgb[3 * i + ch];
    dy_rgb[(long long)i * ld_rgb + ch] = (g_surfs_rgb[3 * e + ch] * (1.0f - sg)) * sg;     // at::sigmoid_backward's order
  }
  for (int ch = 3; ch < ld_rgb; ++ch) dy_rgb[(long long)i * ld_rgb + ch] = 0.0f;
  if (dy_alpha && i >= a0) {
    const float sa = sig_alpha[i - a0];
    const float g = g_surfs_alpha[e] * decay[i - a0];
    dy_alpha[(long long)(i - a0) * ld_alpha] = (g * (1.0f - sa)) * sa;
    for (int ch = 1; ch < ld_alpha; ++ch) dy_alpha[(long long)(i - a0) * ld_alpha + ch] = 0.0f;
  }
}

}  // namespace

extern "C" int vsa_legacy_hit_prep(const float* rays_o, const float* rays_d, const float* hit_t, const int32_t* hit_slot,
                                   const float* tris, const int64_t* hit_shell, const int64_t* hit_ray, int nr_hits,
                                   int nr_rays, float* pts,
                                   float* dirs, float* normals, void* stream) {
  if (nr_hits < 0 || nr_rays < 0) return VSA_ERR_ARG;
  if (nr_hits == 0) return VSA_OK;
  if (!rays_o || !rays_d || !hit_t || !hit_slot || !tris || !hit_shell || !hit_ray || !pts || !dirs || !normals)
    return VSA_ERR_ARG;
  hipLaunchKernelGGL(legacy_hit_prep_kernel, dim3(vsa_div_up(nr_hits, LG_BLOCK)), dim3(LG_BLOCK), 0, (hipStream_t)stream,
                     rays_o, rays_d, hit_t, hit_slot, reinterpret_cast<const float4*>(tris),
                     reinterpret_cast<const long long*>(hit_shell), reinterpret_cast<const long long*>(hit_ray), nr_hits,
                     nr_rays, pts, dirs, normals);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_legacy_shade_out_fwd(const float* y_rgb, int ld_rgb, const float* y_alpha, int ld_alpha,
                                        int alpha_first_row, const int64_t* hit_shell, const int64_t* hit_ray,
                                        const float* dirs, const float* normals,
                                        int nr_hits, int nr_rays, int nr_shells, int with_alpha_decay, float* surfs_rgb,
                                        float* surfs_alpha, float* surfs_normals, float* sig_rgb, float* sig_alpha,
                                        float* decay, void* stream) {
  if (nr_hits < 0 || nr_rays < 0 || nr_shells < 1 || nr_shells > VSA_MAX_SHELLS || ld_rgb < 3 || alpha_first_row < 0 ||
      (y_alpha && ld_alpha < 1))
    return VSA_ERR_ARG;
  if (!surfs_rgb || !surfs_alpha || !surfs_normals) return VSA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (nr_hits == 0) return VSA_OK;
  if (!y_rgb || !hit_shell || !hit_ray || !dirs || !normals || !sig_rgb || (y_alpha && (!sig_alpha || !decay)))
    return VSA_ERR_ARG;
  hipLaunchKernelGGL(legacy_shade_out_fwd_kernel, dim3(vsa_div_up(nr_hits, LG_BLOCK)), dim3(LG_BLOCK), 0, st, y_rgb,
                     ld_rgb, y_alpha, ld_alpha, alpha_first_row, reinterpret_cast<const long long*>(hit_shell),
                     reinterpret_cast<const long long*>(hit_ray), dirs, normals,
                     nr_hits, nr_shells, with_alpha_decay, surfs_rgb, surfs_alpha, surfs_normals, sig_rgb, sig_alpha,
                     decay);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_legacy_shade_out_bwd(const float* g_surfs_rgb, const float* g_surfs_alpha, const int64_t* hit_shell,
                                        const int64_t* hit_ray,
                                        const float* sig_rgb, const float* sig_alpha, const float* decay, int nr_hits,
                                        int nr_shells, int alpha_first_row, float* dy_rgb, int ld_rgb, float* dy_alpha,
                                        int ld_alpha, void* stream) {
  if (nr_hits < 0 || nr_shells < 1 || nr_shells > VSA_MAX_SHELLS || ld_rgb < 3 || alpha_first_row < 0 ||
      (dy_alpha && ld_alpha < 1))
    return VSA_ERR_ARG;
  if (nr_hits == 0) return VSA_OK;
  if (!g_surfs_rgb || !g_surfs_alpha || !hit_shell || !hit_ray || !sig_rgb || !dy_rgb ||
      (dy_alpha && (!sig_alpha || !decay)))
    return VSA_ERR_ARG;
  hipLaunchKernelGGL(legacy_shade_out_bwd_kernel, dim3(vsa_div_up(nr_hits, LG_BLOCK)), dim3(LG_BLOCK), 0,
                     (hipStream_t)stream, g_surfs_rgb, g_surfs_alpha, reinterpret_cast<const long long*>(hit_shell),
                     reinterpret_cast<const long long*>(hit_ray), sig_rgb,
                     sig_alpha, decay, nr_hits, nr_shells, alpha_first_row, dy_rgb, ld_rgb, dy_alpha, ld_alpha);
  VSA_RETURN_LAUNCH_STATUS();
}
