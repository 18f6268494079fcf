// The two scalar read-outs of a training iteration, one launch each and no fill in front of it:
//   vsa_count_hits — how many (shell, ray) pairs hit: the sample count the reference's dynamic ray
//                    count is steered by (/root/reference/volsurfs_py/trainer.py:293-304);
//   vsa_l1_mean    — mean |pred - gt|, the L1 image loss that is logged (utils/losses.py:14-19).
// As torch expressions ((hit_slot >= 0).sum(), (gt - pred).abs().mean()) they were 4 + 3 launches of
// 5-10 us each in an iteration of 0.97 ms (profiles/r05/README.md, timeline of the training loop).
//
// Both are the same two-level sum: a workgroup reduces its grid-stride share to one partial, publishes it and
// takes a ticket; the workgroup that draws the LAST ticket adds the partials in index order (a fixed tree: the
// result does not depend on which workgroup finishes when), writes the result and puts the ticket back to
// zero — the scratch a caller zeroes once serves every later call on the same stream.
#include "common.h"

namespace {

constexpr int RED_BLOCK = 256;
constexpr int RED_MAX_BLOCKS = 256;

struct RedScratch {
  unsigned ticket;
  unsigned pad[3];
  double partial[RED_MAX_BLOCKS];
};

__device__ __forceinline__ double block_sum(double v, double* s_part) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < RED_BLOCK / 64; ++w) r += s_part[w];
  }
  __syncthreads();
  return r;    // valid in thread 0
}

// `mine`: this thread's share of the sum; returns true in thread 0 of the last workgroup with the total in *total
__device__ __forceinline__ bool two_level_sum(double mine, RedScratch* sc, double* total) {
  __shared__ double s_part[RED_BLOCK / 64];
  __shared__ bool s_last;
  const double b = block_sum(mine, s_part);
  if (threadIdx.x == 0) {
    __hip_atomic_store(&sc->partial[blockIdx.x], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned t = __hip_atomic_fetch_add(&sc->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last) return false;
  double v = 0.0;
  if (threadIdx.x < gridDim.x)
    v = __hip_atomic_load(&sc->partial[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const double r = block_sum(v, s_part);
  if (threadIdx.x == 0) {
    __hip_atomic_store(&sc->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
    *total = r;
    return true;
  }
  return false;
}

__global__ __launch_bounds__(RED_BLOCK) void count_hits_kernel(const int* __restrict__ hit_slot, long long n,
                                                               RedScratch* sc, long long* __restrict__ out) {
  long long c = 0;
  const long long n4 = n >> 2;
  const int4* h4 = reinterpret_cast<const int4*>(hit_slot);
  for (long long i = blockIdx.x * (long long)RED_BLOCK + threadIdx.x; i < n4; i += (long long)gridDim.x * RED_BLOCK) {
    const int4 v = h4[i];
    c += (v.x >= 0) + (v.y >= 0) + (v.z >= 0) + (v.w >= 0);
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) c += hit_slot[(n4 << 2) + threadIdx.x] >= 0;
  double total;
  if (two_level_sum((double)c, sc, &total)) *out = (long long)total;     // counts < 2^53: exact in a double
}

__global__ __launch_bounds__(RED_BLOCK) void l1_mean_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            long long n, RedScratch* sc, float* __restrict__ out,
                                                            vsa_train_ctl* __restrict__ ctl) {
  if (ctl) {        // the graph-replayed iteration: the active elements and the result live in the control block
    n = 3ll * ctl->nr_rays;
    out = &ctl->loss;
  }
  double s = 0.0;
  const long long n4 = n >> 2;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  for (long long i = blockIdx.x * (long long)RED_BLOCK + threadIdx.x; i < n4; i += (long long)gridDim.x * RED_BLOCK) {
    const float4 x = a4[i], y = b4[i];
    s += (double)(fabsf(x.x - y.x) + fabsf(x.y - y.y)) + (double)(fabsf(x.z - y.z) + fabsf(x.w - y.w));
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    s += (double)fabsf(a[i] - b[i]);
  }
  double total;
  if (two_level_sum(s, sc, &total)) *out = (float)(total / (double)n);
}

int red_grid(long long n) {
  const long long per_block = (long long)RED_BLOCK * 4 * 4;      // four 16-byte loads per thread and array
  long long g = (n + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > RED_MAX_BLOCKS ? RED_MAX_BLOCKS : g));
}

}  // namespace

extern "C" long long vsa_reduce_scratch_bytes(void) { return (long long)sizeof(RedScratch); }

extern "C" int vsa_count_hits(const int32_t* hit_slot, long long n, void* scratch, int64_t* out, void* stream) {
  if (n < 0 || !scratch || !out || (n > 0 && !hit_slot)) return VSA_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(hit_slot) & 15) || (reinterpret_cast<uintptr_t>(scratch) & 7)) return VSA_ERR_ARG;
  hipLaunchKernelGGL(count_hits_kernel, dim3(red_grid(n)), dim3(RED_BLOCK), 0, (hipStream_t)stream, hit_slot, n,
                     static_cast<RedScratch*>(scratch), reinterpret_cast<long long*>(out));
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_l1_mean(const float* pred, const float* gt, long long n, void* scratch, float* out, void* stream) {
  if (n <= 0 || !pred || !gt || !scratch || !out) return VSA_ERR_ARG;
  if (((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(gt)) & 15) ||
      (reinterpret_cast<uintptr_t>(scratch) & 7))
    return VSA_ERR_ARG;
  hipLaunchKernelGGL(l1_mean_kernel, dim3(red_grid(n)), dim3(RED_BLOCK), 0, (hipStream_t)stream, pred, gt, n,
                     static_cast<RedScratch*>(scratch), out, (vsa_train_ctl*)nullptr);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_l1_mean_ctl(const float* pred, const float* gt, int capacity, void* scratch, vsa_train_ctl* ctl,
                               void* stream) {
  if (capacity < 1 || !pred || !gt || !scratch || !ctl) return VSA_ERR_ARG;
  if (((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(gt)) & 15) ||
      (reinterpret_cast<uintptr_t>(scratch) & 7))
    return VSA_ERR_ARG;
  hipLaunchKernelGGL(l1_mean_kernel, dim3(red_grid(3ll * capacity)), dim3(RED_BLOCK), 0, (hipStream_t)stream, pred, gt,
                     3ll * capacity, static_cast<RedScratch*>(scratch), (float*)nullptr, ctl);
  VSA_RETURN_LAUNCH_STATUS();
}
