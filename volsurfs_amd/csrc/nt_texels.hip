// Neural-texture step 1 + 2: per-hit uv, touched-texel marking, slot compaction.
//
// The reference evaluates its texture network at the 4 lerp corners of every
// hit (models/neural_texture.py:124-153); corners are texel centres, so hits
// that share a texel share the evaluation.  These kernels build the set of
// unique touched texels ("slots") per (shell, degree) without a sort and
// without host syncs: byte marks (plain stores, benign races) + a 3-pass
// exclusive scan that writes slot_of[texel] and texel_of_slot[slot].
#include "nt_common.h"

namespace {

__global__ void nt_mark_kernel(vsa_nt_plan plan, const int* __restrict__ hit_slot,
                               const float* __restrict__ hit_uv,
                               const float* __restrict__ face_uvs, int N,
                               float* __restrict__ tex_uv, unsigned char* __restrict__ marks) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y;
  if (n >= N) return;
  const long long o = (long long)s * N + n;
  const int slot = hit_slot[o];
  float2 uv = make_float2(0.f, 0.f);
  if (slot >= 0) {
    uv = nt_interp_uv(face_uvs + 6 * (long long)slot, hit_uv[2 * o], hit_uv[2 * o + 1]);
    const int D = marks ? max(plan.rgb_degrees, plan.alpha_degrees) : 0;   // marks == NULL: uv only
    for (int d = 0; d < D; ++d) {
      const int R = plan.tex_res[d];
      const int W = R + 2;
      const NtFootprint f = nt_footprint(uv.x, uv.y, R, plan.anchor != 0);
      unsigned char* m = marks + plan.dom_off[s * VSA_NT_MAX_DEG + d] +
                         (long long)(f.j0 + 1) * W + (f.i0 + 1);
      m[0] = 1;
      if (!plan.anchor) {      // lerp: the 2x2 footprint; anchor: the one texel
        m[1] = 1;
        m[W] = 1;
        m[W + 1] = 1;
      }
    }
  }
  tex_uv[2 * o] = uv.x;
  tex_uv[2 * o + 1] = uv.y;
}

__device__ __forceinline__ int count16(const uint4 v) {
  return __popc(v.x & 0x01010101u) + __popc(v.y & 0x01010101u) + __popc(v.z & 0x01010101u) +
         __popc(v.w & 0x01010101u);
}

// pass A: number of marked texels per 4096-texel block
__global__ __launch_bounds__(256) void nt_count_kernel(const uint4* __restrict__ marks,
                                                       int* __restrict__ block_count) {
  __shared__ int s_w[4];
  const uint4 v = marks[(long long)blockIdx.x * 256 + threadIdx.x];
  int c = count16(v);
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// One workgroup per persistent kernel: last launch's busy times -> shares of its cost axis.
// share' = share * clamp(mean / time, 0.8, 1.25) ^ 0.5 (damped: a workgroup's time is not linear in its
// share, and the times carry noise), floored at a quarter of an equal share, renormalised.
__device__ __forceinline__ void nt_rebalance_body(NtBalance* __restrict__ b, int k, float* s) {
  const int t = threadIdx.x;
  const int G = b->tick_wgs[k];
  if (G <= 1 || G > NT_BAL_MAX_WG) return;
  const bool have = b->frac_wgs[k] == G && b->frac[k][G] == (unsigned)NT_BAL_ONE;
  const float equal = (float)NT_BAL_ONE / (float)G;
  const float share = t < G ? (have ? (float)(b->frac[k][t + 1] - b->frac[k][t]) : equal) : 0.f;
#ifndef NT_BAL_GAIN
#define NT_BAL_GAIN 0.5f
#endif
#ifndef NT_BAL_EMA
#define NT_BAL_EMA 1.0f       /* weight of the newest time in the running mean kept in ema[] */
#endif
#ifndef NT_BAL_MASK
#define NT_BAL_MASK 0x30      /* kernels whose shares are corrected (bit = NT_BAL_* id): the two MLP kernels.
                                 Measured (profiles/r03/rebalance.txt): nt_mlp_bwd 0.68-0.70 -> 0.64-0.65 ms as a
                                 stage, nt_mlp_fwd unchanged; the encode kernels get SLOWER with their shares
                                 corrected (backward 0.61-0.63 -> 0.63-0.65 at any gain 0.15-0.5, with or without a
                                 running mean of the times): a piece's cost there is mostly its fixed part (table
                                 staging / plane flush), which a moved boundary duplicates instead of moving */
#endif
  if (!((NT_BAL_MASK >> k) & 1)) return;
  float time = t < G ? (float)b->ticks[k][t] : 0.f;
  if (NT_BAL_EMA < 1.0f && t < G) {
    const float prev = b->ema[k][t];
    time = have && prev > 0.f ? (1.0f - NT_BAL_EMA) * prev + NT_BAL_EMA * time : time;
    b->ema[k][t] = time;
  }
  auto scan = [&](float v) {          // inclusive, Hillis-Steele over the 1024 threads
    s[t] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const float u = t >= off ? s[t - off] : 0.f;
      __syncthreads();
      s[t] += u;
      __syncthreads();
    }
    const float r = s[t];
    __syncthreads();
    return r;
  };
  scan(time);
  const float mean = s[1023] / (float)G;
  __syncthreads();
  float w = 0.f;
  if (t < G) {
    const float r = time > 0.f ? fminf(fmaxf(mean / time, 0.8f), 1.25f) : 1.25f;
    w = fmaxf(share * powf(r, NT_BAL_GAIN), 0.25f * equal);
  }
  const float incl = scan(w);
  const float total = s[1023];
  if (t < G) b->frac[k][t + 1] = t == G - 1 ? (unsigned)NT_BAL_ONE : (unsigned)(incl / total * (float)NT_BAL_ONE);
  if (t == 0) {
    b->frac[k][0] = 0;
    b->frac_wgs[k] = G;
  }
}
__global__ __launch_bounds__(1024) void nt_rebalance_kernel(NtBalance* __restrict__ b) {
  __shared__ float s[1024];
  nt_rebalance_body(b, blockIdx.x, s);
}

// pass B: exclusive scan of the block counts (single workgroup), segment starts
__global__ __launch_bounds__(1024) void nt_scan_blocks_kernel(vsa_nt_plan plan,
                                                              int* __restrict__ block_count,
                                                              int nr_blocks,
                                                              int* __restrict__ seg_start) {
  __shared__ int s_part[1024];
  // workgroups 1 .. NT_BAL_KERNELS of the launch (frame loop with a balance buffer) turn the last
  // frame's workgroup times into shares meanwhile (a 7.8 us launch of its own before)
  if (blockIdx.x > 0) {
    nt_rebalance_body(static_cast<NtBalance*>(plan.balance), (int)blockIdx.x - 1, reinterpret_cast<float*>(s_part));
    return;
  }
  const int t = threadIdx.x;
  const int per = (nr_blocks + 1023) / 1024;
  const int b0 = t * per, b1 = min(nr_blocks, b0 + per);
  int sum = 0;
  for (int b = b0; b < b1; ++b) sum += block_count[b];
  s_part[t] = sum;
  __syncthreads();
  // Hillis-Steele inclusive scan over 1024 partials
  for (int off = 1; off < 1024; off <<= 1) {
    int v = t >= off ? s_part[t - off] : 0;
    __syncthreads();
    s_part[t] += v;
    __syncthreads();
  }
  int run = s_part[t] - sum;  // exclusive
  for (int b = b0; b < b1; ++b) {
    int c = block_count[b];
    block_count[b] = run;
    run += c;
  }
  __syncthreads();
  if (t == 0) block_count[nr_blocks] = s_part[1023];
  __threadfence_block();
  __syncthreads();
  const int nseg = plan.nr_shells * VSA_NT_MAX_DEG;
  for (int i = t; i <= nseg; i += 1024) {
    // block_count[] was rewritten by other lanes of this workgroup: bypass this CU's L1
    seg_start[i] = __hip_atomic_load(&block_count[(int)(plan.dom_off[i] / NT_DOM_BLOCK)],
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// pass C: slot_of[texel], texel_of_slot[slot], slot_xy[slot].  One thread per 4 texels
// (1024 threads = one 4096-texel domain block): the marks come in as one dword and
// slot_of goes out as one int4 per lane, i.e. every wave instruction moves a contiguous
// 256 B / 1 KiB (16 texels per thread made each store touch 64 lines at 16 B).
// SPARSE (vsa_nt_compact_frame): slot_of is only written where a texel is marked (nothing reads the
// slot of an untouched texel: 112 MB of -1 per 800x800 frame otherwise), the marks are cleared on the
// way (the next frame needs no 28 MB fill), texel_of_slot is optional.
template <bool SPARSE>
__global__ __launch_bounds__(1024) void nt_assign_kernel(vsa_nt_plan plan,
                                                         unsigned* __restrict__ marks,
                                                         const int* __restrict__ block_prefix,
                                                         int4* __restrict__ slot_of,
                                                         int* __restrict__ texel_of_slot,
                                                         float2* __restrict__ slot_xy,
                                                         long long slot_capacity) {
  __shared__ int s_w[16];
  // an untouched block (a training batch touches a few per cent of the 28 M texels; the back of every
  // shell is never seen): no marks to clear, no slot to write
  if (SPARSE && block_prefix[blockIdx.x + 1] == block_prefix[blockIdx.x]) return;
  // the (shell, degree) domain this block lies in (domains are block aligned)
  const long long blk0 = (long long)blockIdx.x * NT_DOM_BLOCK;
  int sd = 0;
  const int nseg = plan.nr_shells * VSA_NT_MAX_DEG;
  while (sd + 1 < nseg && plan.dom_off[sd + 1] <= blk0) ++sd;
  const int R = plan.tex_res[sd % VSA_NT_MAX_DEG], W = R + 2;
  const float Rf = (float)R, inv_R = 1.0f / Rf;
  const bool pow2 = (R & (R - 1)) == 0;
  const long long dom0 = plan.dom_off[sd];
  const long long vec = (long long)blockIdx.x * 1024 + threadIdx.x;   // 4-texel group
  const unsigned word = marks[vec] & 0x01010101u;
  const int c = __popc(word);
  // exclusive scan across the 1024 threads (PMC had this kernel 59 % VALU-busy: the wave scan as six
  // ds_bpermute steps and a generic 32-bit division per thread; now a DPP scan and a float-reciprocal
  // quotient, exact below 2^24, with one correction step)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int incl = nt_wave_incl_scan(c);
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  // the waves in front of this one: a 16-lane scan of the wave totals (up to 15 dependent LDS reads and adds before)
  const int wtot = nt_wave_incl_scan(lane < 16 ? s_w[lane] : 0);
  const int before = wave ? __builtin_amdgcn_readlane(wtot, (wave - 1) & 15) : 0;
  int slot = block_prefix[blockIdx.x] + incl - c + before;
  int out[4];
  const int local0 = (int)(vec * 4 - dom0);
  int iy, ix;
  if (local0 < (1 << 24)) {      // (R + 2)^2 <= 2^24 up to R = 4094: exact in fp32, off by at most one row
    iy = (int)((float)local0 * (1.0f / (float)W));
    ix = local0 - iy * W;
    if (ix < 0) --iy, ix += W;
    else if (ix >= W) ++iy, ix -= W;
  } else {
    iy = local0 / W;
    ix = local0 - iy * W;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool m = (word >> (8 * i)) & 1u;
    out[i] = m ? slot : -1;
    if (m) {
      if (slot < slot_capacity) {
        if (!SPARSE || texel_of_slot) texel_of_slot[slot] = (int)(vec * 4 + i);
        // texel centre, normalised exactly like normalize_uv_coord(corner) in the reference.  A power-of-two R
        // (every shipped configuration): the quotient is an exact scaling, i.e. the same bits as a product with
        // 1 / R — two IEEE divisions per marked texel were most of this kernel's vector work
        const float cx = (float)(ix - 1) + 0.5f, cy = (float)(iy - 1) + 0.5f;
        slot_xy[slot] = pow2 ? make_float2(cx * inv_R, cy * inv_R) : make_float2(cx / Rf, cy / Rf);
      }
      ++slot;
    }
    if (++ix == W) {
      ix = 0;
      ++iy;
    }
  }
  if (!SPARSE || word) slot_of[vec] = make_int4(out[0], out[1], out[2], out[3]);
  if (SPARSE && word) marks[vec] = 0;
}

// The SPARSE pass C with ONE WAVE per 4096-texel block (vsa_nt_compact_frame): a block is one dependent chain —
// prefix + marks in, scan, slots out — and with a 1024-thread workgroup per block only 2 chains per CU are in
// flight: 48 us for the 6.8 k blocks of a training batch (~115 marks per block), 3.6 us per block on a CU, all of
// it round trips (rocprofv3, profiles/r06/train_graph_timeline.txt; several blocks per workgroup one after the
// other: slower still).  A wave takes the block alone: lane l owns the 4-texel groups 64 k + l, k = 0 .. 15 (every
// load a contiguous 256 B), all 16 mark words are requested up front, the slot numbers come from 16 DPP scans with a
// running total — no LDS, no barrier, 32 chains per CU.
__global__ __launch_bounds__(256) void nt_assign_wave_kernel(vsa_nt_plan plan, unsigned* __restrict__ marks,
                                                             const int* __restrict__ block_prefix,
                                                             int4* __restrict__ slot_of,
                                                             int* __restrict__ texel_of_slot,
                                                             float2* __restrict__ slot_xy,
                                                             long long slot_capacity, int nr_blocks) {
  const int lane = threadIdx.x & 63;
  const int blk = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blk >= nr_blocks) return;
  const int first = block_prefix[blk];
  if (block_prefix[blk + 1] == first) return;       // untouched: nothing to clear, no slot to write
  const long long vec0 = (long long)blk * 1024 + lane;
  unsigned w[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) w[k] = marks[vec0 + 64 * k];
  const long long blk0 = (long long)blk * NT_DOM_BLOCK;
  int sd = 0;
  const int nseg = plan.nr_shells * VSA_NT_MAX_DEG;
  while (sd + 1 < nseg && plan.dom_off[sd + 1] <= blk0) ++sd;
  const int R = plan.tex_res[sd % VSA_NT_MAX_DEG], W = R + 2;
  const float Rf = (float)R, inv_R = 1.0f / Rf, inv_W = 1.0f / (float)W;
  const bool pow2 = (R & (R - 1)) == 0;
  const long long dom0 = plan.dom_off[sd];
  int run = first;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const unsigned word = w[k] & 0x01010101u;
    const int c = __popc(word);
    const int incl = nt_wave_incl_scan(c);
    int slot = run + incl - c;
    run += __builtin_amdgcn_readlane(incl, 63);
    if (!word) continue;
    const long long vec = vec0 + 64 * k;
    const int local0 = (int)(vec * 4 - dom0);
    int iy, ix;
    if (local0 < (1 << 24)) {      // exact in fp32, off by at most one row (as nt_assign_kernel)
      iy = (int)((float)local0 * inv_W);
      ix = local0 - iy * W;
      if (ix < 0) --iy, ix += W;
      else if (ix >= W) ++iy, ix -= W;
    } else {
      iy = local0 / W;
      ix = local0 - iy * W;
    }
    int out[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool m = (word >> (8 * i)) & 1u;
      out[i] = m ? slot : -1;
      if (m) {
        if (slot < slot_capacity) {
          if (texel_of_slot) texel_of_slot[slot] = (int)(vec * 4 + i);
          const float cx = (float)(ix - 1) + 0.5f, cy = (float)(iy - 1) + 0.5f;
          slot_xy[slot] = pow2 ? make_float2(cx * inv_R, cy * inv_R) : make_float2(cx / Rf, cy / Rf);
        }
        ++slot;
      }
      if (++ix == W) {
        ix = 0;
        ++iy;
      }
    }
    slot_of[vec] = make_int4(out[0], out[1], out[2], out[3]);
    marks[vec] = 0;
  }
}

}  // namespace

static int plan_check(const vsa_nt_plan* p) {
  if (!p) return VSA_ERR_ARG;
  if (p->nr_shells < 1 || p->nr_shells > VSA_MAX_SHELLS) return VSA_ERR_ARG;
  if (p->rgb_degrees < 1 || p->rgb_degrees > VSA_NT_MAX_DEG) return VSA_ERR_ARG;
  if (p->alpha_degrees < 1 || p->alpha_degrees > VSA_NT_MAX_DEG) return VSA_ERR_ARG;
  if (p->n_levels < 1 || p->n_levels > VSA_NT_MAX_LEVELS) return VSA_ERR_ARG;
  const int nseg = p->nr_shells * VSA_NT_MAX_DEG;
  for (int i = 0; i <= nseg; ++i)
    if (p->dom_off[i] % NT_DOM_BLOCK != 0 || (i && p->dom_off[i] < p->dom_off[i - 1]))
      return VSA_ERR_ARG;
  if (p->dom_off[nseg] >= (1ll << 31)) return VSA_ERR_UNSUPPORTED;
  return VSA_OK;
}

extern "C" long long vsa_nt_balance_bytes(void) { return (long long)sizeof(NtBalance); }

extern "C" int vsa_nt_rebalance(const vsa_nt_plan* plan, void* stream) {
  int rc = plan_check(plan);
  if (rc) return rc;
  if (!plan->balance) return VSA_OK;
  hipLaunchKernelGGL(nt_rebalance_kernel, dim3(NT_BAL_KERNELS), dim3(1024), 0, (hipStream_t)stream,
                     static_cast<NtBalance*>(plan->balance));
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_nt_mark(const vsa_nt_plan* plan, const int32_t* hit_slot, const float* hit_uv,
                           const float* face_uvs, int nr_rays, float* tex_uv, uint8_t* marks,
                           void* stream) {
  int rc = plan_check(plan);
  if (rc) return rc;
  if (nr_rays < 0) return VSA_ERR_ARG;
  if (nr_rays == 0) return VSA_OK;
  if (!hit_slot || !hit_uv || !face_uvs || !tex_uv) return VSA_ERR_ARG;
  dim3 grid(vsa_div_up(nr_rays, 256), plan->nr_shells);
  hipLaunchKernelGGL(nt_mark_kernel, grid, dim3(256), 0, (hipStream_t)stream, *plan, hit_slot,
                     hit_uv, face_uvs, nr_rays, tex_uv, marks);
  VSA_RETURN_LAUNCH_STATUS();
}

static int nt_compact(const vsa_nt_plan* plan, uint8_t* marks, int32_t* slot_of, int32_t* texel_of_slot,
                      float* slot_xy, int32_t* seg_start, int32_t* block_scratch, bool sparse, void* stream) {
  int rc = plan_check(plan);
  if (rc) return rc;
  if (!marks || !slot_of || (!texel_of_slot && !sparse) || !slot_xy || !seg_start || !block_scratch)
    return VSA_ERR_ARG;
  const long long total = plan->dom_off[plan->nr_shells * VSA_NT_MAX_DEG];
  const int nr_blocks = (int)(total / NT_DOM_BLOCK);
  if (nr_blocks == 0) return VSA_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(nt_count_kernel, dim3(nr_blocks), dim3(256), 0, st,
                     reinterpret_cast<const uint4*>(marks), block_scratch);
  hipLaunchKernelGGL(nt_scan_blocks_kernel, dim3(sparse && plan->balance ? 1 + NT_BAL_KERNELS : 1), dim3(1024), 0, st,
                     *plan, block_scratch, nr_blocks, seg_start);
  if (sparse)
    hipLaunchKernelGGL(nt_assign_wave_kernel, dim3(vsa_div_up(nr_blocks, 4)), dim3(256), 0, st, *plan,
                       reinterpret_cast<unsigned*>(marks), block_scratch, reinterpret_cast<int4*>(slot_of),
                       texel_of_slot, reinterpret_cast<float2*>(slot_xy), (long long)plan->slot_capacity, nr_blocks);
  else
    hipLaunchKernelGGL(nt_assign_kernel<false>, dim3(nr_blocks), dim3(1024), 0, st, *plan,
                       reinterpret_cast<unsigned*>(marks), block_scratch, reinterpret_cast<int4*>(slot_of),
                       texel_of_slot, reinterpret_cast<float2*>(slot_xy), (long long)plan->slot_capacity);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_nt_compact(const vsa_nt_plan* plan, const uint8_t* marks, int32_t* slot_of,
                              int32_t* texel_of_slot, float* slot_xy, int32_t* seg_start,
                              int32_t* block_scratch, void* stream) {
  return nt_compact(plan, const_cast<uint8_t*>(marks), slot_of, texel_of_slot, slot_xy, seg_start, block_scratch,
                    false, stream);
}

extern "C" int vsa_nt_compact_frame(const vsa_nt_plan* plan, uint8_t* marks, int32_t* slot_of,
                                    int32_t* texel_of_slot, float* slot_xy, int32_t* seg_start,
                                    int32_t* block_scratch, void* stream) {
  return nt_compact(plan, marks, slot_of, texel_of_slot, slot_xy, seg_start, block_scratch, true, stream);
}
