// 2-D / 3-D multiresolution hash-grid encoding with fp32 tables and fp32 output, and
// the SH direction encoding (the legacy appearance branch's
// GridHashEncoder, encodings/gridhash.py:12-92, and NerfHash's position / direction
// encoders, models/nerfhash.py:26-38).
//
// The reference gets the grid from tinycudann (absent: tcnn.Encoding(input_dim, {"otype":
// "Grid", "type": "Hash", 24 levels, 2 features, 2^18 entries, base 16, growth 2, linear
// interpolation}, dtype=float32)); this follows the published tiny-cuda-nn definition
// restated in oracle/tcnn_like.py (grid_forward_f32).  Unlike the 2-D texture grids of
// nt_encode.hip (2^15 entries: a level fits the LDS), a level here is 2 MiB of fp32, so
// the tables are gathered from L2 / the 256 MiB MALL (24 levels = 48 MiB): one thread per
// (sample, level), 2^D float2 gathers, D-linear weights formed in the oracle's order.
#include "common.h"
#include <cstdint>
#include <cstdlib>

namespace {

#ifndef GRID_FWD_PAIRS
#define GRID_FWD_PAIRS 1
#endif
#ifndef GRID_FWD_LV
#define GRID_FWD_LV 2     /* levels per thread: 1 / 2 / 4 = 1.63 / 1.47 / 1.59 ms per 2.1 M-sample batch (bench.py --workload dtu) */
#endif
constexpr unsigned GRID_PRIMES[3] = {1u, 2654435761u, 805459861u};

struct GridLevel {
  float scale;
  unsigned res, size, offset;
};

__device__ __forceinline__ GridLevel grid_level(const vsa_grid_plan& p, int l) {
  GridLevel g;
  g.scale = p.level_scale[l];
  g.res = (unsigned)p.level_res[l];
  g.size = (unsigned)p.level_size[l];
  g.offset = (unsigned)p.level_offset[l];
  return g;
}

// tiny-cuda-nn grid_index: dense strides while they fit the level, otherwise the XOR-prime hash
template <int D>
__device__ __forceinline__ unsigned grid_index(const GridLevel& g, const unsigned c[D]) {
  unsigned stride = 1, index = 0;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (stride <= g.size) {
      index += c[d] * stride;
      stride *= g.res;
    }
  }
  if (g.size < stride) {
    index = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) index ^= c[d] * GRID_PRIMES[d];
  }
  // hashed levels have power-of-two sizes: a mask instead of a ~35-instruction 32-bit modulo
  // (uniform per level; the general form remains for the small dense levels)
  return (g.size & (g.size - 1)) == 0 ? (index & (g.size - 1)) : index % g.size;
}

template <int D>
struct GridCell {
  unsigned c[D];
  float f[D];
};

template <int D>
__device__ __forceinline__ GridCell<D> grid_cell(const GridLevel& g, const float* __restrict__ x) {
  GridCell<D> r;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const float pos = x[d] * g.scale + 0.5f;
    const float fl = floorf(pos);
    r.f[d] = pos - fl;
    r.c[d] = (unsigned)(int)fl;
  }
  return r;
}

// corner weight in the oracle's order: ((w_0 * w_1) * w_2), w_d = f_d or 1 - f_d
template <int D>
__device__ __forceinline__ float corner_weight(const GridCell<D>& cell, int corner) {
  float w = 1.0f;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const float wd = ((corner >> d) & 1) ? cell.f[d] : 1.0f - cell.f[d];
    w = d == 0 ? wd : w * wd;
  }
  return w;
}

// LV levels per thread (blockIdx.y = group of LV levels): a thread's LV feature pairs leave as ONE 8 LV-byte store.  One
// level per thread wrote 8 bytes per lane, 208 bytes apart (a 52-float row): every wave instruction touched 64 lines for
// 512 bytes, 50 M such requests per 2.1 M-sample batch (what that pattern costs: profiles/r06/encode_bwd_train_diag.txt).
template <int D, int LV>
__global__ __launch_bounds__(256) void grid_encode_fwd_kernel(vsa_grid_plan plan,
                                                              const float2* __restrict__ tables,
                                                              const float* __restrict__ x, int B,
                                                              float* __restrict__ out, int out_stride,
                                                              int append_x) {
  const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  float res[2 * LV];
#pragma unroll
  for (int lv = 0; lv < LV; ++lv) {
  const int l = blockIdx.y * LV + lv;
  const GridLevel g = grid_level(plan, l);
  const GridCell<D> cell = grid_cell<D>(g, x + b * D);
  float f0 = 0.f, f1 = 0.f;
#if GRID_FWD_PAIRS
  // The kernel is bound by the rate of divergent 8-byte gathers (~1 address per clock and CU).  On
  // a hashed level the x coordinate enters the index with prime 1, so for an EVEN x the two
  // x-neighbours of a corner pair differ in index bit 0 only: one aligned 16-byte load fetches both
  // (level offsets are multiples of 8 entries).  Same values, same order of the sums.
  bool hashed = false;
  {
    unsigned long long stride = 1;
#pragma unroll
    for (int d = 0; d < D; ++d) stride = stride <= g.size ? stride * g.res : stride;
    hashed = g.size < stride && (g.size & (g.size - 1)) == 0;        // uniform per level
  }
  const bool pair_ok = hashed && (cell.c[0] & 1u) == 0u;
#pragma unroll
  for (int corner = 0; corner < (1 << D); corner += 2) {
    unsigned c[D];
#pragma unroll
    for (int d = 0; d < D; ++d) c[d] = cell.c[d] + ((corner >> d) & 1);
    const unsigned i0 = grid_index<D>(g, c);
    float2 v0, v1;
    if (pair_ok) {
      const float4 q = *reinterpret_cast<const float4*>(tables + g.offset + (i0 & ~1u));
      v0 = (i0 & 1u) ? make_float2(q.z, q.w) : make_float2(q.x, q.y);
      v1 = (i0 & 1u) ? make_float2(q.x, q.y) : make_float2(q.z, q.w);
    } else {
      v0 = tables[g.offset + i0];
      c[0] += 1u;
      v1 = tables[g.offset + grid_index<D>(g, c)];
    }
    const float w0 = corner_weight<D>(cell, corner), w1 = corner_weight<D>(cell, corner + 1);
    f0 = f0 + w0 * v0.x;
    f1 = f1 + w0 * v0.y;
    f0 = f0 + w1 * v1.x;
    f1 = f1 + w1 * v1.y;
  }
#else
#pragma unroll
  for (int corner = 0; corner < (1 << D); ++corner) {
    unsigned c[D];
#pragma unroll
    for (int d = 0; d < D; ++d) c[d] = cell.c[d] + ((corner >> d) & 1);
    const float w = corner_weight<D>(cell, corner);
    const float2 v = tables[g.offset + grid_index<D>(g, c)];
    f0 = f0 + w * v.x;
    f1 = f1 + w * v.y;
  }
#endif
  res[2 * lv] = f0, res[2 * lv + 1] = f1;
  }
  // out row = [2 L features | the D inputs when append_x] (GridHashEncoder's concat_points written
  // here instead of by a torch.cat over the whole feature matrix); rows of odd stride are not
  // 8-byte aligned
  float* o = out + b * out_stride + 2 * LV * blockIdx.y;
  if (LV % 2 == 0 && (out_stride & 3) == 0) {       // 16-byte groups (launched only for 16-byte aligned `out`)
#pragma unroll
    for (int q = 0; q < LV / 2; ++q)
      reinterpret_cast<float4*>(o)[q] = make_float4(res[4 * q], res[4 * q + 1], res[4 * q + 2], res[4 * q + 3]);
  } else if ((out_stride & 1) == 0) {
#pragma unroll
    for (int lv = 0; lv < LV; ++lv) reinterpret_cast<float2*>(o)[lv] = make_float2(res[2 * lv], res[2 * lv + 1]);
  } else {
#pragma unroll
    for (int i = 0; i < 2 * LV; ++i) o[i] = res[i];
  }
  if (append_x && blockIdx.y == 0) {
#pragma unroll
    for (int d = 0; d < D; ++d) out[b * out_stride + 2 * plan.n_levels + d] = x[b * D + d];
  }
}

// Two lanes per (sample, level), one per feature: the two float atomics of an entry sit in
// adjacent lanes and leave the wave as ONE 64-byte request instead of two instructions'
// worth (the kernel is bound by the L2 atomic request rate: 53 ms for 2.1 M samples x 24
// levels before, see tools/bench_bg.py).
template <int D>
__global__ __launch_bounds__(256) void grid_encode_bwd_kernel(vsa_grid_plan plan,
                                                              const float* __restrict__ x,
                                                              const float* __restrict__ g_out,
                                                              int g_stride, int B,
                                                              float* __restrict__ g_tables) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long b = t >> 1;
  const int f = (int)(t & 1);
  const int l = blockIdx.y;
  if (b >= B) return;
  const float go = g_out[b * g_stride + 2 * l + f];
  if (go == 0.f) return;
  const GridLevel g = grid_level(plan, l);
  const GridCell<D> cell = grid_cell<D>(g, x + b * D);
#pragma unroll
  for (int corner = 0; corner < (1 << D); ++corner) {
    unsigned c[D];
#pragma unroll
    for (int d = 0; d < D; ++d) c[d] = cell.c[d] + ((corner >> d) & 1);
    const float w = corner_weight<D>(cell, corner);
    atomicAdd(g_tables + 2 * (long long)(g.offset + grid_index<D>(g, c)) + f, w * go);
  }
}

// ---- large batches: table gradients WITHOUT memory-side atomics.
// The kernel above is bound by the request rate of the memory-side float-atomic path (measured:
// 403 M requests -> 26 ms for the 2.1 M samples x 24 levels of a background batch, ~16 G
// requests/s whatever the occupancy).  A 2^18-entry level does not fit the LDS, but a SLICE of it
// does: a workgroup owns (level, slice of 2^13 entries), scans its share of ALL samples,
// recomputes their corner indices and accumulates only the corners that fall into its slice,
// then flushes the slice once.  The index arithmetic is redone 32x — VALU work the chip has to
// spare — against 26 ms of atomics.  The LDS accumulators are 64-bit FIXED POINT (ds_add_u64:
// LDS float atomics are ~20x slower on gfx950, tools/ubench/lds_atomics.hip; first version of this
// kernel with ds_add_f32: 27.7 ms), scaled by a power of two from max|g| so that
// 8 B contributions cannot overflow: resolution 2^-38 of the level's largest gradient, i.e. finer
// than the fp32 sums it replaces for any level within 2^-14 of that, and independent of the summation order (bit-reproducible).
// g_lm: the output gradient re-laid level-major [L][B] float2 (grid_transpose_kernel, which also
// reduces max|g|) so that a workgroup streams its level's gradients contiguously.
constexpr int GS_SLICE_LOG2 = 13;
constexpr int GS_SLICE = 1 << GS_SLICE_LOG2;
constexpr int GS_THREADS = 1024;

// A workgroup re-lays GT_ROWS samples x L levels through LDS: rows are read as they lie (contiguous
// per sample), and every level's GT_ROWS gradients leave as one contiguous 512-byte run (written
// straight from the row-major order each lane's 8 bytes went to its own 16 MB-strided address).
constexpr int GT_ROWS = 64;
__global__ __launch_bounds__(256) void grid_transpose_kernel(const float* __restrict__ g_out,
                                                             int g_stride, int B, int L,
                                                             float2* __restrict__ g_lm,
                                                             unsigned* __restrict__ max_bits) {
  __shared__ float2 s_tile[VSA_GRID_MAX_LEVELS][GT_ROWS + 1];
  const long long b0 = (long long)blockIdx.x * GT_ROWS;
  const int rows = (int)min((long long)GT_ROWS, (long long)B - b0);
  float m = 0.f;
  for (int idx = threadIdx.x; idx < rows * L; idx += 256) {
    const int r = idx / L, l = idx - r * L;
    const float* gp = g_out + (b0 + r) * g_stride + 2 * l;
    const float2 g = make_float2(gp[0], gp[1]);
    s_tile[l][r] = g;
    float mm = fmaxf(fabsf(g.x), fabsf(g.y));
    if (!(mm < INFINITY)) mm = 0.f;
    m = fmaxf(m, mm);
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < L * GT_ROWS; idx += 256) {
    const int l = idx / GT_ROWS, r = idx - l * GT_ROWS;
    if (r < rows) g_lm[(long long)l * B + b0 + r] = s_tile[l][r];
  }
  // max |g| over everything: wave reduction, one atomic per wave (non-negative floats order like
  // their bit patterns)
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  // (one atomic per wave on ONE address serialises at the memory side: 8.9 ms.  Only a wave that
  // would RAISE the maximum it currently sees issues one — a handful per launch)
  if ((threadIdx.x & 63) == 0 && m > 0.f &&
      __float_as_uint(m) > __hip_atomic_load(max_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(max_bits, __float_as_uint(m));
}

// Only ~1 corner in 32 falls into a workgroup's slice, and an LDS atomic INSTRUCTION costs its
// ~36 cycles whether 2 or 64 of its lanes are active (second version of this kernel: one masked
// ds_add_u64 pair per corner, 20.9 ms, bound by the 352 M LDS-atomic instructions).  So the
// in-slice contributions of a wave are first compacted: each corner's in-slice lanes append
// (entry, w g_x, w g_y) to a per-wave ring buffer in LDS (ballot / mbcnt give the positions),
// and whenever 64 are queued the wave adds them with ONE fully active atomic pair.
constexpr int GS_QUEUE = 128;                      // ring entries per wave (a corner adds <= 64)

// round(v) as a two's-complement 64-bit integer for |v| < 2^62, from two native 32-bit
// conversions (the float -> int64 conversion itself is a ~30-instruction software sequence):
// v = hi * 2^32 + lo with hi = floor(v / 2^32) exact (a power-of-two scaling), lo in [0, 2^32)
__device__ __forceinline__ unsigned long long fixed62(float v) {
  const float r = rintf(v);
  const float hi = floorf(r * 2.3283064365386963e-10f);
  const float lo = r - hi * 4294967296.0f;            // exact: r has 24 significant bits
  return ((unsigned long long)(unsigned)(int)hi << 32) + (unsigned long long)(unsigned)lo;
}

template <int D>
__global__ __launch_bounds__(GS_THREADS) void grid_encode_bwd_sliced_kernel(
    vsa_grid_plan plan, const float* __restrict__ x, const float2* __restrict__ g_lm,
    const unsigned* __restrict__ max_bits, int count_bits, int B, float* __restrict__ g_tables) {
  extern __shared__ unsigned long long s_acc[];    // [GS_SLICE][2] fixed point, then the queues
  const int slice = blockIdx.x, l = blockIdx.y;
  const GridLevel g = grid_level(plan, l);
  if ((unsigned)slice << GS_SLICE_LOG2 >= g.size) return;
  const float gmax = __uint_as_float(max_bits[0]);
  if (!(gmax > 0.f)) return;                         // no gradient at all
  int e;
  frexpf(gmax, &e);                                  // gmax < 2^e
  const float scale = ldexpf(1.0f, 62 - count_bits - e);
  for (int i = threadIdx.x; i < 2 * GS_SLICE; i += GS_THREADS) s_acc[i] = 0ull;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned* s_q = reinterpret_cast<unsigned*>(s_acc + 2 * GS_SLICE) + wave * (GS_QUEUE * 3);
  int q_head = 0, q_count = 0;                       // wave-uniform
  auto drain = [&](int n) {
    __builtin_amdgcn_wave_barrier();
    if (lane < n) {
      const int pos = (q_head + lane) & (GS_QUEUE - 1);
      const unsigned ent = s_q[3 * pos];
      const float vx = __uint_as_float(s_q[3 * pos + 1]), vy = __uint_as_float(s_q[3 * pos + 2]);
      atomicAdd(s_acc + 2 * ent, fixed62(vx * scale));
      atomicAdd(s_acc + 2 * ent + 1, fixed62(vy * scale));
    }
    __builtin_amdgcn_wave_barrier();
    q_head = (q_head + n) & (GS_QUEUE - 1);
    q_count -= n;
  };
  const long long per = (B + gridDim.z - 1) / gridDim.z;
  const long long b0 = blockIdx.z * per, b1 = min((long long)B, b0 + per);
  const float2* gl = g_lm + (long long)l * B;
  // A lane takes GS_PER consecutive samples per trip and the NEXT trip's samples are loaded before
  // this trip's arithmetic: ~5 KiB per wave in flight.  (PMC of the versions with one sample per
  // trip: 56 % of the wave cycles waiting — the ~3 us load latency of 256 workgroups streaming
  // 430 MB exceeds a trip's ~1.2 us of arithmetic, so a single sample of prefetch hid nothing.)
  constexpr int GS_PER = 8;
  auto load_samples = [&](long long bfirst, float2 go[GS_PER], float xv[GS_PER][D]) {
#pragma unroll
    for (int j = 0; j < GS_PER; ++j) {
      const long long b = bfirst + j;
      go[j] = make_float2(0.f, 0.f);
#pragma unroll
      for (int d = 0; d < D; ++d) xv[j][d] = 0.f;
      if (b < b1) {
        go[j] = gl[b];
#pragma unroll
        for (int d = 0; d < D; ++d) xv[j][d] = x[b * D + d];
      }
    }
  };
  float2 go_n[GS_PER];
  float x_n[GS_PER][D];
  constexpr long long GS_STEP = (long long)GS_THREADS * GS_PER;
  load_samples(b0 + ((long long)wave * 64 + lane) * GS_PER, go_n, x_n);
  for (long long base = b0 + (long long)wave * 64 * GS_PER; base < b1; base += GS_STEP) {   // wave-uniform trips
    float2 go_c[GS_PER];
    float x_c[GS_PER][D];
#pragma unroll
    for (int j = 0; j < GS_PER; ++j) {
      go_c[j] = go_n[j];
#pragma unroll
      for (int d = 0; d < D; ++d) x_c[j][d] = x_n[j][d];
    }
    load_samples(base + GS_STEP + (long long)lane * GS_PER, go_n, x_n);
#pragma unroll 1
    for (int j = 0; j < GS_PER; ++j) {
      const float2 go = go_c[j];
      const bool active = go.x != 0.f || go.y != 0.f;
      if (!__ballot(active)) continue;
      const GridCell<D> cell = grid_cell<D>(g, x_c[j]);
#pragma unroll
      for (int corner = 0; corner < (1 << D); ++corner) {
        unsigned idx = 0;
        bool in = false;
        if (active) {
          unsigned c[D];
#pragma unroll
          for (int d = 0; d < D; ++d) c[d] = cell.c[d] + ((corner >> d) & 1);
          idx = grid_index<D>(g, c);
          in = (idx >> GS_SLICE_LOG2) == (unsigned)slice;
        }
        const unsigned long long mask = __ballot(in);
        if (mask) {
          if (in) {
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                       __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            const int pos = (q_head + q_count + rank) & (GS_QUEUE - 1);
            const float w = corner_weight<D>(cell, corner);
            s_q[3 * pos] = idx & (GS_SLICE - 1);
            s_q[3 * pos + 1] = __float_as_uint(w * go.x);
            s_q[3 * pos + 2] = __float_as_uint(w * go.y);
          }
          q_count += __popcll(mask);
          if (q_count >= 64) drain(64);
        }
      }
    }
  }
  if (q_count > 0) drain(q_count);
  __syncthreads();
  float* out = g_tables + 2ll * (g.offset + ((unsigned)slice << GS_SLICE_LOG2));
  const int n = min(GS_SLICE, (int)(g.size - ((unsigned)slice << GS_SLICE_LOG2)));
  const double inv = 1.0 / (double)scale;
  for (int i = threadIdx.x; i < 2 * n; i += GS_THREADS) {
    const long long v = (long long)s_acc[i];
    if (v != 0) atomicAdd(out + i, (float)((double)v * inv));   // += semantics; other sample chunks add theirs
  }
}

// ---- very large batches: bin once, accumulate densely.
// The sliced kernel re-derives every sample's corner indices once per SLICE (32x) and is
// issue-bound at that (14.5 ms for 2.1 M samples).  Here the indices are derived twice:
//   1. grid_bin_count:   per (level, slice) histogram of the contributions (LDS histogram per
//                        workgroup, one global atomic per workgroup and slice);
//   2. (host-free) exclusive scan of the 768 counts -> bin offsets / cursors (grid_bin_scan);
//   3. grid_bin_scatter: a workgroup takes 1024 samples of one level, counting-sorts their 8192
//                        contributions by slice IN LDS (rank = returned LDS atomic), reserves a
//                        run per slice in the global bins with one atomic, and copies the sorted
//                        records out — contiguous runs, 12 B per contribution;
//   4. grid_bin_accumulate: a workgroup per (level, slice) streams its bin (all lanes active) into
//                        the 64-bit fixed-point LDS accumulators of the sliced kernel and flushes.
// Traffic: 2 x 12 B x 8 L B (4.8 GB written + read at 2.1 M samples) instead of 32x the index VALU.
#ifndef VSA_GB_SAMPLES
#define VSA_GB_SAMPLES 512
#endif
constexpr int GB_SAMPLES = VSA_GB_SAMPLES;          // samples per scatter trip = threads per workgroup (512: three workgroups per CU overlap their phases; 1024 = one per CU: 1.82 -> 1.49 ms)
constexpr int GB_MAX_SLICES = 32;                   // 2^18 entries / 2^13
#ifndef VSA_GB_QUANTUM_LOG2
#define VSA_GB_QUANTUM_LOG2 18
#endif
#ifndef GB_RPL
#define GB_RPL 4                                    // records per lane and trip of the accumulation
#endif
constexpr unsigned long long GB_QUANTUM = 1ull << VSA_GB_QUANTUM_LOG2;   // records per accumulate workgroup

template <int D>
__device__ __forceinline__ void grid_corner_indices(const GridLevel& g, const GridCell<D>& cell,
                                                    unsigned idx[1 << D]) {
#pragma unroll
  for (int corner = 0; corner < (1 << D); ++corner) {
    unsigned c[D];
#pragma unroll
    for (int d = 0; d < D; ++d) c[d] = cell.c[d] + ((corner >> d) & 1);
    idx[corner] = grid_index<D>(g, c);
  }
}

template <int D>
__global__ __launch_bounds__(GB_SAMPLES) void grid_bin_count_kernel(
    vsa_grid_plan plan, const float* __restrict__ x, const float2* __restrict__ g_lm, int B,
    unsigned long long* __restrict__ counts) {
  __shared__ unsigned s_cnt[GB_MAX_SLICES];
  const int l = blockIdx.y;
  const GridLevel g = grid_level(plan, l);
  if (threadIdx.x < GB_MAX_SLICES) s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  const float2* gl = g_lm + (long long)l * B;
  for (long long b = (long long)blockIdx.x * GB_SAMPLES + threadIdx.x; b < B;
       b += (long long)gridDim.x * GB_SAMPLES) {
    const float2 go = gl[b];
    if (go.x == 0.f && go.y == 0.f) continue;
    float xv[D];
#pragma unroll
    for (int d = 0; d < D; ++d) xv[d] = x[b * D + d];
    const GridCell<D> cell = grid_cell<D>(g, xv);
    unsigned idx[1 << D];
    grid_corner_indices<D>(g, cell, idx);
#pragma unroll
    for (int corner = 0; corner < (1 << D); ++corner) atomicAdd(&s_cnt[idx[corner] >> GS_SLICE_LOG2], 1u);
  }
  __syncthreads();
  if (threadIdx.x < GB_MAX_SLICES && s_cnt[threadIdx.x])
    atomicAdd(&counts[l * GB_MAX_SLICES + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

// offsets[i] = sum of counts[0..i), i <= n; cursors start at the offsets.  One workgroup, one bin per
// thread (n <= 1024): wave scans + a scan of the 16 wave totals.  (One thread walking the 768 bins
// with a dependent global load each took 85 us.)
__global__ __launch_bounds__(1024) void grid_bin_scan_kernel(const unsigned long long* __restrict__ counts, int n,
                                                             unsigned long long* __restrict__ offsets,
                                                             unsigned long long* __restrict__ cursors) {
  __shared__ unsigned long long s_tot[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const unsigned long long c = t < n ? counts[t] : 0ull;
  unsigned long long incl = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  if (lane == 63) s_tot[wave] = incl;
  __syncthreads();
  unsigned long long base = 0ull;
  for (int w = 0; w < wave; ++w) base += s_tot[w];
  const unsigned long long excl = base + incl - c;
  if (t < n) {
    offsets[t] = excl;
    cursors[t] = excl;
  }
  if (t == n - 1) offsets[n] = excl + c;
}

template <int D>
__global__ __launch_bounds__(GB_SAMPLES) void grid_bin_scatter_kernel(
    vsa_grid_plan plan, const float* __restrict__ x, const float2* __restrict__ g_lm, int B,
    unsigned long long* __restrict__ cursors, unsigned short* __restrict__ rec_idx,
    float* __restrict__ rec_x, float* __restrict__ rec_y) {
  constexpr int NC = 1 << D;
  extern __shared__ unsigned s_raw[];
  unsigned* s_cnt = s_raw;                              // [32] contributions per slice in this trip
  unsigned* s_off = s_raw + GB_MAX_SLICES;              // [33] exclusive offsets in the LDS buffer
  unsigned long long* s_base = reinterpret_cast<unsigned long long*>(s_raw + 2 * GB_MAX_SLICES + 2);  // [32]
  unsigned* s_idx = s_raw + 4 * GB_MAX_SLICES + 4;      // [GB_SAMPLES * NC]
  float* s_vx = reinterpret_cast<float*>(s_idx + GB_SAMPLES * NC);
  float* s_vy = s_vx + GB_SAMPLES * NC;
  const int l = blockIdx.y;
  const GridLevel g = grid_level(plan, l);
  const float2* gl = g_lm + (long long)l * B;
  for (long long base = (long long)blockIdx.x * GB_SAMPLES; base < B; base += (long long)gridDim.x * GB_SAMPLES) {
    if (threadIdx.x < GB_MAX_SLICES) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    const long long b = base + threadIdx.x;
    float2 go = make_float2(0.f, 0.f);
    if (b < B) go = gl[b];
    const bool active = go.x != 0.f || go.y != 0.f;
    unsigned idx[NC], rank[NC];
    float w[NC];
    if (active) {
      float xv[D];
#pragma unroll
      for (int d = 0; d < D; ++d) xv[d] = x[b * D + d];
      const GridCell<D> cell = grid_cell<D>(g, xv);
      grid_corner_indices<D>(g, cell, idx);
#pragma unroll
      for (int corner = 0; corner < NC; ++corner) {
        w[corner] = corner_weight<D>(cell, corner);
        rank[corner] = atomicAdd(&s_cnt[idx[corner] >> GS_SLICE_LOG2], 1u);
      }
    }
    __syncthreads();
    if (threadIdx.x < 64) {                             // exclusive scan of the 32 counts on one wave
      const unsigned cnt = threadIdx.x < GB_MAX_SLICES ? s_cnt[threadIdx.x] : 0u;
      unsigned incl = cnt;
#pragma unroll
      for (int d = 1; d < GB_MAX_SLICES; d <<= 1) {
        const unsigned up = __shfl_up(incl, d, 64);
        if ((int)threadIdx.x >= d) incl += up;
      }
      if (threadIdx.x < GB_MAX_SLICES) s_off[threadIdx.x] = incl - cnt;
      if (threadIdx.x == GB_MAX_SLICES - 1) s_off[GB_MAX_SLICES] = incl;
      if (threadIdx.x < GB_MAX_SLICES && cnt)
        s_base[threadIdx.x] = atomicAdd(&cursors[l * GB_MAX_SLICES + threadIdx.x], (unsigned long long)cnt);
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int corner = 0; corner < NC; ++corner) {
        const unsigned pos = s_off[idx[corner] >> GS_SLICE_LOG2] + rank[corner];
        s_idx[pos] = idx[corner];
        s_vx[pos] = w[corner] * go.x;
        s_vy[pos] = w[corner] * go.y;
      }
    }
    __syncthreads();
    const unsigned total = s_off[GB_MAX_SLICES];
    for (unsigned p = threadIdx.x; p < total; p += GB_SAMPLES) {
      const unsigned id = s_idx[p];
      const unsigned sl = id >> GS_SLICE_LOG2;
      const unsigned long long dst = s_base[sl] + (p - s_off[sl]);
      rec_idx[dst] = (unsigned short)(id & (GS_SLICE - 1));      // 13 bits: 10 B per record instead of 12
      rec_x[dst] = s_vx[p];
      rec_y[dst] = s_vy[p];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(GS_THREADS) void grid_bin_accumulate_kernel(
    vsa_grid_plan plan, const unsigned long long* __restrict__ offsets,
    const unsigned short* __restrict__ rec_idx, const float* __restrict__ rec_x,
    const float* __restrict__ rec_y, const unsigned* __restrict__ max_bits, int count_bits,
    float* __restrict__ g_tables) {
  extern __shared__ unsigned long long s_acc[];    // [GS_SLICE][2] fixed point
  const int slice = blockIdx.x, l = blockIdx.y;
  const GridLevel g = grid_level(plan, l);
  if ((unsigned)slice << GS_SLICE_LOG2 >= g.size) return;
  // a bin is shared out in quanta of GB_QUANTUM records (blockIdx.z): the single slice of a dense
  // 17^3 level holds ALL 8 B contributions of that level — one workgroup on it took 19 ms
  unsigned long long r0 = offsets[l * GB_MAX_SLICES + slice];
  unsigned long long r1 = offsets[l * GB_MAX_SLICES + slice + 1];
  r0 += (unsigned long long)blockIdx.z * GB_QUANTUM;
  if (r0 >= r1) return;
  if (r1 > r0 + GB_QUANTUM) r1 = r0 + GB_QUANTUM;
  const float gmax = __uint_as_float(max_bits[0]);
  int e;
  frexpf(gmax, &e);
  const float scale = ldexpf(1.0f, 62 - count_bits - e);
  for (int i = threadIdx.x; i < 2 * GS_SLICE; i += GS_THREADS) s_acc[i] = 0ull;
  __syncthreads();
  // A lane owns GB_RPL CONSECUTIVE records (vector loads from the 4-record-aligned start of the
  // quantum; records outside [r0, r1) are masked).  Two reasons:
  //  * one 12-byte record in flight per lane left the kernel waiting for memory with its single
  //    workgroup per CU (the accumulators take 128 KiB);
  //  * consecutive records of a bin are the SAME corner of consecutive samples (the scatter ranks a
  //    wave's lanes corner by corner), and on the coarser levels neighbouring samples of a ray sit
  //    in the same cell: 64 lanes adding to a handful of entries serialise in the LDS (measured
  //    with scrambled entries: 1.5 ms instead of 3.4 ms for the 403 M records of a background
  //    batch).  Equal neighbours are summed in registers first — integer sums, so the result does
  //    not depend on the grouping.
  const unsigned long long ra = r0 & ~3ull;
  for (unsigned long long base = ra + (unsigned long long)threadIdx.x * GB_RPL; base < r1;
       base += (unsigned long long)GB_RPL * GS_THREADS) {
    unsigned ent[GB_RPL];
    float vx[GB_RPL], vy[GB_RPL];
#pragma unroll
    for (int q = 0; q < GB_RPL / 4; ++q) {
      const uint2 e2 = *reinterpret_cast<const uint2*>(rec_idx + base + 4 * q);      // four 16-bit entries
      const uint4 e4 = make_uint4(e2.x & 0xffffu, e2.x >> 16, e2.y & 0xffffu, e2.y >> 16);
      const float4 x4 = *reinterpret_cast<const float4*>(rec_x + base + 4 * q);
      const float4 y4 = *reinterpret_cast<const float4*>(rec_y + base + 4 * q);
      ent[4 * q] = e4.x, ent[4 * q + 1] = e4.y, ent[4 * q + 2] = e4.z, ent[4 * q + 3] = e4.w;
      vx[4 * q] = x4.x, vx[4 * q + 1] = x4.y, vx[4 * q + 2] = x4.z, vx[4 * q + 3] = x4.w;
      vy[4 * q] = y4.x, vy[4 * q + 1] = y4.y, vy[4 * q + 2] = y4.z, vy[4 * q + 3] = y4.w;
    }
    unsigned long long sx = 0ull, sy = 0ull;
#pragma unroll
    for (int u = 0; u < GB_RPL; ++u) {
      const unsigned long long ru = base + u;
      const bool ok = ru >= r0 && ru < r1;
      if (ok) {
        sx += fixed62(vx[u] * scale);
        sy += fixed62(vy[u] * scale);
      }
      // last record of a run of equal entries inside this lane's stretch
      const bool next_ok = u + 1 < GB_RPL && ru + 1 >= r0 && ru + 1 < r1;
      const bool flush = ok && !(next_ok && ent[u + 1 < GB_RPL ? u + 1 : u] == ent[u]);
      if (flush) {
        atomicAdd(s_acc + 2 * ent[u], sx);
        atomicAdd(s_acc + 2 * ent[u] + 1, sy);
        sx = 0ull, sy = 0ull;
      }
    }
  }
  __syncthreads();
  float* out = g_tables + 2ll * (g.offset + ((unsigned)slice << GS_SLICE_LOG2));
  const int n = min(GS_SLICE, (int)(g.size - ((unsigned)slice << GS_SLICE_LOG2)));
  const double inv = 1.0 / (double)scale;
  for (int i = threadIdx.x; i < 2 * n; i += GS_THREADS) {
    const long long v = (long long)s_acc[i];
    if (v != 0) atomicAdd(out + i, (float)((double)v * inv));
  }
}

// SHEncoder.__call__ (encodings/sphericalharmonics.py:84-153): the SH basis of a direction,
// degree 0..4, fp32, products formed left to right as the reference writes them.
__global__ void sh_encode_kernel(const float* __restrict__ dirs, int B, int degree,
                                 float* __restrict__ out) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float x = dirs[3 * b], y = dirs[3 * b + 1], z = dirs[3 * b + 2];
  const int n = (degree + 1) * (degree + 1);
  float r[25];
#pragma unroll
  for (int i = 1; i < 25; ++i) r[i] = 0.f;
  r[0] = 0.28209479177387814f;
  if (degree > 0) {
    const float C1 = 0.4886025119029199f;
    r[1] = -C1 * y;
    r[2] = C1 * z;
    r[3] = -C1 * x;
    if (degree > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      r[4] = 1.0925484305920792f * xy;
      r[5] = -1.0925484305920792f * yz;
      r[6] = 0.31539156525252005f * ((2.0f * zz - xx) - yy);
      r[7] = -1.0925484305920792f * xz;
      r[8] = 0.5462742152960396f * (xx - yy);
      if (degree > 2) {
        r[9] = (-0.5900435899266435f * y) * (3.0f * xx - yy);
        r[10] = (2.890611442640554f * xy) * z;
        r[11] = (-0.4570457994644658f * y) * ((4.0f * zz - xx) - yy);
        r[12] = (0.3731763325901154f * z) * ((2.0f * zz - 3.0f * xx) - 3.0f * yy);
        r[13] = (-0.4570457994644658f * x) * ((4.0f * zz - xx) - yy);
        r[14] = (1.445305721320277f * z) * (xx - yy);
        r[15] = (-0.5900435899266435f * x) * (xx - 3.0f * yy);
        if (degree > 3) {
          r[16] = (2.5033429417967046f * xy) * (xx - yy);
          r[17] = (-1.7701307697799304f * yz) * (3.0f * xx - yy);
          r[18] = (0.9461746957575601f * xy) * (7.0f * zz - 1.0f);
          r[19] = (-0.6690465435572892f * yz) * (7.0f * zz - 3.0f);
          r[20] = 0.10578554691520431f * (zz * (35.0f * zz - 30.0f) + 3.0f);
          r[21] = (-0.6690465435572892f * xz) * (7.0f * zz - 3.0f);
          r[22] = (0.47308734787878004f * (xx - yy)) * (7.0f * zz - 1.0f);
          r[23] = (-1.7701307697799304f * xz) * (xx - 3.0f * yy);
          r[24] = 0.6258357354491761f * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy));
        }
      }
    }
  }
  // (compile-time indices: a run-time loop over r[] moved the array to scratch memory.  Degree 1
  //  and 3 rows are 16 / 64 bytes: whole float4 stores)
  float* o = out + b * n;
  if ((n & 3) == 0) {
#pragma unroll
    for (int i = 0; i < 24; i += 4)
      if (i < n) *reinterpret_cast<float4*>(o + i) = make_float4(r[i], r[i + 1], r[i + 2], r[i + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < 25; ++i)
      if (i < n) o[i] = r[i];
  }
}

int plan_ok(const vsa_grid_plan* p) {
  if (!p) return VSA_ERR_ARG;
  if (p->n_dims != 2 && p->n_dims != 3) return VSA_ERR_UNSUPPORTED;
  if (p->n_features != 2) return VSA_ERR_UNSUPPORTED;
  if (p->n_levels < 1 || p->n_levels > VSA_GRID_MAX_LEVELS) return VSA_ERR_ARG;
  for (int l = 0; l < p->n_levels; ++l)
    if (p->level_size[l] < 1 || p->level_res[l] < 1) return VSA_ERR_ARG;
  return VSA_OK;
}

}  // namespace

extern "C" int vsa_grid_encode_fwd_ld(const vsa_grid_plan* plan, const float* tables, const float* x,
                                      int nr_points, float* out, int out_stride, int append_x,
                                      void* stream) {
  int rc = plan_ok(plan);
  if (rc) return rc;
  if (nr_points < 0 || out_stride < 2 * plan->n_levels + (append_x ? plan->n_dims : 0)) return VSA_ERR_ARG;
  if (nr_points == 0) return VSA_OK;
  if (!tables || !x || !out) return VSA_ERR_ARG;
  // levels per thread: 4 (32-byte stores) / 2 where the level count divides and the rows are 16-byte groups
  static const int lv_env = getenv("VSA_GRID_FWD_LV") ? atoi(getenv("VSA_GRID_FWD_LV")) : GRID_FWD_LV;
  const bool wide = (out_stride & 3) == 0 && ((uintptr_t)out & 15) == 0;
  const int lv = wide && lv_env >= 4 && plan->n_levels % 4 == 0 ? 4 : wide && lv_env >= 2 && plan->n_levels % 2 == 0 ? 2 : 1;
  dim3 grid(vsa_div_up(nr_points, 256), plan->n_levels / lv);
#define GRID_FWD_LAUNCH(DD, LL)                                                                                  \
  hipLaunchKernelGGL((grid_encode_fwd_kernel<DD, LL>), grid, dim3(256), 0, (hipStream_t)stream, *plan,          \
                     reinterpret_cast<const float2*>(tables), x, nr_points, out, out_stride, append_x)
  if (plan->n_dims == 2) {
    if (lv == 4) GRID_FWD_LAUNCH(2, 4);
    else if (lv == 2) GRID_FWD_LAUNCH(2, 2);
    else GRID_FWD_LAUNCH(2, 1);
  } else {
    if (lv == 4) GRID_FWD_LAUNCH(3, 4);
    else if (lv == 2) GRID_FWD_LAUNCH(3, 2);
    else GRID_FWD_LAUNCH(3, 1);
  }
#undef GRID_FWD_LAUNCH
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_grid_encode_fwd(const vsa_grid_plan* plan, const float* tables, const float* x,
                                   int nr_points, float* out, void* stream) {
  if (!plan) return VSA_ERR_ARG;
  return vsa_grid_encode_fwd_ld(plan, tables, x, nr_points, out, 2 * plan->n_levels, 0, stream);
}

extern "C" int vsa_grid_encode_bwd_ld(const vsa_grid_plan* plan, const float* x, const float* g_out,
                                      int g_stride, int nr_points, float* grad_tables, void* stream) {
  int rc = plan_ok(plan);
  if (rc) return rc;
  if (nr_points < 0 || g_stride < 2 * plan->n_levels) return VSA_ERR_ARG;
  if (nr_points == 0) return VSA_OK;
  if (!x || !g_out || !grad_tables) return VSA_ERR_ARG;
  dim3 grid(vsa_div_up(2ll * nr_points, 256), plan->n_levels);
  if (plan->n_dims == 2)
    hipLaunchKernelGGL(grid_encode_bwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, *plan, x,
                       g_out, g_stride, nr_points, grad_tables);
  else
    hipLaunchKernelGGL(grid_encode_bwd_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, *plan, x,
                       g_out, g_stride, nr_points, grad_tables);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_grid_encode_bwd(const vsa_grid_plan* plan, const float* x, const float* g_out,
                                   int nr_points, float* grad_tables, void* stream) {
  if (!plan) return VSA_ERR_ARG;
  return vsa_grid_encode_bwd_ld(plan, x, g_out, 2 * plan->n_levels, nr_points, grad_tables, stream);
}

extern "C" int vsa_grid_encode_bwd_sliced_ld(const vsa_grid_plan* plan, const float* x,
                                             const float* g_out, int g_stride, int nr_points,
                                             float* grad_tables, float* workspace, void* stream) {
  int rc = plan_ok(plan);
  if (rc) return rc;
  if (nr_points < 0 || g_stride < 2 * plan->n_levels) return VSA_ERR_ARG;
  if (nr_points == 0) return VSA_OK;
  if (!x || !g_out || !grad_tables || !workspace) return VSA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int L = plan->n_levels;
  // workspace: [L][B] float2, then L words of max|g| bits
  unsigned* level_max = reinterpret_cast<unsigned*>(workspace + 2ll * nr_points * L);
  VSA_HIP_TRY(hipMemsetAsync(level_max, 0, sizeof(unsigned) * VSA_GRID_MAX_LEVELS, st));
  hipLaunchKernelGGL(grid_transpose_kernel, dim3(vsa_div_up(nr_points, GT_ROWS)), dim3(256), 0,
                     st, g_out, g_stride, nr_points, L, reinterpret_cast<float2*>(workspace), level_max);
  int max_size = 1;
  for (int l = 0; l < L; ++l) max_size = plan->level_size[l] > max_size ? plan->level_size[l] : max_size;
  const int slices = (max_size + GS_SLICE - 1) >> GS_SLICE_LOG2;
  const size_t lds = 2ull * GS_SLICE * sizeof(unsigned long long) + (GS_THREADS / 64) * GS_QUEUE * 3 * sizeof(unsigned);
  static bool attr_set = false;
  if (!attr_set) {
    VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(grid_encode_bwd_sliced_kernel<2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(grid_encode_bwd_sliced_kernel<3>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  // sample chunks: enough workgroups for a few rounds over the CUs (one workgroup per CU: 128 KiB of LDS)
  int nr_cus = 0;
  rc = vsa_cu_count(&nr_cus);
  if (rc) return rc;
  int chunks = (6 * nr_cus + slices * L - 1) / (slices * L);
  if (chunks < 1) chunks = 1;
  // an entry receives at most 2^D corners of every sample of a chunk
  long long worst = ((long long)(nr_points + chunks - 1) / chunks) << plan->n_dims;
  int count_bits = 1;
  while ((1ll << count_bits) < worst) ++count_bits;
  dim3 grid(slices, L, chunks);
  const float2* g_lm = reinterpret_cast<const float2*>(workspace);
  if (plan->n_dims == 2)
    hipLaunchKernelGGL(grid_encode_bwd_sliced_kernel<2>, grid, dim3(GS_THREADS), lds, st, *plan, x, g_lm,
                       level_max, count_bits, nr_points, grad_tables);
  else
    hipLaunchKernelGGL(grid_encode_bwd_sliced_kernel<3>, grid, dim3(GS_THREADS), lds, st, *plan, x, g_lm,
                       level_max, count_bits, nr_points, grad_tables);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_grid_encode_bwd_sliced(const vsa_grid_plan* plan, const float* x,
                                          const float* g_out, int nr_points, float* grad_tables,
                                          float* workspace, void* stream) {
  if (!plan) return VSA_ERR_ARG;
  return vsa_grid_encode_bwd_sliced_ld(plan, x, g_out, 2 * plan->n_levels, nr_points, grad_tables, workspace,
                                       stream);
}

extern "C" int vsa_grid_encode_bwd_binned_workspace(const vsa_grid_plan* plan, int nr_points,
                                                    long long* workspace_floats) {
  int rc = plan_ok(plan);
  if (rc) return rc;
  if (nr_points < 0 || !workspace_floats) return VSA_ERR_ARG;
  const long long L = plan->n_levels, B = nr_points;
  // [L][B] float2 | 3 x (8 corners x L x B) records | max bits + counts / offsets / cursors
  // (the record arrays start on a 16-byte boundary: the accumulation reads them as 4-record vectors)
  *workspace_floats = ((2 * B * L + 3) & ~3ll) + 3 * (B << plan->n_dims) * L + 64 + 3 * 2 * (VSA_GRID_MAX_LEVELS * GB_MAX_SLICES + 1);
  return VSA_OK;
}

extern "C" int vsa_grid_encode_bwd_binned_ld(const vsa_grid_plan* plan, const float* x,
                                             const float* g_out, int g_stride, int nr_points,
                                             float* grad_tables, float* workspace, void* stream) {
  int rc = plan_ok(plan);
  if (rc) return rc;
  if (nr_points < 0 || g_stride < 2 * plan->n_levels) return VSA_ERR_ARG;
  if (nr_points == 0) return VSA_OK;
  if (!x || !g_out || !grad_tables || !workspace) return VSA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long long L = plan->n_levels, B = nr_points;
  const long long nrec = (B << plan->n_dims) * L;
  for (int l = 0; l < L; ++l)
    if (((plan->level_size[l] + GS_SLICE - 1) >> GS_SLICE_LOG2) > GB_MAX_SLICES) return VSA_ERR_UNSUPPORTED;
  float2* g_lm = reinterpret_cast<float2*>(workspace);
  if (reinterpret_cast<size_t>(workspace) & 15) return VSA_ERR_ARG;
  const long long g_floats = (2 * B * L + 3) & ~3ll;
  unsigned short* rec_idx = reinterpret_cast<unsigned short*>(workspace + g_floats);   // (uses half of its nrec floats)
  float* rec_x = workspace + g_floats + nrec;
  float* rec_y = rec_x + nrec;
  unsigned* max_bits = reinterpret_cast<unsigned*>(rec_y + nrec);
  const int nbins = VSA_GRID_MAX_LEVELS * GB_MAX_SLICES;
  unsigned long long* counts = reinterpret_cast<unsigned long long*>(max_bits + 64);
  unsigned long long* offsets = counts + nbins + 1;
  unsigned long long* cursors = offsets + nbins + 1;
  VSA_HIP_TRY(hipMemsetAsync(max_bits, 0, 64 * sizeof(unsigned) + (nbins + 1) * sizeof(unsigned long long), st));
  hipLaunchKernelGGL(grid_transpose_kernel, dim3(vsa_div_up(B, GT_ROWS)), dim3(256), 0, st,
                     g_out, g_stride, nr_points, (int)L, g_lm, max_bits);
  int nr_cus = 0;
  rc = vsa_cu_count(&nr_cus);
  if (rc) return rc;
  int gx = (int)((B + GB_SAMPLES - 1) / GB_SAMPLES);
  const int gx_cap = (4 * (1024 / GB_SAMPLES) * nr_cus + (int)L - 1) / (int)L;
  if (gx > gx_cap) gx = gx_cap;
  const size_t lds_sc = (4 * GB_MAX_SLICES + 4) * sizeof(unsigned) + 3ull * GB_SAMPLES * (1 << plan->n_dims) * sizeof(float);
  const size_t lds_acc = 2ull * GS_SLICE * sizeof(unsigned long long);
  static bool attr_set = false;
  if (!attr_set) {
    VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(grid_bin_scatter_kernel<2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(grid_bin_scatter_kernel<3>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    VSA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(grid_bin_accumulate_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_acc));
    attr_set = true;
  }
  dim3 grid(gx, (unsigned)L);
  if (plan->n_dims == 2) {
    hipLaunchKernelGGL(grid_bin_count_kernel<2>, grid, dim3(GB_SAMPLES), 0, st, *plan, x, g_lm, nr_points, counts);
  } else {
    hipLaunchKernelGGL(grid_bin_count_kernel<3>, grid, dim3(GB_SAMPLES), 0, st, *plan, x, g_lm, nr_points, counts);
  }
  hipLaunchKernelGGL(grid_bin_scan_kernel, dim3(1), dim3(1024), 0, st, counts, nbins, offsets, cursors);
  if (plan->n_dims == 2) {
    hipLaunchKernelGGL(grid_bin_scatter_kernel<2>, grid, dim3(GB_SAMPLES), lds_sc, st, *plan, x, g_lm, nr_points,
                       cursors, rec_idx, rec_x, rec_y);
  } else {
    hipLaunchKernelGGL(grid_bin_scatter_kernel<3>, grid, dim3(GB_SAMPLES), lds_sc, st, *plan, x, g_lm, nr_points,
                       cursors, rec_idx, rec_x, rec_y);
  }
  // an entry receives at most 2^D corners of every sample
  long long worst = B << plan->n_dims;
  int count_bits = 1;
  while ((1ll << count_bits) < worst) ++count_bits;
  const unsigned zmax = (unsigned)(((unsigned long long)worst + GB_QUANTUM - 1) / GB_QUANTUM);   // a level's records may all sit in one bin
  hipLaunchKernelGGL(grid_bin_accumulate_kernel, dim3(GB_MAX_SLICES, (unsigned)L, zmax), dim3(GS_THREADS), lds_acc, st,
                     *plan, offsets, rec_idx, rec_x, rec_y, max_bits, count_bits, grad_tables);
  VSA_RETURN_LAUNCH_STATUS();
}

extern "C" int vsa_grid_encode_bwd_binned(const vsa_grid_plan* plan, const float* x,
                                          const float* g_out, int nr_points, float* grad_tables,
                                          float* workspace, void* stream) {
  if (!plan) return VSA_ERR_ARG;
  return vsa_grid_encode_bwd_binned_ld(plan, x, g_out, 2 * plan->n_levels, nr_points, grad_tables, workspace,
                                       stream);
}

extern "C" int vsa_sh_encode(const float* dirs, int nr_dirs, int degree, float* out, void* stream) {
  if (nr_dirs < 0 || degree < 0 || degree > 4) return VSA_ERR_ARG;
  if (nr_dirs == 0) return VSA_OK;
  if (!dirs || !out) return VSA_ERR_ARG;
  hipLaunchKernelGGL(sh_encode_kernel, dim3(vsa_div_up(nr_dirs, 256)), dim3(256), 0,
                     (hipStream_t)stream, dirs, nr_dirs, degree, out);
  VSA_RETURN_LAUNCH_STATUS();
}
