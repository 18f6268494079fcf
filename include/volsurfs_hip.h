/* volsurfs_hip.h — C-ABI of libvolsurfs_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the VolSurfs K-shell render hot path.
 * The reference's own boundary is the pybind11 module `volsurfs`
 * (/root/reference/src/PyBridge.cxx:19-139) plus three third-party native
 * modules the hot path calls (raytracelib, tinycudann, mvdatasets helpers).
 * Every entry point below is what a binding for one of those call sites would
 * bind; the reference interface it replaces is cited per function.
 *
 * Conventions
 *   - plain pointers + sizes, no torch / STL types; all pointers are DEVICE
 *     pointers unless a parameter is marked [host];
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     every launch is stream-ordered and asynchronous (the reference
 *     synchronises after every kernel, src/VolumeRendering.cu:64; callers only
 *     consume returned tensors, so async is legal — SURVEY §8b "Threading");
 *   - return value: 0 = ok, >0 = hipError_t of the failed call/launch,
 *     <0 = VSA_ERR_* argument / support error.  Nothing is printed-and-ignored;
 *   - caller owns every buffer; the library keeps no global state except
 *     objects created by *_create and destroyed by *_destroy;
 *   - fp32 row-major tensors, int32 indices (reference layout);
 *   - every device buffer starts 16-byte aligned (rows and slabs are moved with 16-byte loads,
 *     stores and LDS-DMA; any allocator's base pointer qualifies, an odd view into one may not).
 */
#ifndef VOLSURFS_HIP_H
#define VOLSURFS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSA_OK 0
#define VSA_ERR_ARG (-1)
#define VSA_ERR_UNSUPPORTED (-2)

#define VSA_MAX_SHELLS 16

/* Library version / build info (sanity check used by the loader). */
int vsa_version(void);

/* ------------------------------------------------------------------------
 * A7  Dense K-shell alpha composite.
 * Replaces the PyTorch op sequence volsurfs_py/methods/volsurfs.py:601-640 and
 * :704-708 (forward) and its autograd replay (backward).
 *   surfs_rgb   [N,K,3] f32, surfs_alpha [N,K] f32 — inner->outer shell order,
 *               zero where the ray misses the shell (volsurfs.py:455-456).
 *   rgb_bg      [N,3] f32, or [1,3] when bg_is_broadcast (volsurfs.py:688).
 *   out_rgb     [N,3]  = rgb_fg + bg_T * rgb_bg             (required)
 *   out_rgb_fg  [N,3], out_bg_transmittance [N], out_weights [N,K],
 *   out_surfs_rgb_h [N,K,3], out_surfs_alpha_h [N,K]: optional (NULL to skip);
 *               the last two are the fp16-rounded copies the reference returns
 *               (volsurfs.py:732-733).
 *   carry_f16   0: running transmittance carried in fp32, rounded per store
 *               (ATen CPU cumprod; pinned by tests/golden); 1: carried in fp16
 *               (ATen CUDA scan).
 *   1 <= K <= 9 (reference configs ship K = 1,3,5,7,9).
 */
int vsa_composite_dense_fwd(const float* surfs_rgb, const float* surfs_alpha,
                            const float* rgb_bg, int bg_is_broadcast, float* out_rgb,
                            float* out_rgb_fg, float* out_bg_transmittance, float* out_weights,
                            float* out_surfs_rgb_h, float* out_surfs_alpha_h, int nr_rays,
                            int nr_shells, int carry_f16, void* stream);

/* Backward of out_rgb w.r.t. surfs_rgb, surfs_alpha (and per-ray rgb_bg when
 * g_rgb_bg != NULL, [N,3]); forward is recomputed in registers. */
int vsa_composite_dense_bwd(const float* surfs_rgb, const float* surfs_alpha,
                            const float* rgb_bg, int bg_is_broadcast, const float* g_rgb,
                            float* g_surfs_rgb, float* g_surfs_alpha, float* g_rgb_bg,
                            int nr_rays, int nr_shells, int carry_f16, void* stream);

/* The same with the L1 image loss of utils/losses.py:14-19 fused in: the upstream gradient
 * is formed in the kernel as loss_scale * sign(pred_rgb - gt_rgb) (loss_scale = 1 / (3 N_global)
 * for the mean), saving three elementwise passes over [N,3]. */
int vsa_composite_dense_bwd_l1(const float* surfs_rgb, const float* surfs_alpha,
                               const float* rgb_bg, int bg_is_broadcast, const float* pred_rgb,
                               const float* gt_rgb, float loss_scale, float* g_surfs_rgb,
                               float* g_surfs_alpha, int nr_rays, int nr_shells, int carry_f16,
                               void* stream);

/* Forward, L1 loss and backward of a training step in ONE pass over the shells' colours: writes the
 * composited out_rgb [N,3] (bit-identical to vsa_composite_dense_fwd) and the gradients of
 * loss_scale * sum |out_rgb - gt_rgb| w.r.t. surfs_rgb / surfs_alpha (as vsa_composite_dense_bwd_l1). */
int vsa_composite_dense_fwd_bwd_l1(const float* surfs_rgb, const float* surfs_alpha,
                                   const float* rgb_bg, int bg_is_broadcast, const float* gt_rgb,
                                   float loss_scale, float* out_rgb, float* g_surfs_rgb,
                                   float* g_surfs_alpha, int nr_rays, int nr_shells, int carry_f16,
                                   void* stream);

/* The two scalar read-outs of a training iteration (csrc/reduce.hip), ONE launch each and no fill in front:
 *   vsa_count_hits: *out (i64, device) = number of entries >= 0 among hit_slot[0..n) — the sample count that
 *     steers the reference's dynamic ray count (trainer.py:293-304; there `samples_3d.shape[0]` after a
 *     boolean-mask compaction, volsurfs.py:713-716);
 *   vsa_l1_mean: *out (f32, device) = mean |pred[i] - gt[i]| over n elements (utils/losses.py:14-19, loss_l1
 *     without mask), summed in f64 in a fixed order (the same bits every call).
 * scratch: device memory of vsa_reduce_scratch_bytes() bytes, 8-byte aligned, ZEROED before its first use and
 * left ready for the next call by every call; calls that share it must be ordered on one stream.
 * hit_slot / pred / gt: 16-byte aligned. */
long long vsa_reduce_scratch_bytes(void);
int vsa_count_hits(const int32_t* hit_slot, long long n, void* scratch, int64_t* out, void* stream);
int vsa_l1_mean(const float* pred, const float* gt, long long n, void* scratch, float* out, void* stream);

/* ------------------------------------------------------------------------
 * A2  BVH build (host) + K-shell closest-hit traversal (device).
 * Replaces raytracelib.RayTracer(tensor_meshes) / .trace(rays_o, rays_d, mesh_id)
 * at volsurfs_py/methods/volsurfs.py:128 and :476-485 (raytracelib is an
 * un-vendored submodule, .gitmodules:14-17; a reference binding would wrap
 * these in a class exposing `trace`).
 *
 * vsa_bvh_build  [host pointers]  verts [nv,3] f32, faces [nf,3] i32.
 *   Binned-SAH binary BVH, leaves <= leaf_size (1..8) triangles.
 * vsa_bvh_sizes  node / triangle counts and tree depth of a built BVH.
 * vsa_bvh_export [host pointers]  nodes_out [nr_nodes,16] f32 (64-B nodes:
 *   child0 box (6), child1 box (6), ref0, ref1, cnt0, cnt1 as i32 bits;
 *   ref >= 0 inner node index, ref < 0 leaf with first triangle ~ref),
 *   tris_out [nr_tris,12] f32 in leaf order: v0.xyz, original face id (i32
 *   bits), e1.xyz, 0, e2.xyz, 0.  node_base / tri_base are added to the
 *   references so that K meshes can be concatenated into one array pair.
 */
typedef struct vsa_bvh vsa_bvh;
int vsa_bvh_build(const float* verts, const int32_t* faces, int nr_verts, int nr_faces,
                  int leaf_size, vsa_bvh** out_bvh);
int vsa_bvh_sizes(const vsa_bvh* bvh, int* nr_nodes, int* nr_tris, int* max_depth);
int vsa_bvh_export(const vsa_bvh* bvh, float* nodes_out, float* tris_out, int node_base,
                   int tri_base);
/* Quantised export for vsa_trace_q: qnodes_out [nr_nodes,8] u32 (32-B nodes: per child three
 * dwords of 6 x u16 box coordinates on the mesh's grid, rounded outward; dwords 6, 7 = the
 * children: node index >= 0, leaf code ~((first_tri << 4) | count), or 0x7fffffff = none),
 * frame_out [6] = grid origin xyz and step xyz, tris_out as vsa_bvh_export. */
int vsa_bvh_export_q(const vsa_bvh* bvh, uint32_t* qnodes_out, float* tris_out, int node_base,
                     int tri_base, float* frame_out);
/* Refit after the vertices moved (same faces, same nr_verts as the build): triangle records and
 * child boxes are recomputed in place, bottom-up; leaf order — hence every triangle slot id and any
 * per-slot table of the caller — is unchanged.  Re-export afterwards.  Closest hits through the
 * refitted tree are bit-identical to those of a fresh build (the boxes only prune).
 * Replaces re-running `RayTracer(tensor_meshes)` (volsurfs.py:82-128) when only positions changed. */
int vsa_bvh_refit(vsa_bvh* bvh, const float* verts, int nr_verts);
int vsa_bvh_destroy(vsa_bvh* bvh);

/* vsa_trace: closest hit of every ray against each of nr_meshes BVHs in ONE
 * launch (grid.y = mesh).  mesh_roots [host, nr_meshes] = root node index of
 * each mesh in `nodes`; max_depth = deepest tree (must be < 48).
 *   rays_o, rays_d [N,3] f32 (rays_d need not be normalised; t is in units of
 *   |rays_d|).  Hit iff t > t_min; closest = smallest t, ties -> smallest
 *   original face id (order independent, bit-identical to the brute-force
 *   oracle).  Outputs, mesh-major [nr_meshes, N]: hit_t (0 on miss),
 *   hit_slot (index into `tris`, -1 on miss), hit_uv [.,.,2] = barycentric
 *   weights of v1 and v2. */
int vsa_trace(const float* nodes, const float* tris, const int32_t* mesh_roots, int nr_meshes,
              int max_depth, const float* rays_o, const float* rays_d, int nr_rays, float t_min,
              float* hit_t, int32_t* hit_slot, float* hit_uv, void* stream);

/* vsa_trace on the quantised nodes of vsa_bvh_export_q (half the node bytes; identical
 * results).  mesh_frames [host, nr_meshes*6].  Valid while ray origins stay within ~60 mesh
 * extents of the mesh (fp32 error of the origin in grid units < the boxes' outward margin);
 * beyond that use vsa_trace. */
int vsa_trace_q(const uint32_t* qnodes, const float* tris, const int32_t* mesh_roots,
                const float* mesh_frames, int nr_meshes, int max_depth, const float* rays_o,
                const float* rays_d, int nr_rays, float t_min, float* hit_t, int32_t* hit_slot,
                float* hit_uv, void* stream);
/* The same launch with NARROW waves: rays_per_wave (1 .. 64) rays per 64-lane wave, the other lanes idle.  For small
 * batches of incoherent rays (a training batch: a few ten thousand random pixels of many views): a wave walks until
 * its slowest ray is done, so fewer rays per wave mean shorter dependent chains and more waves to hide them behind —
 * the chip is mostly empty at that size anyway.  Same hits (a ray's walk does not depend on its wave). */
int vsa_trace_q_narrow(const uint32_t* qnodes, const float* tris, const int32_t* mesh_roots,
                       const float* mesh_frames, int nr_meshes, int max_depth, const float* rays_o,
                       const float* rays_d, int nr_rays, float t_min, float* hit_t, int32_t* hit_slot,
                       float* hit_uv, int rays_per_wave, void* stream);
/* vsa_trace_q with the launch order taken from the previous call's measured cost (identical results):
 * every wave files itself, by the trips its walk took, into one of three lists of the NEXT call's
 * order; the next call dispatches the lists first (longest walks first), then everything else in the
 * natural order.  A wave's trips depend on its rays only, so for the same rays the prediction is
 * exact, and for a camera that moved a little it is close; for unrelated rays it is as good as any
 * order.  Why: the few waves that hold grazing rays walk 10-20x the median; dispatched late they leave
 * the chip draining for a third of the launch (profiles/NOTEBOOK.md A9.4).  feedback: device memory, 16-byte
 * aligned, >= vsa_trace_feedback_bytes(nr_rays, nr_meshes) (< 0 on bad arguments), ZEROED by the
 * caller before its first use, then owned by this sequence of calls with the SAME feedback_bytes
 * (a buffer sized for more rays serves fewer: a half written for another item count is recognised
 * and ignored): phase alternates 0, 1, 0, ... (the half written by one call is read by the next),
 * or phase = 2: the phase lives in the buffer itself and a one-wave kernel in front of the launch
 * flips it — the form to use inside a captured graph, where a host-side toggle would be frozen and
 * every replay would read the half the last eager call wrote;
 * calls that share a feedback buffer must be ordered on one stream.  The lists and flags a
 * call reads always partition the items, so the hits do not depend on what they were measured on. */
long long vsa_trace_feedback_bytes(int nr_rays, int nr_meshes);
int vsa_trace_q_fb(const uint32_t* qnodes, const float* tris, const int32_t* mesh_roots,
                   const float* mesh_frames, int nr_meshes, int max_depth, const float* rays_o,
                   const float* rays_d, int nr_rays, float t_min, float* hit_t, int32_t* hit_slot,
                   float* hit_uv, void* feedback, long long feedback_bytes, int phase, void* stream);
/* The cooperative finish of vsa_trace_q_fb (same call site, volsurfs.py:476-485): a wave walks one ray per lane until
 * its slowest ray is done — a grazing ray takes 100-360 trips of the walk against a median of 9, and a small launch is
 * as long as its longest wave.  In launches of at most `max_waves` waves (nr_meshes x ceil(nr_rays / 64): training
 * batches; larger launches keep the plain walk, whose register budget holds five waves per SIMD) a wave looks every
 * `chunk` trips at how many of its lanes are still walking, and once they are at most `lanes` the WHOLE wave finishes
 * those rays together: their pending subtrees go into a queue of (node, ray) entries, every lane takes one entry per
 * round and appends the children that survive.  The hits are bit for bit those of the plain walk (the closest hit is
 * a minimum over (t, face id); a stale bound only visits more).  Process-wide setting, defaults 16 / 24 / 4096
 * (environment VSA_TRACE_COOP="chunk,lanes", VSA_TRACE_COOP_WAVES); lanes = 0 switches it off.  Returns
 * VSA_ERR_ARG for chunk < 1, lanes outside 0..64 or max_waves < 0. */
int vsa_trace_coop_config(int chunk, int lanes, long long max_waves);
/* Measurement aid (bench.py's stage_roofline.trace; not on the product path): the walk of vsa_trace_q with
 * counters.  stats (device, 5 x u64, overwritten): lane-level node visits (one 32-byte node fetch each),
 * lane-level triangle tests (48 bytes each), wave-level trips of the walk loop summed over the waves (one trip =
 * one dependent node fetch of a whole wave), waves, and the longest wave's trips. */
int vsa_trace_q_stats(const uint32_t* qnodes, const float* tris, const int32_t* mesh_roots,
                      const float* mesh_frames, int nr_meshes, int max_depth, const float* rays_o,
                      const float* rays_d, int nr_rays, float t_min, uint64_t* stats, void* stream);
/* vsa_hit_attributes: expands one mesh's hit records [N] into the dict
 * raytracelib returns (volsurfs.py:496-501): is_hit [N] u8, triangles_id [N]
 * i32 (original face index, -1 on miss), positions [N,3] = o + t d, normals
 * [N,3] = unit geometric face normal (e1 x e2), barycentric [N,3] =
 * (1-u-v, u, v).  Any output may be NULL. */
int vsa_hit_attributes(const float* tris, const float* rays_o, const float* rays_d,
                       const float* hit_t, const int32_t* hit_slot, const float* hit_uv,
                       int nr_rays, uint8_t* is_hit, int32_t* tri_id, float* positions,
                       float* normals, float* barycentric, void* stream);

/* ------------------------------------------------------------------------
 * A3/A4/A6  Neural-texture appearance of the K shells (the "shade" stage).
 * Replaces, per shell and per model, SHNeuralTextures.forward
 * (volsurfs_py/models/sh_neural_textures.py:64-97) -> NeuralTexture.forward
 * (models/neural_texture.py:81-197) -> tcnn HashGrid + FullyFusedMLP
 * (neural_texture.py:54-77; tiny-cuda-nn is a pip dependency, absent), plus the
 * uv interpolation / alpha decay / scatter around it (methods/volsurfs.py:504-516,
 * 539-596) and the autograd replay of all of it.
 *
 * MI355X design (DESIGN.md "Texel-deduplicated shading"): NeuralTexture only
 * ever evaluates its network at TEXEL CENTRES (the 4 lerp corners), so the
 * frame's hits are reduced to the set of unique touched texels per (shell,
 * degree) first; hash encoding + MLP run once per unique texel ("slot") and
 * write the 8-bit quantised texel (exactly the value the reference computes for
 * every hit that touches it); hits then gather 4 corner rows per degree.  Hash
 * levels are processed level-major with the whole level table resident in LDS
 * (2^15 entries x half2 = 128 KiB <= 160 KiB), forward gathers and backward
 * scatter-adds never touch HBM atomics.
 *
 * Indexing: texture x = (shell*2 + type)*4 + degree, type 0 = rgb, 1 = alpha.
 * Texel domain of (shell, degree): (R_d+2)^2 texels (one-texel apron for
 * corners outside [0,1]), padded to a multiple of 4096, concatenated:
 * dom_off[shell*4 + degree].  Slots are numbered globally in domain order.
 */
#define VSA_NT_MAX_LEVELS 16
#define VSA_NT_MAX_DEG 4
#define VSA_NT_WEIGHTS_PER_TEX 8192 /* W1[64][32] W2[64][64] W3[32][64] (rows >= C' zero) */
/* Texel rows (u8) and gradient rows (f32) share one per-degree layout, counted in QUADS
 * (4 elements: 4 bytes of a texel row, 4 floats of a gradient row).  A slot of SH band d
 * (n = 2d+1 coefficients) owns VSA_NT_ROW_QUADS(d) quads: rgb coefficient c (0..3n-1) at
 * element c, alpha coefficient c (0..n-1) at element 4*VSA_NT_ALPHA_QUAD(d) + c; the rest
 * is zero padding.  The row of slot i of segment (shell, d) starts at quad
 * row_base[shell*4+d] + (i - seg_start[shell*4+d]) * VSA_NT_ROW_QUADS(d). */
#define VSA_NT_ROW_QUADS(d) ((d) == 0 ? 2 : ((d) == 1 ? 4 : 8))
#define VSA_NT_ALPHA_QUAD(d) ((d) == 0 ? 1 : ((d) == 1 ? 3 : ((d) == 2 ? 4 : 6)))

typedef struct vsa_nt_plan {
  int32_t nr_shells;                 /* K */
  int32_t rgb_degrees;               /* sh_degree+1 of the rgb models, 1..4 */
  int32_t alpha_degrees;             /* sh_degree+1 of the alpha models, 1..4 */
  int32_t inner_solid;               /* 1: shell 0 has no alpha model (alpha = 1), volsurfs.py:181-183 */
  int32_t with_alpha_decay;          /* volsurfs.py:585-594 */
  int32_t n_levels;                  /* 16 */
  int32_t tex_res[VSA_NT_MAX_DEG];   /* R_d (square textures) */
  float sh_lo[VSA_NT_MAX_DEG];       /* val_range[0] = -sh_range[d] */
  float sh_span[VSA_NT_MAX_DEG];     /* val_range[1]-val_range[0] */
  float level_scale[VSA_NT_MAX_LEVELS];
  int32_t level_res[VSA_NT_MAX_LEVELS];
  int32_t level_size[VSA_NT_MAX_LEVELS];      /* entries */
  int32_t level_offset[VSA_NT_MAX_LEVELS + 1]; /* entries */
  int64_t dom_off[VSA_MAX_SHELLS * VSA_NT_MAX_DEG + 1];
  int64_t slot_capacity;             /* rows allocated in every per-slot buffer */
  int32_t max_rays;                  /* N the buffers were sized for: a (shell,degree)
                                        segment holds <= min(4*max_rays, (R_d+2)^2) slots */
  int32_t anchor;                    /* 0: lerp of the 2x2 texel footprint (the shipped configs, neural_texture.py:106-139);
                                        1: anchor — the sample takes the ONE texel it falls in, unblended (:88-104) */
  int64_t row_base[VSA_MAX_SHELLS * VSA_NT_MAX_DEG + 1]; /* quads; multiples of 8; segment sd
                                        reserves min(4*max_rays,(R_d+2)^2)*VSA_NT_ROW_QUADS(d);
                                        the last entry is the allocation size */
  void* balance;                     /* NULL, or device memory of vsa_nt_balance_bytes() bytes, ZEROED before
                                        its first use: the persistent kernels (encode / MLP, forward and
                                        backward) stamp every workgroup's busy time there and split their
                                        work by the shares vsa_nt_rebalance derived from the previous
                                        launch's times (same pieces, same results: only who does which) */
  int32_t row_format;                /* 0: 8-bit quantised texel rows (using_sh_quantization = 1: every shipped config);
                                        1: f16 rows holding sigmoid(x) un-quantised (using_sh_quantization = 0,
                                        using_sh_squeezing = 1; neural_texture.py:159-164, 183-187) — `texels` is then an
                                        f16 array with the SAME quad layout (a quad = 4 halves = 8 bytes); forward
                                        kernels only differ, the backward is the quantised one's (round is a
                                        straight-through estimator);
                                        2: f16 rows holding the RAW network output (using_sh_squeezing = 0: no sigmoid, no
                                        quantiser, no expansion to val_range; neural_texture.py:157-169, 181-187) — layout as
                                        format 1; the backward passes the row gradient through unchanged (no sigmoid', no span).
                                        vsa_nt_encode_mlp_fwd supports format 0 only */
  int32_t grads_zeroed;              /* 1: the caller vouches that grad_tables is ALL ZERO on entry to vsa_nt_encode_bwd /
                                        _range / _phased (it cleared the buffer, or the fused Adam step did, and nothing
                                        has been accumulated since): a table plane whose slots ONE workgroup walks is then
                                        written with plain stores instead of float atomics (same values: 0 + v).  0: the
                                        launch accumulates into whatever the buffer holds */
  int32_t shared_rgb;                /* 1: are_volsurfs_colors_indep = 0 (methods/volsurfs.py:159-165, 524-527): ONE colour model
                                        `models["rgb"]` for all shells — every shell's colour textures read the parameters of
                                        texture (0*2 + 0)*4 + degree and the gradients of all K shells accumulate there.  Slots,
                                        feature planes and texel rows stay per (shell, degree): only the PARAMETER index maps */
  int32_t shared_alpha;              /* 1: are_volsurfs_alphas_indep = 0 (volsurfs.py:200-206, 553-556): the same for the alpha
                                        model (parameters of texture (0*2 + 1)*4 + degree).  With inner_solid the reference's
                                        loop stores models["alpha"] = None and leaves: NO shell has an alpha model (alpha = 1,
                                        no decay, on every shell) */
} vsa_nt_plan;

/* Measured-time rebalancing of the persistent kernels' work split (profiles/NOTEBOOK.md A9.0): once per frame,
 * before the first of them, turns the busy times the previous frame's launches stamped into
 * plan->balance into per-workgroup shares of each kernel's cost axis (a workgroup that took longer
 * than the mean gets a smaller share, damped).  No-op while nothing has been stamped.
 * vsa_nt_compact_frame does the same inside its own scan launch when plan->balance is set: a frame loop
 * built on it does not call vsa_nt_rebalance. */
long long vsa_nt_balance_bytes(void);
int vsa_nt_rebalance(const vsa_nt_plan* plan, void* stream);

/* Step 1 (per frame): per-hit texture uv + mark touched texels.
 *   hit_slot [K,N] i32, hit_uv [K,N,2] f32 from vsa_trace; face_uvs [nr_tris,6]
 *   f32 = per-corner uvs in leaf (slot) order.  Writes tex_uv [K,N,2] and sets
 *   marks[dom_off[..] + texel] = 1 for the 4 lerp corners of every degree.
 *   marks (u8 [dom_off[K*4]]) must be zero on entry; marks == NULL: tex_uv only (shading from
 *   baked textures needs no compaction). */
int vsa_nt_mark(const vsa_nt_plan* plan, const int32_t* hit_slot, const float* hit_uv,
                const float* face_uvs, int nr_rays, float* tex_uv, uint8_t* marks, void* stream);

/* Step 2: compact marks into slots.  slot_of [dom total] i32 (-1 = untouched),
 * texel_of_slot [slot_capacity] i32 (domain index), slot_xy [slot_capacity,2] f32
 * (normalised texel-centre coordinates = the network input of the slot),
 * seg_start [K*4+1] i32 (first slot of each (shell,degree); last = total),
 * block_scratch [dom total/4096 + 1] i32. */
int vsa_nt_compact(const vsa_nt_plan* plan, const uint8_t* marks, int32_t* slot_of,
                   int32_t* texel_of_slot, float* slot_xy, int32_t* seg_start,
                   int32_t* block_scratch, void* stream);
/* vsa_nt_compact for a frame loop: the same slots, seg_start and slot_xy, but slot_of is written only
 * where a texel is marked (entries of untouched texels keep whatever they held: nothing on the path
 * reads them - shading only looks up the corners it marked), the marks are CLEARED on the way (the
 * next vsa_nt_mark needs no fill), and texel_of_slot may be NULL.  At 800x800, K = 5 that is 112 MB
 * of -1 and a 28 MB fill less per frame.  The frame loop's invariant is "marks are zero between
 * frames": a vsa_nt_mark that is NOT followed by a successful vsa_nt_compact_frame (an error in
 * between, or the dense vsa_nt_compact above, which leaves the marks as they are) must be followed by
 * a fill of the marks before the next frame - stale marks inflate every later frame's slot counts. */
int vsa_nt_compact_frame(const vsa_nt_plan* plan, uint8_t* marks, int32_t* slot_of,
                         int32_t* texel_of_slot, float* slot_xy, int32_t* seg_start,
                         int32_t* block_scratch, void* stream);

/* Step 3: hash-grid encode every slot of every texture, level-major with the
 * level table in LDS.  tables_h: f16 [n_tex][level_offset[n]*2];
 * features: f16x2, blocked [2 types][slot_capacity/256][n_levels][256] (slot_capacity
 * is a multiple of 256): a 32-slot MLP tile's 16 levels stay within 16 KiB. */
int vsa_nt_encode_fwd(const vsa_nt_plan* plan, const void* tables_h, const float* slot_xy,
                      const int32_t* seg_start, void* features, void* stream);

/* Step 4: MLP 32->64->64->C' of every slot of every texture on MFMA, fused with
 * sigmoid / x255 / round (neural_texture.py:156-169).  weights_h: f16
 * [n_tex][VSA_NT_WEIGHTS_PER_TEX]; texels: u8 [row_base[last]*4] (quantised texel rows,
 * layout above).  pre_out (optional, tests): f16 [slot_capacity][32], the network output
 * before the sigmoid: rgb channel c at [slot][c], alpha channel c at [slot][24+c]. */
int vsa_nt_mlp_fwd(const vsa_nt_plan* plan, const void* weights_h, const void* features,
                   const int32_t* seg_start, uint8_t* texels, void* pre_out, void* stream);

/* Steps 3 + 4 in ONE launch (the reference's `Sequential(encoding, network)` is one tiny-cuda-nn
 * call per corner batch, models/neural_texture.py:63-79, 153): a wave hash-grid encodes 64
 * consecutive slots (a lane per slot, all levels, table entries gathered through L2) and runs
 * the MLP on them straight from registers (csrc/nt_fused.hip).  Bit-identical to
 * vsa_nt_encode_fwd + vsa_nt_mlp_fwd.  features: NULL, or the blocked feature planes of
 * vsa_nt_encode_fwd, which are then written as well (vsa_nt_mlp_bwd recomputes the forward
 * from them).  pre_out: as vsa_nt_mlp_fwd (optional, tests). */
int vsa_nt_encode_mlp_fwd(const vsa_nt_plan* plan, const void* tables_h, const void* weights_h,
                          const float* slot_xy, const int32_t* seg_start, void* features,
                          uint8_t* texels, void* pre_out, void* stream);

/* Backward of step 4: recomputes the forward per 32-slot tile, back-propagates
 * grad_rows (f16, already multiplied by grad_scale) through sigmoid (round = STE)
 * and the three layers on MFMA.  Overwrites `features` IN PLACE with the feature
 * gradients (f16x2, still scaled) and accumulates grad_weights (f32
 * [n_tex][VSA_NT_WEIGHTS_PER_TEX]) += weight_grad_scale x (the scaled weight gradients), one
 * flush per workgroup; weight_grad_scale = 1 / grad_scale adds the true gradients straight to
 * weights.grad.
 * grad_rows is CONSUMED: every row read is reset to zero, so the buffer (zero
 * at allocation) is zero again outside a [vsa_nt_shade_bwd, vsa_nt_mlp_bwd] pair.
 * dfeat_abs_sum (f32 [n_tex][32], zeroed by the caller) += sum over slots of |dF| per
 * feature row: the overflow bound of vsa_nt_encode_bwd's fixed-point accumulation. */
int vsa_nt_mlp_bwd(const vsa_nt_plan* plan, const void* weights_h, void* features,
                   const int32_t* seg_start, uint16_t* grad_rows, float* grad_weights,
                   float* dfeat_abs_sum, float weight_grad_scale, void* stream);

/* Step 5: per-hit shading from the texel rows (expand LUT -> lerp -> fp16 SH
 * coefficients -> SH eval -> sigmoid -> alpha decay), scattered dense:
 * surfs_rgb [N,K,3], surfs_alpha [N,K] (zero on miss; inner->outer), optional
 * surfs_normals [N,K,3] and coeffs_out [K,N,64] (tests: 48 rgb [ch][16] + 16
 * alpha lerped fp16 SH coefficients).  tris = the tracer's triangle array.
 * act_out (optional, [K,N,4] f32): the three rgb sigmoids and the alpha sigmoid (before
 * the decay) of every hit (entries of misses are NOT written), which vsa_nt_shade_bwd can take
 * back as act_in instead of re-gathering the texel rows and re-evaluating the SH sums. */
int vsa_nt_shade_fwd(const vsa_nt_plan* plan, const int32_t* hit_slot, const float* tex_uv,
                     const float* rays_d, const float* tris, const int32_t* slot_of,
                     const int32_t* seg_start, const uint8_t* texels, int nr_rays,
                     float* surfs_rgb, float* surfs_alpha,
                     float* surfs_normals, float* coeffs_out, float* act_out, void* stream);

/* Backward of step 5: grad_rows (f16 [row_base[last]*4], same row layout as
 * texels; zero on entry, see vsa_nt_mlp_bwd) += grad_scale * dL/d(q/255)
 * (round is a straight-through estimator, utils/math.py:5-18).  act_in: NULL, or the
 * act_out of the forward call on the SAME frame and parameters. */
int vsa_nt_shade_bwd(const vsa_nt_plan* plan, const int32_t* hit_slot, const float* tex_uv,
                     const float* rays_d, const float* tris, const int32_t* slot_of,
                     const int32_t* seg_start, const uint8_t* texels, int nr_rays,
                     const float* g_surfs_rgb,
                     const float* g_surfs_alpha, float grad_scale, uint16_t* grad_rows,
                     const float* act_in, void* stream);

/* Backward of step 3: grad_tables (f32 [n_tex][level_offset[n]][2]) +=
 * transpose-interpolation of dfeatures (f16x2, same layout as features, holding
 * grad * grad_scale).  Accumulates (caller zeroes grad_tables per optimiser step; plan->grads_zeroed
 * tells the launch that it has, see there). */
int vsa_nt_encode_bwd(const vsa_nt_plan* plan, const void* dfeatures, const float* dfeat_abs_sum,
                      float grad_scale, const float* slot_xy, const int32_t* seg_start,
                      float* grad_tables, void* stream);

/* The same, restricted to the textures of shells [shell_begin, shell_end): lets a
 * data-parallel caller start the all-reduce of one shell's table gradients (a contiguous
 * slice of grad_tables) while the next shell's are still being accumulated. */
int vsa_nt_encode_bwd_range(const vsa_nt_plan* plan, const void* dfeatures,
                            const float* dfeat_abs_sum, float grad_scale, const float* slot_xy,
                            const int32_t* seg_start, float* grad_tables, int shell_begin,
                            int shell_end, void* stream);

/* Data-parallel form of vsa_nt_encode_bwd (rays shard by tile, the ranks all-reduce the
 * gradients; the reference is single-GPU, so this replaces nothing — it is what keeps the all-reduce
 * of the 113 MB of table gradients off the critical path of trainer.py:249-264's backward).
 *
 * vsa_dp_flags: n (<= VSA_MAX_SHELLS + 1) completion words, zero at creation, each its own 8-byte
 * hipMallocSignalMemory allocation (plain device memory if the runtime refuses), so that another stream
 * can wait for one through the command processor — a waiting KERNEL would park a wave on a CU whose
 * registers the persistent MLP kernels need entirely.  vsa_dp_flags_read: [host] copy of the n words
 * (synchronises the device; tests). */
typedef struct vsa_dp_flags vsa_dp_flags;
int vsa_dp_flags_create(int n, vsa_dp_flags** out);
int vsa_dp_flags_destroy(vsa_dp_flags* flags);
int vsa_dp_flags_read(const vsa_dp_flags* flags, uint32_t* host_out);

/* ONE launch: the shells are cut into n_phases groups, phase p = shells [phase_shell_end[p-1],
 * phase_shell_end[p]) ([host] array, strictly increasing, last = nr_shells); every workgroup walks its
 * share of the dense levels of all shells, then finishes its share of phase p's hashed levels before
 * it touches phase p+1, and when the LAST workgroup is through phase p the kernel stores
 * word p of `flags` = *epoch (system scope; that phase's slice of grad_tables — textures
 * [8*begin, 8*end) — is final and visible).  counters: n_phases device words, zero on entry, zero again
 * on exit.  Results equal vsa_nt_encode_bwd's up to the order of the float atomics that join two
 * workgroups' shares of one table plane (pieces are cut at other slots).
 * Another stream waits for a phase with vsa_dp_stream_wait(flags, p, epoch value).
 * reserve_cus (>= 0): the launch uses that many workgroups fewer than the device has compute units.  A
 * workgroup takes 16 waves, 128 KiB of LDS and most of a CU's vector registers: small kernels run beside it
 * (measured: a 64-workgroup and a 256 MB elementwise kernel finish 0.1-0.3 ms into the 0.65 ms launch), a kernel
 * whose waves need more than the ~64 registers per lane it leaves on a SIMD does not (the traversal: 96) — whether a
 * collective's kernels do is the communication library's business, and this is the knob for it. */
int vsa_nt_encode_bwd_phased(const vsa_nt_plan* plan, const void* dfeatures,
                             const float* dfeat_abs_sum, float grad_scale, const float* slot_xy,
                             const int32_t* seg_start, float* grad_tables, int n_phases,
                             const int32_t* phase_shell_end, vsa_dp_flags* flags, uint32_t* counters,
                             const uint32_t* epoch, int reserve_cus, void* stream);

/* Stream-ordered signal: (*epoch += 1 when advance_epoch), then word `index` of flags = *epoch at system
 * scope — a one-lane kernel, so it can sit inside a captured HIP graph (hipStreamWriteValue32 cannot).
 * The step calls it once behind vsa_nt_mlp_bwd (weights.grad is final), advancing the epoch the phased
 * encode backward then publishes. */
int vsa_dp_signal(vsa_dp_flags* flags, int index, uint32_t* epoch, int advance_epoch, void* stream);

/* Make `stream` wait until word `index` >= value.  mode 1: hipStreamWaitValue32 (on signal memory the
 * command processor polls: no compute resource), mode 2: a one-lane polling kernel ((int32)(word - value)
 * >= 0; it occupies registers of one SIMD: see above), mode 0: 1 where the words are signal memory and
 * hipDeviceAttributeCanUseStreamWaitValue says so, else 2.  ENQUEUE IT AFTER THE PRODUCER: HIP
 * multiplexes streams onto a few hardware queues, and a wait queued ahead of the kernel that
 * satisfies it on the same queue would never return. */
int vsa_dp_stream_wait(vsa_dp_flags* flags, int index, uint32_t value, int mode, void* stream);

/* ------------------------------------------------------------------------
 * A5 / A10  Encoders of the legacy appearance branch and of the background field:
 * the 2-D / 3-D multiresolution hash grid behind GridHashEncoder
 * (volsurfs_py/encodings/gridhash.py:12-92: tcnn.Encoding "Grid"/"Hash", fp32) and
 * SHEncoder.__call__ (volsurfs_py/encodings/sphericalharmonics.py:84-153).
 * tables: fp32 [level_offset[n_levels]][2]; x: fp32 [nr_points][n_dims] in [0,1];
 * out / g_out: fp32 [nr_points][n_levels*2] (level-major, as tcnn returns it).
 */
#define VSA_GRID_MAX_LEVELS 32
typedef struct vsa_grid_plan {
  int32_t n_dims;      /* 2 or 3 */
  int32_t n_levels;
  int32_t n_features;  /* 2 */
  int32_t reserved0;
  float level_scale[VSA_GRID_MAX_LEVELS];
  int32_t level_res[VSA_GRID_MAX_LEVELS];
  int32_t level_size[VSA_GRID_MAX_LEVELS];        /* entries */
  int32_t level_offset[VSA_GRID_MAX_LEVELS + 1];  /* entries */
} vsa_grid_plan;

int vsa_grid_encode_fwd(const vsa_grid_plan* plan, const float* tables, const float* x,
                        int nr_points, float* out, void* stream);
/* grad_tables (fp32, same shape as tables) += transpose-interpolation of g_out. */
int vsa_grid_encode_bwd(const vsa_grid_plan* plan, const float* x, const float* g_out,
                        int nr_points, float* grad_tables, void* stream);
/* The same gradients for LARGE batches without memory-side atomics: workgroups own (level, 2^14-entry
 * slice) fixed-point accumulators in LDS and scan the samples (csrc/grid_encode.hip).  workspace:
 * nr_points * 2 * n_levels + 32 floats (the output gradient re-laid level-major + max|g| per level). */
int vsa_grid_encode_bwd_sliced(const vsa_grid_plan* plan, const float* x, const float* g_out,
                               int nr_points, float* grad_tables, float* workspace, void* stream);
/* The same gradients for VERY large batches: the 2^D x n_levels contributions of every sample are
 * binned by (level, 2^13-entry slice) once (LDS counting sort, contiguous runs) and each bin is
 * accumulated densely in LDS fixed point.  workspace: vsa_grid_encode_bwd_binned_workspace floats
 * (~14 x 2^D x n_levels x nr_points bytes: 4.8 GB for 2.1 M samples). */
int vsa_grid_encode_bwd_binned_workspace(const vsa_grid_plan* plan, int nr_points,
                                         long long* workspace_floats);
int vsa_grid_encode_bwd_binned(const vsa_grid_plan* plan, const float* x, const float* g_out,
                               int nr_points, float* grad_tables, float* workspace, void* stream);
/* The same four with a row stride (floats) on out / g_out, for callers that keep the features
 * inside a wider row: GridHashEncoder's `torch.cat([enc, points], 1)` (gridhash.py:88-90) becomes
 * out_stride = 2 * n_levels + n_dims with append_x = 1 (the kernel writes x behind the features),
 * and the gradient of that matrix is read in place (g_stride; the x columns are ignored: positions
 * carry no gradient on this path). */
int vsa_grid_encode_fwd_ld(const vsa_grid_plan* plan, const float* tables, const float* x,
                           int nr_points, float* out, int out_stride, int append_x, void* stream);
int vsa_grid_encode_bwd_ld(const vsa_grid_plan* plan, const float* x, const float* g_out,
                           int g_stride, int nr_points, float* grad_tables, void* stream);
int vsa_grid_encode_bwd_sliced_ld(const vsa_grid_plan* plan, const float* x, const float* g_out,
                                  int g_stride, int nr_points, float* grad_tables, float* workspace,
                                  void* stream);
int vsa_grid_encode_bwd_binned_ld(const vsa_grid_plan* plan, const float* x, const float* g_out,
                                  int g_stride, int nr_points, float* grad_tables, float* workspace,
                                  void* stream);
/* out [nr_dirs][(degree+1)^2]: SH basis of each direction, degree 0..4. */
int vsa_sh_encode(const float* dirs, int nr_dirs, int degree, float* out, void* stream);

/* A10  The elementwise glue between NerfHash's two MLPs (volsurfs_py/models/nerfhash.py:72-91), one
 * pass each way instead of eleven torch launches over [samples, 64..80] floats:
 *   fwd: density [n] = softplus(y1[:, 0]);  x2 [n][nr_feat + nr_dir] = cat(gelu(y1[:, 1:1+nr_feat]), dirs_enc)
 *   bwd: dy1 [n][1 + nr_feat] from dx2 (gradient of x2; its dirs columns are ignored) and d_density
 *        (either may be NULL = zero).
 * y1 [n][y1_stride >= 1 + nr_feat] is the first MLP's output (r6: a row stride, so that the rows can be padded to a
 * multiple of 4 floats — the MLP kernels then move them as 16-byte accesses), dy1 has the same stride and its padding
 * columns are written as zeros; dirs_enc [n][nr_dir] the encoded directions. */
int vsa_field_head_fwd(const float* y1, int y1_stride, const float* dirs_enc, long long nr_points, int nr_feat,
                       int nr_dir, float* x2, float* density, void* stream);
int vsa_field_head_bwd(const float* y1, int y1_stride, const float* dx2, const float* d_density, long long nr_points,
                       int nr_feat, int nr_dir, float* dy1, void* stream);

/* A5 / A10  Fused fp32 MLP (Linear + bias, exact GELU between layers, last layer linear) on the
 * fp32-input matrix cores: `MLP` (volsurfs_py/models/mlp.py:8-69) as used by RGB (models/rgb.py:139),
 * ColorSH (models/color_sh.py) and NerfHash (models/nerfhash.py:44-56).  Layer l maps dims[l] ->
 * dims[l+1] with w[l] [dims[l+1]][dims[l]] row-major (torch.nn.Linear.weight) and b[l] [dims[l+1]]
 * or NULL; every width <= 128, hidden widths multiples of 32.  All pointers are device pointers.
 *   vsa_mlp_workspace: sizes (floats) of packed_ws (weights in MFMA fragment order, rewritten by
 *     every call), of each of z_ws / a_ws / dz_ws (nr_points x sum of hidden widths) and of
 *     partial_ws (per-workgroup weight-gradient blocks).
 *   vsa_mlp_fwd: y [nr_points][y_stride] = MLP(x [nr_points][x_stride]); z_ws receives the hidden
 *     pre-activations (NULL: inference), a_ws the activations GELU(z) when the backward of this network is
 *     the two-kernel one (vsa_mlp_bwd_needs_act() == 1; otherwise pass NULL: nothing is stored).
 *   vsa_mlp_bwd: from dy = dL/dy and the z_ws (/ a_ws) of the matching forward: dx (optional), and
 *     grads->dw[l] / db[l] (NULL entries are skipped): overwritten, or added to what the buffers
 *     hold when grads->accumulate != 0 (a caller that owns persistent .grad buffers lets the
 *     kernel add into them instead of running one accumulation kernel per parameter).
 *     Networks up to 96 wide whose weights fit the LDS (NerfHash's two, models/nerfhash.py:44-56) take ONE fused
 *     persistent launch — data gradients, weight gradients and bias sums from z alone (GELU and GELU' from one
 *     evaluation), nothing but dx written per sample (csrc/mlp_f32_fused.h): a_ws and dz_ws are then not read and
 *     may be NULL.  It needs 16-byte rows: x, dy (and dx) 16-byte aligned with a stride that is a multiple of 4 floats
 *     (pad the rows: the padding columns of x are ignored, those of dx receive zeros).  Wider networks (RGB / ColorSH:
 *     128) and unaligned rows run mlp_dgrad + mlp_wgrad as before.  Every kernel moves x / y / dy rows as 16-byte
 *     groups when their stride allows it and element by element otherwise.
 *   vsa_mlp_bwd_needs_act(plan, x_stride, dx_stride (0: no dx)): 1 if vsa_mlp_bwd of this plan with rows of these
 *     strides (bases 16-byte aligned) reads a_ws / dz_ws, 0 if not, < 0: VSA_ERR_*. */
#define VSA_MLP_MAX_LAYERS 6
typedef struct vsa_mlp_plan {
  int32_t n_layers;
  int32_t dims[VSA_MLP_MAX_LAYERS + 1];
  const float* w[VSA_MLP_MAX_LAYERS];
  const float* b[VSA_MLP_MAX_LAYERS];
} vsa_mlp_plan;
typedef struct vsa_mlp_grads {
  float* dw[VSA_MLP_MAX_LAYERS];
  float* db[VSA_MLP_MAX_LAYERS];
  int32_t accumulate; /* 0: dw / db are overwritten, else added to */
} vsa_mlp_grads;

int vsa_mlp_workspace(const vsa_mlp_plan* plan, long long nr_points, long long* packed_floats,
                      long long* act_floats, long long* partial_floats);
int vsa_mlp_bwd_needs_act(const vsa_mlp_plan* plan, int x_stride, int dx_stride);
int vsa_mlp_fwd(const vsa_mlp_plan* plan, const float* x, int x_stride, int nr_points, float* y,
                int y_stride, float* z_ws, float* a_ws, float* packed_ws, void* stream);
int vsa_mlp_bwd(const vsa_mlp_plan* plan, const float* x, int x_stride, int nr_points,
                const float* dy, int dy_stride, const float* z_ws, float* dz_ws, const float* a_ws,
                float* packed_ws, float* partial_ws, float* dx, int dx_stride,
                const vsa_mlp_grads* grads, void* stream);

/* The same two for up to 8 networks of ONE architecture in one set of launches: the K per-shell
 * models of the legacy appearance branch (volsurfs_py/methods/volsurfs.py:402-470), each applied to
 * its own shell's hits.  Group g owns rows [sum nr_points[0..g), +nr_points[g]) of x / y / dy / dx
 * and the matching rows x (sum of hidden widths) floats of z_ws / dz_ws / a_ws; packed_ws holds
 * nr_groups x packed_floats, partial_ws nr_groups x the partial_floats vsa_mlp_workspace reports
 * for the largest group; grads [nr_groups] (one `accumulate` setting for all).  plans / nr_points /
 * grads are HOST arrays.  vsa_mlp_fwd / vsa_mlp_bwd are the one-group case. */
int vsa_mlp_fwd_grouped(const vsa_mlp_plan* plans, int nr_groups, const int* nr_points,
                        const float* x, int x_stride, float* y, int y_stride, float* z_ws,
                        float* a_ws, float* packed_ws, float* packed_bwd_ws, void* stream);
/* packed_bwd_ws (optional, as large as packed_ws): the forward's packing launch also writes the
 * transposed fragment order the backward needs; handed to vsa_mlp_bwd_grouped as packed_ws with
 * packed_ready = 1 (the weights must not have changed in between) it saves that pass its own. */
int vsa_mlp_bwd_grouped(const vsa_mlp_plan* plans, int nr_groups, const int* nr_points,
                        const float* x, int x_stride, const float* dy, int dy_stride,
                        const float* z_ws, float* dz_ws, const float* a_ws, float* packed_ws,
                        int packed_ready, float* partial_ws, float* dx, int dx_stride,
                        const vsa_mlp_grads* grads, void* stream);

/* A13  Fused multi-tensor Adam step: apex.optimizers.FusedAdam(betas (0.9, 0.99), eps 1e-15,
 * weight_decay 0) of volsurfs_py/methods/base_method.py:87-94, stepped at trainer.py:278 (the
 * same update as torch.optim.Adam).  One launch for all parameter tensors:
 *   tensors_dev [T] descriptors in DEVICE memory (param / grad / exp_avg / exp_avg_sq: fp32, n
 *   elements, 16-byte aligned; param_f16: optional f16 compute copy refreshed from the new
 *   parameter, or NULL);  chunks_dev [nr_chunks][2] int32 in device memory = (tensor index,
 *   chunk index within the tensor), one workgroup per chunk of vsa_adam_chunk_elems() elements.
 *   step = 1 for the first update (bias correction 1 - beta^step); grad_scale multiplies the
 *   gradient as it is read (1 = as is); zero_grads != 0 clears the gradient after reading it
 *   (the next iteration's zero_grad, trainer.py:118). */
typedef struct vsa_adam_tensor {
  float* param;
  float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  void* param_f16; /* _Float16* or NULL */
  int64_t n;
} vsa_adam_tensor;

int vsa_adam_chunk_elems(void);
int vsa_adam_step(const vsa_adam_tensor* tensors_dev, const int32_t* chunks_dev, int nr_chunks,
                  float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                  int zero_grads, void* stream);
/* The same update from at most max_workgroups workgroups (256 threads each) that stride over the chunks
 * (0: one workgroup per chunk = vsa_adam_step).  One workgroup per chunk queues thousands of workgroups that
 * take every wave slot of the chip until the update drains; a bounded grid leaves room on every CU for
 * the kernels of ANOTHER stream — the next iteration's ray batch, traversal and texel compaction, which
 * read no parameter (optim.FusedAdam.step(stream=...)). */
int vsa_adam_step_shared(const vsa_adam_tensor* tensors_dev, const int32_t* chunks_dev, int nr_chunks,
                         float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                         int zero_grads, int max_workgroups, void* stream);

/* The glue of the legacy appearance branch around its encoders and MLPs, one launch each way (csrc/legacy_glue.hip;
 * volsurfs.py:486-599; as torch expressions: 18 + 21 launches per training iteration of BASELINE configs[2]).
 *   hit_shell / hit_ray [nr_hits] i64 = shell and ray of every hit, sorted by shell then ray (the two columns of
 *   torch.nonzero(hit_slot >= 0), which torch lays out column by column).
 * vsa_legacy_hit_prep: pts = rays_o + hit_t * rays_d (:507), dirs = rays_d of the hit's ray, normals =
 *   normalize(cross(e1, e2)) of the hit triangle (tris as vsa_bvh_export; F.normalize's 1e-12 floor); [nr_hits,3] each.
 * vsa_legacy_shade_out_fwd: into surfs_rgb [N,K,3] / surfs_alpha [N,K] / surfs_normals [N,K,3] — ZEROED by the caller
 *   (entries where nothing was hit keep that zero, volsurfs.py:486-490; one fill of one allocation in the mirror) — at
 *   every hit: sigmoid(y_rgb[i][0..3)) (row stride ld_rgb >= 3), the hit's normal, and alpha = 1 for rows < alpha_first_row
 *   (a solid inner shell) or when y_alpha is NULL, else sigmoid(y_alpha[i - alpha_first_row][0]) times, if
 *   with_alpha_decay, 2 sigmoid(10 clamp(-d.n, 0, 1)) - 1 (:585-594).  sig_rgb [nr_hits,3], sig_alpha / decay
 *   [nr_hits - alpha_first_row]: what the backward needs.
 * vsa_legacy_shade_out_bwd: dy_rgb / dy_alpha (same strides; channels beyond the used ones zeroed) from the gradients
 *   of the dense arrays: g * s (1 - s), the alpha one through the decay. */
int vsa_legacy_hit_prep(const float* rays_o, const float* rays_d, const float* hit_t, const int32_t* hit_slot,
                        const float* tris, const int64_t* hit_shell, const int64_t* hit_ray, int nr_hits, int nr_rays,
                        float* pts, float* dirs, float* normals, void* stream);
int vsa_legacy_shade_out_fwd(const float* y_rgb, int ld_rgb, const float* y_alpha, int ld_alpha, int alpha_first_row,
                             const int64_t* hit_shell, const int64_t* hit_ray, const float* dirs, const float* normals,
                             int nr_hits, int nr_rays,
                             int nr_shells, int with_alpha_decay, float* surfs_rgb, float* surfs_alpha,
                             float* surfs_normals, float* sig_rgb, float* sig_alpha, float* decay, void* stream);
int vsa_legacy_shade_out_bwd(const float* g_surfs_rgb, const float* g_surfs_alpha, const int64_t* hit_shell,
                             const int64_t* hit_ray, const float* sig_rgb, const float* sig_alpha, const float* decay, int nr_hits, int nr_shells,
                             int alpha_first_row, float* dy_rgb, int ld_rgb, float* dy_alpha, int ld_alpha, void* stream);

/* A5  Permutohedral-lattice hash encoding: `PermutoHashEncoder`
 * (volsurfs_py/encodings/permutohash.py:28-37, 68-96) = permutohedral_encoding.PermutoEncoding
 * (un-vendored fork, .gitmodules:7-9; published algorithm restated, parity unpinned).
 * lattice_values [n_levels][capacity][2] f32; x [N][pos_dim] f32 (the wrapper maps the bounding
 * box to [0,1] first); window [n_levels] f32 (Coarse2Fine) or NULL (= ones);
 * out[b*out_stride + 2*l + f] (out_stride >= 2*n_levels, even: the caller may own a wider row
 * that also holds the concatenated points).  scale_factor[l][i] = 1 / (sigma_l * sqrt((i+1)(i+2))),
 * sigma_l = the level's scale (np.geomspace(coarsest, finest, n_levels), permutohash.py:27). */
typedef struct vsa_permuto_plan {
  int32_t pos_dim;     /* 2..4 */
  int32_t n_levels;    /* <= VSA_GRID_MAX_LEVELS */
  int32_t n_features;  /* 2 */
  int32_t capacity;    /* entries per level (2^18) */
  float scale_factor[VSA_GRID_MAX_LEVELS][4];
  float random_shift[VSA_GRID_MAX_LEVELS][4];
} vsa_permuto_plan;

int vsa_permuto_encode_fwd(const vsa_permuto_plan* plan, const float* lattice_values,
                           const float* x, const float* window, int nr_points, float* out,
                           int out_stride, void* stream);
/* grad_values (fp32, same shape as lattice_values) += transpose of the interpolation applied to
 * g_out[b*g_stride + 2*l + f].  Positions carry no gradient on this path (hit points). */
int vsa_permuto_encode_bwd(const vsa_permuto_plan* plan, const float* x, const float* window,
                           const float* g_out, int g_stride, int nr_points, float* grad_values,
                           void* stream);

/* Up to 8 encodings of one geometry (the position encoders of the K per-shell models,
 * volsurfs_py/methods/volsurfs.py:402-470) in one launch each way: group g owns the rows after the
 * first g groups' in x / out / g_out.  plan_host: group 0's plan (levels, dimension, capacity are
 * shared); plans_dev: the nr_groups plans in DEVICE memory (the per-level shifts may differ);
 * lattice_values / grad_values: HOST arrays of nr_groups device pointers; nr_points: host [nr_groups]. */
int vsa_permuto_encode_fwd_grouped(const vsa_permuto_plan* plan_host, const vsa_permuto_plan* plans_dev,
                                   const float* const* lattice_values, int nr_groups,
                                   const int* nr_points, const float* x, const float* window,
                                   float* out, int out_stride, void* stream);
int vsa_permuto_encode_bwd_grouped(const vsa_permuto_plan* plan_host, const vsa_permuto_plan* plans_dev,
                                   int nr_groups, const int* nr_points, const float* x,
                                   const float* window, const float* g_out, int g_stride,
                                   float* const* grad_values, void* stream);

/* ------------------------------------------------------------------------
 * A8 / A9 / A11  Packed (ragged) per-ray sample ops of the background path:
 * what render_contracted_bg (volsurfs_py/utils/background.py:31-141) calls in the
 * reference's pybind module `volsurfs` (src/PyBridge.cxx:70-138).  A pack is the
 * SoA of include/volsurfs/RaySamplesPacked.cuh:7-80; here its tensors are passed
 * individually: ray_start_end_idx [N,2] i32, per-sample [S,*] f32, per-ray [N,*] f32.
 * A ray is owned by a 32-lane half-wave (lanes = consecutive samples).
 */
/* VolumeRendering::cumprod_one_minus_alpha_to_transmittance (src/VolumeRendering.cu:30-78):
 * T_i = prod_{j<i} a_j, bg_T = T_{n-1}; caller pre-fills T with 0 and bg_T with 1. */
int vsa_packed_cumprod_fwd(const int32_t* start_end, const float* one_minus_alpha,
                           float* transmittance, float* bg_transmittance, int nr_rays,
                           void* stream);
/* ..._backward (:671-718): g_a_i = (cumsumLV_{i+1} + g_bgT*bgT)/max(a_i,1e-6), 0 for the last. */
int vsa_packed_cumprod_bwd(const int32_t* start_end, const float* g_bg_transmittance,
                           const float* one_minus_alpha, const float* bg_transmittance,
                           const float* cumsum_lv, float* g_one_minus_alpha, int nr_rays,
                           void* stream);
/* VolumeRendering::cumsum_over_rays (:326-370), inverse = suffix sums. */
int vsa_packed_cumsum(const int32_t* start_end, const float* values, int inverse, float* out,
                      int nr_rays, void* stream);
/* integrate_with_weights_{1d,3d} (:80-176) and their backward (:720-818); dim in {1,3};
 * bug_compat=1 reproduces VolumeRenderingGPU.cuh:1021 (z lane reads column 1). */
int vsa_packed_integrate_fwd(const int32_t* start_end, const float* values, const float* weights,
                             float* out, int nr_rays, int dim, void* stream);
int vsa_packed_integrate_bwd(const int32_t* start_end, const float* g_out, const float* values,
                             const float* weights, float* g_values, float* g_weights,
                             int nr_rays, int dim, int bug_compat, void* stream);
/* The background composite of render_contracted_bg (volsurfs_py/utils/background.py:93-111) as one
 * launch each way: alpha = 1 - exp(-density dt), T = cumprod_one_minus_alpha_to_transmittance(
 * (1 - alpha) + 1e-6) (VolumeRenderingGPU.cuh:28-78), w = alpha T, pred_rgb =
 * integrate_with_weights_3d(rgb, w) (:127-177); backward = integrate_with_weights_3d_backward
 * (:987-1033, bug_compat as above), the suffix sums of cumsum_over_rays(inverse) (:305-361) and
 * cumprod_..._backward (:896-943) with a zero bg-transmittance gradient.  Bit-identical to the chain
 * of the single ops above.  density, dt: [S] (or [S,1]); rgb, g_rgb: [S,3]; weights (optional out,
 * what median_depth_over_rays takes): [S]; scratch: 2 S floats. */
int vsa_packed_composite_fwd(const int32_t* start_end, const float* density, const float* dt,
                             const float* rgb, float* pred_rgb, float* weights, int nr_rays,
                             void* stream);
int vsa_packed_composite_bwd(const int32_t* start_end, const float* density, const float* dt,
                             const float* rgb, const float* g_pred_rgb, float* g_rgb,
                             float* g_density, float* scratch, int nr_rays, int bug_compat,
                             void* stream);
/* median_depth_over_rays (:372-416); fallback_compat=1 reproduces VolumeRenderingGPU.cuh:407. */
int vsa_packed_median_depth(const int32_t* start_end, const float* samples_z,
                            const float* weights, float threshold, float* out, int nr_rays,
                            int fallback_compat, void* stream);
/* RaySamplesPacked::update_dt (src/RaySamplesPacked.cu:396-461). */
int vsa_packed_update_dt(const int32_t* start_end, const float* ray_max_dt, const float* ray_exit,
                         const float* samples_z, int is_background, float* samples_dt,
                         int nr_rays, void* stream);
/* RaySampler::compute_samples_bg (src/RaySampler.cu:70-156): n samples per ray at
 * t = t_start + 1/(s+1e-6) - 1, s: 1 -> 0; jitter uses PCG32 (state, inc) exactly as the
 * reference's by-value copy of m_rng (advance(ray) before every draw). */
int vsa_sample_bg(const float* rays_o, const float* rays_d, const float* ray_t_start,
                  float ray_t_far, int nr_samples_per_ray, int jitter, uint64_t rng_state,
                  uint64_t rng_inc, float* ray_max_dt, float* samples_3d, float* samples_dirs,
                  float* samples_z, int32_t* ray_start_end_idx, int nr_rays, void* stream);
/* RaySampler::contract_samples kernel (src/RaySampler.cu:336-381; update_dt is a separate call). */
int vsa_contract_samples(const float* ray_o, const int32_t* start_end, const float* samples_3d,
                         const float* samples_z, float* out_samples_3d, float* out_samples_z,
                         int nr_rays, void* stream);

/* Ops of the sibling methods reached through the same pybind module.
 * VolumeRendering::sum_over_rays (src/VolumeRendering.cu:231-324): values [S,dim], dim in
 * {1,2,3,32} -> sum_per_ray [N,dim] and the same sum repeated per sample [S,dim]; backward
 * g_values = g_sum_per_ray[ray] + g_sum_per_sample (kernels/volsurfs/VolumeRenderingGPU.cuh:1036-1077). */
int vsa_packed_sum_over_rays(const int32_t* start_end, const float* values, float* sum_per_ray,
                             float* sum_per_sample, int nr_rays, int dim, void* stream);
int vsa_packed_sum_over_rays_bwd(const int32_t* start_end, const float* g_sum_per_ray,
                                 const float* g_sum_per_sample, float* g_values, int nr_rays,
                                 int dim, void* stream);
/* VolumeRendering::sdf2alpha (src/VolumeRendering.cu:178-229): alpha [S,1] (zero-initialised by
 * the caller; the last sample of a ray is not written) from samples_dt, samples_sdf and the
 * per-sample logistic beta. */
int vsa_packed_sdf2alpha(const int32_t* start_end, const float* samples_dt, const float* samples_sdf,
                         const float* logistic_beta, float* alpha, int nr_rays, void* stream);
/* VolumeRendering::compute_cdf (src/VolumeRendering.cu:418-465): cdf [S,1], zero-initialised by
 * the caller. */
int vsa_packed_compute_cdf(const int32_t* start_end, const float* weights, float* cdf, int nr_rays,
                           void* stream);

/* RaySampler::compute_samples_fg (src/RaySampler.cu:158-240, kernels/volsurfs/RaySamplerGPU.cuh:141-270):
 * uniform steps >= min_dist between t_entry and t_exit, at most max_n per ray, rays with fewer
 * than min_n samples get none; ray i writes slots [i*max_n, ...).  Outputs as the
 * RaySamplesPacked fields (caller-initialised to the constructor's values); compact afterwards. */
int vsa_sample_fg(const float* rays_o, const float* rays_d, const float* ray_t_entry,
                  const float* ray_t_exit, float min_dist_between_samples,
                  int min_nr_samples_per_ray, int max_nr_samples_per_ray, int jitter,
                  uint64_t rng_state, uint64_t rng_inc, float* ray_max_dt, int32_t* samples_idx,
                  float* samples_3d, float* samples_dirs, float* samples_z,
                  int32_t* ray_start_end_idx, int nr_rays, void* stream);
/* RaySamplesPacked::compact_to_valid_samples (src/RaySamplesPacked.cu:188-273): out_start [N] =
 * exclusive scan of the per-ray sample counts (caller). */
int vsa_pack_compact(const int32_t* start_end, const int32_t* out_start, const int32_t* samples_idx,
                     const float* samples_3d, const float* samples_dirs, const float* samples_z,
                     const float* samples_dt, const float* samples_values, int values_dim,
                     int32_t* out_idx, float* out_3d, float* out_dirs, float* out_z, float* out_dt,
                     float* out_values, int32_t* out_start_end, int nr_rays, void* stream);
/* VolumeRendering::importance_sample (src/VolumeRendering.cu:467-560): nr_importance_samples new
 * depths per ray by inverting the ray's cdf; ray i writes slots [i*n, (i+1)*n). */
int vsa_importance_sample(const float* rays_o, const float* rays_d, const int32_t* start_end,
                          const float* samples_z, const float* samples_cdf,
                          int nr_importance_samples, int jitter, uint64_t rng_state,
                          uint64_t rng_inc, float* out_3d, float* out_dirs, float* out_z,
                          int32_t* out_start_end, int nr_rays, void* stream);

/* RaySampler::uncontract_samples (src/RaySampler.cu:383-428): inverse of vsa_contract_samples. */
int vsa_uncontract_samples(const float* ray_o, const int32_t* start_end, const float* samples_3d,
                           const float* samples_z, float* out_samples_3d, float* out_samples_z,
                           int nr_rays, void* stream);

/* VolumeRendering::combine_ray_samples_packets (src/VolumeRendering.cu:562-670): depth-ordered
 * merge of two compacted packs over the same rays, dropping samples closer than min_dist to the
 * previous kept one.  out_start [N] = exclusive scan of (count_1 + count_2); compact afterwards. */
int vsa_combine_packs(const int32_t* start_end_1, const int32_t* idx_1, const float* s3d_1,
                      const float* dirs_1, const float* z_1, const float* values_1,
                      const int32_t* start_end_2, const int32_t* idx_2, const float* s3d_2,
                      const float* dirs_2, const float* z_2, const float* values_2,
                      const int32_t* out_start, float min_dist_between_samples, int values_dim,
                      int32_t* out_idx, float* out_3d, float* out_dirs, float* out_z,
                      float* out_values, int32_t* out_start_end, int nr_rays, void* stream);

/* A1  Ray / bounding-primitive intersection: `intersect_bounding_primitive`
 * (volsurfs_py/utils/raycasting.py:4-36) -> mvdatasets BoundingBox / BoundingSphere .intersect
 * (absent: parity unpinned; the primitive is chosen at utils/volsurfs_utils.py:234-272).
 * kind 0 = origin-centred cube of HALF side `size`, kind 1 = origin-centred sphere of radius
 * `size`.  is_hit [N] u8, t_near / t_far [N] (0 on a miss, t_near clamped to >= 0),
 * points_near / points_far [N,3] (optional).  VolSurfs consumes t_far (volsurfs.py:688-693). */
int vsa_intersect_primitive(const float* rays_o, const float* rays_d, int nr_rays, int kind,
                            float size, uint8_t* is_hit, float* t_near, float* t_far,
                            float* points_near, float* points_far, void* stream);

/* ---- ray generation (mvdatasets is an empty submodule: parity unpinned) ----
 * Pinhole rays of one camera, replacing mvdatasets.utils.raycasting.get_camera_rays as called at
 * methods/base_method.py:389-394 and renderers/base_renderer.py:59.  c2w [3,4] and
 * intrinsics_inv [3,3] are DEVICE arrays, row-major.  Ray i = (row * width + col) *
 * nr_rays_per_pixel + s goes through (col + jx, row + jy), (jx, jy) = (0.5, 0.5) or two PCG32
 * draws of the stream (rng_state, rng_inc) advanced by 2 i.  rays_o / rays_d [n,3], points_2d
 * [n,2] (may be NULL). */
int vsa_camera_rays(const float* c2w, const float* intrinsics_inv, int height, int width,
                    int nr_rays_per_pixel, int jitter_pixels, uint64_t rng_state, uint64_t rng_inc,
                    float* rays_o, float* rays_d, float* points_2d, void* stream);

/* Training batch, replacing TensorReel.get_next_rays_batch as called at trainer.py:176-190:
 * batch_size uniformly drawn (camera, pixel) pairs of a resident stack of nr_cameras equally
 * sized views (c2w_all [C,3,4], intrinsics_inv_all [C,3,3], rgb_all [C,H,W,3], mask_all [C,H,W]
 * or NULL), nr_rays_per_pixel rays each.  camera_idx [B], rays_o / rays_d [B*R,3], gt_rgb [B,3]
 * and gt_mask [B] (each may be NULL), points_2d [B*R,2] (may be NULL). */
int vsa_reel_next_rays_batch(const float* c2w_all, const float* intrinsics_inv_all,
                             const float* rgb_all, const float* mask_all, int nr_cameras, int height,
                             int width, int batch_size, int nr_rays_per_pixel, int jitter_pixels,
                             uint64_t rng_state, uint64_t rng_inc, int32_t* camera_idx,
                             float* rays_o, float* rays_d, float* gt_rgb, float* gt_mask,
                             float* points_2d, void* stream);

/* ------------------------------------------------------------------------
 * The training iteration as ONE replayed HIP graph (round 6): device-side control block.
 *
 * The reference's loop (volsurfs_py/trainer.py:118-308) changes four things from one iteration to the next that a
 * launch sequence would carry as kernel ARGUMENTS — the ray count of the dynamic batch (:288-304), the loss
 * normalisation 1 / (3 n), the learning rate of the warm-up + MultiStepLR schedule (:306-308, schedulers/warmup.py,
 * base_method.py:71-76) with Adam's step count, and the sampler's random stream (:176-190) — and reads the hit count
 * back on the host to size the next batch.  Here they live in device memory: every kernel of the iteration runs at
 * a fixed CAPACITY of rays, the `_ctl` entry points read what varies from this block, and a one-lane kernel at the
 * end of the iteration (vsa_train_ctl_tick) applies the reference's rules for the next one.  The iteration is then
 * a static graph: no host read, no host write, no launch gap inside it.
 * Rays beyond nr_rays (up to the capacity) are DUMMY rays: the sampler points them away from the scene (they hit
 * nothing: no texel is marked, nothing is shaded), the composite gives them no gradient, the loss does not count them. */
typedef struct vsa_train_ctl {
  int32_t iter;            /* 0-based number of the iteration that runs next (= times the schedule has been stepped) */
  int32_t nr_rays;         /* active rays of that iteration, 1 .. capacity */
  int32_t capacity;        /* rays every launch is sized for */
  int32_t adam_step;       /* Adam updates applied or pending so far: the pending one's 1-based step count */
  float loss_scale;        /* loss_weight / (3 nr_rays): d mean|gt - pred| / d pred, as trainer.train_step forms it */
  float adam_lr;           /* learning rate of the pending Adam update = lr_at(iteration it belongs to) */
  float loss;              /* mean |gt - pred| of the last finished iteration (vsa_l1_mean_ctl) */
  int32_t clamped;         /* how often the dynamic rule asked for more than `capacity` rays (the host re-captures) */
  uint64_t rng_state, rng_inc;   /* the sampler's PCG32 stream (advanced by 2^32 per iteration, as TensorReel does) */
  int64_t nr_hits;         /* hits of the last finished iteration (vsa_count_hits writes here) */
  int32_t target_hits;     /* target_nr_of_training_samples (0: the ray count stays) */
  int32_t nr_warmup;       /* warm-up iterations of the schedule */
  int32_t nr_milestones;   /* <= 8 */
  int32_t milestone[8];    /* MultiStepLR milestones, sorted */
  float lr_stage[9];       /* (float)(base_lr * gamma^k), k = 0 .. nr_milestones: formed on the host, in double */
  int32_t adam_pending;    /* 1: an iteration has finished whose update vsa_adam_step_ctl has not applied yet (set by the tick) */
  double lr_base;          /* base learning rate (the warm-up ramp is lr_base * (it / nr_warmup) in double, as the host's) */
  double loss_weight;      /* 1 (a data-parallel rank: its share of the global batch) */
  int64_t sum_rays, sum_hits;    /* running totals over the finished iterations (the tick adds; a bench reads the difference) */
} vsa_train_ctl;

/* End of an iteration (stream-ordered behind vsa_count_hits / vsa_l1_mean_ctl of that iteration): the pending Adam
 * update becomes this iteration's (adam_step += 1, adam_lr = lr_at(iter)); nr_rays = int(nr_rays * (target_hits /
 * nr_hits)) (trainer.py:288-304; clamped to 1 .. capacity, counted in `clamped`); loss_scale follows; iter += 1; the
 * random stream advances by 2^32.  One lane. */
int vsa_train_ctl_tick(vsa_train_ctl* ctl, void* stream);

/* vsa_reel_next_rays_batch at a fixed launch size: `capacity` (the host's copy of ctl->capacity: it sizes the grids of
 * the `_ctl` launches) samples are written, the first
 * ctl->nr_rays drawn exactly as vsa_reel_next_rays_batch(batch_size = nr_rays) draws them from (ctl->rng_state,
 * ctl->rng_inc); the rest are dummy rays: origin dummy_o[3], direction dummy_d[3] (host arrays: a ray that misses
 * every shell), gt_rgb = dummy_rgb[3]. */
int vsa_reel_next_rays_batch_ctl(const float* c2w_all, const float* intrinsics_inv_all, const float* rgb_all,
                                 const float* mask_all, int nr_cameras, int height, int width, int capacity,
                                 int nr_rays_per_pixel, int jitter_pixels, const vsa_train_ctl* ctl,
                                 const float* dummy_o, const float* dummy_d, const float* dummy_rgb,
                                 int32_t* camera_idx, float* rays_o, float* rays_d, float* gt_rgb,
                                 float* gt_mask, float* points_2d, void* stream);

/* vsa_composite_dense_fwd_bwd_l1 over ctl->capacity rays with loss_scale = ctl->loss_scale; rays >= ctl->nr_rays get
 * zero gradients (their colour is still written). */
int vsa_composite_dense_fwd_bwd_l1_ctl(const float* surfs_rgb, const float* surfs_alpha, const float* rgb_bg,
                                       int bg_is_broadcast, const float* gt_rgb, const vsa_train_ctl* ctl,
                                       float* out_rgb, float* g_surfs_rgb, float* g_surfs_alpha, int capacity,
                                       int nr_shells, int carry_f16, void* stream);

/* vsa_l1_mean over the first 3 * ctl->nr_rays elements of pred / gt ([capacity, 3]); the mean goes to ctl->loss. */
int vsa_l1_mean_ctl(const float* pred, const float* gt, int capacity, void* scratch, vsa_train_ctl* ctl, void* stream);

/* vsa_adam_step_shared with lr = ctl->adam_lr and step = ctl->adam_step read on the device (the bias corrections
 * are formed there, in double); a launch that finds adam_pending == 0 (the first replay of a loop) does nothing. */
int vsa_adam_step_ctl(const vsa_adam_tensor* tensors_dev, const int32_t* chunks_dev, int nr_chunks, float beta1,
                      float beta2, float eps, float grad_scale, int zero_grads, int max_workgroups,
                      const vsa_train_ctl* ctl, void* stream);

/* Re-orders a row-major per-pixel array [height*width, channels] (channels 1..4, f32) into
 * 8x8-pixel-tile-major order (inverse = 0), or back (inverse = 1).  height and width must be
 * multiples of 8.  Inside a tile the pixels run boustrophedon: element j of a tile is pixel row
 * j/8 and pixel column j%8 on even rows, 7 - j%8 on odd rows, so consecutive elements are always
 * neighbouring pixels.  Used on the rays of a full frame before the traversal (a wave then covers
 * a square patch of pixels instead of a 64x1 strip) and on the rendered colours after it. */
int vsa_tile_order(const float* src, float* dst, int height, int width, int channels, int inverse,
                   void* stream);
/* The same forward re-ordering for the three per-ray inputs of a frame ([H*W,3] each; gt_rgb
 * and gt_rgb_tiled may be NULL) in one launch. */
int vsa_tile_order_rays(const float* rays_o, const float* rays_d, const float* gt_rgb,
                        float* rays_o_tiled, float* rays_d_tiled, float* gt_rgb_tiled, int height,
                        int width, void* stream);

/* ---- OccupancyGrid: include/volsurfs/OccupancyGrid.cuh:9-68 -----------------
 * A cubic grid of nr_voxels_per_dim^3 voxels (a power of two <= 1024) in Morton order, centred on
 * the origin, with side lengths extent_{x,y,z}: grid_values f32 [n^3], grid_occupancy / grid_roi
 * u8 (torch.bool) [n^3].  Kernels: kernels/volsurfs/OccupancyGridGPU.cuh, occ_grid_helpers.h.
 * Every ray marcher is capped at 65536 steps (the reference's loops are unbounded). */

/* get_grid_lower_left_voxels_vertices (centre_of_voxel = 0) / get_grid_samples,
 * get_random_grid_samples(_in_roi) (centre_of_voxel = 1, optional jitter inside the voxel):
 * positions [nr_points,3] of the voxels point_indices (NULL = 0..nr_points-1).
 * OccupancyGridGPU.cuh:31-119. */
int vsa_occ_grid_points(const int32_t* point_indices, int nr_points, int nr_voxels_per_dim,
                        float extent_x, float extent_y, float extent_z, int centre_of_voxel, int jitter,
                        uint64_t rng_state, uint64_t rng_inc, float* out_points, void* stream);
/* update_grid_values: grid[idx] = max(values, grid[idx] * decay).  OccupancyGridGPU.cuh:122-151. */
int vsa_occ_update_values(const int32_t* point_indices, const float* values, int nr_points, float decay,
                          float* grid_values, void* stream);
/* update_grid_occupancy_with_density_values (:153-225): occupied = value > thresh (any of the 27
 * neighbours when check_neighbours). */
int vsa_occ_update_occupancy_density(const int32_t* point_indices, int nr_points, int nr_voxels_per_dim,
                                     float extent_x, float extent_y, float extent_z,
                                     float occupancy_thresh, int check_neighbours,
                                     const float* grid_values, uint8_t* grid_occupancy, void* stream);
/* update_grid_occupancy_with_sdf_values (:229-315): NeuS logistic density of the smallest |sdf|
 * reachable in the voxel > thresh; logistic_beta [nr_points]. */
int vsa_occ_update_occupancy_sdf(const int32_t* point_indices, const float* logistic_beta, int nr_points,
                                 int nr_voxels_per_dim, float extent_x, float extent_y, float extent_z,
                                 float occupancy_thresh, const float* grid_values,
                                 uint8_t* grid_occupancy, void* stream);
/* check_occupancy (:376-413): per point (roi && occupied) and the voxel's value (false / 0 outside). */
int vsa_occ_check(const float* points, int nr_points, int nr_voxels_per_dim, float extent_x,
                  float extent_y, float extent_z, const float* grid_values,
                  const uint8_t* grid_occupancy, const uint8_t* grid_roi, uint8_t* out_occupancy,
                  float* out_values, void* stream);
/* get_rays_t_near_t_far (:318-374). */
int vsa_occ_rays_t_near_t_far(const float* rays_o, const float* rays_d, const float* ray_t_entry,
                              const float* ray_t_exit, int nr_rays, int nr_voxels_per_dim,
                              float extent_x, float extent_y, float extent_z,
                              const uint8_t* grid_occupancy, const uint8_t* grid_roi, float* out_t_near,
                              float* out_t_far, void* stream);
/* get_first_rays_sample_start_of_grid_occupied_regions (:505-581): pack of one sample per ray. */
int vsa_occ_first_sample(const float* rays_o, const float* rays_d, const float* ray_t_entry,
                         const float* ray_t_exit, int nr_rays, int nr_voxels_per_dim, float extent_x,
                         float extent_y, float extent_z, const uint8_t* grid_occupancy,
                         const uint8_t* grid_roi, float* samples_3d, float* samples_dirs,
                         float* samples_z, float* samples_dt, int32_t* ray_start_end_idx, void* stream);
/* advance_ray_sample_to_next_occupied_voxel (:415-503); new_samples_3d may alias samples_3d. */
int vsa_occ_advance_samples(const float* samples_dirs, const float* samples_3d, int nr_points,
                            int nr_voxels_per_dim, float extent_x, float extent_y, float extent_z,
                            const uint8_t* grid_occupancy, const uint8_t* grid_roi,
                            float* new_samples_3d, uint8_t* is_within_bounds, void* stream);
/* RaySampler::compute_samples_fg_in_grid_occupied_regions (src/RaySampler.cu:243-334, kernel
 * RaySamplerGPU.cuh:275-457): equidistant samples in occupied-space arc length; outputs as
 * vsa_sample_fg. */
int vsa_sample_fg_occupied(const float* rays_o, const float* rays_d, const float* ray_t_entry,
                           const float* ray_t_exit, float min_dist_between_samples,
                           int min_nr_samples_per_ray, int max_nr_samples_per_ray, int jitter_samples,
                           uint64_t rng_state, uint64_t rng_inc, int nr_voxels_per_dim, float extent_x,
                           float extent_y, float extent_z, const uint8_t* grid_occupancy,
                           const uint8_t* grid_roi, float* ray_max_dt, int32_t* samples_idx,
                           float* samples_3d, float* samples_dirs, float* samples_z,
                           int32_t* ray_start_end_idx, int nr_rays, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VOLSURFS_HIP_H */
