"""KShellPipeline — the fused host path of VolSurfs.render_rays + L1 loss +
backward, one stream, no host syncs.

Stage order mirrors /root/reference/volsurfs_py/methods/volsurfs.py:423-761:
trace (K shells, one launch) -> shade (neural textures) -> dense composite ->
loss_l1 (utils/losses.py:14-19) -> backward of the same chain.
"""
import torch

from . import _lib
from .camera import pinhole_rays
from .mesh import nested_shells, stress_shells
from .raytrace import RayTracer


# Scale of the f16 gradient chain relative to the ray count of a mean-L1 loss (the analogue of tiny-cuda-nn's
# loss scale; divided out before anything is accumulated in fp32).  With 1 x N the chain carried 1/3 per
# ray and channel: fine up to K = 7, but at K = 9 the shells behind eight others underflowed f16 — every
# MLP-weight gradient within 3.6e-3 of its tensor's largest entry instead of 3.1e-4 at 16 x N, table entries over
# 1e-3: 2.7e-5 -> 1.3e-6 of them (profiles/r05/parity_report.json; 16 and 256 measure the same, so 16 it is:
# the headroom to f16's 65 504 stays > 10x even with hundreds of hits on one texel).  A power of two: exact.
GRAD_CHAIN_GAIN = 16.0

# Graph capture in "thread_local" error mode: under the default ("global") ANY thread's HIP call that is illegal during
# a capture fails — and torch.distributed's ProcessGroupNCCL watchdog thread polls its Work events with hipEventQuery
# whenever a collective is still outstanding: a step captured right after eager data-parallel warm-up steps aborted the
# process about once in twenty runs ("operation not permitted when stream is capturing" from the watchdog).
_CAPTURE_MODE = "thread_local"


class _no_gc:
    """No garbage collection inside a stream capture: a cycle collected there may destroy ANOTHER pipeline's HIP graph (or a
    signal allocation), which the runtime refuses during a capture — and a failed destroy inside a C++ destructor aborts
    the process (seen once in a full `pytest -m gpu` run: 'Fatal Python error: Aborted ... Garbage-collecting' under
    capture_graph, profiles/r06/README.md).  Collect first, then hold the collector off until the capture has ended."""

    def __enter__(self):
        import gc
        gc.collect()
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()


class StageTimer:
    def __init__(self):
        self.records = []          # (name, start_evt, end_evt)
        self.meta = {}             # name -> dict(bytes=, flops=, bound=)

    def run(self, name, fn, record, **meta):
        if not record:
            return fn()
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn()
        e.record()
        self.records.append((name, s, e))
        self.meta[name] = meta
        return out

    def report(self):
        torch.cuda.synchronize()
        acc, cnt = {}, {}
        for name, s, e in self.records:
            acc[name] = acc.get(name, 0.0) + s.elapsed_time(e)
            cnt[name] = cnt.get(name, 0) + 1
        return {k: dict(ms=acc[k] / cnt[k], **self.meta[k]) for k in acc}


class KShellPipeline:
    dtype_desc = "f16 (hash features, MLP on MFMA, composite: as the reference), f32 accumulate and I/O"

    def __init__(self, meshes, rays_o, rays_d, gt_rgb, bg_color=(1.0, 1.0, 1.0), seed=42,
                 init="tcnn", image_hw=None, tracer=None):
        """image_hw = (H, W): the rays are the row-major pixels of one full frame; every step
        then first re-orders them (and the ground truth) into 8x8-pixel tiles, so that a wave
        of any per-ray kernel covers a square patch, and returns the colours in the caller's
        order (vsa_tile_order; both passes are inside the step).
        tracer: a RayTracer already built over `meshes` (another pipeline's: the BVH of a 1.3 M-triangle
        shell takes seconds to build) instead of building one."""
        from .neural_textures import NeuralTextureBank
        self.meshes = meshes
        self.K = len(meshes)
        self.tracer = RayTracer(meshes) if tracer is None else tracer
        self.rays_o, self.rays_d, self.gt = rays_o, rays_d, gt_rgb
        self.nr_rays = rays_o.shape[0]
        import os
        self.image_hw = None
        if image_hw is not None and image_hw[0] % 8 == 0 and image_hw[1] % 8 == 0 and \
                image_hw[0] * image_hw[1] == self.nr_rays and os.environ.get("VSA_TILE_ORDER", "1") != "0":
            self.image_hw = (int(image_hw[0]), int(image_hw[1]))
            self._o_t, self._d_t, self._gt_t = (torch.empty_like(x) for x in (rays_o, rays_d, gt_rgb))
            self._rgb_out = torch.empty_like(gt_rgb)
            H, W = self.image_hw
            tiles = torch.arange(H * W, device=rays_o.device).reshape(H // 8, 8, W // 8, 8).permute(0, 2, 1, 3).clone()
            tiles[:, :, 1::2] = tiles[:, :, 1::2].flip(-1)          # odd pixel rows of a tile run right to left (raygen.hip)
            self._tile_idx = tiles.reshape(-1)            # element i of a tile-major array = pixel _tile_idx[i]
        dev = rays_o.device
        self.bg = torch.tensor([bg_color], device=dev, dtype=torch.float32)
        self.timer = StageTimer()
        N, K = self.nr_rays, self.K
        # per-corner uvs in the tracer's leaf (slot) order
        fu = []
        for m, off, n in zip(meshes, self.tracer.mesh_tri_offset, self.tracer.mesh_nr_tris):
            ids = self.tracer.slot_face_id[off:off + n].long()
            fu.append(m.faces_uvs.reshape(-1, 6)[ids])
        self.face_uvs = torch.cat(fu, 0).contiguous()
        self.bank = NeuralTextureBank(K, N, device=dev, seed=seed)
        if init == "spread":
            # larger parameters: textures with visible structure (parity tests)
            g = torch.Generator().manual_seed(seed)
            with torch.no_grad():
                self.bank.tables.copy_((torch.rand(self.bank.tables.shape, generator=g) * 2 - 1).to(dev))
            self.bank.refresh_half_params()
        self.shading = "neural_textures"
        self.grad_scale = GRAD_CHAIN_GAIN * float(N)
        # rays the mean of the L1 loss runs over: this pipeline's own, or — when it renders one
        # rank's share of a frame (bench.py --scaling strong) — the whole frame's
        self.loss_rays = N
        self.surfs_rgb = self.surfs_alpha = None
        # per-hit output sigmoids kept from shade_fwd for shade_bwd of the same frame
        self._act = torch.empty(K, N, 4, device=dev)

    def to_ray_order(self, x, dim=0):
        """Tests / inspection: a per-ray internal buffer (surfs_rgb, surfs_alpha, _hit_slot: in
        tile order when image_hw is set) re-indexed in the caller's ray order."""
        if self.image_hw is None:
            return x
        out = torch.empty_like(x)
        out.index_copy_(dim, self._tile_idx, x)
        return out

    @classmethod
    def synthetic(cls, K=5, subdiv=6, res=800, device="cuda", seed=42, rows=None, gt_seed=None,
                  noise=0.0, atlas_charts=0, stress=False, cam_pos=(0.0, 0.0, -1.5), **kw):
        """res: an int (square frame) or (H, W).  rows: optional LongTensor of image rows (whole
        8-row bands, parallel.shard_bands): the pipeline then renders only those rows of the
        frame — one rank's share under strong scaling — with the loss still the frame's mean."""
        if stress:      # non-convex lobed shells, 12x triangle-area spread, 256 randomly packed charts (mesh.stress_shells)
            meshes = stress_shells(K=K, subdiv=subdiv, device=device)
        else:
            meshes = nested_shells(K=K, subdiv=subdiv, device=device, noise=noise, atlas_charts=atlas_charts)
        H, W = (res, res) if isinstance(res, int) else res
        o, d = pinhole_rays(H, W, focal=1111.1 * min(H, W) / 800.0, cam_pos=cam_pos, device=device)
        # (gt_seed: data-parallel ranks share the parameters — `seed` — and differ in their data)
        g = torch.Generator(device=device).manual_seed(seed if gt_seed is None else gt_seed)
        gt = torch.rand(o.shape[0], 3, device=device, generator=g)
        n_frame, h_local = H * W, H
        if rows is not None:
            rows = rows.to(device)
            o, d, gt = (x.view(H, W, 3)[rows].reshape(-1, 3).contiguous() for x in (o, d, gt))
            h_local = int(rows.numel())
        p = cls(meshes, o, d, gt, seed=seed, image_hw=(h_local, W), **kw)
        p.loss_rays = n_frame
        p.grad_scale = GRAD_CHAIN_GAIN * float(n_frame)      # the f16 gradient chain sees 16 / 3 per ray and channel
        p.res = res
        p.subdiv = subdiv
        p.scene_desc = ("stress shells: 4 lobes of depth 0.3 r across the view axis (non-convex, up to 6 crossings per "
                        "ray), triangle areas spread 12x, 256 randomly packed uv charts per shell, noise 0.05, "
                        f"parameters {kw.get('init', 'tcnn')}-initialised") if stress else \
            (f"noise {noise}, {atlas_charts}x{atlas_charts} randomly packed uv charts per shell, "
             f"parameters {kw.get('init', 'tcnn')}-initialised") if (noise or atlas_charts) else \
            "perfect spheres, one continuous octahedral uv chart, tcnn-initialised parameters"
        return p

    def reset_stage_timers(self):
        self.timer = StageTimer()

    def stage_report(self):
        return self.timer.report()

    @staticmethod
    def stage_roofline(st, hbm_peak, mfma_peak):
        sec = st["ms"] * 1e-3
        out = {}
        if st.get("bytes"):
            out["GB/s"] = round(st["bytes"] / sec / 1e9, 1)
            out["hbm_frac"] = round(st["bytes"] / sec / 1e9 / hbm_peak, 4)
        if st.get("flops"):
            out["TFLOP/s"] = round(st["flops"] / sec / 1e12, 2)
            out["mfma_frac"] = round(st["flops"] / sec / 1e12 / mfma_peak, 4)
        return out

    def config_desc(self, world):
        return {
            "workload": f"synthetic kitten-like: {self.res if isinstance(self.res, int) else self.res[1]}x"
                        f"{self.res if isinstance(self.res, int) else self.res[0]} rays, K={self.K} nested "
                        f"icospheres subdiv {self.subdiv} ({self.tracer.mesh_nr_tris[0]} tris each), "
                        "SH-degree-3 neural textures (rgb + alpha per shell, res 2048/1024/512/256, "
                        "16-level 2-D hash grid + 32-64-64-C MLP, 8-bit quantised, lerp), alpha decay, "
                        "white bg, L1 loss, gradients to all hash tables and MLP weights",
            "rays_per_gpu": self.nr_rays,
            "global_rays": self.loss_rays if self.loss_rays != self.nr_rays else self.nr_rays * world,
            "parallelism": f"tile-parallel x{world}",
            "hits_per_frame": getattr(self, "last_hits", None),
            "unique_texels_per_frame": getattr(self, "last_slots", None),
            # the traversal's launch order comes from the previous frame's measured wave cost (same hits;
            # VSA_TRACE_FEEDBACK=0 = the stateless launch: profiles/NOTEBOOK.md A9.4)
            "trace_launch_order": "cost feedback from the previous frame (static camera: the previous frame has "
                                  "the same rays; value_cold is the figure without any inter-frame feedback)"
                                  if self.tracer.cost_feedback and
                                  self.tracer.node_format == "q16" else "natural",
        }

    def stats(self):
        """Host-side read-back of frame statistics (outside the timed region) and the
        ALGORITHMIC bytes / FLOPs of every stage for this frame: each
        operand counted once per stage at its stored width, gathers per hit."""
        from .neural_textures import stage_accounting
        bank, N, K = self.bank, self.nr_rays, self.K
        self.last_hits = M = int((self._hit_slot >= 0).sum().item())
        nodes_b = self.tracer.nodes.numel() * 4 + self.tracer.tris.numel() * 4
        self.acct, self.mlp_flops_fwd, self.last_slots = stage_accounting(bank, N, M, nodes_b)
        return self.last_hits, self.last_slots

    def capture_graph_split(self, **step_kw):
        """The step as THREE graphs: `replay_prefix()` = what does not read a parameter or touch a
        gradient (ray tile order, traversal, mark / compact); `replay_mid()` = zero_grad .. MLP backward;
        `replay_tail()` = the hash-grid backward (one launch).  A data-parallel caller launches the prefix
        while the previous step's gradient reduction (and, in training, the optimiser) is still running,
        waits, launches mid, records "weights.grad final", launches tail and queues its flag waits behind
        that record — so that they are pending beside ONE kernel, not beside twenty
        (parallel.OverlappedStep.run_split)."""
        self._static_rgb = None
        self._graph_dp = dp = step_kw.get("dp")
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        hook, _ = (dp.on_weights_final, setattr(dp, "on_weights_final", None)) if dp is not None else (None, None)
        with torch.cuda.stream(s):
            for _ in range(2):
                self.step(**step_kw)      # (eager warm-ups: see capture_graph)
        if dp is not None:
            dp.on_weights_final = hook
        torch.cuda.current_stream().wait_stream(s)
        self._graph_prefix, self._graph_mid, self._graph_tail = (torch.cuda.CUDAGraph() for _ in range(3))
        with _no_gc():
            with torch.cuda.graph(self._graph_prefix, capture_error_mode=_CAPTURE_MODE):
                self.step(part="prefix", **step_kw)
            with torch.cuda.graph(self._graph_mid, pool=self._graph_prefix.pool(), capture_error_mode=_CAPTURE_MODE):
                self.step(part="mid", **step_kw)
            with torch.cuda.graph(self._graph_tail, pool=self._graph_prefix.pool(), capture_error_mode=_CAPTURE_MODE):
                self._static_rgb = self.step(part="tail", **step_kw)

    def replay_prefix(self):
        self._graph_prefix.replay()

    def _count_replay(self):
        dp = getattr(self, "_graph_dp", None)
        if dp is not None:
            dp.epoch_host += 1       # the replayed graph holds one vsa_dp_signal (StepSignals.epoch_host)

    def replay_mid(self):
        self._graph_mid.replay()
        self._count_replay()

    def replay_tail(self):
        self._graph_tail.replay()
        return self._static_rgb

    def replay_rest(self):
        self._graph_mid.replay()
        self._count_replay()
        self._graph_tail.replay()
        return self._static_rgb

    def capture_graph(self, **step_kw):
        """Capture one whole step (zero_grad .. backward: ~25 launches on one stream, no
        host sync) into a HIP graph; `replay()` then costs one graph launch instead of
        the per-kernel host overhead of the eager path."""
        self._static_rgb = None
        self._graph_dp = dp = step_kw.get("dp")
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        hook, _ = (dp.on_weights_final, setattr(dp, "on_weights_final", None)) if dp is not None else (None, None)
        with torch.cuda.stream(s):
            for _ in range(2):
                self.step(**step_kw)      # (eager warm-ups: they advance dp's epoch like any eager step, and leave no event behind)
        if dp is not None:
            dp.on_weights_final = hook
        torch.cuda.current_stream().wait_stream(s)
        self._graph = torch.cuda.CUDAGraph()
        with _no_gc(), torch.cuda.graph(self._graph, capture_error_mode=_CAPTURE_MODE):
            self._static_rgb = self.step(**step_kw)
        return self._graph

    def replay(self):
        self._graph.replay()
        self._count_replay()
        return self._static_rgb

    def step(self, record=False, grad_ready=None, dp=None, part=None):
        """zero_grad -> forward -> L1 loss -> backward (trainer.py:118-264 without
        the optimiser step).  Returns the predicted rgb [N,3].

        grad_ready(tensor), if given, is called as soon as a gradient tensor is final on
        the current stream: once with weights.grad, then once per shell with that shell's
        contiguous slice of tables.grad (the hash-grid backward then runs shell by shell),
        so a data-parallel caller can overlap its all-reduce with the rest of backward.

        dp (parallel.StepSignals): the same overlap WITHOUT splitting anything — the step stays one
        stream of launches (graph-capturable); the device publishes "weights.grad final" and "phase p
        of tables.grad final" in dp's flags and parallel.OverlappedStep's side stream waits on them.

        part: None = the whole step; "prefix" = only its parameter- and gradient-free head (ray tile
        order, traversal, mark / compact); "mid" = zero_grad .. MLP backward of a step whose prefix has run;
        "tail" = the hash-grid backward of a step whose mid has run; "rest" = mid + tail."""
        from .composite import composite_fwd_raw, composite_bwd_raw
        N, K = self.nr_rays, self.K
        T, bank = self.timer, self.bank
        acct = getattr(self, "acct", None) or {}     # algorithmic bytes per stage (stats())

        if part == "tail":
            return self._step_tail(record, acct, grad_ready, dp, self._mid_out)
        if part in ("rest", "mid"):
            rays_d, gt, hit_slot, tex_uv = self._prefix_out
        else:
            rays_d, gt, hit_slot, tex_uv = self._step_prefix(record, acct)
            if part == "prefix":
                self._prefix_out = (rays_d, gt, hit_slot, tex_uv)
                return None
        T.run("zero_grad", bank.zero_grads, record, bytes=(bank.tables.numel() + bank.weights.numel()) * 4)
        rgb = self._step_rest(record, acct, dp, rays_d, gt, hit_slot, tex_uv)
        if part == "mid":
            self._mid_out = rgb
            return None
        if dp is not None and dp.on_weights_final is not None and not torch.cuda.is_current_stream_capturing():
            dp.on_weights_final()          # (eager step: weights.grad is final on the current stream here)
        return self._step_tail(record, acct, grad_ready, dp, rgb)

    def _step_prefix(self, record, acct):
        N, T, bank = self.nr_rays, self.timer, self.bank
        rays_o, rays_d, gt = self.rays_o, self.rays_d, self.gt
        if self.image_hw is not None:
            H, W = self.image_hw

            T.run("ray_tile_order",
                  lambda: _lib.call("vsa_tile_order_rays", self.rays_o, self.rays_d, self.gt, self._o_t,
                                    self._d_t, self._gt_t, H, W, _lib.stream_ptr()),
                  record, bytes=N * 72)
            rays_o, rays_d, gt = self._o_t, self._d_t, self._gt_t
        hit_t, hit_slot, hit_uv = T.run(
            "trace", lambda: self.tracer.trace_all(rays_o, rays_d), record,
            bytes=acct.get("trace", 0))
        self._hit_slot = hit_slot
        tex_uv = T.run("nt_mark_compact",
                       lambda: bank.mark_and_compact(hit_slot, hit_uv, self.face_uvs), record,
                       bytes=acct.get("nt_mark_compact", 0))
        return rays_d, gt, hit_slot, tex_uv

    def _step_rest(self, record, acct, dp, rays_d, gt, hit_slot, tex_uv):
        N, K = self.nr_rays, self.K
        T, bank = self.timer, self.bank
        mlp_flops = getattr(self, "mlp_flops_fwd", 0)
        from . import neural_textures as _nt
        if _nt.FUSED_FORWARD is True:      # encode + MLP as one launch (csrc/nt_fused.hip): VSA_NT_FUSED=1
            T.run("nt_encode_mlp_fwd", bank.encode_mlp, record, bytes=acct.get("nt_encode_mlp_fwd", 0),
                  flops=mlp_flops, bound="mfma")
        else:
            T.run("nt_encode_fwd", bank.encode, record, bytes=acct.get("nt_encode_fwd", 0))
            T.run("nt_mlp_fwd", bank.mlp, record, bytes=acct.get("nt_mlp_fwd", 0), flops=mlp_flops,
                  bound="mfma")
        rgb_k, alpha_k, _, _ = T.run(
            "nt_shade_fwd", lambda: bank.shade(hit_slot, tex_uv, rays_d, self.tracer.tris,
                                               act_out=self._act),
            record, bytes=acct.get("nt_shade_fwd", 0))
        self.surfs_rgb, self.surfs_alpha = rgb_k, alpha_k
        # forward composite, d mean|gt - pred| / d pred (utils/losses.py:14-19) and the composite
        # backward in one pass over the shells' colours (vsa_composite_dense_fwd_bwd_l1)
        from .composite import composite_fwd_bwd_l1_raw
        rgb, g_c, g_a = T.run("composite_fwd_bwd",
                              lambda: composite_fwd_bwd_l1_raw(rgb_k, alpha_k, self.bg, gt,
                                                               1.0 / (3.0 * self.loss_rays)),
                              record, bytes=N * (12 + 12 + 32 * K))
        if self.image_hw is not None:      # colours back in the caller's (row-major) order
            T.run("rgb_row_order", lambda: _lib.call("vsa_tile_order", rgb, self._rgb_out, self.image_hw[0],
                                                     self.image_hw[1], 3, 1, _lib.stream_ptr()),
                  record, bytes=N * 24)
            rgb = self._rgb_out
        tris = self.tracer.tris
        T.run("nt_shade_bwd", lambda: bank.backward_shade(hit_slot, tex_uv, rays_d, tris, g_c, g_a,
                                                           self.grad_scale, self._act), record,
              bytes=acct.get("nt_shade_bwd", 0), bound="atomic")
        T.run("nt_mlp_bwd", lambda: bank.backward_mlp(self.grad_scale), record,
              bytes=acct.get("nt_mlp_bwd", 0), flops=2 * mlp_flops, bound="mfma")
        if dp is not None:
            dp.signal_weights()            # epoch += 1 (the hash-grid backward publishes it), weights word = epoch
        return rgb

    def _step_tail(self, record, acct, grad_ready, dp, rgb):
        """The hash-grid backward: the last launch of the step."""
        K, T, bank = self.K, self.timer, self.bank
        enc_bytes = acct.get("nt_encode_bwd", 0)
        if dp is not None:
            T.run("nt_encode_bwd", lambda: bank.backward_encode_phased(self.grad_scale, dp), record,
                  bytes=enc_bytes)
        elif grad_ready is None:
            T.run("nt_encode_bwd", lambda: bank.backward_encode(self.grad_scale), record,
                  bytes=enc_bytes)
        else:
            grad_ready(bank.weights.grad)

            def by_shell():
                for s in range(K):
                    bank.backward_encode(self.grad_scale, shells=(s, s + 1))
                    grad_ready(bank.tables.grad[s * 8:(s + 1) * 8])
            T.run("nt_encode_bwd", by_shell, record, bytes=enc_bytes)
        return rgb
