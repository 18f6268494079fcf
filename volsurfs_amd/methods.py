"""VolSurfs — host-side mirror of the reference's K-shell method for the hot path
(SURVEY §8a row H): same entry points, argument meaning and output dict as
/root/reference/volsurfs_py/methods/volsurfs.py (render_rays :423-761, forward
:763-816) and base_method.py (render :366-541, chunking :407-418; optimiser
:60-94), implemented on the HIP kernels of libvolsurfs_hip.so.  No CPU fallback.
"""
import math

import torch

from . import _lib
from .composite import composite_dense
from .neural_textures import NeuralTextureBank
from .raytrace import RayTracer


class _ShadeStage(torch.autograd.Function):
    """tables, weights -> surfs_rgb [N,K,3], surfs_alpha [N,K] for the traced hits
    (volsurfs.py:492-599), differentiable w.r.t. every texture's parameters."""

    @staticmethod
    def forward(ctx, tables, weights, method, hit_slot, hit_uv, rays_d):
        bank = method.bank
        tex_uv = bank.mark_and_compact(hit_slot, hit_uv, method.face_uvs)
        # (torch.is_grad_enabled() is always False inside Function.forward and parameters keep
        # requires_grad under no_grad: render_rays records whether the call is differentiated)
        need = getattr(method, "_grad_on", True) and (tables.requires_grad or weights.requires_grad)
        bank.evaluate(need_features=need)
        act = torch.empty(hit_slot.shape[0], hit_slot.shape[1], 4, device=hit_slot.device) if need else None
        rgb, alpha, normals, _ = bank.shade(hit_slot, tex_uv, rays_d, method.raytracer.tris,
                                            want_normals=True, act_out=act)
        # backward reads the bank's per-frame state (slot_of, seg_start, texels, features): stamp it
        bank.frame_generation = getattr(bank, "frame_generation", 0) + 1
        ctx.generation = bank.frame_generation
        ctx.method, ctx.saved = method, (hit_slot, tex_uv, rays_d, act)
        ctx.mark_non_differentiable(normals, tex_uv)
        return rgb, alpha, normals, tex_uv

    @staticmethod
    def backward(ctx, g_rgb, g_alpha, *unused):
        method = ctx.method
        bank = method.bank
        hit_slot, tex_uv, rays_d, act = ctx.saved
        if bank.frame_generation != ctx.generation:
            raise _lib.VolsurfsHipError(
                "backward of a render_rays call whose per-frame texel state was overwritten by a "
                "later render_rays: call backward before rendering again")
        # The kernels ACCUMULATE into bank.tables.grad / bank.weights.grad (autograd semantics), so
        # the gradients are written straight into the persistent .grad buffers and None is returned
        # for the two parameters: no 113 MB temporary, no second pass adding it to .grad.
        bank._ensure_grads()
        opt = getattr(method, "optimizer", None)
        # the fused Adam step (or zero_grad) has just cleared them — and nothing has touched the buffers since
        zeroed = bool(opt.grads_are_clean()) if hasattr(opt, "grads_are_clean") else bool(getattr(opt, "_grads_clean", False))
        if hasattr(opt, "mark_grads_dirty"):
            opt.mark_grads_dirty()
        # scale of the fp16 gradient chain (tcnn's loss scale).  By default a power of two that
        # brings the largest incoming per-ray gradient to 4: whatever the loss (mean- or
        # sum-reduced, any batch size) the chain neither underflows nor overflows in f16.  One
        # device->host read per backward call of this autograd path (KShellPipeline, the bench
        # path, knows its loss and uses the ray count without a read-back).
        scale = method.grad_scale
        if scale is None:
            gmax = max(g_rgb.abs().max().item(), g_alpha.abs().max().item())
            scale = 2.0 ** min(max(math.floor(math.log2(4.0 / gmax)), -24), 40) if gmax > 0 else 1.0
        bank.backward(hit_slot, tex_uv, rays_d, method.raytracer.tris, g_rgb.contiguous(),
                      g_alpha.contiguous(), scale, act, grads_zeroed=zeroed)
        return None, None, None, None, None, None


class _LegacyShadeOut(torch.autograd.Function):
    """Behind the legacy models (volsurfs.py:521-599): sigmoid of the colour / alpha outputs, the alpha decay, and the
    scatter of every hit's values to the dense [N,K,.] arrays — one launch forward, one backward
    (vsa_legacy_shade_out_fwd / _bwd) instead of 21 + 12 torch launches and as many autograd nodes.
    y_rgb [M, >= 3], y_alpha [M - a0, >= 1] or None (pre-sigmoid), shell_of / ray_of [M] i64, dirs / nrm [M,3]."""

    @staticmethod
    def forward(ctx, y_rgb, y_alpha, shell_of, ray_of, dirs, nrm, N, K, a0, with_decay):
        y_rgb = y_rgb.contiguous()
        ya = None if y_alpha is None else y_alpha.contiguous()
        M, dev = shell_of.shape[0], y_rgb.device
        dense = torch.zeros(N * K * 7, device=dev)          # one fill for the three dense arrays
        surfs_rgb, surfs_alpha = dense[:N * K * 3].view(N, K, 3), dense[N * K * 3:N * K * 4].view(N, K)
        surfs_normals = dense[N * K * 4:].view(N, K, 3)
        sig_rgb = torch.empty(M, 3, device=dev)
        sig_a = torch.empty(max(M - a0, 1), device=dev) if ya is not None else None
        dec = torch.empty_like(sig_a) if ya is not None else None
        _lib.call("vsa_legacy_shade_out_fwd", y_rgb, y_rgb.shape[1], ya, ya.shape[1] if ya is not None else 0, int(a0),
                  shell_of, ray_of, dirs, nrm, M, N, K, bool(with_decay), surfs_rgb, surfs_alpha, surfs_normals, sig_rgb,
                  sig_a, dec, _lib.stream_ptr())
        ctx.save_for_backward(shell_of, ray_of, sig_rgb, sig_a, dec)
        ctx.meta = (M, N, K, int(a0), y_rgb.shape[1], ya.shape[1] if ya is not None else 0)
        ctx.mark_non_differentiable(surfs_normals)
        return surfs_rgb, surfs_alpha, surfs_normals

    @staticmethod
    def backward(ctx, g_rgb, g_alpha, _g_normals):
        shell_of, ray_of, sig_rgb, sig_a, dec = ctx.saved_tensors
        M, N, K, a0, ld_rgb, ld_a = ctx.meta
        dev = sig_rgb.device
        # (a dense array that did not take part in the loss arrives as None: zeros of its shape)
        g_rgb = torch.zeros(N, K, 3, device=dev) if g_rgb is None else g_rgb.contiguous()
        g_alpha = torch.zeros(N, K, device=dev) if g_alpha is None else g_alpha.contiguous()
        dy_rgb = torch.empty(M, ld_rgb, device=dev)
        dy_a = torch.empty(M - a0, ld_a, device=dev) if ld_a else None
        _lib.call("vsa_legacy_shade_out_bwd", g_rgb, g_alpha, shell_of, ray_of, sig_rgb, sig_a, dec, M, K, a0, dy_rgb,
                  ld_rgb, dy_a, ld_a, _lib.stream_ptr())
        return dy_rgb, dy_a, None, None, None, None, None, None, None, None


class VolSurfs(torch.nn.Module):
    """K nested mesh shells with SH neural-texture appearance.

    tensor_meshes: list (inner -> outer, utils/mesh_loaders.py:28-30) of objects with
    .vertices [V,3], .faces [F,3], .get_faces_uvs() [F,3,2] (volsurfs.py:82-117, 511).
    hyper-parameters follow config/volsurfs/base_5.cfg."""

    def __init__(self, tensor_meshes, max_rays=16384, sh_degree=3, transp_view_dep=True,
                 sh_range=(15, 15, 15, 15), textures_res=(2048, 1024, 512, 256),
                 is_inner_mesh_solid=False, with_alpha_decay=True, bg_color=(1.0, 1.0, 1.0),
                 bg_model=None, bounding_primitive=None, nr_samples_bg=32, lr=1e-3, seed=42,
                 using_neural_textures=True, appearance_predict_sh_coeffs=False,
                 rgb_mlp_layers_dims=(128, 128, 64), rgb_pos_encoder_type="gridhash",
                 rgb_dir_encoder_type="spherical_harmonics", rgb_view_dep=True,
                 rgb_normal_dep=False, transp_normal_dep=False, rgb_nr_iters_for_c2f=0,
                 are_volsurfs_colors_indep=True, are_volsurfs_alphas_indep=True, bb_sides=2.0,
                 lr_milestones=(100000, 150000, 180000, 190000), nr_warmup_iters=3000,
                 using_neural_textures_anchor=False, using_neural_textures_lerp=True,
                 using_sh_quantization=True, using_sh_squeezing=True):
        """using_neural_textures_anchor / _lerp, using_sh_quantization, using_sh_squeezing: the reference's
        hyper-parameters of the same names (config/volsurfs/base_5.cfg:16-19 -> volsurfs.py:149-153); every
        combination the reference accepts is built, the ones it exits on raise in NeuralTextureBank.
        are_volsurfs_colors_indep / are_volsurfs_alphas_indep = 0 (volsurfs.py:159-165, 200-206): ONE colour / alpha
        model for all shells, on both appearance branches (neural textures: NeuralTextureBank(shared_rgb / shared_alpha))."""
        super().__init__()
        self.using_neural_textures = using_neural_textures
        self.with_alpha_decay = with_alpha_decay
        self.tensor_meshes = tensor_meshes
        self.nr_meshes = len(tensor_meshes)
        dev = tensor_meshes[0].vertices.device
        self.raytracer = RayTracer(tensor_meshes)                      # volsurfs.py:128
        fu = []
        for m, off, n in zip(tensor_meshes, self.raytracer.mesh_tri_offset, self.raytracer.mesh_nr_tris):
            ids = self.raytracer.slot_face_id[off:off + n].long()
            fu.append(m.get_faces_uvs().reshape(-1, 6)[ids])
        self.face_uvs = torch.cat(fu, 0).contiguous()
        self.bank, self.models = None, torch.nn.ModuleDict()
        if using_neural_textures:
            self.bank = NeuralTextureBank(self.nr_meshes, max_rays, sh_degree=sh_degree,
                                          alpha_sh_degree=sh_degree if transp_view_dep else 0,
                                          sh_range=sh_range, textures_res=textures_res,
                                          inner_solid=is_inner_mesh_solid,
                                          with_alpha_decay=with_alpha_decay, device=dev, seed=seed,
                                          anchor=bool(using_neural_textures_anchor),
                                          lerp=bool(using_neural_textures_lerp),
                                          quantize_output=bool(using_sh_quantization),
                                          squeeze_output=bool(using_sh_squeezing),
                                          shared_rgb=not are_volsurfs_colors_indep,
                                          shared_alpha=not are_volsurfs_alphas_indep)
            self.colors_indep, self.alphas_indep = bool(are_volsurfs_colors_indep), bool(are_volsurfs_alphas_indep)
        else:
            # legacy appearance branch (volsurfs.py:208-300): one RGB / ColorSH per shell (or one
            # for all shells), alpha model None for a solid inner mesh
            from .models import ColorSH, RGB

            def make(out_channels, view_dep, normal_dep):
                if appearance_predict_sh_coeffs:
                    return ColorSH(in_channels=3, out_channels=out_channels,
                                   mlp_layers_dims=list(rgb_mlp_layers_dims),
                                   pos_encoder_type=rgb_pos_encoder_type, sh_deg=sh_degree,
                                   normal_dep=normal_dep, nr_iters_for_c2f=rgb_nr_iters_for_c2f,
                                   bb_sides=bb_sides, device=dev)
                return RGB(in_channels=3, out_channels=out_channels,
                           mlp_layers_dims=list(rgb_mlp_layers_dims),
                           pos_encoder_type=rgb_pos_encoder_type,
                           dir_encoder_type=rgb_dir_encoder_type, sh_deg=sh_degree,
                           view_dep=view_dep, normal_dep=normal_dep,
                           nr_iters_for_c2f=rgb_nr_iters_for_c2f, bb_sides=bb_sides, device=dev)
            for i in range(self.nr_meshes):
                self.models[f"rgb_{i}" if are_volsurfs_colors_indep else "rgb"] = \
                    make(3, rgb_view_dep, rgb_normal_dep)
                if not are_volsurfs_colors_indep:
                    break
            self.solid_inner = is_inner_mesh_solid
            for i in range(self.nr_meshes):
                key = f"alpha_{i}" if are_volsurfs_alphas_indep else "alpha"
                if not (is_inner_mesh_solid and i == 0):
                    self.models[key] = make(1, transp_view_dep, transp_normal_dep)
                if not are_volsurfs_alphas_indep:
                    break
            self.colors_indep, self.alphas_indep = are_volsurfs_colors_indep, are_volsurfs_alphas_indep
        self.max_rays = max_rays
        # volsurfs.py:686-702: constant colour, or (bg_color None) a learned contracted
        # background model rendered through the packed ops (utils/background.py)
        self.bg_color = None if bg_color is None else torch.tensor([bg_color], device=dev,
                                                                   dtype=torch.float32)
        self.bg_model, self.bounding_primitive, self.nr_samples_bg = bg_model, bounding_primitive, nr_samples_bg
        if self.bg_color is None and (bg_model is None or bounding_primitive is None):
            raise _lib.VolsurfsHipError("bg_color=None needs bg_model and bounding_primitive")
        self.grad_scale = None      # None = chosen per backward call from the incoming gradients (see _ShadeStage)
        self.save_checkpoints_path = self.load_checkpoints_path = None
        self.profiler = None        # a trainer.Profiler (or any object with start(name) / end(name))
        self.baked = None
        self.is_training = True
        self.lr = lr
        self.optimizer = None
        # params/hyper_params.py:8-11; the schedulers are created like base_method.py:60-76
        # (decay at init_optim) and volsurfs.py:774-783 (warm-up at the first forward)
        self.lr_milestones, self.nr_warmup_iters = list(lr_milestones), nr_warmup_iters
        self.scheduler_lr_decay = self.lr_scheduler = None

    @classmethod
    def from_meshes_path(cls, meshes_path, load_checkpoints_path, meshes_indices=None, start_iter_nr=0,
                         device="cuda", **kwargs):
        """The constructor path of methods/volsurfs.py:62-128: shells from `meshes_path` (.obj / .ply,
        sorted by isolevel, optionally a subset), copied into `<checkpoints>/meshes/` at the first
        iteration and loaded from there when a run resumes; then the BVHs are built."""
        from .mesh import prepare_run_meshes
        meshes = prepare_run_meshes(meshes_path, meshes_indices, load_checkpoints_path, start_iter_nr,
                                    require_uvs=kwargs.get("using_neural_textures", True), device=device)
        m = cls(meshes, **kwargs)
        m.load_checkpoints_path = m.save_checkpoints_path = load_checkpoints_path
        return m

    # -- optimiser: apex FusedAdam(betas (0.9, 0.99), eps 1e-15, wd 0) of
    # base_method.py:87-94 == Adam with the same hyper-parameters
    def init_optim(self, world=1, rank=0, group=None, sharded=False):
        """sharded=True (world > 1): optimiser state and update sharded over the ranks
        (optim.ShardedFusedAdam: reduce-scatter -> Adam on a slice -> all-gather of the f16 copies);
        its step() replaces the gradient all-reduce + step of trainer.train_step."""
        from .optim import FusedAdam, ShardedFusedAdam
        self._dist_rank, self._dist_world = int(rank), int(world)
        half = {}
        if self.bank is not None:
            params = [self.bank.tables, self.bank.weights]
            half = {self.bank.tables: self.bank.tables_h, self.bank.weights: self.bank.weights_h}
        else:
            params = list(self.models.parameters())
        if isinstance(self.bg_model, torch.nn.Module):
            params += list(self.bg_model.parameters())
        # one HIP launch per step for every tensor, fused with the f16 refresh of the texture
        # parameters and the next iteration's zero_grad (volsurfs_amd/optim.py, csrc/adam.hip)
        if sharded and world > 1:
            self.optimizer = ShardedFusedAdam(params, world, rank, group, lr=self.lr, betas=(0.9, 0.99),
                                              eps=1e-15, weight_decay=0.0, half_copies=half)
        else:
            self.optimizer = FusedAdam(params, lr=self.lr, betas=(0.9, 0.99), eps=1e-15,
                                       weight_decay=0.0, half_copies=half)
        from .schedulers import MultiStepLR
        self.scheduler_lr_decay = MultiStepLR(self.optimizer, milestones=self.lr_milestones, gamma=0.3)
        return self.optimizer

    def optim_step(self, overlap=False):
        """overlap=True (neural textures only): the Adam launch goes to a side stream and the next
        reader of the texture parameters on the current stream waits for it (bank.wait_params, in
        encode / mlp): the next iteration's ray batch, traversal and texel compaction run beside the
        HBM-bound update.  Code that reads `bank.tables` / `.weights` directly with torch ops must
        call `bank.wait_params()` first."""
        if overlap and self.bank is not None and self.bg_model is None:
            if getattr(self, "_optim_stream", None) is None:
                self._optim_stream = torch.cuda.Stream()
            self.bank._params_event = self.optimizer.step(stream=self._optim_stream)
        else:
            self.optimizer.step()    # (refreshes bank.tables_h / weights_h inside the kernel)

    def sync_params(self):
        """Every reader of the parameters or the optimiser state outside the texture kernels
        (save / load / bake / state_dict) goes through here: an `optim_step(overlap=True)` may
        still be writing them on its side stream."""
        if self.bank is not None:
            self.bank.wait_params()
        if hasattr(getattr(self, "optimizer", None), "gather_masters"):
            self.optimizer.gather_masters()      # sharded Adam: fp32 masters are per-slice until gathered

    legacy_grouped = True     # class-wide switch: False = the per-shell loop (tests compare the two)
    legacy_fused_step = __import__("os").environ.get("VSA_LEGACY_FUSED_STEP", "1") != "0"   # trainer: no autograd at all
    legacy_two_streams = __import__("os").environ.get("VSA_LEGACY_TWO_STREAMS", "1") != "0"  # ... colour / alpha chains side by side
    legacy_fused_glue = __import__("os").environ.get("VSA_LEGACY_FUSED_GLUE", "1") != "0"   # grouped path: hit preparation,
    # sigmoid / decay / scatter and (forward()) composite + L1 as one launch each instead of torch expressions

    def _legacy_groupable(self, x_probe):
        """The grouped path covers the configuration BASELINE configs[2] trains: every model an `RGB`
        with a position encoder, no geometry features, and per type (rgb / alpha) one MLP architecture
        and one set of input flags."""
        from .models import RGB, mlps_groupable
        for typ in ("rgb", "alpha"):
            mods = [m for k, m in self.models.items() if k.split("_")[0] == typ]
            if not mods:
                continue
            if not all(isinstance(m, RGB) and m.pos_dep and not m.geom_feat_dep for m in mods):
                return False
            f0 = (mods[0].view_dep, mods[0].normal_dep, mods[0].sh_deg, mods[0].dir_encoder_type)
            if any((m.view_dep, m.normal_dep, m.sh_deg, m.dir_encoder_type) != f0 for m in mods):
                return False
            if not mlps_groupable([m.mlp for m in mods], x_probe):
                return False
        return True

    def _shade_legacy_grouped(self, rays_o, rays_d, hit_t, hit_slot, iter_nr, ahead=None, tape=None):
        """The same arithmetic as the per-shell loop with the hits of ALL shells prepared at once and
        each model type evaluated as one grouped op (models._FusedMLPGrouped): ~80 torch ops per
        call instead of ~350 — the legacy training loop is bound by the host's op dispatch."""
        from .encodings import permuto_hash_encode_grouped, permuto_hash_encoders_groupable
        from .models import fused_mlp_grouped
        N, K = rays_o.shape[0], self.nr_meshes
        dev = rays_o.device

        def dense_zeros():
            return torch.zeros(N, K, 3, device=dev), torch.zeros(N, K, device=dev), torch.zeros(N, K, 3, device=dev)
        if ahead is not None:          # compacted when the traversal was queued (trace_ahead)
            counts = ahead.counts()
            M = int(sum(counts))
            shell_of, ray_of = ahead.idx[:M, 0], ahead.idx[:M, 1]
        else:
            shell_of, ray_of = (hit_slot >= 0).nonzero(as_tuple=True)      # sorted by shell, then ray
            counts = torch.bincount(shell_of, minlength=K).tolist()
            M = int(sum(counts))
        if M == 0:
            return dense_zeros()
        begin = [0]
        for c in counts:
            begin.append(begin[-1] + c)
        fused_glue = VolSurfs.legacy_fused_glue
        if fused_glue:
            # the hits' points, directions and face normals in one launch (18 torch launches otherwise)
            # (torch.nonzero lays its [n, 2] result out column by column: the two columns are contiguous as they are)
            shell_of, ray_of = shell_of.contiguous(), ray_of.contiguous()
            pts, d, nrm = (torch.empty(M, 3, device=dev) for _ in range(3))
            _lib.call("vsa_legacy_hit_prep", rays_o, rays_d, hit_t, hit_slot, self.raytracer.tris, shell_of, ray_of, M, N,
                      pts, d, nrm, _lib.stream_ptr())
        else:
            slots = hit_slot[shell_of, ray_of].long()
            tri = self.raytracer.tris[slots]
            nrm = torch.nn.functional.normalize(torch.cross(tri[:, 4:7], tri[:, 8:11], dim=1), dim=1)
            d = rays_d[ray_of]
            pts = rays_o[ray_of] + hit_t[shell_of, ray_of][:, None] * d

        def evaluate(typ, indep, first_shell):
            """sigmoid(MLP(cat(pos enc, dir enc, normals))) for the hits of shells >= first_shell (fused glue: the
            MLP's output; the sigmoid is taken in _LegacyShadeOut)."""
            a0 = begin[first_shell]
            if a0 == M:
                return None
            if indep:
                mods = [self.models[f"{typ}_{i}"] for i in range(first_shell, K)]
                sizes = counts[first_shell:]
            else:
                mods, sizes = [self.models[typ]], [M - a0]
            m0 = mods[0]
            pos_encs = [mod.pos_encoder for mod in mods]
            if tape is not None:
                # the autograd-free step (fused_legacy_forward): the same grouped launches called directly, their
                # backward closures kept on the tape
                from .encodings import permuto_hash_encode_grouped_manual
                from .models import cat_rows16, fused_mlp_grouped_manual
                enc, enc_bwd = permuto_hash_encode_grouped_manual(pos_encs, pts[a0:], sizes, iter_nr=iter_nr)
                parts = [enc]
                if m0.view_dep:
                    parts.append(m0.dir_encoder(d[a0:], iter_nr=iter_nr))
                if m0.normal_dep:
                    parts.append(nrm[a0:])
                # (rows padded to a multiple of 4 floats — 66 -> 68: the MLP kernels move them as 16-byte groups)
                y, mlp_bwd = fused_mlp_grouped_manual([mod.mlp for mod in mods], cat_rows16(parts), sizes)
                n_enc = enc.shape[1]
                tape[typ] = lambda gy: enc_bwd(mlp_bwd(gy)[:, :n_enc])
                return y
            if VolSurfs.legacy_grouped_encode and len(mods) > 1 and permuto_hash_encoders_groupable(pos_encs, pts):
                # one autograd node for all shells' position encodings (encodings._PermutoEncodeGrouped)
                parts = [permuto_hash_encode_grouped(pos_encs, pts[a0:], sizes, iter_nr=iter_nr)]
            else:
                encs, a = [], a0
                for mod, n in zip(mods, sizes):
                    if n:
                        f = mod.pos_encoder(pts[a:a + n], iter_nr=iter_nr)
                        encs.append(f[0] if isinstance(f, tuple) else f)
                    a += n
                parts = [torch.cat(encs, 0) if len(encs) > 1 else encs[0]]
            if m0.view_dep:
                with torch.no_grad():
                    parts.append(m0.dir_encoder(d[a0:], iter_nr=iter_nr))
            if m0.normal_dep:
                parts.append(nrm[a0:])
            y = fused_mlp_grouped([mod.mlp for mod in mods], torch.cat(parts, 1), sizes)
            return y if fused_glue else torch.sigmoid(y)
        first = 1 if self.solid_inner else 0
        has_alpha = any(k.split("_")[0] == "alpha" for k in self.models)
        if fused_glue:
            two = tape is not None and has_alpha and VolSurfs.legacy_two_streams
            if two:
                # the colour and the alpha models are independent chains of small launches (49 k rows in ten groups
                # fill the chip only in part): the alpha chain on a side stream beside the colour chain
                side = getattr(self, "_alpha_stream", None)
                if side is None:
                    side = self._alpha_stream = torch.cuda.Stream(device=dev)
                main = torch.cuda.current_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    y_alpha = evaluate("alpha", self.alphas_indep, first)
                y_rgb = evaluate("rgb", self.colors_indep, 0)
                main.wait_stream(side)
                if y_alpha is not None:
                    y_alpha.record_stream(main)
                tape["side"] = side
            else:
                y_rgb = evaluate("rgb", self.colors_indep, 0)
                y_alpha = evaluate("alpha", self.alphas_indep, first) if has_alpha else None
            if tape is not None:
                from .encodings import ManualCtx
                ctx = ManualCtx()
                out = _LegacyShadeOut.forward(ctx, y_rgb, y_alpha, shell_of, ray_of, d, nrm, N, K, begin[first],
                                              bool(self.with_alpha_decay))
                tape["out"] = lambda g_rgb, g_alpha: _LegacyShadeOut.backward(ctx, g_rgb, g_alpha, None)[:2]
                return out
            return _LegacyShadeOut.apply(y_rgb, y_alpha, shell_of, ray_of, d, nrm, N, K, begin[first],
                                         bool(self.with_alpha_decay))
        surfs_rgb, surfs_alpha, surfs_normals = dense_zeros()
        pred = evaluate("rgb", self.colors_indep, 0)
        surfs_rgb = surfs_rgb.index_put((ray_of, shell_of), pred[:, :3])
        alpha = torch.ones(M, device=dev)
        if has_alpha:
            pa = evaluate("alpha", self.alphas_indep, first)
            if pa is not None:
                av = pa[:, 0]
                if self.with_alpha_decay:
                    with torch.no_grad():
                        dot = torch.sum(-d[begin[first]:] * nrm[begin[first]:], dim=1).clamp(0.0, 1.0)
                        decay = torch.sigmoid(10.0 * dot) * 2.0 - 1.0
                    av = av * decay
                alpha = torch.cat([alpha[:begin[first]], av]) if begin[first] else av
        surfs_alpha = surfs_alpha.index_put((ray_of, shell_of), alpha)
        surfs_normals = surfs_normals.index_put((ray_of, shell_of), nrm)
        return surfs_rgb, surfs_alpha, surfs_normals

    def supports_fused_legacy_step(self, rays_o, gt_mask=None, is_training_masked=False):
        """The legacy branch's training step without autograd (fused_legacy_forward / _backward): the configuration
        the grouped launches cover (BASELINE configs[2]), constant background, unmasked L1, every model's encoder a
        groupable permutohedral one."""
        if (self.using_neural_textures or not (VolSurfs.legacy_fused_glue and VolSurfs.legacy_grouped
                                               and VolSurfs.legacy_grouped_encode and VolSurfs.legacy_fused_step)
                or self.bg_color is None or (is_training_masked and gt_mask is not None)
                or rays_o.shape[0] > self.max_rays or not self._legacy_groupable(rays_o)):
            return False
        from .encodings import permuto_hash_encoders_groupable
        for typ, indep in (("rgb", self.colors_indep), ("alpha", self.alphas_indep)):
            mods = [m for k, m in self.models.items() if k.split("_")[0] == typ]
            if mods and not permuto_hash_encoders_groupable([m.pos_encoder for m in mods], rays_o):
                return False
        return True

    @torch.no_grad()
    def fused_legacy_forward(self, rays_o, rays_d, gt_rgb, iter_nr=0, ahead=None, loss_weight=1.0, is_first_iter=False):
        """Forward pass + loss of the legacy training step with every launch called directly (no autograd graph):
        traversal (or the look-ahead context), hit preparation, grouped encoders / MLPs, sigmoid-decay-scatter,
        composite + L1 (which already leaves the gradients w.r.t. the dense colour / alpha arrays, scaled by
        loss_weight).  Returns (loss [] = the unweighted mean, state for fused_legacy_backward)."""
        from .composite import composite_fwd_bwd_l1_raw, l1_mean
        self._warmup_scheduler(is_first_iter)
        rays_o, rays_d = rays_o.contiguous(), rays_d.contiguous()
        if ahead is not None and (ahead.rays_o.data_ptr() != rays_o.data_ptr() or ahead.rays_o.shape != rays_o.shape):
            ahead = None
        if ahead is not None:
            ahead.join()
            hit_t, hit_slot = ahead.hit_t, ahead.hit_slot
        else:
            hit_t, hit_slot, _ = self._trace_now(rays_o, rays_d)
        tape = {}
        rgb_k, alpha_k, _ = self._shade_legacy(rays_o, rays_d, hit_t, hit_slot, iter_nr, ahead, tape=tape)
        if ahead is not None:
            self.last_nr_hits = int(sum(ahead.counts()))
        N = rays_o.shape[0]
        rgb, g_c, g_a = composite_fwd_bwd_l1_raw(rgb_k, alpha_k, self.bg_color, gt_rgb.contiguous(),
                                                 float(loss_weight) / (3.0 * N))
        return l1_mean(rgb, gt_rgb), (tape, g_c, g_a)

    @torch.no_grad()
    def fused_legacy_backward(self, state):
        """Backward pass of fused_legacy_forward: the gradients are added into the parameters' .grad buffers."""
        tape, g_c, g_a = state
        if "out" not in tape:            # no ray hit anything
            return
        dy_rgb, dy_alpha = tape["out"](g_c, g_a)
        side = tape.get("side")
        if side is not None and "alpha" in tape and dy_alpha is not None:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            dy_alpha.record_stream(side)
            with torch.cuda.stream(side):
                tape["alpha"](dy_alpha)
            tape["rgb"](dy_rgb)
            main.wait_stream(side)
            return
        tape["rgb"](dy_rgb)
        if "alpha" in tape and dy_alpha is not None:
            tape["alpha"](dy_alpha)

    def _shade_legacy(self, rays_o, rays_d, hit_t, hit_slot, iter_nr, ahead=None, tape=None):
        """volsurfs.py:486-599, legacy branch: per shell, the hit points / view directions /
        face normals go through that shell's RGB (or ColorSH) models; alpha decay; dense scatter."""
        if VolSurfs.legacy_grouped and self._legacy_groupable(rays_o):
            return self._shade_legacy_grouped(rays_o, rays_d, hit_t, hit_slot, iter_nr, ahead, tape)
        N, K = rays_o.shape[0], self.nr_meshes
        dev = rays_o.device
        surfs_rgb = torch.zeros(N, K, 3, device=dev)
        surfs_alpha = torch.zeros(N, K, device=dev)
        surfs_normals = torch.zeros(N, K, 3, device=dev)
        # ONE compaction for all shells (the reference synchronises per shell: `any_hit`, then a
        # boolean-mask gather per buffer, volsurfs.py:481-507): the hits sorted by shell, and the
        # per-shell counts read back once
        shell_of, ray_of = (hit_slot >= 0).nonzero(as_tuple=True)
        counts = torch.bincount(shell_of, minlength=K).tolist()
        off = 0
        for i in range(K):
            c = counts[i]
            if c == 0:
                continue
            rows = ray_of[off:off + c]
            off += c
            slots = hit_slot[i][rows].long()
            tri = self.raytracer.tris[slots]                                    # [M,12]: v0+id, e1, e2
            nrm = torch.nn.functional.normalize(torch.cross(tri[:, 4:7], tri[:, 8:11], dim=1), dim=1)
            d = rays_d[rows]
            pts = rays_o[rows] + hit_t[i][rows][:, None] * d
            m_rgb = self.models[f"rgb_{i}" if self.colors_indep else "rgb"]
            pred = m_rgb(points=pts, samples_dirs=d, normals=nrm, iter_nr=iter_nr)
            col = torch.full_like(rows, i)
            surfs_rgb = surfs_rgb.index_put((rows, col), pred[:, :3])
            key = f"alpha_{i}" if self.alphas_indep else "alpha"
            if key not in self.models or (self.solid_inner and i == 0):
                a = torch.ones(c, device=dev)
            else:
                a = self.models[key](points=pts, samples_dirs=d, normals=nrm, iter_nr=iter_nr)[:, 0]
                if self.with_alpha_decay:
                    with torch.no_grad():
                        dot = torch.sum(-d * nrm, dim=1).clamp(0.0, 1.0)
                        decay = torch.sigmoid(10.0 * dot) * 2.0 - 1.0
                    a = a * decay
            surfs_alpha = surfs_alpha.index_put((rows, col), a)
            surfs_normals = surfs_normals.index_put((rows, col), nrm)
        return surfs_rgb, surfs_alpha, surfs_normals

    legacy_grouped_encode = __import__("os").environ.get("VSA_GROUPED_ENCODE", "1") != "0"   # A/B switch
    look_ahead = True      # trainer.train_step_from_reel queues the next batch's traversal a step ahead (legacy models)
    look_ahead_stream = __import__("os").environ.get("VSA_LOOK_AHEAD_STREAM", "1") != "0"   # ... on a side stream

    class _TraceAhead:
        """The traversal and hit compaction of a batch, queued ahead of time (trace_ahead)."""

        def __init__(self, rays_o, rays_d, hit, idx, counts_dev):
            self.rays_o, self.rays_d = rays_o, rays_d
            self.hit_t, self.hit_slot, self.hit_uv = hit
            self.idx = idx                                  # [K*N, 2] (shell, ray), sorted, zero padded
            self._host = torch.zeros(counts_dev.shape[0], dtype=torch.int64).pin_memory()
            self._host.copy_(counts_dev, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()                            # (on the stream the context was built on)
            self._counts = None
            self._joined = False

        def join(self):
            """Built on a side stream (trace_ahead with look_ahead_stream): the consumer's stream waits for it once,
            and the allocator is told that the context's arrays are now in use there."""
            if self._joined:
                return
            self._joined = True
            cur = torch.cuda.current_stream()
            cur.wait_event(self._event)
            for t in (self.hit_t, self.hit_slot, self.hit_uv, self.idx):
                t.record_stream(cur)

        def counts(self):
            """Hits per shell as Python ints: waits for the traversal only, not for whatever was
            queued behind it."""
            if self._counts is None:
                self._event.synchronize()
                self._counts = self._host.tolist()
            return self._counts

    def _trace_now(self, rays_o, rays_d):
        """The traversal on the current stream — behind whatever look-ahead traversal is still in flight on the side
        stream: the tracer's launch-order feedback buffer belongs to one launch at a time."""
        side = getattr(self, "_ahead_stream", None)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        return self.raytracer.trace_all(rays_o, rays_d)

    def trace_ahead(self, rays_o, rays_d):
        """Queue the traversal of a batch and the compaction of its hits NOW and read the hit
        counts later (`render_rays(..., ahead=ctx)`).  The legacy shading needs the counts on the
        host (one model per shell); read right after the traversal they stall the host until
        everything queued before — the previous iteration's backward pass and optimiser step — has
        run.  The training loop therefore queues the NEXT batch's traversal between this batch's
        forward and backward passes: by the time the host has queued the backward pass, the counts
        of the next batch are already there, and host and device work overlap (the traversal reads
        no parameter, so the order does not matter).  No sync in here: `nonzero_static` over the
        whole [K, N] mask, padded."""
        rays_o, rays_d = rays_o.contiguous(), rays_d.contiguous()
        if not VolSurfs.look_ahead_stream:
            hit = self.raytracer.trace_all(rays_o, rays_d)
            mask = hit[1] >= 0
            idx = torch.nonzero_static(mask, size=mask.numel(), fill_value=0)
            ctx = VolSurfs._TraceAhead(rays_o, rays_d, hit, idx, mask.sum(1))
            ctx._joined = True
            return ctx
        # on a side stream: the traversal of a training batch is one long latency chain on a nearly empty chip
        # (profiles/NOTEBOOK.md round 5) — beside this batch's backward pass it costs the step nothing
        side = getattr(self, "_ahead_stream", None)
        if side is None:
            side = self._ahead_stream = torch.cuda.Stream(device=rays_o.device)
        ready = torch.cuda.Event()
        ready.record()                                      # the rays were written on the current stream
        with torch.cuda.stream(side):
            side.wait_event(ready)
            hit = self.raytracer.trace_all(rays_o, rays_d)
            mask = hit[1] >= 0
            idx = torch.nonzero_static(mask, size=mask.numel(), fill_value=0)
            return VolSurfs._TraceAhead(rays_o, rays_d, hit, idx, mask.sum(1))

    def render_rays(self, rays_o, rays_d, iter_nr=None, return_samples=True, ahead=None, **kwargs):
        """volsurfs.py:423-761: returns {"renders": {"ray_traced": {...}}, "samples_3d",
        "samples_grad"} with the reference's keys, shapes and dtypes."""
        N, K = rays_o.shape[0], self.nr_meshes
        if N > self.max_rays:
            raise _lib.VolsurfsHipError(f"{N} rays > max_rays={self.max_rays}; use render()")
        rays_o, rays_d = rays_o.contiguous(), rays_d.contiguous()
        prof = getattr(self, "profiler", None)      # section names of the reference
        if prof is not None:
            prof.start("meshes_raytracing")
        if ahead is not None and (ahead.rays_o.data_ptr() != rays_o.data_ptr() or ahead.rays_o.shape != rays_o.shape):
            ahead = None                                                       # not this batch's: trace now
        if ahead is not None:
            ahead.join()
            hit_t, hit_slot, hit_uv = ahead.hit_t, ahead.hit_slot, ahead.hit_uv
        else:
            hit_t, hit_slot, hit_uv = self._trace_now(rays_o, rays_d)              # :476-485, one launch
        if prof is not None:
            prof.end("meshes_raytracing")
            prof.start("ray_color_inference")
        if self.using_neural_textures:
            self._grad_on = torch.is_grad_enabled()
            rgb_k, alpha_k, normals, tex_uv = _ShadeStage.apply(self.bank.tables, self.bank.weights,
                                                                self, hit_slot, hit_uv, rays_d)
        else:
            rgb_k, alpha_k, normals = self._shade_legacy(rays_o, rays_d, hit_t, hit_slot, iter_nr, ahead)
            tex_uv = None
        if prof is not None:
            prof.end("ray_color_inference")
        if self.bg_color is not None:
            rgb_bg = self.bg_color
        else:
            from .background import intersect_bounding_primitive, render_contracted_bg
            raycast = intersect_bounding_primitive(self.bounding_primitive, rays_o, rays_d)   # :434
            if prof is not None:
                prof.start("render_contracted_bg")
            rgb_bg = render_contracted_bg(self.bg_model, raycast, self.nr_samples_bg,
                                          jitter_samples=self.is_training, iter_nr=iter_nr)["pred_rgb"]
            if prof is not None:
                prof.end("render_contracted_bg")
        if prof is not None:
            prof.start("render_fg")
        out = composite_dense(rgb_k, alpha_k, rgb_bg)                          # :601-640, 704-708
        if prof is not None:
            prof.end("render_fg")
        renders = {
            "rgb": out["rgb"], "rgb_fg": out["rgb_fg"], "rgb_bg": out["rgb_bg"],
            "surfs_alpha": out["surfs_alpha"], "surfs_rgb": out["surfs_rgb"],
            "surfs_normals": normals, "surfs_blending_weights": out["surfs_blending_weights"],
            "bg_transmittance": out["bg_transmittance"],
            "surfs_uvs": None if tex_uv is None else tex_uv.permute(1, 0, 2).contiguous(),  # [N,K,2], :509-516
        }
        res = {"renders": {"ray_traced": renders}, "samples_3d": None, "samples_grad": None}
        if ahead is not None and not self.using_neural_textures:
            self.last_nr_hits = int(sum(ahead.counts()))     # the trainer's sample count without the compaction
        if return_samples:       # :713-716 (boolean compaction = a host sync, as in the reference)
            hits = (hit_slot >= 0).t()
            pts = rays_o[:, None, :] + hit_t.t()[..., None] * rays_d[:, None, :]
            res["samples_3d"] = pts[hits].reshape(-1, 3)
            res["samples_grad"] = normals[hits].reshape(-1, 3)
        return res

    # -- baked-texture inference (renderers/mesh_renderer.py:113-201 extended
    # to K shells: 8-bit SH-coefficient textures -> bilinear fetch -> SH eval -> composite)
    @torch.no_grad()
    def bake(self):
        """Evaluate all texels of all 2K neural textures once into 8-bit texel rows."""
        self.sync_params()
        b = self.bank
        full = NeuralTextureBank.full_capacity_rays(b.tex_res)
        baked = NeuralTextureBank(self.nr_meshes, full, sh_degree=b.rgb_degrees - 1,
                                  alpha_sh_degree=b.alpha_degrees - 1,
                                  sh_range=[-float(b.plan.sh_lo[d]) for d in range(4)],
                                  textures_res=b.tex_res, inner_solid=bool(b.plan.inner_solid),
                                  with_alpha_decay=bool(b.plan.with_alpha_decay),
                                  device=b.tables.device, training=False, anchor=b.anchor, lerp=not b.anchor,
                                  shared_rgb=b.shared_rgb, shared_alpha=b.shared_alpha)
        baked.tables.copy_(b.tables)
        baked.weights.copy_(b.weights)
        baked.refresh_half_params()
        baked.bake_all()
        baked.features = None            # 128 B per (texel, model): only the 8-bit rows are kept
        self.baked = baked
        return baked

    @torch.no_grad()
    def render_baked(self, rays_o, rays_d, chunk=1 << 20):
        """Render from the baked textures: trace -> per-hit uv -> shade -> composite (no hash
        grid, no MLP).  Returns the `ray_traced` dict of render()."""
        outs = []
        for a in range(0, rays_o.shape[0], chunk):
            o, d = rays_o[a:a + chunk].contiguous(), rays_d[a:a + chunk].contiguous()
            hit_t, hit_slot, hit_uv = self._trace_now(o, d)
            tex_uv = self.baked.tex_uv_only(hit_slot, hit_uv, self.face_uvs)
            rgb_k, alpha_k, normals, _ = self.baked.shade(hit_slot, tex_uv, d, self.raytracer.tris,
                                                          want_normals=True)
            bg = self.bg_color if self.bg_color is not None else torch.ones(1, 3, device=o.device)
            out = composite_dense(rgb_k, alpha_k, bg)
            out["surfs_normals"] = normals
            outs.append(out)
        return outs[0] if len(outs) == 1 else {k: torch.cat([o_[k] for o_ in outs], 0) for k in outs[0]}

    # -- fused training path: the same forward + mean-L1 + backward as `forward(...)` followed by
    # `loss.backward()`, as ONE sequence of C-ABI launches on the current stream (no autograd
    # graph, no temporaries for the gradients, no host synchronisation): trace -> mark/compact ->
    # encode -> MLP -> shade -> composite+L1+its backward (one launch) -> shade_bwd -> MLP bwd ->
    # encode bwd, accumulating into the persistent bank.tables.grad / bank.weights.grad.
    def supports_fused_step(self, gt_mask=None, is_training_masked=False):
        return (self.bank is not None and self.bg_color is not None and self.baked is None
                and not (is_training_masked and gt_mask is not None))

    def _warmup_scheduler(self, is_first_iter):
        if is_first_iter and self.scheduler_lr_decay is not None:              # volsurfs.py:774-783
            from .schedulers import GradualWarmupScheduler
            if self.nr_warmup_iters > 0:
                self.lr_scheduler = GradualWarmupScheduler(self.optimizer, multiplier=1,
                                                           total_epoch=self.nr_warmup_iters,
                                                           after_scheduler=self.scheduler_lr_decay)
            else:
                self.lr_scheduler = self.scheduler_lr_decay

    def fused_forward_backward(self, rays_o, rays_d, gt_rgb, loss_weight=1.0, is_first_iter=False):
        """Gradients of `loss_weight * mean|gt_rgb - rgb|` (utils/losses.py:14-19) w.r.t. every
        texture parameter, accumulated into .grad.  Returns (loss [] device tensor — the
        unweighted mean, nr_hits [] int64 device tensor, rgb [N,3])."""
        from .composite import composite_fwd_bwd_l1_raw, count_hits, l1_mean
        self._warmup_scheduler(is_first_iter)
        N = rays_o.shape[0]
        if N > self.max_rays:
            raise _lib.VolsurfsHipError(f"{N} rays > max_rays={self.max_rays}")
        bank = self.bank
        bank._ensure_grads()
        opt = getattr(self, "optimizer", None)
        # the fused Adam step (or zero_grad) has just cleared them — and nothing has touched the buffers since
        zeroed = bool(opt.grads_are_clean()) if hasattr(opt, "grads_are_clean") else bool(getattr(opt, "_grads_clean", False))
        if hasattr(opt, "mark_grads_dirty"):
            opt.mark_grads_dirty()
        bank.frame_generation = getattr(bank, "frame_generation", 0) + 1
        rays_o, rays_d = rays_o.contiguous(), rays_d.contiguous()
        hit_t, hit_slot, hit_uv = self._trace_now(rays_o, rays_d)
        nr_hits = count_hits(hit_slot)               # one launch (the torch expression: compare, cast, fill, reduce)
        tex_uv = bank.mark_and_compact(hit_slot, hit_uv, self.face_uvs)
        bank.evaluate()
        act = torch.empty(self.nr_meshes, N, 4, device=rays_o.device)
        tris = self.raytracer.tris
        rgb_k, alpha_k, _, _ = bank.shade(hit_slot, tex_uv, rays_d, tris, act_out=act)
        loss_scale = float(loss_weight) / (3.0 * N)
        rgb, g_c, g_a = composite_fwd_bwd_l1_raw(rgb_k, alpha_k, self.bg_color, gt_rgb.contiguous(),
                                                 loss_scale)
        from .pipeline import GRAD_CHAIN_GAIN
        scale = self.grad_scale if self.grad_scale is not None else GRAD_CHAIN_GAIN / (3.0 * loss_scale)
        bank.backward(hit_slot, tex_uv, rays_d, tris, g_c, g_a, scale, act, grads_zeroed=zeroed)
        loss = l1_mean(rgb, gt_rgb)                  # the logged value, one launch
        return loss, nr_hits, rgb

    def forward(self, rays_o, rays_d, gt_rgb, gt_mask=None, iter_nr=0, is_first_iter=False,
                is_training_masked=False, ahead=None, return_samples=True):
        """volsurfs.py:763-816: L1 rgb loss (utils/losses.py:14-19).  `ahead`: this batch's
        trace_ahead context (then samples_3d is not gathered: `last_nr_hits` has the count);
        return_samples=False skips the boolean compaction of the hit points (two host syncs) when
        the caller has no use for them."""
        self._warmup_scheduler(is_first_iter)                                   # :774-783
        want_samples = return_samples and (ahead is None or self.using_neural_textures)
        if (VolSurfs.legacy_fused_glue and VolSurfs.legacy_grouped and not self.using_neural_textures and not want_samples
                and self.bg_color is not None and not (is_training_masked and gt_mask is not None)
                and rays_o.shape[0] <= self.max_rays and self._legacy_groupable(rays_o)):
            # the legacy training step with its glue fused (BASELINE configs[2]): traversal (or the look-ahead
            # context), grouped shading, then composite + L1 as ONE node (composite.composite_l1) — the same values as
            # render_rays + (gt - pred).abs().mean(), 55 launches and autograd nodes less per iteration
            from .composite import composite_l1
            rays_o, rays_d = rays_o.contiguous(), rays_d.contiguous()
            if ahead is not None and (ahead.rays_o.data_ptr() != rays_o.data_ptr() or ahead.rays_o.shape != rays_o.shape):
                ahead = None
            if ahead is not None:
                ahead.join()
                hit_t, hit_slot = ahead.hit_t, ahead.hit_slot
            else:
                hit_t, hit_slot, _ = self._trace_now(rays_o, rays_d)
            rgb_k, alpha_k, _ = self._shade_legacy(rays_o, rays_d, hit_t, hit_slot, iter_nr, ahead)
            if ahead is not None:
                self.last_nr_hits = int(sum(ahead.counts()))
            loss_rgb, _ = composite_l1(rgb_k, alpha_k, self.bg_color, gt_rgb)
            return {"loss": loss_rgb, "rgb": loss_rgb}, {}, None
        res = self.render_rays(rays_o=rays_o, rays_d=rays_d, iter_nr=iter_nr, ahead=ahead,
                               return_samples=want_samples)
        pred = res["renders"]["ray_traced"]["rgb"]
        if is_training_masked and gt_mask is not None:
            loss_rgb = ((gt_rgb - pred).abs() * gt_mask).mean()
        else:
            loss_rgb = (gt_rgb - pred).abs().mean()
        return {"loss": loss_rgb, "rgb": loss_rgb}, {}, res["samples_3d"]

    # ---- checkpoints (base_method.py:118-264): <path>/<iter:07d>/models/{key}.pt per model
    # (rgb_i / alpha_i / bg), the optimiser state under its lower-cased class name
    def _model_states(self):
        out = {}
        if self.bank is not None:
            b = self.bank
            # the reference's keys (volsurfs.py:159-165, 200-206): rgb_i / alpha_i, or "rgb" / "alpha" for a model
            # all shells share (its parameters are shell 0's rows of the bank)
            for typ, name, shared in ((0, "rgb", b.shared_rgb), (1, "alpha", b.shared_alpha)):
                for i in range(1 if shared else self.nr_meshes):
                    a = b.tex_index(i, typ, 0)
                    if any(b.tex_channels(a + d) for d in range(4)):
                        out[name if shared else f"{name}_{i}"] = (b.tables[a:a + 4], b.weights[a:a + 4])
        return out

    def save(self, iter_nr):
        """Data-parallel runs: a COLLECTIVE when the optimiser is sharded — every rank calls it
        (`sync_params` all-gathers the fp32 masters, `ShardedFusedAdam.state_dict` the moment
        slices); rank 0 alone writes the files, which hold full tensors and therefore load under
        any world size, sharded or not.  Replicated (all-reduce) runs: rank 0's call suffices."""
        import os
        # every rank reaches the collectives, whatever its own path setting: a rank that returned early
        # here left the others hanging in the all-gathers
        sharded = hasattr(getattr(self, "optimizer", None), "gather_masters")
        root = getattr(self, "save_checkpoints_path", None)
        if root is None and not sharded:
            return None
        self.sync_params()       # parameters and moments are final on the current stream
        opt_state = self.optimizer.state_dict() if getattr(self, "optimizer", None) is not None else None
        if root is None:
            return None
        path = os.path.join(root, format(iter_nr, "07d"), "models")
        if getattr(self, "_dist_rank", 0) != 0:
            return path          # took part in the collectives; the files are rank 0's
        os.makedirs(path, exist_ok=True)
        for key, (t, w) in self._model_states().items():
            torch.save({"tables": t.detach().cpu(), "weights": w.detach().cpu()}, os.path.join(path, f"{key}.pt"))
        for key, model in self.models.items():
            torch.save(model.state_dict(), os.path.join(path, f"{key}.pt"))
        if isinstance(self.bg_model, torch.nn.Module):
            torch.save(self.bg_model.state_dict(), os.path.join(path, "bg.pt"))
        if opt_state is not None:
            # one file name for both optimiser classes (the reference's: base_method.py:246-253)
            torch.save(opt_state, os.path.join(path, "fusedadam.pt"))
        return path

    def load(self, iter_nr):
        """Missing files are skipped, as in the reference (a model absent from the checkpoint
        keeps its initialisation)."""
        import os
        if getattr(self, "load_checkpoints_path", None) is None:
            return None
        path = os.path.join(self.load_checkpoints_path, format(iter_nr, "07d"), "models")
        self.sync_params()       # no optimiser step is still writing what is about to be overwritten
        with torch.no_grad():
            for key, (t, w) in self._model_states().items():
                f = os.path.join(path, f"{key}.pt")
                if os.path.exists(f):
                    st = torch.load(f, map_location=t.device)
                    from .checkpoint import is_reference_state_dict, load_reference_state_dict
                    if is_reference_state_dict(st):      # a checkpoint written by the reference itself
                        name, _, i = key.partition("_")           # "rgb_3", or "rgb" (one model for all shells)
                        load_reference_state_dict(self.bank, int(i or 0), 0 if name == "rgb" else 1, st)
                        continue
                    t.copy_(st["tables"])
                    w.copy_(st["weights"])
        if self.bank is not None:
            self.bank.refresh_half_params()
            self.baked = None                                   # baked textures are stale now
        for key, model in self.models.items():
            f = os.path.join(path, f"{key}.pt")
            if os.path.exists(f):
                model.load_state_dict(torch.load(f, map_location="cuda"))
        f = os.path.join(path, "bg.pt")
        if isinstance(self.bg_model, torch.nn.Module) and os.path.exists(f):
            self.bg_model.load_state_dict(torch.load(f, map_location="cuda"))
        if getattr(self, "optimizer", None) is not None:
            # "fusedadam.pt" since round 4; sharded runs of earlier rounds wrote it under the class name
            names = ["fusedadam.pt", f"{type(self.optimizer).__name__.lower()}.pt"]
            f = next((os.path.join(path, n) for n in names if os.path.exists(os.path.join(path, n))), None)
            if f is not None:
                self.optimizer.load_state_dict(torch.load(f, map_location="cuda"))
            elif os.path.isdir(path):
                # (base_method.py:192 prints the same warning: training resumes with zero moments)
                import warnings
                warnings.warn(f"checkpoint {path} holds no optimiser state ({' / '.join(names)}): resuming with zero moments")
        return path

    @torch.no_grad()
    def render_camera(self, camera, nr_rays_per_pixel=1, jitter_pixels=False, chunk=None):
        """base_method.py:366-541 from the camera down: device ray generation (`ray_gen`,
        :386-402) then the chunked render, reshaped to [H, W, C] images."""
        from .camera import get_camera_rays
        rays_o, rays_d, _ = get_camera_rays(camera, nr_rays_per_pixel, jitter_pixels)
        full = self.render(rays_o, rays_d, nr_rays_per_pixel, chunk)
        return {k: None if v is None else v.reshape(camera.height, camera.width, *v.shape[1:])
                for k, v in full.items()}

    @torch.no_grad()
    def render(self, rays_o, rays_d, nr_rays_per_pixel=1, chunk=None):
        """base_method.py:366-541: chunked full-frame render, supersample mean over
        nr_rays_per_pixel; buffers stay on the device.  chunk: rays per render_rays call — the
        reference's test_rays_batch_size is 16 384; the default here is what this method's buffers
        hold (`max_rays`), so a method built for whole frames renders them in one launch sequence
        (chunking never changes a pixel: test_render_chunks_equal_one_shot)."""
        chunk = int(chunk) if chunk else self.max_rays
        outs = []
        for a in range(0, rays_o.shape[0], chunk):
            r = self.render_rays(rays_o[a:a + chunk], rays_d[a:a + chunk], return_samples=False)
            outs.append(r["renders"]["ray_traced"])
        # keys a configuration does not produce are None (surfs_uvs on the legacy branch)
        # (one chunk: the buffers as they are - `torch.cat` of a single tensor is a copy, and the output
        # dict holds a dozen per-ray buffers: 8 % of a one-chunk frame's time in rocprofv3)
        full = outs[0] if len(outs) == 1 else \
            {k: None if outs[0][k] is None else torch.cat([o[k] for o in outs], 0) for k in outs[0]}
        if nr_rays_per_pixel > 1:
            full = {k: None if v is None else v.reshape(-1, nr_rays_per_pixel, *v.shape[1:]).mean(1)
                    for k, v in full.items()}
        return full
