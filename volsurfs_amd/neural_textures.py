"""Neural-texture bank: parameters, plan and per-frame buffers of the K shells'
SH neural textures, and the host side of the shade stage.

Parameter shapes mirror what the reference instantiates per shell
(/root/reference/volsurfs_py/methods/volsurfs.py:143-206): an rgb
SHNeuralTextures (3 channels) and an alpha one (1 channel), each a ModuleList of
sh_degree+1 NeuralTexture = tcnn HashGrid (16 levels x 2 features, 2^15 entries,
base 16, x1.5; models/neural_texture.py:54-63) + FullyFusedMLP 32->64->64->C
(:65-77).  Here all of them live in two stacked tensors so that one launch
serves every texture:  tables [n_tex, E, 2],  weights [n_tex, 8192]
(W1[64,32] | W2[64,64] | W3[32,64], rows >= C zero), texture index
x = (shell*2 + type)*4 + degree.
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import _lib

MAX_SHELLS, MAX_DEG, MAX_LEVELS = 16, 4, 16
DOM_BLOCK = 4096
WEIGHTS_PER_TEX = 8192
# encode + MLP as one launch (csrc/nt_fused.hip; bit-identical to the two kernels).  Measured on
# MI355X (profiles/r03): forward-only evaluation, where the 730 MB of feature planes are neither
# written nor read, 1.30 -> 1.18 ms per 800x800 frame (+10.5 % Mrays/s); the training step, whose
# backward needs the planes, 0.65 vs 0.61-0.64 ms and 1 043 vs 1 061 it/s — the vector cache's
# look-up rate bounds the gathers (profiles/NOTEBOOK.md A9.1a).  Hence "auto": fused exactly when the feature
# planes are not needed.  VSA_NT_FUSED=1 / 0 force it on / off (A/B switch, tools/README).
# work split of the persistent kernels corrected by the previous frame's measured workgroup times
# (vsa_nt_rebalance; same results); "0" = the fitted cost model alone
REBALANCE = os.environ.get("VSA_NT_REBALANCE", "1") != "0"
_DENSE_COMPACT = os.environ.get("VSA_NT_DENSE_COMPACT", "0") == "1"    # A/B switch: rounds 1-2's fill + dense slot_of
FUSED_FORWARD = {"0": False, "1": True}.get(os.environ.get("VSA_NT_FUSED", "auto"), "auto")

class Plan(ctypes.Structure):
    """Mirror of `vsa_nt_plan` (include/volsurfs_hip.h)."""
    _fields_ = [
        ("nr_shells", ctypes.c_int32), ("rgb_degrees", ctypes.c_int32),
        ("alpha_degrees", ctypes.c_int32), ("inner_solid", ctypes.c_int32),
        ("with_alpha_decay", ctypes.c_int32), ("n_levels", ctypes.c_int32),
        ("tex_res", ctypes.c_int32 * MAX_DEG), ("sh_lo", ctypes.c_float * MAX_DEG),
        ("sh_span", ctypes.c_float * MAX_DEG), ("level_scale", ctypes.c_float * MAX_LEVELS),
        ("level_res", ctypes.c_int32 * MAX_LEVELS), ("level_size", ctypes.c_int32 * MAX_LEVELS),
        ("level_offset", ctypes.c_int32 * (MAX_LEVELS + 1)),
        ("dom_off", ctypes.c_int64 * (MAX_SHELLS * MAX_DEG + 1)),
        ("slot_capacity", ctypes.c_int64),
        ("max_rays", ctypes.c_int32), ("anchor", ctypes.c_int32),
        ("row_base", ctypes.c_int64 * (MAX_SHELLS * MAX_DEG + 1)),
        ("balance", ctypes.c_void_p),
        ("row_format", ctypes.c_int32), ("grads_zeroed", ctypes.c_int32),
        ("shared_rgb", ctypes.c_int32), ("shared_alpha", ctypes.c_int32),
    ]


ROW_QUADS = (2, 4, 8, 8)      # VSA_NT_ROW_QUADS(d): quads (4 elements) per texel / gradient row
ALPHA_QUAD = (1, 3, 4, 6)     # VSA_NT_ALPHA_QUAD(d): first alpha quad of a row


def grid_geometry(n_levels=16, log2_hashmap_size=15, base_resolution=16, per_level_scale=1.5):
    """Level geometry of tiny-cuda-nn's multiresolution grid (published formula;
    neural_texture.py:54-61 fixes the arguments)."""
    log2_pls = np.float32(math.log2(per_level_scale))
    scale, res, size, offset = [], [], [], [0]
    for l in range(n_levels):
        s = np.float32(np.exp2(np.float32(l) * log2_pls)) * np.float32(base_resolution) - np.float32(1.0)
        r = int(np.ceil(s)) + 1
        n = min((r * r + 7) // 8 * 8, 1 << log2_hashmap_size)
        scale.append(float(s))
        res.append(r)
        size.append(n)
        offset.append(offset[-1] + n)
    return scale, res, size, offset


class NeuralTextureBank(torch.nn.Module):
    def __init__(self, nr_shells, max_rays, sh_degree=3, alpha_sh_degree=3,
                 sh_range=(15.0, 15.0, 15.0, 15.0), textures_res=(2048, 1024, 512, 256),
                 inner_solid=False, with_alpha_decay=True, device="cuda", seed=42,
                 training=True, anchor=False, lerp=True, quantize_output=True, squeeze_output=True,
                 grid=None, shared_rgb=False, shared_alpha=False):
        """anchor / lerp / quantize_output / squeeze_output: NeuralTexture's switches
        (models/neural_texture.py:19-52; config keys using_neural_textures_anchor / _lerp,
        using_sh_quantization, using_sh_squeezing).  Built: lerp (the shipped configs) and anchor, each with 8-bit
        quantised texel rows (row_format 0), f16 rows of the un-quantised sigmoid (quantize_output=False: 1) or
        f16 rows of the raw network output (squeeze_output=False: 2; neural_texture.py:157-169, 181-187 skipped).
        shared_rgb / shared_alpha: are_volsurfs_colors_indep = 0 / are_volsurfs_alphas_indep = 0
        (methods/volsurfs.py:159-165, 200-206, 524-527, 553-556): ONE model for all shells — every shell reads
        the parameters of shell 0's textures of that type and the K shells' gradients accumulate there (the
        other shells' rows of `tables` / `weights` are unused and stay zero).  With inner_solid, shared_alpha
        means NO alpha model at all: the reference's loop stores models["alpha"] = None at i = 0 and leaves.
        grid: keyword arguments of grid_geometry() for a hash grid other than the reference's
        (16 levels always: the MLP reads 32 features); levels of more than 2^15 entries are refused
        by the library (one level = one LDS plane)."""
        super().__init__()
        if bool(anchor) == bool(lerp):
            raise ValueError("NeuralTexture is either anchor or lerp (neural_texture.py:47-51, 141-147)")
        if quantize_output and not squeeze_output:
            raise ValueError("quantize_output requires squeeze_output (sh_neural_textures.py:32-36)")
        # 0: 8-bit quantised rows (every shipped config); 1: f16 rows of sigmoid(x), un-quantised; 2: f16 rows of x
        self.row_format = 0 if quantize_output else (1 if squeeze_output else 2)
        self.shared_rgb, self.shared_alpha = bool(shared_rgb), bool(shared_alpha)
        self.anchor = bool(anchor)
        K = nr_shells
        self.K, self.max_rays = K, max_rays
        self.rgb_degrees, self.alpha_degrees = sh_degree + 1, alpha_sh_degree + 1
        self.D = max(self.rgb_degrees, self.alpha_degrees)
        scale, res, size, offset = grid_geometry(**(grid or {}))
        self.n_entries = offset[-1]
        self.n_tex = K * 2 * MAX_DEG
        p = Plan()
        p.nr_shells, p.rgb_degrees, p.alpha_degrees = K, self.rgb_degrees, self.alpha_degrees
        p.inner_solid, p.with_alpha_decay, p.n_levels = int(inner_solid), int(with_alpha_decay), 16
        for d in range(MAX_DEG):
            p.tex_res[d] = int(textures_res[d])
            p.sh_lo[d] = -float(sh_range[d])
            p.sh_span[d] = 2.0 * float(sh_range[d])
        for l in range(16):
            p.level_scale[l], p.level_res[l], p.level_size[l] = scale[l], res[l], size[l]
        for l in range(17):
            p.level_offset[l] = offset[l]
        off, cap, quads = 0, 0, 0
        for s in range(K):
            for d in range(MAX_DEG):
                p.dom_off[s * MAX_DEG + d] = off
                p.row_base[s * MAX_DEG + d] = quads
                if d < self.D:
                    T = (textures_res[d] + 2) ** 2
                    off += (T + DOM_BLOCK - 1) // DOM_BLOCK * DOM_BLOCK
                    cap += min(4 * max_rays, T)
                    quads += (min(4 * max_rays, T) * ROW_QUADS[d] + 7) // 8 * 8
        for i in range(K * MAX_DEG, MAX_SHELLS * MAX_DEG + 1):
            p.dom_off[i] = off
            p.row_base[i] = quads
        self.row_quads_total = quads
        cap = (cap + 255) // 256 * 256 + 256      # feature planes are blocked by 256 slots
        p.slot_capacity = cap
        p.max_rays = max_rays
        p.anchor = int(self.anchor)
        p.row_format = int(self.row_format)
        p.shared_rgb, p.shared_alpha = int(self.shared_rgb), int(self.shared_alpha)
        self.plan, self.dom_total, self.slot_capacity = p, off, cap
        self.tex_res = tuple(int(r) for r in textures_res)

        # ---- parameters (fp32 masters), tcnn-style init: U(-1e-4, 1e-4) tables,
        # Xavier-uniform weights
        g = torch.Generator().manual_seed(seed)
        tables = (torch.rand(self.n_tex, self.n_entries, 2, generator=g) * 2 - 1) * 1e-4
        weights = torch.zeros(self.n_tex, WEIGHTS_PER_TEX)
        for x in range(self.n_tex):
            C = self.tex_channels(x)
            if C and self.param_tex(x) != x:     # (a shell that reads shell 0's shared model owns no parameters)
                tables[x] = 0
                continue
            if C == 0:
                continue

            def xav(o, i):
                s_ = math.sqrt(6.0 / (i + o))
                return (torch.rand(o, i, generator=g) * 2 - 1) * s_
            w3 = torch.zeros(32, 64)
            pad = (C + 15) // 16 * 16
            w3[:C] = xav(pad, 64)[:C]
            weights[x] = torch.cat([xav(64, 32).flatten(), xav(64, 64).flatten(), w3.flatten()])
        self.tables = torch.nn.Parameter(tables.to(device))
        self.weights = torch.nn.Parameter(weights.to(device))
        self._alloc(device, training)

    # -- bookkeeping -------------------------------------------------------
    def tex_channels(self, x):
        deg, typ, shell = x % MAX_DEG, (x // MAX_DEG) & 1, x // (2 * MAX_DEG)
        if typ == 0:
            return 3 * (2 * deg + 1) if deg < self.rgb_degrees else 0
        if self.plan.inner_solid and (shell == 0 or self.shared_alpha):     # nt_shell_has_alpha (csrc/nt_common.h)
            return 0
        return (2 * deg + 1) if deg < self.alpha_degrees else 0

    def param_tex(self, x):
        """The texture whose rows of `tables` / `weights` (and of their gradients) logical texture x uses:
        itself, or shell 0's of the same type and degree when that type's model is shared (nt_param_tex)."""
        typ = (x // MAX_DEG) & 1
        return x % (2 * MAX_DEG) if (self.shared_alpha if typ else self.shared_rgb) else x

    @staticmethod
    def tex_index(shell, typ, deg):
        return (shell * 2 + typ) * MAX_DEG + deg

    def _alloc(self, dev, training):
        i32, u8 = torch.int32, torch.uint8
        cap, K = self.slot_capacity, self.K
        if REBALANCE:
            fn = _lib.lib().vsa_nt_balance_bytes
            fn.restype = ctypes.c_longlong
            self.balance = torch.zeros(int(fn()), dtype=u8, device=dev)
            self.plan.balance = self.balance.data_ptr()
        self.marks = torch.zeros(self.dom_total, dtype=u8, device=dev)
        self.slot_of = torch.full((self.dom_total,), -1, dtype=i32, device=dev)
        self.texel_of_slot = torch.zeros(cap, dtype=i32, device=dev)
        self.slot_xy = torch.zeros(cap, 2, device=dev)
        self.seg_start = torch.zeros(K * MAX_DEG + 1, dtype=i32, device=dev)
        self.block_scratch = torch.zeros(self.dom_total // DOM_BLOCK + 1, dtype=i32, device=dev)
        # blocked layout [type][slot/256][level][slot%256][2]  (nt_common.h: nt_feat_index)
        self.features = torch.empty(2, cap // 256, 16, 256, 2, dtype=torch.float16, device=dev)
        self.tables_h = torch.empty(self.n_tex, self.n_entries, 2, dtype=torch.float16, device=dev)
        self.weights_h = torch.empty(self.n_tex, WEIGHTS_PER_TEX, dtype=torch.float16, device=dev)
        # per-degree row widths (include/volsurfs_hip.h: VSA_NT_ROW_QUADS), 4 elements per quad
        # texel rows: 4 elements per quad — bytes (8-bit quantised) or halves (row_format 1)
        self.texels = torch.zeros(self.row_quads_total * 4, dtype=u8 if self.row_format == 0 else torch.float16,
                                  device=dev)
        self.grad_rows = torch.zeros(self.row_quads_total * 4, dtype=torch.float16, device=dev) if training else None
        self.refresh_half_params()

    @torch.no_grad()
    def refresh_half_params(self):
        """fp16 compute copies of the fp32 masters (tcnn keeps the same pair)."""
        self.wait_params()
        self.tables_h.copy_(self.tables)
        self.weights_h.copy_(self.weights)

    # -- stages --------------------------------------------------------------
    def mark_and_compact(self, hit_slot, hit_uv, face_uvs, want_texel_of_slot=False):
        """hit_slot [K,N] i32, hit_uv [K,N,2], face_uvs [nr_tris,6] (leaf order).
        Returns tex_uv [K,N,2].  slot_of is valid for the touched texels only (vsa_nt_compact_frame);
        texel_of_slot (the inverse map: nothing on the path reads it) is written on request."""
        K, N = hit_slot.shape
        assert K == self.K and N <= self.max_rays
        if getattr(self, "baked", False):
            raise _lib.VolsurfsHipError("this bank holds baked textures: use tex_uv_only + shade")
        st = _lib.stream_ptr()
        tex_uv = torch.empty(K, N, 2, device=hit_slot.device)
        if self.plan.balance and _DENSE_COMPACT:     # (vsa_nt_compact_frame rebalances inside its scan launch)
            _lib.call("vsa_nt_rebalance", ctypes.byref(self.plan), st)
        # Invariant: the marks are zero between frames — allocated so, and vsa_nt_compact_frame clears
        # what it reads.  `_marks_dirty` is set while a mark has not been followed by a compaction that
        # returned OK (an exception between the two, a failed launch, the dense vsa_nt_compact, which
        # does not clear): stale marks would silently inflate every later frame's slot counts.
        if _DENSE_COMPACT or getattr(self, "_marks_dirty", False):
            self.marks.zero_()
        self._marks_dirty = True
        _lib.call("vsa_nt_mark", ctypes.byref(self.plan), hit_slot, hit_uv, face_uvs, N, tex_uv,
                  self.marks, st)
        if _DENSE_COMPACT:
            _lib.call("vsa_nt_compact", ctypes.byref(self.plan), self.marks, self.slot_of,
                      self.texel_of_slot, self.slot_xy, self.seg_start, self.block_scratch, st)
            return tex_uv                  # (marks stay dirty: the next frame zeroes them)
        _lib.call("vsa_nt_compact_frame", ctypes.byref(self.plan), self.marks, self.slot_of,
                  self.texel_of_slot if want_texel_of_slot else None, self.slot_xy, self.seg_start,
                  self.block_scratch, st)
        self._marks_dirty = False
        return tex_uv

    # -- baking (the deploy format, sh_neural_textures.py:99-114 /
    # neural_texture.py:200-251 evaluate every texel once and store it as an 8-bit texture)
    @staticmethod
    def full_capacity_rays(textures_res):
        """max_rays for which every (shell, degree) segment can hold ALL its texels."""
        return ((max(int(r) for r in textures_res) + 2) ** 2 + 3) // 4

    @torch.no_grad()
    def bake_all(self):
        """Evaluate EVERY texel of every texture (instead of the texels a frame touches): after
        this, `shade` works for any ray without mark/compact/encode/mlp.  Needs a bank built
        with max_rays >= full_capacity_rays(textures_res)."""
        if self.row_format != 0:
            raise _lib.VolsurfsHipError("baked textures are the 8-bit deploy format: using_sh_quantization=1 only")
        self.marks.zero_()
        for s in range(self.K):
            for d in range(self.D):
                R = self.tex_res[d]
                W = R + 2
                if min(4 * self.max_rays, W * W) < W * W:
                    raise _lib.VolsurfsHipError("bake_all needs a full-capacity bank "
                                                "(max_rays >= NeuralTextureBank.full_capacity_rays)")
                off = int(self.plan.dom_off[s * MAX_DEG + d])
                self.marks[off:off + W * W].view(W, W).fill_(1)   # interior + the one-texel apron
        _lib.call("vsa_nt_compact", ctypes.byref(self.plan), self.marks, self.slot_of,
                  self.texel_of_slot, self.slot_xy, self.seg_start, self.block_scratch,
                  _lib.stream_ptr())
        self.marks.zero_()
        self.evaluate(need_features=False)
        self.baked = True
        return self

    @torch.no_grad()
    def baked_textures(self):
        """{(shell, type, degree): uint8 [R, R, C]} indexed [iy, ix] in the network's texel
        coordinates (texel centre = ((ix + 0.5) / R, (iy + 0.5) / R), see nt_footprint: the
        reference's 90-degree-rotated uv, neural_texture.py:114-121); C = channels x (2d+1)."""
        dense = self.rows_dense(self.texels)
        out = {}
        for s in range(self.K):
            for d in range(self.D):
                R = self.tex_res[d]
                W, n = R + 2, 2 * d + 1
                off = int(self.plan.dom_off[s * MAX_DEG + d])
                slots = self.slot_of[off:off + W * W].view(W, W)[1:R + 1, 1:R + 1].long()
                rows = dense[slots]                                  # [R, R, 32]
                if self.tex_channels(self.tex_index(s, 0, d)):
                    out[(s, 0, d)] = rows[..., :3 * n].contiguous()
                if self.tex_channels(self.tex_index(s, 1, d)):
                    out[(s, 1, d)] = rows[..., 24:24 + n].contiguous()
        return out

    def tex_uv_only(self, hit_slot, hit_uv, face_uvs):
        """Per-hit texture uv without touching the compaction (inference from a baked bank)."""
        K, N = hit_slot.shape
        tex_uv = torch.empty(K, N, 2, device=hit_slot.device)
        _lib.call("vsa_nt_mark", ctypes.byref(self.plan), hit_slot, hit_uv, face_uvs, N, tex_uv,
                  None, _lib.stream_ptr())
        return tex_uv

    def wait_params(self):
        """An optimiser step may still be running on a side stream (VolSurfs.optim_step(overlap=True)):
        the first reader of the parameters on the current stream waits for it here."""
        ev = getattr(self, "_params_event", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._params_event = None

    def encode(self):
        self.wait_params()
        _lib.call("vsa_nt_encode_fwd", ctypes.byref(self.plan), self.tables_h, self.slot_xy,
                  self.seg_start, self.features, _lib.stream_ptr())
        return self.features

    def features_level_major(self):
        """[type, level, slot, 2] view-copy of the blocked feature planes (tests)."""
        f = self.features.permute(0, 2, 1, 3, 4)            # [type, level, block, 256, 2]
        return f.reshape(2, 16, self.slot_capacity, 2)

    def mlp(self, want_pre=False):
        self.wait_params()
        pre = None
        if want_pre:
            pre = torch.zeros(self.slot_capacity, 32, dtype=torch.float16, device=self.texels.device)
        _lib.call("vsa_nt_mlp_fwd", ctypes.byref(self.plan), self.weights_h, self.features,
                  self.seg_start, self.texels, pre, _lib.stream_ptr())
        return (self.rows_dense(self.texels), pre) if want_pre else self.texels

    def encode_mlp(self, write_features=True, want_pre=False):
        """encode() + mlp() as ONE launch (csrc/nt_fused.hip; bit-identical results).
        write_features=False: inference — the feature planes are neither written nor read."""
        self.wait_params()
        pre = None
        if want_pre:
            pre = torch.zeros(self.slot_capacity, 32, dtype=torch.float16, device=self.texels.device)
        _lib.call("vsa_nt_encode_mlp_fwd", ctypes.byref(self.plan), self.tables_h, self.weights_h,
                  self.slot_xy, self.seg_start, self.features if write_features else None,
                  self.texels, pre, _lib.stream_ptr())
        return (self.rows_dense(self.texels), pre) if want_pre else self.texels

    def evaluate(self, need_features=True):
        """Texel rows of the compacted slots.  need_features=False (inference, baking): the fused
        launch; True (a backward pass follows): the level-major encode kernel, then the MLP kernel."""
        if self.row_format == 0 and (FUSED_FORWARD is True or (FUSED_FORWARD == "auto" and not need_features)):
            return self.encode_mlp(write_features=need_features)
        self.encode()
        return self.mlp()

    def rows_dense(self, buf):
        """Tests / export: the per-degree rows of `texels` or `grad_rows` re-laid as
        [slot_capacity, 32] with rgb coefficient c at column c and alpha coefficient c at 24+c."""
        seg = self.seg_start.cpu().tolist()
        dense = torch.zeros(self.slot_capacity, 32, dtype=buf.dtype, device=buf.device)
        for s in range(self.K):
            for d in range(self.D):
                sd, n, q = s * MAX_DEG + d, 2 * d + 1, ROW_QUADS[d]
                a, b = seg[sd], seg[sd + 1]
                r0 = int(self.plan.row_base[sd]) * 4
                rows = buf[r0:r0 + (b - a) * q * 4].view(b - a, q * 4)
                dense[a:b, :3 * n] = rows[:, :3 * n]
                dense[a:b, 24:24 + n] = rows[:, 4 * ALPHA_QUAD[d]:4 * ALPHA_QUAD[d] + n]
        return dense

    def shade(self, hit_slot, tex_uv, rays_d, tris, want_coeffs=False, want_normals=False,
              act_out=None):
        """act_out: optional [K,N,4] f32 buffer that receives the per-hit output sigmoids;
        pass it to backward_shade(act=...) of the same frame to skip the re-gather."""
        K, N = hit_slot.shape
        dev = hit_slot.device
        rgb = torch.empty(N, K, 3, device=dev)
        alpha = torch.empty(N, K, device=dev)
        normals = torch.empty(N, K, 3, device=dev) if want_normals else None
        coeffs = torch.empty(K, N, 64, device=dev) if want_coeffs else None
        _lib.call("vsa_nt_shade_fwd", ctypes.byref(self.plan), hit_slot, tex_uv, rays_d, tris,
                  self.slot_of, self.seg_start, self.texels, N, rgb, alpha, normals, coeffs,
                  act_out, _lib.stream_ptr())
        return rgb, alpha, normals, coeffs

    def backward(self, hit_slot, tex_uv, rays_d, tris, g_surfs_rgb, g_surfs_alpha, grad_scale,
                 act=None, grads_zeroed=None):
        """Back-propagates d loss / d surfs_rgb [N,K,3], d surfs_alpha [N,K] to
        self.tables.grad / self.weights.grad (accumulating, like autograd).
        grad_scale keeps the fp16 intermediate gradients in range (the analogue of
        tiny-cuda-nn's loss scale); it is divided out before accumulation.
        grads_zeroed: see backward_encode."""
        self.backward_shade(hit_slot, tex_uv, rays_d, tris, g_surfs_rgb, g_surfs_alpha, grad_scale,
                            act)
        self.backward_mlp(grad_scale)
        self.backward_encode(grad_scale, grads_zeroed=grads_zeroed)

    def _ensure_grads(self):
        if self.tables.grad is None:
            self.tables.grad = torch.zeros_like(self.tables)
        if self.weights.grad is None:
            self.weights.grad = torch.zeros_like(self.weights)

    def zero_grads(self):
        """tables.grad, weights.grad and the per-texture sum |dF| of the MLP backward live in ONE
        allocation so that a training step clears them with one fill (three small launches
        otherwise).  Returns False if the gradient tensors are not (any more) those views —
        e.g. an optimiser's zero_grad(set_to_none=True) dropped them — and clears what exists."""
        flat = getattr(self, "_grad_flat", None)
        nt, nw = self.tables.numel(), self.weights.numel()
        if flat is None:
            flat = torch.zeros(nt + nw + self.n_tex * 32, device=self.tables.device)
            self._grad_flat = flat
            self._flat_t, self._flat_w = flat[:nt].view_as(self.tables), flat[nt:nt + nw].view_as(self.weights)
            self._dfsum = flat[nt + nw:].view(self.n_tex, 32)
        if self.tables.grad is None and self.weights.grad is None:
            self.tables.grad, self.weights.grad = self._flat_t, self._flat_w
        self._tables_grad_zero = True        # (backward_encode: a sole writer's table plane is stored, not added)
        if self.tables.grad is self._flat_t and self.weights.grad is self._flat_w:
            flat.zero_()
            self._dfsum_clean = True
            self._tables_grad_sig = (id(self.tables.grad), self.tables.grad._version)
            return True
        self.tables.grad.zero_()
        self.weights.grad.zero_()
        self._tables_grad_sig = (id(self.tables.grad), self.tables.grad._version)
        return False

    def _take_grads_zeroed(self, grads_zeroed):
        """plan.grads_zeroed of the next hash-grid backward launch: the caller's word (it knows its optimiser
        has just cleared the gradients), else what zero_grads() left; either way the buffer holds something
        after the launch."""
        z = bool(getattr(self, "_tables_grad_zero", False) if grads_zeroed is None else grads_zeroed)
        if grads_zeroed is None and z:       # zero_grads() said so: still the same buffer, untouched by torch since?
            g = self.tables.grad
            z = g is not None and getattr(self, "_tables_grad_sig", None) == (id(g), g._version)
        # (shared models: the kernel itself keeps a shared plane's flushes atomic — K shells write it)
        self._tables_grad_zero = False
        self.plan.grads_zeroed = int(z)

    def backward_shade(self, hit_slot, tex_uv, rays_d, tris, g_surfs_rgb, g_surfs_alpha, grad_scale,
                       act=None):
        self._ensure_grads()
        if act is not None:
            _lib.check_f32(act, hit_slot.shape[0], hit_slot.shape[1], 4)
        _lib.call("vsa_nt_shade_bwd", ctypes.byref(self.plan), hit_slot, tex_uv, rays_d, tris,
                  self.slot_of, self.seg_start, self.texels, hit_slot.shape[1],
                  g_surfs_rgb.contiguous(),
                  g_surfs_alpha.contiguous(), float(grad_scale), self.grad_rows, act,
                  _lib.stream_ptr())

    def backward_mlp(self, grad_scale):
        if getattr(self, "_dfsum", None) is None:
            self._dfsum = torch.zeros(self.n_tex, 32, device=self.weights.device)
        elif not getattr(self, "_dfsum_clean", False):
            self._dfsum.zero_()
        self._dfsum_clean = False            # (zero_grads() has just cleared it together with the gradients)
        _lib.call("vsa_nt_mlp_bwd", ctypes.byref(self.plan), self.weights_h, self.features,
                  self.seg_start, self.grad_rows, self.weights.grad, self._dfsum,
                  1.0 / float(grad_scale), _lib.stream_ptr())

    def backward_encode(self, grad_scale, shells=None, grads_zeroed=None):
        """shells=(begin, end) restricts the launch to those shells' textures (their table
        gradients are the contiguous slice tables.grad[begin*8:end*8]).  grads_zeroed=True: the caller
        vouches that tables.grad is all zero (vsa_nt_plan.grads_zeroed); None: true right after zero_grads()."""
        self._take_grads_zeroed(grads_zeroed)
        if shells is None:
            _lib.call("vsa_nt_encode_bwd", ctypes.byref(self.plan), self.features, self._dfsum,
                      float(grad_scale), self.slot_xy, self.seg_start, self.tables.grad,
                      _lib.stream_ptr())
        else:
            _lib.call("vsa_nt_encode_bwd_range", ctypes.byref(self.plan), self.features,
                      self._dfsum, float(grad_scale), self.slot_xy, self.seg_start,
                      self.tables.grad, int(shells[0]), int(shells[1]), _lib.stream_ptr())

    def backward_encode_phased(self, grad_scale, signals, grads_zeroed=None):
        """The hash-grid backward as ONE launch that finishes the shells phase by phase and publishes
        each phase's completion in signals.flags (parallel.StepSignals; vsa_nt_encode_bwd_phased)."""
        self._take_grads_zeroed(grads_zeroed)
        _lib.call("vsa_nt_encode_bwd_phased", ctypes.byref(self.plan), self.features, self._dfsum,
                  float(grad_scale), self.slot_xy, self.seg_start, self.tables.grad, signals.n,
                  signals.phase_end_c, signals._flags, signals.counters, signals.epoch, int(signals.reserve_cus),
                  _lib.stream_ptr())


def stage_accounting(bank, nr_rays, nr_hits, tracer_bytes=0):
    """ALGORITHMIC bytes / FLOPs of every neural-texture stage for the frame the bank last compacted
    (DESIGN.md §5): each operand counted once per stage at its stored width, gathers per hit.  Host-side
    read-back of the slot counts: outside any timed region.  Returns (bytes per stage, unpadded forward FLOPs
    of the MLP over the unique texels, slots)."""
    N, K, M = nr_rays, bank.K, nr_hits
    P = int(bank.seg_start[K * 4].item())
    seg = bank.seg_start.cpu().tolist()
    fl, row_quads, slot_models = 0, 0, 0
    for s_ in range(K):
        for d in range(4):
            P_sd = seg[s_ * 4 + d + 1] - seg[s_ * 4 + d]
            row_quads += P_sd * ROW_QUADS[d]
            for typ in range(2):
                C = bank.tex_channels(bank.tex_index(s_, typ, d))
                if C:
                    fl += P_sd * 2 * (32 * 64 + 64 * 64 + 64 * C)
                    slot_models += P_sd
    ntex = sum(1 for x in range(bank.n_tex) if bank.tex_channels(x))
    rows_u8, rows_f16 = row_quads * 4, row_quads * 8      # texel rows u8, gradient rows f16
    feats = slot_models * 64                      # 16 levels x f16x2 per (slot, model)
    gathers = M * (16 * 4 + 4 * sum(ROW_QUADS) * 4 + 8) + N * 12   # slot ids + 16 texel rows + uv, dirs
    acct = {
        "trace": N * (24 + 16 * K) + tracer_bytes,
        # marks read twice (count, assign); per slot: mark set + cleared, slot_of, slot_xy
        # (vsa_nt_compact_frame: nothing is written for untouched texels)
        "nt_mark_compact": N * K * 20 + bank.dom_total * (1 + 1) + P * (1 + 1 + 4 + 8),
        "nt_encode_fwd": P * 8 + feats + ntex * bank.n_entries * 4,
        "nt_mlp_fwd": feats + rows_u8 + ntex * 8192 * 2,
        # fused: texel centre in (per texture), feature planes + texel rows out, parameters once
        "nt_encode_mlp_fwd": slot_models * 8 + feats + rows_u8 + ntex * (bank.n_entries * 4 + 8192 * 2),
        "nt_shade_fwd": gathers + N * K * 16,
        "nt_shade_bwd": gathers + N * K * 16 + rows_f16,
        "nt_mlp_bwd": 2 * feats + 2 * rows_f16 + ntex * 8192 * (2 + 4),
        "nt_encode_bwd": feats + P * 8 + ntex * bank.n_entries * 8,
    }
    return acct, fl, P
