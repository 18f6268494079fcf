"""Encoders of the legacy appearance branch and the background field (SURVEY §8a rows A5,
A10), mirroring /root/reference/volsurfs_py/encodings/*.py and utils/encoder.py:8-48:
same class names, constructor arguments, call signatures and `output_dim`.

* GridHashEncoder (encodings/gridhash.py:12-92): the reference wraps tcnn.Encoding
  ("Grid"/"Hash", fp32); here the grid is `vsa_grid_encode_fwd/bwd` (csrc/grid_encode.hip).
* SHEncoder (encodings/sphericalharmonics.py:36-229): `__call__` = `vsa_sh_encode`.
* FrequencyEncoder / IdentityEncoder: elementwise torch ops, as in the reference.
* PermutoHashEncoder (encodings/permutohash.py:10-99): the reference wraps the un-vendored
  `permutohedral_encoding` fork (parity unpinned); `PermutoEncoding` here is the
  published permutohedral-lattice encoding on `vsa_permuto_encode_fwd/bwd`
  (csrc/permuto_encode.hip), restated in oracle/permuto.py.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib

GRID_MAX_LEVELS = 32
SLICED_BWD_MIN_POINTS = 1 << 18     # below this the plain atomic scatter is cheaper (fixed scan cost)
BINNED_BWD_MIN_POINTS = 1 << 20     # above this: bin once + dense accumulation (4.8 GB of records at 2.1 M points)


class GridPlan(ctypes.Structure):
    """Mirror of `vsa_grid_plan` (include/volsurfs_hip.h)."""
    _fields_ = [
        ("n_dims", ctypes.c_int32), ("n_levels", ctypes.c_int32),
        ("n_features", ctypes.c_int32), ("reserved0", ctypes.c_int32),
        ("level_scale", ctypes.c_float * GRID_MAX_LEVELS),
        ("level_res", ctypes.c_int32 * GRID_MAX_LEVELS),
        ("level_size", ctypes.c_int32 * GRID_MAX_LEVELS),
        ("level_offset", ctypes.c_int32 * (GRID_MAX_LEVELS + 1)),
    ]


def grid_plan(n_dims, n_levels, log2_hashmap_size, base_resolution, per_level_scale):
    """Level geometry of tiny-cuda-nn's GridEncoding (published formula)."""
    p = GridPlan()
    p.n_dims, p.n_levels, p.n_features = n_dims, n_levels, 2
    log2_pls = np.float32(math.log2(per_level_scale))
    off = 0
    for l in range(n_levels):
        s = np.float32(np.exp2(np.float32(l) * log2_pls)) * np.float32(base_resolution) - np.float32(1.0)
        r = int(np.ceil(s)) + 1
        n = min((min(r ** n_dims, 1 << 62) + 7) // 8 * 8, 1 << log2_hashmap_size)
        p.level_scale[l], p.level_res[l], p.level_size[l], p.level_offset[l] = float(s), r, n, off
        off += n
    p.level_offset[n_levels] = off
    return p, off


class _GridEncode(torch.autograd.Function):
    """append=True: the output rows are [2 L features | x] (GridHashEncoder's concat_points,
    gridhash.py:88-90) written by the encode kernel itself, and the backward reads the gradient of
    that wider matrix in place (`vsa_grid_encode_*_ld`): no torch.cat forward, no slice copy back."""

    @staticmethod
    def forward(ctx, tables, x, plan, owner=None, append=False):
        x = _lib.check_f32(x.contiguous(), x.shape[0], plan.n_dims)
        ctx.owner = owner        # the nn.Parameter behind `tables` (direct gradient accumulation)
        width = plan.n_levels * 2 + (plan.n_dims if append else 0)
        # rows padded to a multiple of 4 floats (51 -> 52): the MLP that reads them — and writes their gradient —
        # moves 16-byte groups (models.padded_rows; csrc/mlp_f32_fused.h), and an even stride keeps the float2 stores aligned
        ld = (width + 3) // 4 * 4
        buf = torch.empty(x.shape[0], ld, device=x.device)
        out = buf[:, :width] if ld != width else buf
        _lib.call("vsa_grid_encode_fwd_ld", ctypes.byref(plan), tables, x, x.shape[0], out, ld,
                  1 if append else 0, _lib.stream_ptr())
        ctx.save_for_backward(x)
        ctx.plan, ctx.shape, ctx.width = plan, tables.shape, width
        return out

    @staticmethod
    def backward(ctx, g_out):
        (x,) = ctx.saved_tensors
        from .optim import accumulate_into_grad
        direct = accumulate_into_grad(ctx.owner) if ctx.owner is not None else None
        g_tables = direct if direct is not None else torch.zeros(ctx.shape, device=x.device)
        if not (g_out.dim() == 2 and g_out.stride(1) == 1 and g_out.stride(0) >= ctx.width):
            g_out = g_out.contiguous()
        g_ld = g_out.stride(0)          # (padded rows are read in place)
        # (the binned path slices a level into <= 32 runs of 2^13 entries: tables up to 2^18 entries)
        if x.shape[0] >= BINNED_BWD_MIN_POINTS and max(ctx.plan.level_size[:ctx.plan.n_levels]) <= (1 << 18):
            # very large batches: bin the contributions by table slice once, accumulate densely
            n = ctypes.c_longlong()
            _lib.call("vsa_grid_encode_bwd_binned_workspace", ctypes.byref(ctx.plan), x.shape[0],
                      ctypes.byref(n))
            ws = torch.empty(n.value, device=x.device)
            _lib.call("vsa_grid_encode_bwd_binned_ld", ctypes.byref(ctx.plan), x, g_out, g_ld,
                      x.shape[0], g_tables, ws, _lib.stream_ptr())
        elif x.shape[0] >= SLICED_BWD_MIN_POINTS:
            # large batches: LDS-resident table slices instead of memory-side float atomics
            ws = torch.empty(x.shape[0] * ctx.plan.n_levels * 2 + 32, device=x.device)
            _lib.call("vsa_grid_encode_bwd_sliced_ld", ctypes.byref(ctx.plan), x, g_out, g_ld,
                      x.shape[0], g_tables, ws, _lib.stream_ptr())
        else:
            _lib.call("vsa_grid_encode_bwd_ld", ctypes.byref(ctx.plan), x, g_out, g_ld, x.shape[0],
                      g_tables, _lib.stream_ptr())
        # positions carry no gradient on this path
        return (None if direct is not None else g_tables), None, None, None, None


class HashGrid(torch.nn.Module):
    """tcnn.Encoding(input_dim, {"otype": "Grid", "type": "Hash", ...}, dtype=float32)
    shaped: callable on [B, D] in [0,1], `.n_output_dims`, fp32 parameters U(-1e-4, 1e-4)."""

    def __init__(self, n_input_dims, config, seed=1337, device="cuda"):
        super().__init__()
        self.plan, n_entries = grid_plan(n_input_dims, config["n_levels"],
                                         config["log2_hashmap_size"], config["base_resolution"],
                                         config["per_level_scale"])
        assert config.get("n_features_per_level", 2) == 2
        g = torch.Generator().manual_seed(seed)
        self.params = torch.nn.Parameter(
            ((torch.rand(n_entries, 2, generator=g) * 2 - 1) * 1e-4).to(device))
        self.n_output_dims = 2 * config["n_levels"]

    def forward(self, x, append_points=False):
        return _GridEncode.apply(self.params, x.float(), self.plan, self.params, append_points)


def _inv_half_sides(enc):
    """1 / (bb_sides / 2) of an encoder, formed once (two launches per call otherwise — the training
    loop of the per-shell models is bound by the host's op dispatch); same tensor ops, same values."""
    bb = enc.bb_sides
    key = (bb.data_ptr(), bb._version)
    memo = getattr(enc, "_inv_half_memo", None)
    if memo is None or memo[0] != key:
        memo = (key, 1 / (bb / 2))
        enc._inv_half_memo = memo
    return memo[1]


class Encoder(torch.nn.Module):
    # hash encoders return (features, points_out_of_bounds) like the reference's; the models that own
    # them (RGB / ColorSH / NerfHash, here as there) use only the features and switch the mask off
    compute_out_of_bounds = True

    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim


class IdentityEncoder(Encoder):
    def __init__(self, input_dim=3, **kwargs):
        super().__init__(input_dim, input_dim)

    def forward(self, x, **kwargs):
        return x


class FrequencyEncoder(Encoder):
    """encodings/frequency.py:10-59."""

    def __init__(self, input_dim=3, multires=6, include_input=True, **kwargs):
        super().__init__(input_dim, input_dim * multires * 2 + (input_dim if include_input else 0))
        self.multires, self.include_input = multires, include_input

    def forward(self, x, **kwargs):
        parts = [x] if self.include_input else []
        for l in range(self.multires):
            freq = 2.0 ** l
            parts += [torch.sin(x * freq), torch.cos(x * freq)]
        return torch.cat(parts, -1)


class SHEncoder(Encoder):
    """encodings/sphericalharmonics.py:36-153 (`__call__`: SH basis of unit directions)."""

    def __init__(self, input_dim=3, degree=3):
        assert input_dim == 3, "SH encoding only supports 3D inputs"
        assert 0 <= degree <= 4, "SH degree must be 0-4"
        super().__init__(input_dim, (degree + 1) ** 2)
        self.degree = degree

    def forward(self, dirs, **kwargs):
        dirs = _lib.check_f32(dirs.contiguous(), dirs.shape[0], 3)
        out = torch.empty(dirs.shape[0], self.output_dim, device=dirs.device)
        _lib.call("vsa_sh_encode", dirs, dirs.shape[0], self.degree, out, _lib.stream_ptr())
        return out


class Coarse2Fine:
    """permutohedral_encoding.Coarse2Fine(nr_levels)(t) -> window [nr_levels]: the package is
    absent; restated from its published form (a nerfies-style cosine ramp,
    alpha = t * nr_levels, w_i = (1 - cos(pi * clamp(alpha - i, 0, 1))) / 2).  t = 1 (the
    evaluation path and `iter_nr=None`) gives all ones exactly.  PARITY UNPINNED for t < 1."""

    def __init__(self, nr_levels):
        self.nr_levels = nr_levels
        self._memo = (None, None)

    def __call__(self, t):
        t = float(t)
        if self._memo[0] != t:           # (t is constant for whole phases of training: 1.0 once c2f is over)
            alpha = t * self.nr_levels
            i = torch.arange(self.nr_levels, dtype=torch.float32)
            w = 0.5 * (1.0 - torch.cos(math.pi * torch.clamp(alpha - i, 0.0, 1.0)))
            self._memo = (t, w)
            self.all_open = bool((w == 1.0).all())
        return self._memo[1]

    all_open = False                     # the last window returned is all ones (a no-op)


def map_range_val(input_val, input_start, input_end, output_start, output_end):
    """utils/common.py:94-100: clamped linear remap."""
    input_clamped = max(input_start, min(input_end, input_val))
    if input_start >= input_end:
        return output_end
    return output_start + ((output_end - output_start) / (input_end - input_start)) * (
        input_clamped - input_start)


class GridHashEncoder(Encoder):
    """encodings/gridhash.py:12-92: 3-D hash grid (24 levels, 2^18, base 16, growth 2) with the
    coarse-to-fine window, the bounding-box normalisation and the concatenated points."""

    def __init__(self, input_dim=3, nr_levels=24, log2_hashmap_size=18, nr_feat_per_level=2,
                 base_resolution=16, growth_factor=2, nr_iters_for_c2f=0, concat_points=True,
                 bb_sides=2.0, device="cuda"):
        self.config = {
            "otype": "Grid", "type": "Hash", "n_levels": nr_levels,
            "n_features_per_level": nr_feat_per_level, "log2_hashmap_size": log2_hashmap_size,
            "base_resolution": base_resolution, "per_level_scale": growth_factor,
            "interpolation": "Linear",
        }
        encoder = HashGrid(input_dim, self.config, device=device)
        super().__init__(input_dim, encoder.n_output_dims + input_dim)
        self.encoder = encoder
        self.concat_points = concat_points
        self.bb_sides = bb_sides
        if self.bb_sides is not None:
            if isinstance(self.bb_sides, float):
                self.bb_sides = np.array([self.bb_sides] * input_dim)
            if isinstance(self.bb_sides, np.ndarray):
                self.bb_sides = torch.tensor(self.bb_sides, dtype=torch.float32)
            self.bb_sides = self.bb_sides.to(device)
        self.c2f = Coarse2Fine(nr_levels)
        self.nr_iters_for_c2f = nr_iters_for_c2f

    def forward(self, points, iter_nr=None, **kwargs):
        if iter_nr is None or iter_nr < 0:
            t = 1.0
        else:
            t = map_range_val(iter_nr, 0.0, self.nr_iters_for_c2f, 0.3, 1.0)
        window = self.c2f(t)
        all_open = self.c2f.all_open                  # t = 1 (evaluation, c2f off): the window is a no-op
        if not all_open:                              # (no host-to-device copy otherwise)
            window = window.to(points.device).repeat_interleave(self.config["n_features_per_level"])
        out_of_bounds = None
        if self.bb_sides is not None:
            if self.compute_out_of_bounds:
                out_of_bounds = torch.logical_or((points <= -self.bb_sides / 2).any(dim=1),
                                                 (points >= self.bb_sides / 2).any(dim=1))
            points = points * _inv_half_sides(self)
            points = (points + 1) / 2
        if all_open and self.concat_points and GridHashEncoder.fused_concat and points.is_cuda \
                and not points.requires_grad:
            # the encode kernel writes the points behind the features (and its backward reads the
            # wider gradient in place): no cat, no slice copy
            return self.encoder(points, append_points=True), out_of_bounds
        enc = self.encoder(points)
        if not all_open:                              # (x * 1 is exact: skipping it changes no bit, and
            enc = enc * window                        #  saves two passes over [samples, 48] per step)
        if self.concat_points:
            enc = torch.cat([enc, points], dim=1)
        return enc, out_of_bounds

    fused_concat = True       # class-wide switch: False = torch.cat (tests compare the two)

    def reset(self):
        pass


class PermutoPlan(ctypes.Structure):
    """Mirror of `vsa_permuto_plan` (include/volsurfs_hip.h)."""
    _fields_ = [
        ("pos_dim", ctypes.c_int32), ("n_levels", ctypes.c_int32),
        ("n_features", ctypes.c_int32), ("capacity", ctypes.c_int32),
        ("scale_factor", (ctypes.c_float * 4) * GRID_MAX_LEVELS),
        ("random_shift", (ctypes.c_float * 4) * GRID_MAX_LEVELS),
    ]


class _PermutoEncode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, values, x, window, enc, extra):
        """values [L, capacity, 2]; x [B, D] in [0,1]; window [L] or None.  Returns
        [B, 2L + extra]: the encoding in the first 2L columns of a row that the caller fills up
        (concatenated points) — written in place by the kernel, no torch.cat copy."""
        plan = enc.plan
        x = _lib.check_f32(x.contiguous(), x.shape[0], plan.pos_dim)
        width = 2 * plan.n_levels + extra
        stride = width + (width & 1)                  # float2 stores: even row stride
        buf = torch.empty(x.shape[0], stride, device=x.device)
        _lib.call("vsa_permuto_encode_fwd", ctypes.byref(plan), values, x, window, x.shape[0], buf,
                  stride, _lib.stream_ptr())
        if extra:                       # concatenated points (no gradient path: hit points)
            buf[:, 2 * plan.n_levels:width] = x * enc.concat_points_scaling
        ctx.save_for_backward(x, window)
        ctx.plan, ctx.shape, ctx.owner = plan, values.shape, enc.lattice_values
        return buf[:, :width]

    @staticmethod
    def backward(ctx, g_out):
        x, window = ctx.saved_tensors
        from .optim import accumulate_into_grad
        direct = accumulate_into_grad(ctx.owner)
        g_values = direct if direct is not None else torch.zeros(ctx.shape, device=x.device)
        g_out = g_out.contiguous()
        _lib.call("vsa_permuto_encode_bwd", ctypes.byref(ctx.plan), x, window, g_out,
                  g_out.shape[1], x.shape[0], g_values, _lib.stream_ptr())
        return (None if direct is not None else g_values), None, None, None, None   # positions: no gradient here


_PERMUTO_MAX_GROUPS = 8
_plans_dev_cache = {}


def _plans_on_device(encs, device):
    """The encoders' plans as one device array (vsa_permuto_encode_*_grouped read the per-group
    shifts from there: eight 1 KiB plans do not fit the kernel arguments).  Cached on the identity of
    the plan objects, which an encoder replaces when its shift buffer changes."""
    plans = [e.plan for e in encs]
    key = (tuple(id(p) for p in plans), str(device))
    hit = _plans_dev_cache.get(key)
    if hit is None:
        raw = b"".join(bytes(p) for p in plans)
        dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        if len(_plans_dev_cache) > 64:
            _plans_dev_cache.clear()
        hit = _plans_dev_cache[key] = (dev, plans)       # (keeps the plan objects alive: ids stay unique)
    return hit[0]


class _PermutoEncodeGrouped(torch.autograd.Function):
    """G encodings of one geometry (the per-shell models' position encoders) on G consecutive row
    segments of x as ONE autograd node, one output matrix and one launch each way for up to 8 groups
    (`vsa_permuto_encode_fwd_grouped` / `_bwd_grouped`: group = blockIdx.z) instead of G x (apply,
    output allocation, point columns, slice and cat bookkeeping, launches)."""

    @staticmethod
    def _runs(sizes):
        out, a = [], 0
        for g0 in range(0, len(sizes), _PERMUTO_MAX_GROUPS):
            n = sum(sizes[g0:g0 + _PERMUTO_MAX_GROUPS])
            out.append((g0, min(_PERMUTO_MAX_GROUPS, len(sizes) - g0), a, n))
            a += n
        return out

    @staticmethod
    def forward(ctx, x, sizes, window, extra, encs, *values):
        plan0 = encs[0].plan
        x = _lib.check_f32(x.contiguous(), x.shape[0], plan0.pos_dim)
        width = 2 * plan0.n_levels + extra
        stride = width + (width & 1)                  # float2 stores: even row stride
        buf = torch.empty(x.shape[0], stride, device=x.device)
        for g0, ng, a, n in _PermutoEncodeGrouped._runs(sizes):
            if n == 0:
                continue
            run = encs[g0:g0 + ng]
            ptrs = (ctypes.c_void_p * ng)(*[v.data_ptr() for v in values[g0:g0 + ng]])
            cnt = (ctypes.c_int * ng)(*sizes[g0:g0 + ng])
            _lib.call("vsa_permuto_encode_fwd_grouped", ctypes.byref(run[0].plan), _plans_on_device(run, x.device),
                      ptrs, ng, cnt, x[a:a + n], window, buf[a:a + n], stride, _lib.stream_ptr())
        if extra:                       # concatenated points (no gradient path: hit points)
            buf[:, 2 * plan0.n_levels:width] = x * encs[0].concat_points_scaling
        ctx.save_for_backward(x, window)
        ctx.encs, ctx.sizes = encs, sizes
        return buf[:, :width]

    @staticmethod
    def backward(ctx, g_out):
        x, window = ctx.saved_tensors
        from .optim import accumulate_into_grad
        if not (g_out.dim() == 2 and g_out.stride(1) == 1 and g_out.stride(0) >= g_out.shape[1]):
            g_out = g_out.contiguous()          # (a column slice of wider rows is read in place: the kernel takes the row stride)
        encs, sizes = ctx.encs, ctx.sizes
        g_values, grads = [], []
        for enc in encs:
            direct = accumulate_into_grad(enc.lattice_values)
            g_values.append(direct if direct is not None else torch.zeros_like(enc.lattice_values))
            grads.append(None if direct is not None else g_values[-1])
        for g0, ng, a, n in _PermutoEncodeGrouped._runs(sizes):
            if n == 0:
                continue
            run = encs[g0:g0 + ng]
            ptrs = (ctypes.c_void_p * ng)(*[v.data_ptr() for v in g_values[g0:g0 + ng]])
            cnt = (ctypes.c_int * ng)(*sizes[g0:g0 + ng])
            _lib.call("vsa_permuto_encode_bwd_grouped", ctypes.byref(run[0].plan), _plans_on_device(run, x.device),
                      ng, cnt, x[a:a + n], window, g_out[a:a + n], g_out.stride(0), ptrs, _lib.stream_ptr())
        return (None, None, None, None, None, *grads)      # positions: no gradient here


def permuto_hash_encoders_groupable(encs, points):
    """PermutoHashEncoders of one configuration (what `get_encoder("permutohash", ...)` gives every
    per-shell model) on CUDA points without a gradient path."""
    e0 = encs[0]
    if not all(isinstance(e, PermutoHashEncoder) for e in encs) or not points.is_cuda or points.requires_grad:
        return False

    def sig(e):
        got = getattr(e, "_group_sig", None)
        if got is None:               # formed once per encoder (the bounding box is a device read)
            bb = None if e.bb_sides is None else tuple(e.bb_sides.tolist())
            got = e._group_sig = (e.nr_levels, e.capacity, tuple(np.asarray(e.scale_list).tolist()),
                                  e.concat_points, e.concat_points_scaling, e.remove_last_element,
                                  e.nr_iters_for_c2f, e.encoder.pos_dim, bb)
        return got
    s0 = sig(e0)
    return all(sig(e) == s0 for e in encs[1:])


def permuto_hash_encode_grouped(encs, points, sizes, iter_nr=None):
    """[sum(sizes), output_dim]: rows of segment g through encs[g] — PermutoHashEncoder.forward's
    arithmetic (window, bounding-box normalisation, concatenated points, last channel dropped) done
    once for all segments, the lattice lookups per segment.  out_of_bounds is not computed (the
    per-shell models do not use it)."""
    e0 = encs[0]
    if iter_nr is None or iter_nr < 0:
        t = 1.0
    else:
        t = map_range_val(iter_nr, 0.0, e0.nr_iters_for_c2f, 0.3, 1.0)
    window = e0.c2f(t)
    window = None if e0.c2f.all_open else window.view(-1).to(points.device, torch.float32).contiguous()
    if e0.bb_sides is not None:
        points = points * _inv_half_sides(e0)
        points = (points + 1) / 2
    extra = e0.encoder.pos_dim if e0.concat_points else 0
    inner = tuple(e.encoder for e in encs)
    enc = _PermutoEncodeGrouped.apply(points.float(), tuple(int(n) for n in sizes), window, extra, inner,
                                      *[e.lattice_values for e in inner])
    if e0.remove_last_element:
        enc = enc[:, :-1]
    return enc


class ManualCtx:
    """What an autograd Function's forward / backward need from their ctx, for calling them WITHOUT the autograd
    engine (the fused legacy training step, methods.VolSurfs.fused_legacy_forward: same kernels, same order, no
    graph nodes, no engine pass — the loop of BASELINE configs[2] is as much host as device time)."""

    def __init__(self, n_inputs=64):
        self.needs_input_grad = (True,) * n_inputs
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass


def permuto_hash_encode_grouped_manual(encs, points, sizes, iter_nr=None):
    """permuto_hash_encode_grouped without autograd: returns (encoding, backward) where backward(g_encoding) adds the
    lattice gradients into the encoders' .grad buffers (they must own persistent ones: optim.FusedAdam)."""
    e0 = encs[0]
    t = 1.0 if iter_nr is None or iter_nr < 0 else map_range_val(iter_nr, 0.0, e0.nr_iters_for_c2f, 0.3, 1.0)
    window = e0.c2f(t)
    window = None if e0.c2f.all_open else window.view(-1).to(points.device, torch.float32).contiguous()
    if e0.bb_sides is not None:
        points = points * _inv_half_sides(e0)
        points = (points + 1) / 2
    extra = e0.encoder.pos_dim if e0.concat_points else 0
    inner = tuple(e.encoder for e in encs)
    ctx = ManualCtx()
    enc = _PermutoEncodeGrouped.forward(ctx, points.float(), tuple(int(n) for n in sizes), window, extra, inner,
                                        *[e.lattice_values for e in inner])
    drop = bool(e0.remove_last_element)

    def backward(g_enc):
        if drop:
            g_enc = torch.nn.functional.pad(g_enc, (0, 1))
        grads = _PermutoEncodeGrouped.backward(ctx, g_enc)[5:]
        for e, g in zip(inner, grads):          # (None when the kernel added straight into .grad)
            if g is not None:
                e.lattice_values.grad = g if e.lattice_values.grad is None else e.lattice_values.grad + g
    return (enc[:, :-1] if drop else enc), backward


class PermutoEncoding(torch.nn.Module):
    """`permutohedral_encoding.PermutoEncoding(pos_dim, capacity, nr_levels, nr_feat_per_level,
    scale_list, appply_random_shift_per_level=, concat_points=, concat_points_scaling=)` shaped
    (permutohash.py:28-37): callable `(points, window)`, `.output_dims()`, `.reset()`.
    Parameters: `lattice_values` [nr_levels, capacity, nr_feat] ~ N(0, 1e-5^2); buffer
    `random_shift_per_level` [nr_levels, pos_dim] ~ 10 N(0,1) (zeros when the shift is off)."""

    def __init__(self, pos_dim, capacity, nr_levels, nr_feat_per_level, scale_list,
                 appply_random_shift_per_level=True, concat_points=False,
                 concat_points_scaling=1.0, init_scale=1e-5, seed=None, device="cuda"):
        super().__init__()
        if nr_feat_per_level != 2 or not 2 <= pos_dim <= 4 or nr_levels > GRID_MAX_LEVELS:
            raise _lib.VolsurfsHipError("PermutoEncoding: 2 features per level, pos_dim 2..4, "
                                        f"<= {GRID_MAX_LEVELS} levels")
        if len(scale_list) != nr_levels:
            raise _lib.VolsurfsHipError("scale_list must have one entry per level")
        self.pos_dim, self.capacity, self.nr_levels = pos_dim, int(capacity), nr_levels
        self.nr_feat_per_level = nr_feat_per_level
        self.scale_list = np.asarray(scale_list, np.float64)
        self.concat_points, self.concat_points_scaling = concat_points, concat_points_scaling
        if seed is None:
            # every instance draws its own values and per-level shifts from the global torch RNG, as
            # the wrapped package does (the per-shell rgb / alpha models must not share their hash
            # collisions); torch.manual_seed(...) before construction makes a model reproducible
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        self.init_scale, self._seed = init_scale, seed
        g = torch.Generator().manual_seed(seed)
        self.lattice_values = torch.nn.Parameter(
            (torch.randn(nr_levels, self.capacity, nr_feat_per_level, generator=g) * init_scale).to(device))
        shift = torch.randn(nr_levels, pos_dim, generator=g) * 10.0 if appply_random_shift_per_level \
            else torch.zeros(nr_levels, pos_dim)
        self.register_buffer("random_shift_per_level", shift.to(device))
        self._plan, self._plan_key = None, None

    @property
    def plan(self):
        """The kernel's view of the geometry; rebuilt if the shift buffer was replaced
        (load_state_dict)."""
        key = self.random_shift_per_level._version, self.random_shift_per_level.data_ptr()
        if self._plan is None or key != self._plan_key:
            p = PermutoPlan()
            p.pos_dim, p.n_levels, p.n_features, p.capacity = self.pos_dim, self.nr_levels, 2, self.capacity
            shift = self.random_shift_per_level.detach().cpu().numpy()
            for l in range(self.nr_levels):
                for i in range(self.pos_dim):
                    p.scale_factor[l][i] = float(np.float32(
                        1.0 / math.sqrt((i + 1) * (i + 2)) / self.scale_list[l]))
                    p.random_shift[l][i] = float(shift[l, i])
            self._plan, self._plan_key = p, key
        return self._plan

    def output_dims(self):
        return self.nr_levels * self.nr_feat_per_level + (self.pos_dim if self.concat_points else 0)

    def forward(self, positions, anneal_window=None):
        extra = self.pos_dim if self.concat_points else 0
        if anneal_window is not None:
            anneal_window = anneal_window.to(positions.device, torch.float32).contiguous().view(-1)
        return _PermutoEncode.apply(self.lattice_values, positions.float(), anneal_window, self, extra)

    @torch.no_grad()
    def reset(self):
        g = torch.Generator().manual_seed(self._seed + 1)
        self.lattice_values.copy_((torch.randn(self.lattice_values.shape, generator=g) * self.init_scale))


class PermutoHashEncoder(Encoder):
    """encodings/permutohash.py:10-99: 24 levels x 2 features, capacity 2^18, sigma from 1.0 to
    1e-4 (np.geomspace), random shift per level, coarse-to-fine window, bounding-box
    normalisation, concatenated points with the LAST channel dropped (`remove_last_element`)."""

    def __init__(self, input_dim=3, nr_levels=24, log2_hashmap_size=18, nr_feat_per_level=2,
                 coarsest_scale=1.0, finest_scale=0.0001, nr_iters_for_c2f=0,
                 appply_random_shift_per_level=True, concat_points=True, concat_points_scaling=1.0,
                 remove_last_element=True, bb_sides=2.0, device="cuda"):
        capacity = pow(2, log2_hashmap_size)
        scale_list = np.geomspace(coarsest_scale, finest_scale, num=nr_levels)
        encoder = PermutoEncoding(input_dim, capacity, nr_levels, nr_feat_per_level, scale_list,
                                  appply_random_shift_per_level=appply_random_shift_per_level,
                                  concat_points=concat_points,
                                  concat_points_scaling=concat_points_scaling, device=device)
        super().__init__(input_dim, encoder.output_dims() - (1 if remove_last_element else 0))
        self.remove_last_element = remove_last_element
        self.encoder = encoder
        self.capacity, self.nr_levels, self.nr_feat_per_level = capacity, nr_levels, nr_feat_per_level
        self.scale_list = scale_list
        self.appply_random_shift_per_level = appply_random_shift_per_level
        self.concat_points, self.concat_points_scaling = concat_points, concat_points_scaling
        self.bb_sides = bb_sides
        if self.bb_sides is not None:
            if isinstance(self.bb_sides, float):
                self.bb_sides = np.array([self.bb_sides] * input_dim)
            if isinstance(self.bb_sides, np.ndarray):
                self.bb_sides = torch.tensor(self.bb_sides, dtype=torch.float32)
            self.bb_sides = self.bb_sides.to(device)
        self.c2f = Coarse2Fine(nr_levels)
        self.nr_iters_for_c2f = nr_iters_for_c2f

    def forward(self, points, iter_nr=None, **kwargs):
        if iter_nr is None or iter_nr < 0:
            t = 1.0
        else:
            t = map_range_val(iter_nr, 0.0, self.nr_iters_for_c2f, 0.3, 1.0)
        window = self.c2f(t)
        # t = 1 (evaluation, c2f off): all ones — the kernel takes NULL for that, which also spares
        # a host-to-device copy (an implicit synchronisation) per model and forward
        window = None if self.c2f.all_open else window.view(-1)
        out_of_bounds = None
        if self.bb_sides is not None:
            if self.compute_out_of_bounds:
                out_of_bounds = torch.logical_or((points <= -self.bb_sides / 2).any(dim=1),
                                                 (points >= self.bb_sides / 2).any(dim=1))
            points = points * _inv_half_sides(self)
            points = (points + 1) / 2
        enc = self.encoder(points, window)
        if self.remove_last_element:
            enc = enc[:, :-1]
        return enc, out_of_bounds

    def reset(self):
        self.encoder.reset()


def get_encoder(encoding, **kwargs):
    """utils/encoder.py:8-48."""
    if encoding == "none":
        return IdentityEncoder(input_dim=kwargs["input_dim"])
    if encoding == "frequency":
        return FrequencyEncoder(input_dim=kwargs["input_dim"], multires=kwargs["multires"])
    if encoding == "spherical_harmonics":
        return SHEncoder(input_dim=kwargs["input_dim"], degree=kwargs["degree"])
    if encoding == "gridhash":
        return GridHashEncoder(input_dim=kwargs["input_dim"], nr_levels=kwargs["nr_levels"],
                               nr_iters_for_c2f=kwargs["nr_iters_for_c2f"],
                               bb_sides=kwargs.get("bb_sides"))
    if encoding == "permutohash":
        return PermutoHashEncoder(input_dim=kwargs["input_dim"], nr_levels=kwargs["nr_levels"],
                                  nr_iters_for_c2f=kwargs["nr_iters_for_c2f"],
                                  bb_sides=kwargs.get("bb_sides"))
    raise NotImplementedError(
        "Unknown encoding mode, choose from [None, frequency, spherical_harmonics, permutohash, gridhash]")
