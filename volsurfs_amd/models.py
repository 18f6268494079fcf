"""Models of the legacy appearance branch and the background field (SURVEY §8a rows A5,
A10): `MLP`, `RGB`, `ColorSH`, `NerfHash` with the reference's constructor arguments,
forward signatures and module layout (`mlp.layers.<i>.weight/bias` state-dict keys match
/root/reference/volsurfs_py/models/{mlp,rgb,color_sh,nerfhash}.py, so the reference's
checkpoints load).

The encoders run on the HIP kernels of csrc/grid_encode.hip / permuto_encode.hip
(volsurfs_amd/encodings.py).  The MLPs (fp32 Linear + bias, exact GELU) run as ONE fused
launch on the fp32-input matrix cores (csrc/mlp_f32.hip: `vsa_mlp_fwd` / `vsa_mlp_bwd`),
activations never leaving registers between layers; `MLP.fused = False` selects the
reference's own torch op sequence (library GEMMs), which the tests compare against.
"""
import copy
import ctypes

import numpy as np
import torch

from . import _lib
from .encodings import SHEncoder, get_encoder
from .neural_textures import MAX_DEG  # noqa: F401  (re-exported for symmetry)


class _LinearBiasByGemv(torch.autograd.Function):
    """y = x W^T + b with the bias gradient formed as ones[1,B] @ g (a rocBLAS GEMV).  ATen's
    column-sum reduction took 6.6 ms per layer on the background path's [2.1 M, 64] gradients
    (27 % of that step, tools/bench_bg.py); forward and the other gradients are F.linear's."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ weight if ctx.needs_input_grad[0] else None
        gw = g.t() @ x if ctx.needs_input_grad[1] else None
        gb = (torch.ones(1, g.shape[0], device=g.device, dtype=g.dtype) @ g)[0] \
            if ctx.needs_input_grad[2] else None
        return gx, gw, gb


MLP_MAX_LAYERS = 6


class MlpPlan(ctypes.Structure):
    """Mirror of `vsa_mlp_plan` (include/volsurfs_hip.h)."""
    _fields_ = [("n_layers", ctypes.c_int32), ("dims", ctypes.c_int32 * (MLP_MAX_LAYERS + 1)),
                ("w", ctypes.c_void_p * MLP_MAX_LAYERS), ("b", ctypes.c_void_p * MLP_MAX_LAYERS)]


class MlpGrads(ctypes.Structure):
    """Mirror of `vsa_mlp_grads`."""
    _fields_ = [("dw", ctypes.c_void_p * MLP_MAX_LAYERS), ("db", ctypes.c_void_p * MLP_MAX_LAYERS),
                ("accumulate", ctypes.c_int32)]


def _mlp_plan(weights, biases):
    p = MlpPlan()
    p.n_layers = len(weights)
    p.dims[0] = weights[0].shape[1]
    for l, w in enumerate(weights):
        p.dims[l + 1] = w.shape[0]
        p.w[l] = w.data_ptr()
        p.b[l] = biases[l].data_ptr() if biases[l] is not None else None
    return p


# torch.is_grad_enabled() is always False INSIDE Function.forward, and parameters keep
# requires_grad=True under torch.no_grad(): the callers of the fused nodes record here whether the
# call is being differentiated, so that inference (render, render_camera, background evaluation)
# does not allocate and write the backward workspaces (z and GELU(z): M x sum(hidden) fp32 each).
_CALL = {"grad": True}


def _needs_backward(x, params):
    return _CALL["grad"] and (x.requires_grad or any(p_.requires_grad for p_ in params if p_ is not None))


def fused_mlp_supported(dims, x):
    """csrc/mlp_f32.hip: <= 6 linear layers, widths <= 128, hidden widths multiples of 32."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and 1 <= len(dims) - 1 <= MLP_MAX_LAYERS
            and all(1 <= d <= 128 for d in dims) and all(d % 32 == 0 for d in dims[1:-1]))


def _pad4(n):
    return (n + 3) // 4 * 4


def rows16(t):
    """t [M, W] can be handed to the MLP / field-head kernels IN PLACE as rows of 16-byte groups: unit column stride, a
    row stride that is a multiple of 4 floats, a 16-byte aligned base (include/volsurfs_hip.h, vsa_mlp_bwd)."""
    return (t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.stride(0) >= t.shape[1]
            and t.data_ptr() % 16 == 0)


def padded_rows(M, W, device, zero=False):
    """[M, W] view of a fresh [M, pad4(W)] buffer: the producer of a matrix the MLP kernels will read pads its rows so that
    they move as 16-byte accesses (a 51-wide feature row, a 65-wide output row: rows of 52 / 68 floats)."""
    buf = (torch.zeros if zero else torch.empty)(M, _pad4(W), device=device)
    return buf[:, :W] if W % 4 else buf


class _FusedMLP(torch.autograd.Function):
    """y = W_L(... GELU(W_1 x + b_1) ...) + b_L in ONE launch on the fp32 matrix cores
    (vsa_mlp_fwd); backward = vsa_mlp_bwd (data-gradient chain, weight gradients, bias sums).
    params = (w_0, b_0 or None, w_1, b_1, ...)."""

    @staticmethod
    def forward(ctx, x, has_bias, *params):
        nl = len(params) // 2 if has_bias else len(params)
        ws = [params[2 * l] if has_bias else params[l] for l in range(nl)]
        bs = [params[2 * l + 1] if has_bias else None for l in range(nl)]
        ws = [w.contiguous() for w in ws]
        bs = [b.contiguous() if b is not None else None for b in bs]
        x = x if rows16(x) else x.contiguous()        # (a producer's padded rows are read in place)
        M = x.shape[0]
        plan = _mlp_plan(ws, bs)
        sizes = [ctypes.c_longlong() for _ in range(3)]
        _lib.call("vsa_mlp_workspace", ctypes.byref(plan), ctypes.c_longlong(M),
                  *[ctypes.byref(v) for v in sizes])
        need = _needs_backward(x, params)
        dev = x.device
        packed = torch.empty(max(sizes[0].value, 1), device=dev)
        z = torch.empty(max(sizes[1].value, 1), device=dev) if need else None
        # GELU(z), the weight gradients' other operand: only the two-kernel backward reads it (the fused backward of
        # networks up to 96 wide forms it from z: include/volsurfs_hip.h, vsa_mlp_bwd_needs_act)
        xs = x.stride(0)
        fused_bwd = rows16(x) and int(_lib.lib().vsa_mlp_bwd_needs_act(
            ctypes.byref(plan), ctypes.c_int(xs), ctypes.c_int(xs if ctx.needs_input_grad[0] else 0))) == 0
        act = torch.empty_like(z) if need and not fused_bwd else None
        y = padded_rows(M, ws[-1].shape[0], dev)
        _lib.call("vsa_mlp_fwd", ctypes.byref(plan), x, xs, M, y, y.stride(0), z, act,
                  packed, _lib.stream_ptr())
        ctx.xs = xs
        ctx.save_for_backward(x, z, act, *ws, *[b for b in bs if b is not None])
        ctx.meta = (nl, has_bias, sizes[1].value, sizes[2].value)
        return y

    @staticmethod
    def backward(ctx, gy):
        nl, has_bias, act_n, part_n = ctx.meta
        x, z, act = ctx.saved_tensors[:3]
        ws = list(ctx.saved_tensors[3:3 + nl])
        bs = list(ctx.saved_tensors[3 + nl:]) if has_bias else [None] * nl
        M, dev = x.shape[0], x.device
        plan = _mlp_plan(ws, bs)
        if not rows16(gy):          # (e.g. the contiguous [M, 3] gradient of a torch op: re-laid once as padded rows)
            g_ = padded_rows(M, gy.shape[1], dev)
            g_.copy_(gy)
            gy = g_
        dz = torch.empty(max(act_n, 1), device=dev) if act is not None else None
        packed = torch.empty(sum(((w.shape[0] + 31) // 32) * ((w.shape[1] + 31) // 32) * 1024 for w in ws),
                             device=dev)
        partial = torch.empty(max(part_n, 1), device=dev)
        xs = ctx.xs
        dx = None
        if ctx.needs_input_grad[0]:     # rows of x's own stride (the fused backward stores them as 16-byte groups)
            dxb = torch.empty(M, xs, device=dev)
            dx = dxb[:, :x.shape[1]] if xs != x.shape[1] else dxb
        # (an empty batch: the entry point returns without a launch — the gradients are zero)
        alloc = torch.zeros_like if M == 0 else torch.empty_like
        gw = [alloc(w) for w in ws]
        gb = [alloc(b) if b is not None else None for b in bs]
        grads = MlpGrads()
        for l in range(nl):
            grads.dw[l] = gw[l].data_ptr()
            grads.db[l] = gb[l].data_ptr() if gb[l] is not None else None
        _lib.call("vsa_mlp_bwd", ctypes.byref(plan), x, xs, M, gy, gy.stride(0), z, dz, act,
                  packed, partial, dx, xs, ctypes.byref(grads), _lib.stream_ptr())
        out = []
        for l in range(nl):
            out.append(gw[l])
            if has_bias:
                out.append(gb[l])
        return (dx, None, *out)


def _direct_grads(params):
    """The parameters' own .grad buffers if EVERY one of them has a usable one (the conditions of
    optim.accumulate_into_grad, checked in one pass; the optimisers are told once that their
    gradients are dirty), else None (then the Function returns freshly allocated gradients as usual)."""
    if not params:
        return None
    got, opts = [], set()
    for p_ in params:
        if not (isinstance(p_, torch.nn.Parameter) and p_.requires_grad):
            return None
        g = p_.grad
        if g is None or g.dtype != torch.float32 or g.shape != p_.shape or not g.is_contiguous():
            return None
        got.append(g)
        o = getattr(p_, "_vsa_optimizer", None)
        if o is not None:
            opts.add(o)
    for o in opts:
        o = o()
        if o is not None:
            o.mark_grads_dirty()
    return got


MLP_MAX_GROUPS = 8
_NO_CACHE = __import__("os").environ.get("VSA_NO_DESC_CACHE", "0") == "1"     # A/B switch


class _FusedMLPGrouped(torch.autograd.Function):
    """G MLPs of ONE architecture (the per-shell models of the legacy appearance branch) applied to
    G consecutive row segments of x: one autograd node and ONE set of launches for up to 8 groups
    (`vsa_mlp_fwd_grouped` / `vsa_mlp_bwd_grouped`: group = blockIdx.y) — launched one by one the
    networks fill a third of the chip each (~80 workgroups for the 10 k hits of a shell) and cost five
    times the host calls; the training loop of BASELINE configs[2] is bound by both.
    params = group-major (w_0, b_0, w_1, b_1, ...) per group."""

    @staticmethod
    def _batches(sizes):
        """[(first group, nr groups, first row, rows)] in runs of at most MLP_MAX_GROUPS groups."""
        out, a = [], 0
        for g0 in range(0, len(sizes), MLP_MAX_GROUPS):
            n = sum(sizes[g0:g0 + MLP_MAX_GROUPS])
            out.append((g0, min(MLP_MAX_GROUPS, len(sizes) - g0), a, n))
            a += n
        return out

    # The ctypes descriptors of a set of networks (plans: weight / bias pointers; gradient pointer
    # tables) are kept between calls, keyed on the parameter objects and validated by their data
    # pointers: building them field by field for 5 x 8 tensors cost the host ~0.3 ms per iteration
    # of a loop that is bound by it.
    _desc_cache = {}

    @staticmethod
    def _descriptors(params, nl, has_bias, G):
        """(plans per run of <= 8 groups, shapes of group 0's weights) for contiguous parameters."""
        key = (tuple(map(id, params)), nl, has_bias)
        ptrs = tuple(p_.data_ptr() for p_ in params)
        hit = _FusedMLPGrouped._desc_cache.get(key)
        if hit is None or hit[0] != ptrs or _NO_CACHE:
            per = nl * (2 if has_bias else 1)
            groups = []
            for g in range(G):
                ps = params[g * per:(g + 1) * per]
                ws = [ps[2 * l] if has_bias else ps[l] for l in range(nl)]
                bs = [ps[2 * l + 1] if has_bias else None for l in range(nl)]
                groups.append((ws, bs))
            runs = []
            for g0 in range(0, G, MLP_MAX_GROUPS):
                ng = min(MLP_MAX_GROUPS, G - g0)
                runs.append((MlpPlan * ng)(*[_mlp_plan(*groups[g0 + i]) for i in range(ng)]))
            w0 = groups[0][0]
            dims = (w0[-1].shape[0], sum(w.shape[0] for w in w0[:-1]),
                    sum(((w.shape[0] + 31) // 32) * ((w.shape[1] + 31) // 32) * 1024 for w in w0))
            if len(_FusedMLPGrouped._desc_cache) > 32:
                _FusedMLPGrouped._desc_cache.clear()
            hit = _FusedMLPGrouped._desc_cache[key] = (ptrs, runs, dims, {}, tuple(params))
        return hit

    @staticmethod
    def forward(ctx, x, sizes, has_bias, nl, *params):
        x = x if rows16(x) else x.contiguous()        # (padded rows — cat_rows16 — are read in place: 16-byte accesses)
        G = len(sizes)
        if not all(p_.is_contiguous() for p_ in params):
            raise _lib.VolsurfsHipError("fused_mlp_grouped: contiguous parameters only")
        _, runs, (out_dim, hidden, packed_n), _, _ = _FusedMLPGrouped._descriptors(params, nl, has_bias, G)
        need = _needs_backward(x, params)
        dev = x.device
        M = x.shape[0]
        y = padded_rows(M, out_dim, dev)
        z = torch.empty(max(M * hidden, 1), device=dev) if need else None
        act = torch.empty_like(z) if need else None       # GELU(z): the weight gradients' other operand
        packed = torch.empty(max(packed_n, 1) * min(G, MLP_MAX_GROUPS), device=dev)
        # one run of groups and a backward to come: the packing launch also writes the order the
        # backward needs (kept on ctx: the parameters only change in the optimiser step after it)
        packed_bwd = torch.empty_like(packed) if need and len(runs) == 1 else None
        for (g0, ng, a, n), plans in zip(_FusedMLPGrouped._batches(sizes), runs):
            if n == 0:
                continue
            cnt = (ctypes.c_int * ng)(*sizes[g0:g0 + ng])
            _lib.call("vsa_mlp_fwd_grouped", plans, ng, cnt, x[a:a + n], x.stride(0), y[a:a + n], y.stride(0),
                      z[a * hidden:] if z is not None else None, act[a * hidden:] if act is not None else None,
                      packed, packed_bwd, _lib.stream_ptr())
        ctx.packed_bwd = packed_bwd
        ctx.param_versions = [p_._version for p_ in params] if packed_bwd is not None else None
        ctx.save_for_backward(x, z, act)
        ctx.meta = (tuple(sizes), has_bias, nl, hidden, packed_n)
        ctx.param_objs = params        # the Parameter objects themselves: backward reads their weights and may add into their .grad
        return y

    @staticmethod
    def backward(ctx, gy):
        sizes, has_bias, nl, hidden, packed_n = ctx.meta
        x, z, act = ctx.saved_tensors
        params = ctx.param_objs
        G = len(sizes)
        dev = x.device
        if not rows16(gy):
            g_ = padded_rows(gy.shape[0], gy.shape[1], dev)
            g_.copy_(gy)
            gy = g_
        dx = None
        if ctx.needs_input_grad[0]:        # rows of x's stride
            dxb = torch.empty(x.shape[0], x.stride(0), device=dev)
            dx = dxb[:, :x.shape[1]] if x.stride(0) != x.shape[1] else dxb
        nmax = max(sizes) if sizes else 0
        dz = torch.empty(max(x.shape[0] * hidden, 1), device=dev)
        _, runs, _, grad_cache, _ = _FusedMLPGrouped._descriptors(params, nl, has_bias, G)
        # the transposed weights the forward's packing launch left (unless a parameter was written since)
        packed = ctx.packed_bwd
        ready = packed is not None and ctx.param_versions == [p_._version for p_ in params]
        if not ready:
            packed = torch.empty(max(packed_n, 1) * min(G, MLP_MAX_GROUPS), device=dev)
        # parameters that own a persistent .grad (FusedAdam): the reduce kernel adds straight into it
        # and autograd gets None — one accumulation kernel per parameter less (80 per step).  One
        # setting for the whole launch: every parameter of every group has to qualify.
        direct = _direct_grads(list(params))
        if direct is not None:
            targets = direct
        else:
            targets = [torch.empty_like(p_) for p_ in params]
        # gradient pointer tables: kept while the target buffers stay where they are (persistent .grad)
        tptrs = tuple(t.data_ptr() for t in targets)
        got = grad_cache.get(direct is not None)
        if got is None or got[0] != tptrs or _NO_CACHE:
            per = nl * (2 if has_bias else 1)
            tables = []
            for g0 in range(0, G, MLP_MAX_GROUPS):
                ng = min(MLP_MAX_GROUPS, G - g0)
                arr = (MlpGrads * ng)()
                for i in range(ng):
                    ts = tptrs[(g0 + i) * per:(g0 + i + 1) * per]
                    arr[i].accumulate = 1 if direct is not None else 0
                    for l in range(nl):
                        arr[i].dw[l] = ts[2 * l] if has_bias else ts[l]
                        arr[i].db[l] = ts[2 * l + 1] if has_bias else None
                tables.append(arr)
            got = (tptrs, tables)
            if direct is not None:
                grad_cache[True] = got
        partial = None
        for (g0, ng, a, n), plans, grads in zip(_FusedMLPGrouped._batches(sizes), runs, got[1]):
            if n == 0:
                if direct is None:
                    per = nl * (2 if has_bias else 1)
                    for t in targets[g0 * per:(g0 + ng) * per]:
                        t.zero_()
                continue
            cnt = (ctypes.c_int * ng)(*sizes[g0:g0 + ng])
            if partial is None:
                sz = ctypes.c_longlong()
                _lib.call("vsa_mlp_workspace", ctypes.byref(plans[0]), ctypes.c_longlong(nmax), None, None,
                          ctypes.byref(sz))
                partial = torch.empty(max(sz.value, 1) * min(G, MLP_MAX_GROUPS), device=dev)
            _lib.call("vsa_mlp_bwd_grouped", plans, ng, cnt, x[a:a + n], x.stride(0), gy[a:a + n], gy.stride(0),
                      z[a * hidden:], dz[a * hidden:], act[a * hidden:], packed, 1 if ready else 0, partial,
                      dx[a:a + n] if dx is not None else None, x.stride(0), grads, _lib.stream_ptr())
        if direct is not None:
            return (dx, None, None, None, *([None] * len(params)))
        return (dx, None, None, None, *targets)


def cat_rows16(parts):
    """torch.cat(parts, 1) into rows padded to a multiple of 4 floats (a [M, W] view of a [M, pad4(W)] buffer): the MLP
    kernels read — and write the gradient of — such rows as 16-byte groups (include/volsurfs_hip.h, vsa_mlp_fwd)."""
    M, W = parts[0].shape[0], sum(p_.shape[1] for p_ in parts)
    out = padded_rows(M, W, parts[0].device)
    # (no autograd through `out=`: for the autograd-free step — fused_legacy_forward's tape — only)
    with torch.no_grad():
        torch.cat([p_.detach() for p_ in parts], 1, out=out)     # one launch, straight into the strided view
    return out


def fused_mlp_grouped(mlps, x, sizes):
    """mlps: G `MLP` modules of one architecture; x [sum(sizes), in]; segment g goes through
    mlps[g].  Returns [sum(sizes), out]."""
    first = [m for m in mlps[0].layers if isinstance(m, torch.nn.Linear)]
    nl, has_bias = len(first), bool(mlps[0].bias)
    params = []
    for mlp in mlps:
        lin = [m for m in mlp.layers if isinstance(m, torch.nn.Linear)]
        for m in lin:
            params.append(m.weight)
            if has_bias:
                params.append(m.bias)
    _CALL["grad"] = torch.is_grad_enabled()
    return _FusedMLPGrouped.apply(x, tuple(int(n) for n in sizes), has_bias, nl, *params)


def fused_mlp_grouped_manual(mlps, x, sizes):
    """fused_mlp_grouped without autograd (encodings.ManualCtx): returns (y, backward) where backward(gy) returns dx
    and adds the weight / bias gradients into the parameters' .grad buffers."""
    from .encodings import ManualCtx
    first = [m for m in mlps[0].layers if isinstance(m, torch.nn.Linear)]
    nl, has_bias = len(first), bool(mlps[0].bias)
    params = []
    for mlp in mlps:
        for m in (l for l in mlp.layers if isinstance(l, torch.nn.Linear)):
            params.append(m.weight)
            if has_bias:
                params.append(m.bias)
    _CALL["grad"] = True
    ctx = ManualCtx()
    y = _FusedMLPGrouped.forward(ctx, x, tuple(int(n) for n in sizes), has_bias, nl, *params)

    def backward(gy):
        out = _FusedMLPGrouped.backward(ctx, gy)
        for p_, g in zip(params, out[4:]):      # (None when the kernel added straight into .grad)
            if g is not None:
                p_.grad = g if p_.grad is None else p_.grad + g
        return out[0]
    return y, backward


def mlps_groupable(mlps, x):
    """All MLPs share one architecture the fused kernel covers."""
    def sig(m):
        lin = [l for l in m.layers if isinstance(l, torch.nn.Linear)]
        return ([lin[0].in_features] + [l.out_features for l in lin], bool(m.bias), m.last_layer_linear)
    s0 = sig(mlps[0])
    # (per INSTANCE: an MLP that opted out with `mlp.fused = False` must not ride a grouped fused launch)
    return (all(m.fused for m in mlps) and s0[2] and all(sig(m) == s0 for m in mlps) and fused_mlp_supported(s0[0], x))


class MLP(torch.nn.Module):
    """models/mlp.py:8-69: Linear + GELU stack, optional linear last layer."""

    def __init__(self, in_channels, nr_out_channels_per_layer, last_layer_linear, bias=True):
        super().__init__()
        self.last_layer_linear = last_layer_linear
        self.in_channels = in_channels
        self.nr_out_channels_per_layer = nr_out_channels_per_layer
        self.bias = bias
        in_channels_ = in_channels
        modules = []
        for i, cur in enumerate(nr_out_channels_per_layer):
            modules.append(torch.nn.Linear(in_channels_, cur, bias=bias))
            is_last = i == len(nr_out_channels_per_layer) - 1
            if is_last and last_layer_linear:
                continue
            modules.append(torch.nn.GELU())
            in_channels_ = cur
        self.layers = torch.nn.Sequential(*modules)

    fused = True      # class-wide switch: False = the torch op sequence (tests compare the two)

    def forward(self, x):
        linears = [m for m in self.layers if isinstance(m, torch.nn.Linear)]
        dims = [linears[0].in_features] + [m.out_features for m in linears]
        if self.fused and self.last_layer_linear and fused_mlp_supported(dims, x):
            params = []
            for m in linears:
                params.append(m.weight)
                if self.bias:
                    params.append(m.bias)
            _CALL["grad"] = torch.is_grad_enabled()
            return _FusedMLP.apply(x, bool(self.bias), *params)
        if self.fused and x.is_cuda:
            # No library fallback on the product path: a CUDA MLP outside what
            # csrc/mlp_f32.hip covers (wider than 128, hidden widths not multiples of 32, GELU after the
            # last layer, non-fp32) is an error unless the caller asks for the torch op sequence
            # (rocBLAS GEMMs) explicitly with `MLP.fused = False` (class-wide) or `mlp.fused = False`.
            raise _lib.VolsurfsHipError(
                f"volsurfs_amd.models.MLP {dims} (last_layer_linear={self.last_layer_linear}, dtype {x.dtype}): "
                "outside the fused HIP kernel's shapes; set `.fused = False` to run the torch op sequence")
        for layer in self.layers:
            if isinstance(layer, torch.nn.Linear) and layer.bias is not None and x.dim() == 2 \
                    and torch.is_grad_enabled():
                x = _LinearBiasByGemv.apply(x, layer.weight, layer.bias)
            else:
                x = layer(x)
        return x

    def reset(self):
        for layer in self.layers:
            if isinstance(layer, torch.nn.Linear):
                layer.reset_parameters()


def _bb_sides(bb_sides, in_channels, device):
    if isinstance(bb_sides, float):
        bb_sides = np.array([bb_sides] * in_channels)
    if isinstance(bb_sides, np.ndarray):
        bb_sides = torch.tensor(bb_sides, dtype=torch.float32)
    return bb_sides.to(device) if bb_sides is not None else None


class RGB(torch.nn.Module):
    """models/rgb.py:13-149 (use_lipshitz_mlp is not built: raises)."""

    def __init__(self, in_channels, mlp_layers_dims, pos_encoder_type, dir_encoder_type,
                 out_channels=3, pos_dep=True, view_dep=True, geom_feat_dep=False,
                 normal_dep=False, sh_deg=5, in_geom_feat_size=32, nr_iters_for_c2f=0,
                 use_lipshitz_mlp=False, bb_sides=2.0, device="cuda"):
        super().__init__()
        assert (pos_dep and in_channels > 0) or (not pos_dep and in_channels == 0), \
            "pos_dep and in_channels must be consistent"
        if use_lipshitz_mlp:
            raise NotImplementedError("LipshitzMLP (models/lipshitz_mlp.py) is outside SURVEY §8")
        self.in_channels = in_channels
        self.mlp_layers_dims = copy.deepcopy(mlp_layers_dims)
        self.out_channels = out_channels
        self.pos_encoder_type, self.dir_encoder_type = pos_encoder_type, dir_encoder_type
        self.sh_deg = sh_deg
        self.pos_dep, self.view_dep = pos_dep, view_dep
        self.normal_dep, self.geom_feat_dep = normal_dep, geom_feat_dep
        self.in_geom_feat_size = in_geom_feat_size
        self.bb_sides = _bb_sides(bb_sides, in_channels, device)
        mlp_in = 0
        if pos_dep:
            self.pos_encoder = get_encoder(pos_encoder_type, input_dim=in_channels, nr_levels=24,
                                           nr_iters_for_c2f=nr_iters_for_c2f, multires=6,
                                           bb_sides=self.bb_sides)
            self.pos_encoder.compute_out_of_bounds = False       # (models/rgb.py:112-116 drops it too)
            mlp_in += self.pos_encoder.output_dim
        if view_dep:
            self.dir_encoder = get_encoder(dir_encoder_type, input_dim=3, degree=sh_deg)
            mlp_in += self.dir_encoder.output_dim
        if normal_dep:
            mlp_in += 3
        if geom_feat_dep:
            mlp_in += in_geom_feat_size
        self.mlp = MLP(mlp_in, self.mlp_layers_dims + [out_channels], last_layer_linear=True).to(device)
        self.sigmoid = torch.nn.Sigmoid()

    def forward(self, points=None, samples_dirs=None, normals=None, iter_nr=None, geom_feat=None):
        parts = []
        if self.pos_dep:
            feats = self.pos_encoder(points, iter_nr=iter_nr)
            parts.append(feats[0] if isinstance(feats, tuple) else feats)
        if self.view_dep:
            with torch.no_grad():
                parts.append(self.dir_encoder(samples_dirs, iter_nr=iter_nr))
        if self.normal_dep:
            parts.append(normals)
        if self.geom_feat_dep and self.in_geom_feat_size > 0:
            if geom_feat is None:
                raise ValueError("geom_feat is required")     # the reference prints and exit(1)s
            parts.append(geom_feat)
        return self.sigmoid(self.mlp(torch.cat(parts, 1)))


def sh_eval(sh, dirs, degree):
    """SHEncoder.eval (encodings/sphericalharmonics.py:155-229): sh [M, C, (deg+1)^2],
    dirs [M, 3] -> [M, C]; band 0 first, then the basis-weighted sum in index order."""
    basis = SHEncoder(3, degree)(dirs)                     # [M, n]
    result = basis[:, None, 0] * sh[..., 0]
    for i in range(1, (degree + 1) ** 2):
        result = result + basis[:, None, i] * sh[..., i]
    return result


class ColorSH(torch.nn.Module):
    """models/color_sh.py:15-143: position -> SH coefficients -> view-dependent colour."""

    def __init__(self, in_channels, mlp_layers_dims, pos_encoder_type, out_channels=3, sh_deg=3,
                 geom_feat_dep=False, normal_dep=False, in_geom_feat_size=0, nr_iters_for_c2f=0,
                 bb_sides=2.0, device="cuda"):
        super().__init__()
        self.in_channels, self.sh_deg = in_channels, sh_deg
        self.mlp_layers_dims = copy.deepcopy(mlp_layers_dims)
        self.nr_coeffs = (sh_deg + 1) ** 2
        self.color_channels = out_channels
        self.out_channels = self.nr_coeffs * out_channels
        self.pos_encoder_type = pos_encoder_type
        self.pos_dep, self.normal_dep, self.geom_feat_dep = True, normal_dep, geom_feat_dep
        self.in_geom_feat_size = in_geom_feat_size
        self.bb_sides = _bb_sides(bb_sides, in_channels, device)
        # the reference passes `points_scaling=` here, which get_encoder ignores: bb_sides=None
        self.pos_encoder = get_encoder(pos_encoder_type, input_dim=in_channels, nr_levels=24,
                                       nr_iters_for_c2f=nr_iters_for_c2f, multires=6)
        self.pos_encoder.compute_out_of_bounds = False
        mlp_in = self.pos_encoder.output_dim + (3 if normal_dep else 0) + \
            (in_geom_feat_size if geom_feat_dep else 0)
        self.mlp = MLP(mlp_in, self.mlp_layers_dims + [self.out_channels], last_layer_linear=True).to(device)
        self.sigmoid = torch.nn.Sigmoid()

    def forward(self, points, samples_dirs=None, normals=None, geom_feat=None, iter_nr=None):
        feats = self.pos_encoder(points, iter_nr=iter_nr)
        data = feats[0] if isinstance(feats, tuple) else feats
        if self.normal_dep:
            if normals is None:
                raise ValueError("normals are required for normal dependent model")
            data = torch.cat([data, normals], 1)
        if self.geom_feat_dep and self.in_geom_feat_size > 0:
            if geom_feat is None:
                raise ValueError("geom_feat is required")
            data = torch.cat([data, geom_feat], 1)
        pred = self.mlp(data)
        if samples_dirs is None:
            return pred
        sh = pred.reshape(-1, self.color_channels, self.nr_coeffs)
        return self.sigmoid(sh_eval(sh, samples_dirs, self.sh_deg))


class _FieldHead(torch.autograd.Function):
    """nerfhash.py:72-91 between the two MLPs, one kernel each way (csrc/field_head.hip):
    (y1 [n, 1+F], dirs_enc [n, E]) -> (x2 = cat(gelu(y1[:, 1:]), dirs_enc), softplus(y1[:, :1]))."""

    @staticmethod
    def forward(ctx, y1, dirs_enc):
        y1 = y1 if rows16(y1) and y1.dtype == torch.float32 else _lib.check_f32(y1.contiguous())
        dirs_enc = _lib.check_f32(dirs_enc.contiguous(), y1.shape[0], dirs_enc.shape[1])
        n, F, E = y1.shape[0], y1.shape[1] - 1, dirs_enc.shape[1]
        x2 = padded_rows(n, F + E, y1.device)
        density = torch.empty(n, 1, device=y1.device)
        ctx.s1 = s1 = y1.stride(0)
        _lib.call("vsa_field_head_fwd", y1, s1, dirs_enc, ctypes.c_longlong(n), F, E, x2, density,
                  _lib.stream_ptr())
        ctx.save_for_backward(y1)
        ctx.dims = (F, E)
        return x2, density

    @staticmethod
    def backward(ctx, g_x2, g_density):
        (y1,) = ctx.saved_tensors
        F, E = ctx.dims
        g_x2 = g_x2.contiguous() if g_x2 is not None else None
        g_density = g_density.contiguous() if g_density is not None else None
        dyb = torch.empty(y1.shape[0], ctx.s1, device=y1.device)       # rows of y1's stride (padding columns: zeros)
        dy1 = dyb[:, :1 + F] if ctx.s1 != 1 + F else dyb
        _lib.call("vsa_field_head_bwd", y1, ctx.s1, g_x2, g_density, ctypes.c_longlong(y1.shape[0]), F, E, dy1,
                  _lib.stream_ptr())
        return dy1, None


class NerfHash(torch.nn.Module):
    """models/nerfhash.py:11-91: the background radiance field render_contracted_bg evaluates
    at 32 contracted samples per ray (utils/background.py:72-80)."""

    def __init__(self, in_channels, pos_encoder_type, dir_encoder_type, nr_iters_for_c2f=0,
                 device="cuda"):
        super().__init__()
        self.in_channels = in_channels
        self.pos_encoder_type, self.dir_encoder_type = pos_encoder_type, dir_encoder_type
        self.pos_encoder = get_encoder(pos_encoder_type, input_dim=in_channels, nr_levels=24,
                                       nr_iters_for_c2f=nr_iters_for_c2f, multires=6, bb_sides=2.0)
        self.pos_encoder.compute_out_of_bounds = False
        self.pos_encoder_output_dims = self.pos_encoder.output_dim
        self.dir_encoder = get_encoder(dir_encoder_type, input_dim=3, degree=3)
        self.dir_encoder_output_dims = self.dir_encoder.output_dim
        self.nr_feat_for_rgb = 64
        self.mlp_feat_and_density = MLP(self.pos_encoder_output_dims,
                                        [64, 64, 64, self.nr_feat_for_rgb + 1],
                                        last_layer_linear=True).to(device)
        self.mlp_rgb = MLP(self.nr_feat_for_rgb + self.dir_encoder_output_dims, [64, 64, 3],
                           last_layer_linear=True).to(device)
        self.softplus, self.sigmoid, self.gelu = torch.nn.Softplus(), torch.nn.Sigmoid(), torch.nn.GELU()

    def _features(self, points, iter_nr):
        feats = self.pos_encoder(points, iter_nr=iter_nr)
        return feats[0] if isinstance(feats, tuple) else feats

    def forward(self, samples_3d, samples_dirs, iter_nr=None):
        assert samples_3d.shape[1] == self.in_channels, "points should be N x in_channels"
        point_features = self._features(samples_3d, iter_nr)
        with torch.no_grad():
            dirs_enc = self.dir_encoder(samples_dirs)
        feat_and_density = self.mlp_feat_and_density(point_features)
        if NerfHash.fused_head and feat_and_density.is_cuda and feat_and_density.dtype == torch.float32 \
                and dirs_enc.dtype == torch.float32 and not dirs_enc.requires_grad:
            x_rgb, density = _FieldHead.apply(feat_and_density, dirs_enc)
            return self.sigmoid(self.mlp_rgb(x_rgb)), density
        density = feat_and_density[:, 0:1]
        feat_rgb = feat_and_density[:, 1:self.nr_feat_for_rgb + 1]
        rgb = self.mlp_rgb(torch.cat([self.gelu(feat_rgb), dirs_enc], 1))
        return self.sigmoid(rgb), self.softplus(density)

    fused_head = True      # class-wide switch: False = the torch op sequence (tests compare the two)

    def get_only_density(self, ray_samples, iter_nr=None):
        points = ray_samples.view(-1, ray_samples.shape[-1])
        feat_and_density = self.mlp_feat_and_density(self._features(points, iter_nr))
        return self.softplus(feat_and_density[:, 0:1])
