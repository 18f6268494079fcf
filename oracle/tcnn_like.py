"""Oracle: 2-D multiresolution hash grid + fully-fused MLP, as configured at
/root/reference/volsurfs_py/models/neural_texture.py:54-77.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference gets both from `tinycudann` (NVlabs/tiny-cuda-nn master, pip dep
with no version pin: README.md:45-46), which is absent.  This restates the
published algorithm (Instant-NGP / tiny-cuda-nn `GridEncoding` with
`GridType::Hash`, `InterpolationType::Linear`; `FullyFusedMLP`):
  scale_l = 2^(l*log2(per_level_scale)) * base_resolution - 1
  res_l   = ceil(scale_l) + 1
  size_l  = min(round_up(res_l^D, 8), 2^log2_hashmap_size)
  pos     = x*scale_l + 0.5 (rounded product, then rounded sum; tiny-cuda-nn uses one
            fmaf — immaterial for an unpinned restatement) ; cell = floor(pos) ; frac = pos - cell
  index   = dense (x + y*res_l) while the stride fits size_l, else
            XOR_d(cell_d * prime_d), primes (1, 2654435761) ; index %= size_l
  feature = bilinear over the 4 corners, ACCUMULATED IN THE PARAMETER TYPE (half), as the
            published kernel does (tiny-cuda-nn include/tiny-cuda-nn/encodings/grid.h,
            kernel_grid: `result = fma((T)weight, grid_val(pos_grid_local), result)` with
            T = __half, weight = prod_d (1 - frac_d | frac_d) in float, corners in index
            order, bit 0 = x): four half-precision FMAs, one rounding each.
            `accumulate="f32"` keeps round 1-2's restatement (fp32 sum, rounded once) so that
            the texel-flip rate between the two can be reported (tests/test_parity_report.py).
            The backward follows kernel_grid_backward: d table[index_c] += weight (float) * dL/dfeature.
  MLP     = bias-free, ReLU hidden, no output activation, fp16 weights and
            activations (this restatement accumulates each layer in fp32 and
            rounds the activations to fp16; tiny-cuda-nn accumulates in fp16 —
            PARITY UNPINNED, and no bit-level claim is made against it).
Negative cells follow CUDA's float->int->uint32 wrap of tiny-cuda-nn's
`pos_fract` ((uint32_t)(int)floorf(pos)).
"""
import math

import numpy as np
import torch

PRIME_Y = 2654435761


class GridGeometry:
    def __init__(self, n_levels=16, n_features_per_level=2, log2_hashmap_size=15,
                 base_resolution=16, per_level_scale=1.5, n_dims=2):
        assert n_features_per_level == 2 and n_dims == 2
        self.n_levels = n_levels
        self.n_features = n_features_per_level
        self.hashmap_size = 1 << log2_hashmap_size
        log2_pls = np.float32(math.log2(per_level_scale))
        self.scale, self.res, self.size, self.offset = [], [], [], [0]
        for l in range(n_levels):
            s = np.float32(np.exp2(np.float32(l) * log2_pls)) * np.float32(base_resolution) - np.float32(1.0)
            r = int(np.ceil(s)) + 1
            n = r ** n_dims
            n = (n + 7) // 8 * 8
            n = min(n, self.hashmap_size)
            self.scale.append(float(np.float32(s)))
            self.res.append(r)
            self.size.append(n)
            self.offset.append(self.offset[-1] + n)
        self.n_params = self.offset[-1] * self.n_features
        self.n_output_dims = n_levels * n_features_per_level

    def index(self, l, cx, cy):
        """cx, cy int64 tensors holding uint32 values -> entry index in level l."""
        res, size = self.res[l], self.size[l]
        M = 0xFFFFFFFF
        if res <= size:  # stride after dim 0 (= res) still fits: add the y term
            idx = (cx + cy * res) & M
            if res * res > size:  # stride overflowed the table -> hash instead
                idx = (cx ^ ((cy * PRIME_Y) & M)) & M
        else:
            idx = cx
            idx = (cx ^ ((cy * PRIME_Y) & M)) & M
        return idx % size


GRID_ACCUMULATE = "f16"     # module default: the published kernel's half-precision FMA chain


def _grid_cells(geom, l, x):
    """Level l: corner indices [4,B] int64 and corner weights [4,B] fp32 (index order, bit 0 = x)."""
    pos = x * np.float32(geom.scale[l]) + np.float32(0.5)
    cell = torch.floor(pos)
    frac = pos - cell
    c = cell.to(torch.int64) & 0xFFFFFFFF
    idx, w = [], []
    for corner in range(4):
        dx, dy = corner & 1, (corner >> 1) & 1
        wx = frac[:, 0] if dx else 1 - frac[:, 0]
        wy = frac[:, 1] if dy else 1 - frac[:, 1]
        idx.append(geom.index(l, (c[:, 0] + dx) & 0xFFFFFFFF, (c[:, 1] + dy) & 0xFFFFFFFF))
        w.append(wx * wy)
    return torch.stack(idx), torch.stack(w)


class _HashGridF16(torch.autograd.Function):
    """Forward: per level `acc = fma(half(w_c), table_h[idx_c], acc)` over the corners in half
    precision (each FMA evaluated exactly in float64 — an 11-bit x 11-bit product plus an 11-bit
    addend of comparable exponent fits 53 bits — and rounded ONCE to half by numpy's direct
    double -> half conversion).  Backward: float corner weights, as kernel_grid_backward."""

    @staticmethod
    def forward(ctx, geom, table, x):
        tab = table.detach().half().numpy()
        outs, saved = [], []
        for l in range(geom.n_levels):
            idx, w = _grid_cells(geom, l, x.detach())
            w16 = w.numpy().astype(np.float16).astype(np.float64)
            acc = np.zeros((x.shape[0], 2), dtype=np.float16)
            for corner in range(4):
                val = tab[geom.offset[l] + idx[corner].numpy()].astype(np.float64)
                acc = (w16[corner][:, None] * val + acc.astype(np.float64)).astype(np.float16)
            outs.append(torch.from_numpy(acc))
            saved.append((idx, w))
        ctx.geom, ctx.saved, ctx.shape, ctx.dtype = geom, saved, table.shape, table.dtype
        return torch.cat(outs, dim=1)

    @staticmethod
    def backward(ctx, g):
        geom = ctx.geom
        gt = torch.zeros(ctx.shape, dtype=torch.float32)
        g = g.float()
        for l, (idx, w) in enumerate(ctx.saved):
            gl = g[:, 2 * l:2 * l + 2]
            for corner in range(4):
                gt.index_add_(0, geom.offset[l] + idx[corner], w[corner][:, None] * gl)
        return None, gt.to(ctx.dtype), None


def hashgrid_forward(geom, table, x, accumulate=None):
    """table [n_entries, 2] (any float dtype; used as fp16 values), x [B,2] fp32.
    Returns features [B, 32] fp16 (level-major).  accumulate: "f16" (the published kernel's
    half FMA chain; module default GRID_ACCUMULATE) or "f32" (fp32 sum rounded once)."""
    accumulate = accumulate or GRID_ACCUMULATE
    if accumulate == "f16":
        return _HashGridF16.apply(geom, table, x)
    assert accumulate == "f32"
    tab = table.half().float()
    outs = []
    for l in range(geom.n_levels):
        idx, w = _grid_cells(geom, l, x)
        feat = torch.zeros(x.shape[0], 2, dtype=torch.float32)
        for corner in range(4):
            feat = feat + w[corner][:, None] * tab[geom.offset[l] + idx[corner]]
        outs.append(feat)
    return torch.cat(outs, dim=1).half()


def mlp_forward(w1, w2, w3, x_h, n_out, accumulate="f32"):
    """w1 [64,32], w2 [64,64], w3 [P,64] (fp16 values), x_h [B,32] fp16.  fp16 activations.
    accumulate="f32" (default, what the MFMA path does): each layer's dot products in fp32,
    rounded to fp16 once.  accumulate="f16": a MODEL of FullyFusedMLP's half accumulator
    fragments (wmma m16n16k16 with `fragment<accumulator, ..., __half>`): the running sum is
    rounded to half after every 16-deep k-step, products inside a step summed in fp32 — the
    tensor core's internal order is not published, so this only brackets the effect (the
    texel-flip rate between the two modes is reported by tests/test_parity_report.py).
    Returns [B, n_out] fp16."""
    f = lambda t: t.half().float()

    def layer(a_h, w, relu):
        a, wf = f(a_h), f(w)
        if accumulate == "f32":
            o = a @ wf.t()
        else:
            assert accumulate == "f16"
            o = torch.zeros(a.shape[0], wf.shape[0])
            for k in range(0, a.shape[1], 16):
                o = (o + a[:, k:k + 16] @ wf[:, k:k + 16].t()).half().float()
        return (torch.relu(o) if relu else o).half()

    h = layer(x_h, w1, True)
    h = layer(h, w2, True)
    return layer(h, w3, False)[:, :n_out]


# ---- A MODEL of tiny-cuda-nn's half-precision BACKWARD arithmetic (reported by
# tests/test_parity_report.py beside the kernel's own error; PARITY UNPINNED like the rest of this file).
# What north_star calls "the reference CUDA path" back-propagates in half as well:
#   * FullyFusedMLP's backward keeps dL/d(activation) in the network precision (__half) between
#     layers (tiny-cuda-nn src/fully_fused_mlp.cu, kernel_mlp_fused_backward: output fragments of
#     type __half, ReLU applied through the forward activations);
#   * GridEncoding's backward (include/tiny-cuda-nn/encodings/grid.h, kernel_grid_backward) accumulates
#     the table gradient with `atomicAdd((__half2*)&grid_gradient[index], {(T)((float)grad[f] * weight),
#     ...})` whenever N_FEATURES_PER_LEVEL > 1 and the gradient type is __half (grad_t =
#     conditional_t<N_FEATURES_PER_LEVEL == 1, float, T>): every contribution is rounded to half AND
#     the running sum lives in half, in the (nondeterministic) order the atomics arrive.
# The model below is CONSERVATIVE (it rounds less than the real path): one sample per UNIQUE texel
# instead of four per hit, fp32 dot products inside a layer, contributions added in slot order.
def mlp_backward_half(w1, w2, w3, h1_h, h2_h, dout, n_out):
    """dL/d(network input) [B,32] fp16 for dL/d(output) `dout` [B,n_out] fp32: rounded to half at the
    network boundary and after every layer (fp32 dot products), ReLU through the forward activations."""
    f = lambda t: t.half().float()
    g = f(dout)                                            # dL/doutput enters in the network precision
    g = f((g @ f(w3)[:n_out]) * (h2_h.float() > 0))
    g = f((g @ f(w2)) * (h1_h.float() > 0))
    return (g @ f(w1)).half()


def hashgrid_backward_half_atomics(geom, dfeat_h, x):
    """Table gradient [n_entries, 2] as float32 VALUES of a half-precision accumulation: per level and
    corner c, table[idx_c] += half(float(dfeat) * w_c), the running sum rounded to half after every add
    (np.add.at on a float16 array: unbuffered, sequential, in slot order)."""
    out = np.zeros((geom.offset[-1], 2), np.float16)
    d = dfeat_h.float()
    for l in range(geom.n_levels):
        idx, w = _grid_cells(geom, l, x)
        lvl = np.zeros((geom.size[l], 2), np.float16)
        for corner in range(4):
            contrib = (d[:, 2 * l:2 * l + 2] * w[corner][:, None]).half().numpy()
            np.add.at(lvl, idx[corner].numpy(), contrib)
        out[geom.offset[l]:geom.offset[l + 1]] = lvl
    return torch.from_numpy(out.astype(np.float32))


PRIMES = (1, 2654435761, 805459861)


class GridGeometryND:
    """tiny-cuda-nn GridEncoding geometry for n_dims in {2, 3} (the fp32 "Grid"/"Hash"
    encoder of /root/reference/volsurfs_py/encodings/gridhash.py:23-37: 24 levels,
    2 features, 2^18 entries, base 16, growth 2)."""

    def __init__(self, n_dims=3, n_levels=24, log2_hashmap_size=18, base_resolution=16,
                 per_level_scale=2.0):
        self.n_dims, self.n_levels, self.n_features = n_dims, n_levels, 2
        hashmap = 1 << log2_hashmap_size
        log2_pls = np.float32(math.log2(per_level_scale))
        self.scale, self.res, self.size, self.offset = [], [], [], [0]
        for l in range(n_levels):
            s = np.float32(np.exp2(np.float32(l) * log2_pls)) * np.float32(base_resolution) - np.float32(1.0)
            r = int(np.ceil(s)) + 1
            n = min(r ** n_dims, 1 << 62)
            n = min((n + 7) // 8 * 8, hashmap)
            self.scale.append(float(np.float32(s)))
            self.res.append(r)
            self.size.append(n)
            self.offset.append(self.offset[-1] + n)
        self.n_output_dims = 2 * n_levels

    def index(self, l, c):
        """c: list of n_dims int64 tensors holding uint32 values."""
        M = 0xFFFFFFFF
        res, size = self.res[l], self.size[l]
        # tiny-cuda-nn keeps `stride` in a uint32: at res = 2^16 .. 2^18 (levels 12-14 of the
        # 24-level, growth-2 grid) res^2 wraps to 0, the loop runs on with stride 0 and the
        # level falls back to the dense x + y*res index instead of the hash.  Restated as is.
        stride, idx = 1, torch.zeros_like(c[0])
        for d in range(self.n_dims):
            if stride <= size:
                idx = (idx + c[d] * stride) & M
                stride = (stride * res) & M
        if size < stride:
            idx = torch.zeros_like(c[0])
            for d in range(self.n_dims):
                idx = idx ^ ((c[d] * PRIMES[d]) & M)
        return idx % size


def grid_forward_f32(geom, table, x):
    """table [n_entries, 2] fp32, x [B, n_dims] fp32 in [0,1] -> [B, 2*n_levels] fp32.
    Per level: pos = x*scale + 0.5 (two roundings), cell = floor, D-linear interpolation
    with the corner weight formed as ((w0*w1)*w2) and the corners summed in index order."""
    D = geom.n_dims
    outs = []
    for l in range(geom.n_levels):
        pos = x * np.float32(geom.scale[l]) + np.float32(0.5)
        cell = torch.floor(pos)
        frac = pos - cell
        c = cell.to(torch.int64) & 0xFFFFFFFF
        feat = torch.zeros(x.shape[0], 2, dtype=torch.float32)
        for corner in range(1 << D):
            w = None
            cc = []
            for d in range(D):
                bit = (corner >> d) & 1
                wd = frac[:, d] if bit else 1 - frac[:, d]
                w = wd if w is None else w * wd
                cc.append((c[:, d] + bit) & 0xFFFFFFFF)
            feat = feat + w[:, None] * table[geom.offset[l] + geom.index(l, cc)]
        outs.append(feat)
    return torch.cat(outs, dim=1)


class Encoding(torch.nn.Module):
    """tcnn.Encoding(n_input_dims=2, encoding_config) shaped."""

    def __init__(self, n_input_dims, encoding_config, dtype=None, seed=1337):
        super().__init__()
        assert n_input_dims == 2 and encoding_config["otype"] == "HashGrid"
        self.geom = GridGeometry(encoding_config["n_levels"], encoding_config["n_features_per_level"],
                                 encoding_config["log2_hashmap_size"], encoding_config["base_resolution"],
                                 encoding_config["per_level_scale"])
        g = torch.Generator().manual_seed(seed)
        self.params = torch.nn.Parameter(
            (torch.rand(self.geom.offset[-1], 2, generator=g) * 2 - 1) * 1e-4)
        self.n_output_dims = self.geom.n_output_dims

    def forward(self, x):
        return hashgrid_forward(self.geom, self.params, x.float())


class GridEncoding(torch.nn.Module):
    """tcnn.Encoding(n_input_dims, {"otype": "Grid", "type": "Hash", ...}, dtype=float32)
    shaped (gridhash.py:23-37): fp32 parameters and output."""

    def __init__(self, n_input_dims, encoding_config, dtype=None, seed=1337):
        super().__init__()
        assert encoding_config["otype"] == "Grid" and encoding_config["type"] == "Hash"
        assert encoding_config["n_features_per_level"] == 2
        self.geom = GridGeometryND(n_input_dims, encoding_config["n_levels"],
                                   encoding_config["log2_hashmap_size"],
                                   encoding_config["base_resolution"],
                                   encoding_config["per_level_scale"])
        g = torch.Generator().manual_seed(seed)
        self.params = torch.nn.Parameter(
            (torch.rand(self.geom.offset[-1], 2, generator=g) * 2 - 1) * 1e-4)
        self.n_output_dims = self.geom.n_output_dims

    def forward(self, x):
        return grid_forward_f32(self.geom, self.params, x.float())


class Network(torch.nn.Module):
    """tcnn.Network(n_input_dims, n_output_dims, network_config) shaped."""

    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337):
        super().__init__()
        assert network_config["otype"] == "FullyFusedMLP" and network_config["activation"] == "ReLU"
        assert network_config["n_neurons"] == 64 and network_config["n_hidden_layers"] == 2
        assert n_input_dims == 32
        self.n_output_dims = n_output_dims
        pad = (n_output_dims + 15) // 16 * 16
        g = torch.Generator().manual_seed(seed)

        def xavier(o, i):
            s = math.sqrt(6.0 / (i + o))
            return (torch.rand(o, i, generator=g) * 2 - 1) * s

        self.w1 = torch.nn.Parameter(xavier(64, 32))
        self.w2 = torch.nn.Parameter(xavier(64, 64))
        self.w3 = torch.nn.Parameter(xavier(pad, 64))

    def forward(self, x):
        return mlp_forward(self.w1, self.w2, self.w3, x, self.n_output_dims)
