"""Oracle: permutohedral-lattice hash encoding.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference's `PermutoHashEncoder` (/root/reference/volsurfs_py/encodings/permutohash.py:10-99)
wraps `permutohedral_encoding.PermutoEncoding` + `Coarse2Fine` from an un-vendored, unpinned
submodule (s-esposito/permutohedral_encoding, .gitmodules:7-9) whose source is absent from
/root/reference: PARITY UNPINNED.  This restates the published algorithm in numpy fp32:

  * Adams, Baek, Davis, "Fast High-Dimensional Filtering Using the Permutohedral Lattice"
    (Eurographics 2010): elevation of a D-dimensional point onto the hyperplane H_D of
    R^(D+1) with the triangular basis scaled by 1/sqrt((i+1)(i+2)); closest remainder-0
    point by rounding to multiples of D+1; rank of the residuals -> enclosing simplex;
    barycentric weights; vertex keys rem0 + k (- (D+1) where rank > D-k); the lattice hash
    k = (k + key_i) * 2531011.
  * Rosu, Behnke, "PermutoSDF" (CVPR 2023): one lattice per level with its own scale sigma_l
    (position / sigma_l), a random per-level shift added to the position before scaling, one
    hash table of `capacity` feature vectors per level, features = sum of the D+1 vertices'
    vectors weighted by the barycentric coordinates, times the coarse-to-fine window of the
    level; levels concatenated level-major; optionally the (scaled) input points appended.

What the reference's own call sites fix (permutohash.py:26-37, 84-96): capacity 2^18, 24 levels,
2 features, sigma = np.geomspace(1.0, 1e-4, 24), random shift on, points concatenated (scaling 1),
points mapped from the bounding box to [0,1] first, the LAST output channel dropped.
"""
import math

import numpy as np

f32 = np.float32


def scale_factors(scale_list, pos_dim):
    """[n_levels, pos_dim]: 1 / (sigma_l * sqrt((i+1)(i+2)))."""
    out = np.zeros((len(scale_list), pos_dim), np.float64)
    for l, sigma in enumerate(scale_list):
        for i in range(pos_dim):
            out[l, i] = 1.0 / math.sqrt((i + 1) * (i + 2)) / sigma
    return out.astype(f32)


def simplex(x, shift, sf):
    """x [N,D] fp32; shift, sf [D].  Returns rem0 [N,D+1] int, rank [N,D+1] int, bary [N,D+2] fp32."""
    N, D = x.shape
    el = np.zeros((N, D + 1), f32)
    sm = np.zeros(N, f32)
    for i in range(D, 0, -1):
        cf = ((x[:, i - 1] + f32(shift[i - 1])).astype(f32) * f32(sf[i - 1])).astype(f32)
        el[:, i] = (sm - (f32(i) * cf).astype(f32)).astype(f32)
        sm = (sm + cf).astype(f32)
    el[:, 0] = sm
    inv = f32(1.0 / (D + 1))
    v = (el * inv).astype(f32)
    up = (np.ceil(v) * f32(D + 1)).astype(f32)
    down = (np.floor(v) * f32(D + 1)).astype(f32)
    rem0 = np.where((up - el).astype(f32) < (el - down).astype(f32), up, down).astype(np.int64)
    s = rem0.sum(1) // (D + 1)
    rank = np.zeros((N, D + 1), np.int64)
    diff = (el - rem0.astype(f32)).astype(f32)
    for i in range(D):
        for j in range(i + 1, D + 1):
            lt = diff[:, i] < diff[:, j]
            rank[:, i] += lt
            rank[:, j] += ~lt
    rank += s[:, None]
    lo, hi = rank < 0, rank > D
    rank = np.where(lo, rank + D + 1, np.where(hi, rank - (D + 1), rank))
    rem0 = np.where(lo, rem0 + D + 1, np.where(hi, rem0 - (D + 1), rem0))
    bary = np.zeros((N, D + 2), f32)
    rows = np.arange(N)
    for i in range(D + 1):
        delta = ((el[:, i] - rem0[:, i].astype(f32)).astype(f32) * inv).astype(f32)
        a, b = D - rank[:, i], D + 1 - rank[:, i]
        bary[rows, a] = (bary[rows, a] + delta).astype(f32)
        bary[rows, b] = (bary[rows, b] - delta).astype(f32)
    bary[:, 0] = (bary[:, 0] + (f32(1.0) + bary[:, D + 1]).astype(f32)).astype(f32)
    return rem0, rank, bary


def vertex_index(rem0, rank, k, capacity):
    """Entry of simplex vertex k (remainder k) in a level's table."""
    N, D1 = rem0.shape
    D = D1 - 1
    h = np.zeros(N, np.uint64)
    for i in range(D):
        key = rem0[:, i] + k - np.where(rank[:, i] > D - k, D + 1, 0)
        h = (h + (key.astype(np.int64) & 0xFFFFFFFF).astype(np.uint64)) & np.uint64(0xFFFFFFFF)
        h = (h * np.uint64(2531011)) & np.uint64(0xFFFFFFFF)
    return (h % np.uint64(capacity)).astype(np.int64)


def encode(values, x, scale_list, random_shift, window=None):
    """values [L, capacity, 2] fp32; x [N,D] fp32 -> [N, 2L] fp32 (level-major)."""
    L, capacity, F = values.shape
    N, D = x.shape
    sf = scale_factors(scale_list, D)
    out = np.zeros((N, L * F), f32)
    for l in range(L):
        rem0, rank, bary = simplex(x.astype(f32), random_shift[l], sf[l])
        wl = f32(1.0) if window is None else f32(window[l])
        acc = np.zeros((N, F), f32)
        for k in range(D + 1):
            idx = vertex_index(rem0, rank, k, capacity)
            w = (bary[:, k] * wl).astype(f32)
            acc = (acc + (values[l][idx] * w[:, None]).astype(f32)).astype(f32)
        out[:, l * F:(l + 1) * F] = acc
    return out


def encode_backward(g_out, x, scale_list, random_shift, capacity, window=None, F=2):
    """d loss / d values [L, capacity, F] (fp64 accumulation) for d loss / d out = g_out [N, L*F]."""
    N, D = x.shape
    L = len(scale_list)
    sf = scale_factors(scale_list, D)
    g = np.zeros((L, capacity, F), np.float64)
    for l in range(L):
        rem0, rank, bary = simplex(x.astype(f32), random_shift[l], sf[l])
        wl = f32(1.0) if window is None else f32(window[l])
        for k in range(D + 1):
            idx = vertex_index(rem0, rank, k, capacity)
            w = (bary[:, k] * wl).astype(f32)
            np.add.at(g[l], idx, (g_out[:, l * F:(l + 1) * F] * w[:, None]).astype(np.float64))
    return g


def coarse2fine_window(t, nr_levels):
    """permutohedral_encoding.Coarse2Fine(nr_levels)(t): the nerfies cosine-easing window,
    alpha = t * nr_levels, w_i = (1 - cos(pi * clamp(alpha - i, 0, 1))) / 2."""
    alpha = float(t) * nr_levels
    x = np.clip(alpha - np.arange(nr_levels, dtype=np.float64), 0.0, 1.0)
    return (0.5 * (1.0 - np.cos(math.pi * x))).astype(f32)


def permuto_hash_encoder(values, points, random_shift, bb_sides=2.0, window=None,
                         coarsest_scale=1.0, finest_scale=1e-4, concat_points_scaling=1.0,
                         remove_last_element=True):
    """PermutoHashEncoder.__call__ (permutohash.py:68-96): bounding-box normalisation, encoding,
    concatenated points, last channel dropped.  Returns (enc [N, 2L+D(-1)], out_of_bounds [N])."""
    L = values.shape[0]
    scale_list = np.geomspace(coarsest_scale, finest_scale, num=L)
    half = f32(bb_sides) / f32(2)
    oob = (points <= -half).any(1) | (points >= half).any(1)
    p = (points * (f32(1) / half)).astype(f32)
    p = ((p + f32(1)) / f32(2)).astype(f32)
    enc = encode(values, p, scale_list, random_shift, window)
    enc = np.concatenate([enc, (p * f32(concat_points_scaling)).astype(f32)], 1)
    return (enc[:, :-1] if remove_last_element else enc), oob
