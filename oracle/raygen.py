"""CPU restatement of csrc/raygen.hip (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED: the reference takes its rays from mvdatasets (`get_camera_rays`,
methods/base_method.py:389-394; `TensorReel.get_next_rays_batch`, trainer.py:176-190), an
empty submodule in the reference checkout (.gitmodules:10-13).  There is no reference source,
test or golden vector for this step; what is restated here is this library's own pinhole
definition, in fp32 with the kernel's operation order."""
import numpy as np

from .packed import Pcg32

f32 = np.float32


def pinhole_ray(c2w, kinv, x, y):
    c2w, kinv = np.asarray(c2w, f32).reshape(3, 4), np.asarray(kinv, f32).reshape(3, 3)
    x, y = f32(x), f32(y)
    dc = [f32(f32(f32(kinv[i, 0] * x) + f32(kinv[i, 1] * y)) + kinv[i, 2]) for i in range(3)]
    d = [f32(f32(f32(c2w[i, 0] * dc[0]) + f32(c2w[i, 1] * dc[1])) + f32(c2w[i, 2] * dc[2])) for i in range(3)]
    n = np.sqrt(f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2])))
    return c2w[:, 3].copy(), np.array([f32(v / n) for v in d], f32)


def camera_rays(c2w, kinv, H, W, R=1, jitter=False, rng=None):
    n = H * W * R
    o, d, p = np.zeros((n, 3), f32), np.zeros((n, 3), f32), np.zeros((n, 2), f32)
    for i in range(n):
        pixel = i // R
        row, col = pixel // W, pixel % W
        jx = jy = f32(0.5)
        if jitter:
            g = rng.copy()
            g.advance(2 * i)
            jx, jy = g.next_float(), g.next_float()
        x, y = f32(f32(col) + jx), f32(f32(row) + jy)
        o[i], d[i] = pinhole_ray(c2w, kinv, x, y)
        p[i] = (x, y)
    return o, d, p


def reel_batch(c2w_all, kinv_all, rgbs, masks, B, R=1, jitter=False, rng=None):
    C, H, W = rgbs.shape[:3]
    cam = np.zeros(B, np.int32)
    o, d, p = np.zeros((B * R, 3), f32), np.zeros((B * R, 3), f32), np.zeros((B * R, 2), f32)
    gt = np.zeros((B, 3), f32)
    gm = np.zeros((B, 1), f32) if masks is not None else None
    for b in range(B):
        g = rng.copy()
        g.advance(b * (3 + (2 * R if jitter else 0)))
        c = min(int(f32(g.next_float() * f32(C))), C - 1)
        col = min(int(f32(g.next_float() * f32(W))), W - 1)
        row = min(int(f32(g.next_float() * f32(H))), H - 1)
        cam[b] = c
        gt[b] = rgbs[c, row, col]
        if gm is not None:
            gm[b, 0] = masks[c, row, col]
        for s in range(R):
            jx = jy = f32(0.5)
            if jitter:
                jx, jy = g.next_float(), g.next_float()
            x, y = f32(f32(col) + jx), f32(f32(row) + jy)
            o[b * R + s], d[b * R + s] = pinhole_ray(c2w_all[c], kinv_all[c], x, y)
            p[b * R + s] = (x, y)
    return cam, o, d, gt, gm, p


def intersect_primitive(rays_o, rays_d, kind, size):
    """Restatement of vsa_intersect_primitive (utils/raycasting.py:4-36 ->
    mvdatasets BoundingBox / BoundingSphere .intersect, absent: PARITY UNPINNED — this library's
    definition of the slab / quadratic test, fp32, in the kernel's operation order).
    kind 0: origin-centred cube of half side `size`; kind 1: sphere of radius `size`.
    Returns (is_hit, t_near, t_far, p_near, p_far)."""
    o, d = np.asarray(rays_o, f32), np.asarray(rays_d, f32)
    size = f32(size)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        if kind == 0:
            inv = (f32(1.0) / d).astype(f32)
            a = ((-size - o) * inv).astype(f32)
            b = ((size - o) * inv).astype(f32)
            tn = np.full(o.shape[0], -np.inf, f32)
            tf = np.full(o.shape[0], np.inf, f32)
            for i in range(3):                       # fmaxf / fminf drop NaNs (0 * inf of a parallel ray)
                tn = np.fmax(tn, np.fmin(a[:, i], b[:, i]))
                tf = np.fmin(tf, np.fmax(a[:, i], b[:, i]))
            hit = (tn <= tf) & (tf > 0)
        else:
            def dot(u, v):
                return ((u[:, 0] * v[:, 0] + u[:, 1] * v[:, 1]).astype(f32) + u[:, 2] * v[:, 2]).astype(f32)
            a = dot(d, d)
            b = (f32(2.0) * dot(o, d)).astype(f32)
            c = (dot(o, o) - size * size).astype(f32)
            disc = ((b * b).astype(f32) - ((f32(4.0) * a).astype(f32) * c).astype(f32)).astype(f32)
            sq = np.sqrt(np.fmax(disc, f32(0))).astype(f32)
            tn = ((-b - sq).astype(f32) / (f32(2.0) * a).astype(f32)).astype(f32)
            tf = ((-b + sq).astype(f32) / (f32(2.0) * a).astype(f32)).astype(f32)
            hit = (disc >= 0) & (tf > 0)
        tn = np.where(hit, np.fmax(tn, f32(0)), f32(0)).astype(f32)
        tf = np.where(hit, tf, f32(0)).astype(f32)
        pn = (o + (tn[:, None] * d).astype(f32)).astype(f32)
        pf = (o + (tf[:, None] * d).astype(f32)).astype(f32)
    return hit, tn, tf, pn, pf
