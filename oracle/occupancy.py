"""CPU restatement of the reference's OccupancyGrid and grid-occupied sampler kernels (TEST
INFRASTRUCTURE ONLY): kernels/volsurfs/OccupancyGridGPU.cuh:31-581,
kernels/volsurfs/occ_grid_helpers.h:14-180, kernels/volsurfs/RaySamplerGPU.cuh:275-457.

Scalar fp32 loops, one function per reference kernel, each operation rounded as the CUDA source
rounds it (float / int mixing noted inline).  The reference holds no test or golden vector for
these kernels -> "parity unpinned" beyond this restatement (as for the other packed native ops).
Marching loops are capped at MAX_ITERS like csrc/occupancy.hip (the reference's are unbounded)."""
import numpy as np

from .packed import Pcg32

f32 = np.float32
MAX_ITERS = 1 << 16
EPS = f32(1e-6)


def spread3(w):                               # occ_grid_helpers.h:14-23 (expand_bits)
    w &= 0x1fffff
    w = (w | (w << 32)) & 0x001f00000000ffff
    w = (w | (w << 16)) & 0x001f0000ff0000ff
    w = (w | (w << 8)) & 0x010f00f00f00f00f
    w = (w | (w << 4)) & 0x10c30c30c30c30c3
    w = (w | (w << 2)) & 0x1249249249249249
    return w


def morton3d(x, y, z):                        # :27-33, uint32 truncation before the shifts
    xx, yy, zz = (spread3(int(v)) & 0xffffffff for v in (x, y, z))
    return (xx | (yy << 1) | (zz << 2)) & 0xffffffff


def morton3d_invert(x):                       # :43-51
    x &= 0x49249249
    x = (x | (x >> 2)) & 0xc30c30c3
    x = (x | (x >> 4)) & 0x0f00f00f
    x = (x | (x >> 8)) & 0xff0000ff
    x = (x | (x >> 16)) & 0x0000ffff
    return x


def _to_u32(v):                               # float -> uint32_t as cvt.rzi.u32.f32: saturating, NaN -> 0
    v = float(v)
    if not v > 0.0:
        return 0
    return min(int(v), 0xffffffff)


def pos_to_lin_idx(pos, n, extent):           # :55-78 (returned as a signed int)
    c = []
    for k in range(3):
        v = f32(f32(pos[k]) / f32(extent[k]))
        v = f32(v + f32(0.5))
        c.append(f32(v * f32(n)))
    m = morton3d(_to_u32(c[0]), _to_u32(c[1]), _to_u32(c[2]))
    return m - (1 << 32) if m & 0x80000000 else m


def voxel_ok(v, n):
    return 0 <= v < n * n * n


def lin_idx_to_3d(idx, n, extent, center_grid, center_of_voxel):   # :80-120
    out = []
    voxel = f32(1.0 / n)
    half = f32(voxel / f32(2))
    for k in range(3):
        x = f32(f32(morton3d_invert(idx >> k)) / f32(n))
        if center_grid:
            x = f32(x - f32(0.5))
        if center_of_voxel:
            x = f32(x + half)
        out.append(f32(x * f32(extent[k])))
    return out


def _sign(x):
    return 1 if x > 0 else (-1 if x < 0 else 0)


def distance_to_next_voxel(pos, d, n, extent):                    # :131-180
    if all(abs(f32(v)) < EPS for v in d):
        return f32(1e10)
    t = []
    for k in range(3):
        p = f32(f32(f32(pos[k]) / f32(extent[k])) * f32(n))
        tk = f32(1e10)
        if abs(f32(d[k])) > EPS:
            prime = np.floor(f32(p + f32(f32(1.0) * f32(_sign(d[k])))))
            tk = f32(f32(abs(f32(prime - p)) / f32(n)) * f32(extent[k]))
        t.append(tk)
    return f32(min(min(t[0], t[1]), t[2]) + EPS)


def _at(o, d, t):
    return [f32(f32(o[k]) + f32(f32(t) * f32(d[k]))) for k in range(3)]


def _clamp(x, lo, hi):
    return f32(min(max(f32(x), f32(lo)), f32(hi)))


def grid_points(indices, n, extent, center_of_voxel, jitter=False, rng=None):   # OccupancyGridGPU.cuh:31-119
    out = np.zeros((len(indices), 3), f32)
    for i, v in enumerate(indices):
        p = lin_idx_to_3d(int(v), n, extent, True, center_of_voxel)
        if jitter:
            g = rng.copy()
            g.advance(i * 3)
            for k in range(3):
                voxel = f32(f32(extent[k]) / f32(n))
                half = f32(voxel / f32(2))
                p[k] = f32(p[k] + f32(f32(voxel * g.next_float()) - half))
        out[i] = p
    return out


def update_values(grid, indices, values, decay):                  # :122-151
    for i, v in enumerate(indices):
        grid[v] = max(f32(values[i]), f32(grid[v] * f32(decay)))


def update_occupancy_density(occ, grid, indices, n, extent, thresh, neighbours):   # :153-225
    for v in indices:
        v = int(v)
        empty = True
        if neighbours:
            p = [f32(c * f32(n)) for c in lin_idx_to_3d(v, n, extent, False, False)]
            for a in (-1, 0, 1):
                if p[0] + a < 0 or p[0] + a > n - 1:
                    continue
                for b in (-1, 0, 1):
                    if p[1] + b < 0 or p[1] + b > n - 1:
                        continue
                    for c in (-1, 0, 1):
                        if p[2] + c < 0 or p[2] + c > n - 1:
                            continue
                        nb = morton3d(int(p[0] + a), int(p[1] + b), int(p[2] + c))
                        empty = empty and grid[nb] <= f32(thresh)
        else:
            empty = grid[v] <= f32(thresh)
        occ[v] = not empty


def sdf_weight(sdf, beta, n, extent):                             # :229-312; returns the logistic weight
    s = [f32(f32(e) / f32(n)) for e in extent]
    diag = f32(0)
    for a in range(8):
        for b in range(8):
            dd = [f32(f32(f32((b >> k) & 1) * s[k]) - f32(f32((a >> k) & 1) * s[k])) for k in range(3)]
            diag = max(diag, np.sqrt(f32(f32(f32(dd[0] * dd[0]) + f32(dd[1] * dd[1])) + f32(dd[2] * dd[2]))))
    x = _clamp(f32(abs(f32(sdf)) - f32(diag / f32(2))), 0.0, 1e10)
    ex = _clamp(np.exp(f32(-f32(beta) * x), dtype=f32), -1e6, 1e6)
    opx = f32(f32(1) + ex)
    return f32(f32(f32(beta) * ex) / f32(opx * opx))


def check_occupancy(points, n, extent, grid, occ, roi):           # :376-413
    o, v_ = np.zeros(len(points), bool), np.zeros(len(points), f32)
    for i, p in enumerate(points):
        v = pos_to_lin_idx(p, n, extent)
        if voxel_ok(v, n):
            o[i], v_[i] = bool(roi[v] and occ[v]), grid[v]
    return o, v_


def rays_t_near_t_far(o, d, t0, t1, n, extent, occ, roi):         # :318-374
    near, far = np.zeros(len(o), f32), np.zeros(len(o), f32)
    for i in range(len(o)):
        t_start, t_exit = f32(t0[i]), f32(t1[i])
        t, first = t_start, True
        near[i] = far[i] = t_start
        it = 0
        while t < t_exit and it < MAX_ITERS:
            it += 1
            pos = _at(o[i], d[i], t)
            v = pos_to_lin_idx(pos, n, extent)
            if not voxel_ok(v, n):
                break
            inside = bool(roi[v] and occ[v])
            if inside and first:
                near[i], first = t, False
            t = f32(t + distance_to_next_voxel(pos, d[i], n, extent))
            if inside:
                far[i] = _clamp(t, t_start, t_exit)
    return near, far


def first_sample(o, d, t0, t1, n, extent, occ, roi):              # :505-581
    N = len(o)
    se = np.zeros((N, 2), np.int32)
    s3d, z = np.full((N, 3), -1, f32), np.full(N, -1, f32)
    for i in range(N):
        t, t_exit, it = f32(t0[i]), f32(t1[i]), 0
        while t < t_exit and it < MAX_ITERS:
            it += 1
            pos = _at(o[i], d[i], t)
            v = pos_to_lin_idx(pos, n, extent)
            if not voxel_ok(v, n):
                break
            t = f32(f32(t + distance_to_next_voxel(pos, d[i], n, extent)) + EPS)
            if roi[v] and occ[v]:
                se[i] = (i, i + 1)
                s3d[i], z[i] = pos, t
                break
    return se, s3d, z


def advance_samples(dirs, pts, n, extent, occ, roi):              # :415-503
    out, within = np.array(pts, f32).copy(), np.ones(len(pts), bool)
    for i in range(len(pts)):
        prec_t, t, it = f32(0), f32(0), 0
        while within[i] and it < MAX_ITERS:
            it += 1
            pos = _at(pts[i], dirs[i], t)
            v = pos_to_lin_idx(pos, n, extent)
            if not voxel_ok(v, n):
                within[i] = False
                out[i] = _at(pts[i], dirs[i], prec_t)
            else:
                prec_t = t
                t = f32(f32(t + distance_to_next_voxel(pos, dirs[i], n, extent)) + EPS)
                if roi[v] and occ[v]:
                    out[i] = pos
                    break
    return out, within


def sample_fg_occupied(o, d, t0, t1, min_dist, min_nr, max_nr, jitter, rng, n, extent, occ, roi):
    """RaySamplerGPU.cuh:275-457 before compaction: (start_end [N,2] (-1 = no samples), z and
    positions per ray slot [N, max_nr], ray_max_dt)."""
    N = len(o)
    se = np.full((N, 2), -1, np.int32)
    z, s3d = np.full((N, max_nr), -1, f32), np.full((N, max_nr, 3), -1, f32)
    max_dt = np.full(N, -1, f32)
    for i in range(N):
        t_start, t_exit = f32(t0[i]), f32(t1[i])
        t, step, occupied, it = t_start, f32(0), f32(0), 0
        while t < t_exit and it < MAX_ITERS:
            it += 1
            pos = _at(o[i], d[i], t)
            v = pos_to_lin_idx(pos, n, extent)
            if not voxel_ok(v, n):
                break
            if roi[v] and occ[v]:
                occupied = f32(occupied + step)
            step = distance_to_next_voxel(pos, d[i], n, extent)
            t = f32(t + step)
        occupied = _clamp(occupied, 0.0, f32(t_exit - t_start))
        to_create, spacing = 0, f32(0)
        if occupied > 0:
            if occupied > f32(min_dist):
                to_create = min(max(int(f32(occupied / f32(min_dist))), 0), max_nr)
                spacing = f32(occupied / f32(to_create))
            else:
                to_create, spacing = 1, occupied
        created = 0
        if to_create > 0 and to_create >= min_nr:
            to_next, t, it = f32(0), t_start, 0
            if jitter:
                g = rng.copy()
                g.advance(i)
                to_next = f32(spacing * g.next_float())
            while t < t_exit and it < MAX_ITERS:
                it += 1
                t = _clamp(t, t_start, t_exit)
                pos = _at(o[i], d[i], t)
                if created >= to_create:
                    break
                v = pos_to_lin_idx(pos, n, extent)
                if not voxel_ok(v, n):
                    break
                inside = bool(roi[v] and occ[v])
                if inside and to_next == 0:
                    s3d[i, created], z[i, created] = pos, t
                    created += 1
                    to_next = spacing
                to_voxel = distance_to_next_voxel(pos, d[i], n, extent)
                adv = to_voxel
                if inside:
                    adv = min(to_voxel, to_next)
                    to_next = f32(to_next - adv)
                    if to_next <= EPS:
                        to_next = f32(0)
                t = f32(t + adv)
        if created >= min_nr:
            max_dt[i] = spacing
            se[i] = (i * max_nr, i * max_nr + created)
    return se, z, s3d, max_dt
