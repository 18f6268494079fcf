"""Oracle: packed (ragged) per-ray sample ops of the background path, restated
from the reference's CUDA kernels with their serial per-ray loops.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows kernels/volsurfs/VolumeRenderingGPU.cuh:28-78 (cumprod), :80-177
(integrate 1d/3d), :305-361 (cumsum), :364-409 (median depth), :896-1033
(backwards); RaySamplerGPU.cuh:39-139 (compute_samples_bg), :528-592
(contract_samples); RaySamplesPackedGPU.cuh:14-88 (update_dt); pcg32.h (PCG32,
published generator).  These kernels cannot be compiled or imported here (CUDA
only), and the reference has no tests: the only known-answer material is the
worked example at VolumeRenderingGPU.cuh:60-62 -> PARITY UNPINNED beyond it.
The autograd glue around them (volume_rendering_funcs.py:91-241) IS pinned:
tests/golden/packed_glue.npz comes from the reference's own Function classes run
on top of this module (tools/make_golden.py).
Sums are accumulated serially in fp32 like the kernels (nvcc may contract a*b+c
into FMA, which this restatement does not: differences are 1 ulp).
"""
import numpy as np

f32 = np.float32


def _rays(start_end):
    for r in range(start_end.shape[0]):
        a, b = int(start_end[r, 0]), int(start_end[r, 1])
        yield r, a, b


def cumprod_fwd(start_end, a):
    a = np.asarray(a, f32).reshape(-1)
    T = np.zeros_like(a)
    bgT = np.ones(start_end.shape[0], f32)
    for r, i0, i1 in _rays(start_end):
        n = i1 - i0
        if n == 0:
            continue
        t = f32(1.0)
        for i in range(n):
            T[i0 + i] = t
            if i < n - 1:
                t = f32(t * a[i0 + i])
        bgT[r] = t
    return T, bgT


def cumsum(start_end, v, inverse):
    v = np.asarray(v, f32).reshape(-1)
    out = np.zeros_like(v)
    for r, i0, i1 in _rays(start_end):
        seg = v[i0:i1][::-1] if inverse else v[i0:i1]
        c = np.cumsum(seg, dtype=f32)
        out[i0:i1] = c[::-1] if inverse else c
    return out


def integrate_fwd(start_end, values, weights):
    values = np.asarray(values, f32)
    w = np.asarray(weights, f32).reshape(-1)
    D = values.shape[1]
    out = np.zeros((start_end.shape[0], D), f32)
    for r, i0, i1 in _rays(start_end):
        if i1 > i0:
            out[r] = np.cumsum(w[i0:i1, None] * values[i0:i1], axis=0, dtype=f32)[-1]
    return out


def integrate_bwd(start_end, g_out, values, weights, bug_compat=False):
    values = np.asarray(values, f32)
    w = np.asarray(weights, f32).reshape(-1)
    g_values = np.zeros_like(values)
    g_w = np.zeros_like(w)
    for r, i0, i1 in _rays(start_end):
        g = np.asarray(g_out[r], f32)
        g_values[i0:i1] = g[None] * w[i0:i1, None]
        v = values[i0:i1]
        if bug_compat and v.shape[1] == 3:
            v = v[:, [0, 1, 1]]                      # VolumeRenderingGPU.cuh:1021
        acc = np.zeros(i1 - i0, f32)
        for d in range(values.shape[1]):
            acc = acc + g[d] * v[:, d]
        g_w[i0:i1] = acc
    return g_values, g_w


def cumprod_bwd(start_end, g_bgT, a, bgT, cumsum_lv):
    a = np.asarray(a, f32).reshape(-1)
    g = np.zeros_like(a)
    for r, i0, i1 in _rays(start_end):
        n = i1 - i0
        for i in range(n - 1):
            d = max(a[i0 + i], f32(1e-6))
            x = f32(cumsum_lv[i0 + i + 1] / d)
            g[i0 + i] = f32(x + f32(f32(g_bgT[r] * bgT[r]) / d))
    return g


def median_depth(start_end, z, w, thr, fallback_compat=False):
    z = np.asarray(z, f32).reshape(-1)
    w = np.asarray(w, f32).reshape(-1)
    out = np.zeros(start_end.shape[0], f32)
    for r, i0, i1 in _rays(start_end):
        n = i1 - i0
        if n == 0:
            continue
        c = np.cumsum(w[i0:i1], dtype=f32)
        hit = np.nonzero(c >= f32(thr))[0]
        if hit.size:
            out[r] = z[i0 + hit[0]]
        else:
            out[r] = z[n - 1] if fallback_compat else z[i1 - 1]   # :407 reads without idx_start
    return out


def update_dt(start_end, ray_max_dt, ray_exit, z, is_background):
    z = np.asarray(z, f32).reshape(-1)
    dt = np.full_like(z, -1.0)
    for r, i0, i1 in _rays(start_end):
        n = i1 - i0
        if n == 0:
            continue
        m = f32(ray_max_dt[r])
        d = np.clip(z[i0 + 1:i1] - z[i0:i1 - 1], f32(0), m)
        dt[i0:i1 - 1] = d
        dt[i1 - 1] = f32(1e10) if is_background else np.clip(f32(ray_exit[r]) - z[i1 - 1], f32(0), m)
    return dt


class Pcg32:
    MULT = 0x5851f42d4c957f2d
    M64 = (1 << 64) - 1

    def __init__(self, state=0x853c49e6748fea9b, inc=0xda3e39cb94b95bdb):
        self.state, self.inc = state, inc

    def next_uint(self):
        old = self.state
        self.state = (old * self.MULT + self.inc) & self.M64
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF

    def next_float(self):
        u = (self.next_uint() >> 9) | 0x3f800000
        return f32(np.array([u], np.uint32).view(f32)[0] - f32(1.0))

    def advance(self, delta=1 << 32):
        cur_mult, cur_plus, acc_mult, acc_plus = self.MULT, self.inc, 1, 0
        delta &= self.M64
        while delta > 0:
            if delta & 1:
                acc_mult = (acc_mult * cur_mult) & self.M64
                acc_plus = (acc_plus * cur_mult + cur_plus) & self.M64
            cur_plus = ((cur_mult + 1) * cur_plus) & self.M64
            cur_mult = (cur_mult * cur_mult) & self.M64
            delta >>= 1
        self.state = (acc_mult * self.state + acc_plus) & self.M64

    def copy(self):
        return Pcg32(self.state, self.inc)


def sample_bg(rays_o, rays_d, t_start, t_far, n, jitter=False, rng=None):
    rays_o, rays_d = np.asarray(rays_o, f32), np.asarray(rays_d, f32)
    N = rays_o.shape[0]
    z = np.zeros((N, n), f32)
    max_dt = np.zeros(N, f32)
    eps = f32(1e-6)
    delta_s = f32(1.0 / (n - 1))
    for r in range(N):
        s, t_prec, md = f32(1.0), f32(t_start[r]), f32(0.0)
        g = rng.copy() if jitter else None
        for i in range(n):
            t = f32(1.0 / float(f32(s + eps)) - 1.0)
            t = f32(t + f32(t_start[r]))
            t = min(max(t, f32(t_start[r])), f32(t_far))
            if jitter and i != 0 and i != n - 1:
                g.advance(r)
                u = g.next_float()
                t = f32(t_prec + f32(u * f32(t - t_prec)))
            z[r, i] = t
            s = f32(s - delta_s)
            md = max(md, f32(t - t_prec))
            t_prec = t
        max_dt[r] = md
    s3d = (rays_o[:, None, :] + z[..., None] * rays_d[:, None, :]).astype(f32)
    dirs = np.broadcast_to(rays_d[:, None, :], (N, n, 3)).copy()
    se = np.stack([np.arange(N) * n, np.arange(N) * n + n], 1).astype(np.int32)
    return {"samples_z": z.reshape(-1), "samples_3d": s3d.reshape(-1, 3), "samples_dirs": dirs.reshape(-1, 3),
            "ray_max_dt": max_dt, "ray_start_end_idx": se}


def contract(ray_o, start_end, s3d, sz):
    s3d, sz = np.asarray(s3d, f32).copy(), np.asarray(sz, f32).reshape(-1).copy()
    for r, i0, i1 in _rays(start_end):
        p = s3d[i0:i1]
        q = p * f32(2.0)
        norm = np.sqrt((q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]) + q[:, 2] * q[:, 2]).astype(f32)
        m = norm > 1.0
        nz = np.where(m, norm, f32(1.0))
        factor = (f32(2.0) - f32(1.0) / nz).astype(f32)
        pc = ((factor[:, None] * p) / nz[:, None]).astype(f32)
        e = pc - np.asarray(ray_o[r], f32)[None]
        zc = np.sqrt((e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2]).astype(f32)
        s3d[i0:i1] = np.where(m[:, None], pc, p)
        sz[i0:i1] = np.where(m, zc, sz[i0:i1])
    return s3d, sz


# ---- ops of the sibling methods, same serial per-ray loops
def sum_over_rays(start_end, values):
    """VolumeRenderingGPU.cuh:246-303."""
    v = np.asarray(values, f32)
    per_ray = np.zeros((start_end.shape[0], v.shape[1]), f32)
    per_sample = np.zeros_like(v)
    for r, i0, i1 in _rays(start_end):
        acc = np.zeros(v.shape[1], f32)
        for i in range(i0, i1):
            acc = (acc + v[i]).astype(f32)
        if i1 > i0:
            per_ray[r] = acc
            per_sample[i0:i1] = acc
    return per_ray, per_sample


def sum_over_rays_bwd(start_end, g_ray, g_sample):
    """VolumeRenderingGPU.cuh:1036-1077."""
    g = np.zeros_like(np.asarray(g_sample, f32))
    for r, i0, i1 in _rays(start_end):
        g[i0:i1] = (np.asarray(g_ray, f32)[r][None, :] + np.asarray(g_sample, f32)[i0:i1]).astype(f32)
    return g


def _sigmoid_ref(x):
    """:179-183  float res = 1.0 / (1.0 + exp(-x)) with a float argument."""
    with np.errstate(over="ignore"):          # exp overflows to inf like expf does: 1/(1+inf) = 0
        return f32(1.0 / (1.0 + np.float64(np.exp(f32(-x), dtype=f32))))


def sdf2alpha(start_end, dt, sdf, beta):
    """VolumeRenderingGPU.cuh:185-244; double literals promote the marked intermediates."""
    dt, sdf, beta = (np.asarray(a, f32).reshape(-1) for a in (dt, sdf, beta))
    alpha = np.zeros_like(sdf)
    for r, i0, i1 in _rays(start_end):
        for i in range(i0, i1 - 1):
            d, prev, nxt = dt[i], sdf[i], sdf[i + 1]
            mid = f32(np.float64(f32(prev + nxt)) * 0.5)
            cosv = f32(np.float64(f32(nxt - prev)) / (np.float64(d) + 1e-6))
            cosv = f32(min(max(cosv, f32(-1e3)), f32(0.0)))
            half = np.float64(f32(cosv * d)) * 0.5
            prev_e, next_e = f32(np.float64(mid) - half), f32(np.float64(mid) + half)
            pc, nc = _sigmoid_ref(f32(prev_e * beta[i])), _sigmoid_ref(f32(next_e * beta[i]))
            alpha[i] = f32((np.float64(f32(pc - nc)) + 1e-6) / (np.float64(pc) + 1e-6))
    return alpha


def compute_cdf(start_end, weights):
    """VolumeRenderingGPU.cuh:412-460."""
    w = np.asarray(weights, f32).reshape(-1)
    cdf = np.zeros_like(w)
    for r, i0, i1 in _rays(start_end):
        if i1 - i0 < 2:
            continue
        c, tot = f32(0.0), f32(0.0)
        for i in range(i0, i1):
            cdf[i] = c
            tot = f32(tot + w[i])
            c = f32(c + w[i])
        if abs(np.float64(tot) - 1.0) < 1e-3 and abs(np.float64(cdf[i1 - 1]) - 1.0) > 1e-3:
            cdf[i1 - 1] = f32(1.0)
    return cdf


def sample_fg(rays_o, rays_d, t_entry, t_exit, min_dist, min_n, max_n, jitter=False, rng=None):
    """RaySamplerGPU.cuh:141-270 followed by the compaction of RaySamplesPacked.cu:188-273:
    returns the compacted z / 3d / dirs / start_end / ray_max_dt."""
    rays_o, rays_d = np.asarray(rays_o, f32), np.asarray(rays_d, f32)
    N = rays_o.shape[0]
    zs, se, max_dt = [], np.zeros((N, 2), np.int32), np.full(N, -1.0, f32)
    cursor = 0
    for r in range(N):
        t_start, t_end = f32(t_entry[r]), f32(t_exit[r])
        dist = f32(t_end - t_start)
        to_create, step = 0, f32(0.0)
        if dist > 0:
            if dist > f32(min_dist):
                to_create = int(f32(dist / f32(min_dist)))
                to_create = min(max(to_create, 0), max_n)
                step = f32(dist / f32(to_create))
            else:
                to_create, step = 1, dist
        out = []
        if to_create > 0 and to_create >= min_n:
            t = t_start
            if jitter:
                g = rng.copy()
                g.advance(r)
                t = f32(t + f32(step * g.next_float()))
            while t < t_end:
                t = min(max(t, t_start), t_end)
                if len(out) >= to_create:
                    break
                out.append(t)
                t = f32(t + step)
        if len(out) < min_n:
            out = []
        else:
            max_dt[r] = step
        # rays without samples keep the constructor's (-1, -1) (RaySamplesPacked.cu:27-28)
        se[r] = (cursor, cursor + len(out)) if out else (-1, -1)
        cursor += len(out)
        zs.append((r, out))
    z = np.zeros(cursor, f32)
    s3d, dirs = np.zeros((cursor, 3), f32), np.zeros((cursor, 3), f32)
    for r, out in zs:
        for k, t in enumerate(out):
            i = se[r, 0] + k
            z[i] = t
            s3d[i] = (rays_o[r] + f32(t) * rays_d[r]).astype(f32)
            dirs[i] = rays_d[r]
    return {"samples_z": z, "samples_3d": s3d, "samples_dirs": dirs, "ray_start_end_idx": se,
            "ray_max_dt": max_dt}


def _map_range_val(v, i0, i1, o0, o1):
    """VolumeRenderingGPU.cuh:15-21 (all float)."""
    c = max(f32(i0), min(f32(i1), f32(v)))
    if i0 >= i1:
        return f32(o1)
    return f32(f32(o0) + f32(f32(f32(o1) - f32(o0)) / f32(f32(i1) - f32(i0))) * f32(c - f32(i0)))


def importance_sample(rays_o, rays_d, start_end, z, cdf, n_imp, jitter=False, rng=None):
    """VolumeRenderingGPU.cuh:462-678 (+ compaction): every ray with samples gets n_imp new ones."""
    rays_o, rays_d = np.asarray(rays_o, f32), np.asarray(rays_d, f32)
    z, cdf = np.asarray(z, f32).reshape(-1), np.asarray(cdf, f32).reshape(-1)
    out_z, out_se, cursor = [], np.zeros((start_end.shape[0], 2), np.int32), 0
    for r, u0, u1 in _rays(start_end):
        if u1 - u0 == 0:
            out_se[r] = (-1, -1)
            continue
        g = rng.copy() if jitter else None
        dist = f32(1.0 / (n_imp + 1))
        for i in range(n_imp):
            ur = f32(dist + f32(i * dist))
            if jitter:
                g.advance(r)
                rand = g.next_float()
                mov = f32(np.float64(dist) / 2.0)
                ur = f32(ur + _map_range_val(rand, 0.0, 1.0, -mov, mov))
            ur = min(max(ur, f32(0.0 + 1e-6)), f32(1.0 - 1e-6))
            imin, imax = u0, u1 - 1
            while imax >= imin:
                imid = imin + (imax - imin) // 2
                if cdf[imid] > ur:
                    imax = imid
                else:
                    imin = imid
                if imax - imin == 1 or imax == imin:
                    break
            lo = max(imax - 1, 0)
            out_z.append((r, _map_range_val(ur, cdf[lo], cdf[imax], z[lo], z[imax])))
        out_se[r] = (cursor, cursor + n_imp)
        cursor += n_imp
    zz = np.array([t for _, t in out_z], f32)
    s3d = np.stack([(rays_o[r] + f32(t) * rays_d[r]).astype(f32) for r, t in out_z]) if out_z else np.zeros((0, 3), f32)
    return {"samples_z": zz, "samples_3d": s3d, "ray_start_end_idx": out_se}


def uncontract(ray_o, start_end, s3d, sz):
    """RaySamplerGPU.cuh:595-650."""
    s3d, sz = np.asarray(s3d, f32).copy(), np.asarray(sz, f32).reshape(-1).copy()
    for r, i0, i1 in _rays(start_end):
        for i in range(i0, i1):
            p = s3d[i]
            q = (p * f32(2.0)).astype(f32)
            norm = f32(np.sqrt(f32(f32(q[0] * q[0] + q[1] * q[1]) + q[2] * q[2])))
            if norm > 1.0:
                factor = f32(f32(1.0) / f32(f32(2.0) - norm))
                p = ((factor * p).astype(f32) / norm).astype(f32)
                e = (p - np.asarray(ray_o, f32)[r]).astype(f32)
                sz[i] = f32(np.sqrt(f32(f32(e[0] * e[0] + e[1] * e[1]) + e[2] * e[2])))
                s3d[i] = p
    return s3d, sz


def combine_packs(se1, z1, se2, z2, min_dist):
    """VolumeRenderingGPU.cuh:680-895 + compaction: returns (start_end, z, source) where source
    gives (pack, sample index) of every kept sample.  A pack without samples for a ray counts
    as exhausted from the start."""
    z1, z2 = np.asarray(z1, f32).reshape(-1), np.asarray(z2, f32).reshape(-1)
    out_se, out_z, src, cursor = np.full((se1.shape[0], 2), -1, np.int32), [], [], 0
    for r in range(se1.shape[0]):
        a0, na = int(se1[r, 0]), int(se1[r, 1] - se1[r, 0])
        b0, nb = int(se2[r, 0]), int(se2[r, 1] - se2[r, 0])
        if na == 0 and nb == 0:
            continue
        ca = cb = 0
        fa, fb = na == 0, nb == 0
        prec, written = f32(0.0), 0
        for _ in range(na + nb):
            if fa and fb:
                break
            za = f32(1e10) if fa else z1[a0 + ca]
            zb = f32(1e10) if fb else z2[b0 + cb]
            take_a = za < zb
            z = za if take_a else zb
            if not (f32(z - prec) < f32(min_dist)):
                out_z.append(z)
                src.append((0, a0 + ca) if take_a else (1, b0 + cb))
                prec = z
                written += 1
            if take_a:
                if ca + 1 >= na:
                    fa = True
                else:
                    ca += 1
            else:
                if cb + 1 >= nb:
                    fb = True
                else:
                    cb += 1
        if written:
            out_se[r] = (cursor, cursor + written)
        cursor += written
    return out_se, np.array(out_z, f32), src
