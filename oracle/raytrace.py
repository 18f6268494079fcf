"""Oracle wrapper: brute-force closest hit (oracle/raytrace_ref.c) and the
derived per-hit attributes raytracelib's `trace` returns
(/root/reference/volsurfs_py/methods/volsurfs.py:496-501).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the
raytracelib source is absent; semantics follow the call-site contract.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def trace_bruteforce(verts, faces, rays_o, rays_d, t_min=0.0):
    """Returns dict(t [N], tri [N] (-1 miss), uv [N,2])."""
    verts = np.ascontiguousarray(verts, np.float32)
    faces = np.ascontiguousarray(faces, np.int32)
    rays_o = np.ascontiguousarray(rays_o, np.float32)
    rays_d = np.ascontiguousarray(rays_d, np.float32)
    n = rays_o.shape[0]
    t = np.zeros(n, np.float32)
    tri = np.zeros(n, np.int32)
    uv = np.zeros((n, 2), np.float32)
    _load().oracle_trace_bruteforce(
        _p(verts, ctypes.c_float), _p(faces, ctypes.c_int32), ctypes.c_int(faces.shape[0]),
        _p(rays_o, ctypes.c_float), _p(rays_d, ctypes.c_float), ctypes.c_int(n),
        ctypes.c_float(t_min), _p(t, ctypes.c_float), _p(tri, ctypes.c_int32),
        _p(uv, ctypes.c_float))
    return {"t": t, "tri": tri, "uv": uv}


def hit_attributes(verts, faces, rays_o, rays_d, hit):
    """The raytracelib-shaped dict (volsurfs.py:481-501)."""
    verts = np.asarray(verts, np.float32)
    faces = np.asarray(faces, np.int64)
    tri = hit["tri"]
    is_hit = tri >= 0
    tt = np.where(is_hit, tri, 0)
    a = verts[faces[tt, 0]]
    e1 = verts[faces[tt, 1]] - a
    e2 = verts[faces[tt, 2]] - a
    cx = e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1]
    cy = e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2]
    cz = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
    ln = np.sqrt((cx * cx + cy * cy) + cz * cz).astype(np.float32)
    inv = np.where(ln > 0, np.float32(1.0) / np.where(ln > 0, ln, 1), 0).astype(np.float32)
    normals = np.stack([cx * inv, cy * inv, cz * inv], -1) * is_hit[:, None]
    t = hit["t"]
    pos = (np.asarray(rays_o, np.float32) + t[:, None] * np.asarray(rays_d, np.float32)) * is_hit[:, None]
    u, v = hit["uv"][:, 0], hit["uv"][:, 1]
    bary = np.stack([(np.float32(1.0) - u) - v, u, v], -1) * is_hit[:, None]
    return {
        "any_hit": bool(is_hit.any()),
        "is_hit": is_hit,
        "triangles_id": tri,
        "depth": t,
        "positions": pos.astype(np.float32),
        "normals": normals.astype(np.float32),
        "barycentric": bary.astype(np.float32),
    }


def count_crossings(verts, faces, rays_o, rays_d, t_min=0.0):
    """Number of triangles of the mesh each ray crosses (Moeller-Trumbore on every pair, numpy;
    for small ray sets): how non-convex a shell is along the camera's rays — tests of the stress
    scene (mesh.stress_shells) assert that closest hit has to choose among > 2 crossings."""
    v = np.asarray(verts, np.float64)
    f = np.asarray(faces, np.int64)
    v0, e1, e2 = v[f[:, 0]], v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]
    out = np.zeros(len(rays_o), np.int64)
    for i in range(len(rays_o)):
        o, d = np.asarray(rays_o[i], np.float64), np.asarray(rays_d[i], np.float64)
        p = np.cross(d, e2)
        det = (e1 * p).sum(1)
        ok = np.abs(det) > 1e-14
        inv = 1.0 / np.where(ok, det, 1.0)
        t = o - v0
        u = (t * p).sum(1) * inv
        q = np.cross(t, e1)
        w = (q * d).sum(1) * inv
        tt = (q * e2).sum(1) * inv
        out[i] = int((ok & (u >= 0) & (w >= 0) & (u + w <= 1) & (tt > t_min)).sum())
    return out
