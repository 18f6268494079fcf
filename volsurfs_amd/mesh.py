"""Minimal TensorMesh (the shape VolSurfs consumes from mvdatasets:
/root/reference/volsurfs_py/methods/volsurfs.py:82-117, 511) and synthetic
nested shells for tests / bench (datasets are not available offline;
SURVEY.md §8d "Synthetic inputs")."""
import numpy as np
import torch


class TensorMesh:
    """vertices [V,3] f32, faces [F,3] i32, faces_uvs [F,3,2] f32 (per-corner UVs)."""

    def __init__(self, vertices, faces, faces_uvs=None, device="cuda"):
        self.vertices = torch.as_tensor(vertices, dtype=torch.float32).contiguous().to(device)
        self.faces = torch.as_tensor(faces, dtype=torch.int32).contiguous().to(device)
        if faces_uvs is not None:
            faces_uvs = torch.as_tensor(faces_uvs, dtype=torch.float32).contiguous().to(device)
        self.faces_uvs = faces_uvs

    def get_faces_uvs(self):
        return self.faces_uvs


def icosphere(subdiv=2, radius=1.0):
    """Unit icosahedron subdivided `subdiv` times: 20*4^subdiv faces."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t],
                  [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]],
                 np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9],
                  [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2],
                  [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
                  [8, 6, 7], [9, 8, 1]], np.int64)
    for _ in range(subdiv):
        nv = v.shape[0]
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
        e.sort(axis=1)
        key = e[:, 0] * nv + e[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        mid = v[uniq // nv] + v[uniq % nv]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        v = np.concatenate([v, mid], 0)
        F = f.shape[0]
        m01, m12, m20 = nv + inv[:F], nv + inv[F:2 * F], nv + inv[2 * F:]
        f = np.concatenate([np.stack([f[:, 0], m01, m20], 1), np.stack([f[:, 1], m12, m01], 1),
                            np.stack([f[:, 2], m20, m12], 1), np.stack([m01, m12, m20], 1)], 0)
    return (v * radius).astype(np.float32), f.astype(np.int32)


def octahedral_uv(p):
    """Unit vectors [.,3] -> [0,1]^2 octahedral parameterisation."""
    p = p / np.abs(p).sum(-1, keepdims=True)
    u, v = p[..., 0].copy(), p[..., 1].copy()
    neg = p[..., 2] < 0
    uu = (1 - np.abs(v)) * np.where(u >= 0, 1.0, -1.0)
    vv = (1 - np.abs(u)) * np.where(v >= 0, 1.0, -1.0)
    u = np.where(neg, uu, u)
    v = np.where(neg, vv, v)
    return np.stack([u * 0.5 + 0.5, v * 0.5 + 0.5], -1)


def nested_shells(K=5, subdiv=6, r0=0.30, dr=0.01, noise=0.0, seed=0, device="cuda"):
    """K nested (optionally noisy) icospheres, inner -> outer, with per-corner
    octahedral UVs (radii 0.30 + 0.01 k)."""
    rng = np.random.default_rng(seed)
    base_v, f = icosphere(subdiv, 1.0)
    meshes = []
    for k in range(K):
        r = r0 + dr * k
        v = base_v.astype(np.float64)
        if noise > 0:
            bump = 1.0 + noise * np.sin(7.0 * v[:, :1] + k) * np.cos(5.0 * v[:, 1:2]) \
                + 0.1 * noise * rng.standard_normal((v.shape[0], 1))
            v = v * bump
        vv = (v * r).astype(np.float32)
        # per-corner uvs from the face's vertex directions; corners of one face are
        # pulled towards the face centroid's hemisphere so that no face straddles
        # the octahedral fold with wildly different uvs
        uv = octahedral_uv(base_v.astype(np.float64))[f].astype(np.float32)  # [F,3,2]
        meshes.append(TensorMesh(vv, f, uv, device=device))
    return meshes
