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


# ---------------------------------------------------------------------------
# Mesh I/O: the baked shells of a run are
# `<run>/meshes_simplified_uvs/<isolevel>.obj` with per-face-corner UVs from xatlas
# (baker.py:123,151); the reference loads them through mvdatasets.utils.mesh.Mesh (absent)
# in filename order (utils/mesh_loaders.py:22-110) and wraps them as TensorMesh
# (volsurfs.py:82-117).
def load_obj(path, device="cuda"):
    """Wavefront OBJ -> TensorMesh.  `v x y z`, `vt u v`, `f a/at[/an] b/bt[/bn] ...`
    (polygons are fan-triangulated, negative indices are relative).  Faces without `vt`
    indices get uv 0."""
    verts, uvs, faces, face_uvs = [], [], [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                verts.append([float(x) for x in line.split()[1:4]])
            elif line.startswith("vt "):
                uvs.append([float(x) for x in line.split()[1:3]])
            elif line.startswith("f "):
                vi, ti = [], []
                for tok in line.split()[1:]:
                    parts = tok.split("/")
                    i = int(parts[0])
                    vi.append(i - 1 if i > 0 else len(verts) + i)
                    if len(parts) > 1 and parts[1]:
                        t = int(parts[1])
                        ti.append(t - 1 if t > 0 else len(uvs) + t)
                    else:
                        ti.append(-1)
                for k in range(1, len(vi) - 1):
                    faces.append([vi[0], vi[k], vi[k + 1]])
                    face_uvs.append([ti[0], ti[k], ti[k + 1]])
    if not verts or not faces:
        raise ValueError(f"{path}: no geometry")
    v = torch.tensor(verts, dtype=torch.float32)
    fa = torch.tensor(faces, dtype=torch.int32)
    uv_tab = torch.tensor(uvs if uvs else [[0.0, 0.0]], dtype=torch.float32)
    ft = torch.tensor(face_uvs, dtype=torch.long)
    fuv = uv_tab[ft.clamp(min=0)]
    fuv[ft < 0] = 0.0
    mesh = TensorMesh(v, fa, fuv, device=device)
    mesh.has_uvs = bool(uvs) and bool((ft >= 0).all())
    return mesh


def save_obj(path, mesh):
    """TensorMesh -> OBJ with one `vt` per face corner (round-trips through load_obj)."""
    v = mesh.vertices.detach().cpu()
    f = mesh.faces.detach().cpu().long()
    fuv = mesh.get_faces_uvs().detach().cpu().reshape(-1, 3, 2)
    with open(path, "w") as out:
        for p in v.tolist():
            out.write("v %.9g %.9g %.9g\n" % tuple(p))
        for t in fuv.reshape(-1, 2).tolist():
            out.write("vt %.9g %.9g\n" % tuple(t))
        for i, tri in enumerate(f.tolist()):
            out.write("f %d/%d %d/%d %d/%d\n" % (tri[0] + 1, 3 * i + 1, tri[1] + 1, 3 * i + 2,
                                                  tri[2] + 1, 3 * i + 3))


def load_meshes_indexed_from_path(meshes_indices, meshes_path, require_uvs=False, return_paths=False,
                                  device="cuda"):
    """utils/mesh_loaders.py:35-110: the .obj files of a directory sorted by the isolevel in
    their name (inner -> outer), optionally a subset by index.  Errors raise (the reference
    prints and exit(1)s)."""
    import os
    if not os.path.exists(meshes_path):
        raise FileNotFoundError(f"mesh path {meshes_path} does not exist")
    names = [n for n in os.listdir(meshes_path) if n.endswith(".obj")]
    names.sort(key=lambda x: float(x[:-4]))
    if not names:
        raise FileNotFoundError(f"no meshes found in {meshes_path}")
    if meshes_indices is not None:
        idx = sorted(int(i) for i in meshes_indices)
        if not idx:
            raise ValueError("no meshes indices set")
        for i in idx:
            if i < 0 or i >= len(names):
                raise IndexError(f"mesh index {i} out of range")
        names = [names[i] for i in idx]
    paths = [os.path.join(meshes_path, n) for n in names]
    meshes = [load_obj(p, device=device) for p in paths]
    if require_uvs:
        for n, m in zip(names, meshes):
            if not m.has_uvs:
                raise ValueError(f"mesh {n} does not have UVs")
    return (meshes, paths) if return_paths else meshes
