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


def chart_atlas(uv, charts, seed=0, gutter=0.02):
    """Re-pack a continuous per-corner parameterisation [F,3,2] into charts x charts separate
    charts, the way an atlas packer (xatlas in the reference's baker, baker.py:151) leaves real
    shells: a face belongs to the cell of the charts x charts grid that holds its centroid; every
    cell is moved to a random tile of the atlas, with a random flip / transposition and a gutter.
    Neighbouring pixels that cross a chart border then touch texels far apart."""
    rng = np.random.default_rng(seed)
    G = int(charts)
    cen = uv.mean(1)                                             # [F,2]
    cell = np.clip(np.floor(cen * G), 0, G - 1).astype(np.int64)
    cid = cell[:, 1] * G + cell[:, 0]
    local = uv * G - cell[:, None, :]                            # ~[0,1] inside the cell (corners may poke out)
    local = np.clip(local, -gutter / (1 - 2 * gutter) * 0.9, 1 + gutter / (1 - 2 * gutter) * 0.9)
    perm = rng.permutation(G * G)
    flip_u, flip_v, swap = (rng.integers(0, 2, G * G).astype(bool) for _ in range(3))
    lu, lv = local[..., 0].copy(), local[..., 1].copy()
    lu = np.where(flip_u[cid][:, None], 1 - lu, lu)
    lv = np.where(flip_v[cid][:, None], 1 - lv, lv)
    lu, lv = np.where(swap[cid][:, None], lv, lu), np.where(swap[cid][:, None], lu, lv)
    tile = perm[cid]
    tu, tv = (tile % G)[:, None], (tile // G)[:, None]
    out = np.stack([(tu + gutter + lu * (1 - 2 * gutter)) / G, (tv + gutter + lv * (1 - 2 * gutter)) / G], -1)
    return np.clip(out, 0.0, 1.0).astype(np.float32)


def nested_shells(K=5, subdiv=6, r0=0.30, dr=0.01, noise=0.0, seed=0, device="cuda", atlas_charts=0):
    """K nested (optionally noisy) icospheres, inner -> outer, with per-corner
    octahedral UVs (radii 0.30 + 0.01 k).  atlas_charts = G > 0 cuts the
    parameterisation of every shell into G x G charts packed at random (chart_atlas)."""
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
        if atlas_charts:
            uv = chart_atlas(uv.astype(np.float64), atlas_charts, seed=seed + 17 * k)
        meshes.append(TensorMesh(vv, f, uv, device=device))
    return meshes


def stress_shells(K=5, subdiv=6, r0=0.30, dr=0.01, lobes=4, amplitude=0.3, warp=0.55, noise=0.05,
                  charts=16, seed=0, device="cuda"):
    """K nested shells that stress what real baked shells stress (the reference's shells are marching-
    cubes meshes simplified to 2.5 % of their faces with xatlas atlases: baker.py:123,151 — hundreds of
    charts, non-convex, uneven triangles):

      * NON-CONVEX: the radius is modulated by `lobes` lobes of relative depth `amplitude` around an axis
        that is tilted against the bench camera's view axis, so a ray crosses a shell up to `lobes` + 2
        times (closest hit matters beyond front / back) and many rays graze lobe flanks;
      * UNEVEN TRIANGLES: the unit sphere is warped towards a pole before the displacement
        (v -> normalise(v + warp * axis)): triangle areas spread by ((1 + warp) / (1 - warp))^2 ~ 12x
        at warp = 0.55, on top of the stretching along the lobe flanks;
      * FRAGMENTED ATLAS: the parameterisation is cut into charts x charts (= 256) randomly packed,
        flipped and transposed charts (chart_atlas) — neighbouring pixels touch texels far apart;
      * the shell-to-shell noise of `nested_shells(noise=...)` on top.
    """
    rng = np.random.default_rng(seed)
    base_v, f = icosphere(subdiv, 1.0)
    base = base_v.astype(np.float64)
    pole = np.array([0.45, 0.80, 0.40])
    pole /= np.linalg.norm(pole)
    vw = base + warp * pole
    vw /= np.linalg.norm(vw, axis=1, keepdims=True)
    axis = np.array([1.0, 0.3, 0.2])                 # lobes run around this axis: across the view axis (z)
    axis /= np.linalg.norm(axis)
    e1 = np.cross(axis, [0.0, 0.0, 1.0])
    e1 /= np.linalg.norm(e1)
    e2 = np.cross(axis, e1)
    h = vw @ axis
    phi = np.arctan2(vw @ e2, vw @ e1)
    lobe = 1.0 + amplitude * np.sin(lobes * phi) * (1.0 - h * h) + 0.35 * amplitude * np.sin(3.0 * np.pi * h)
    uv0 = octahedral_uv(base)[f]
    meshes = []
    for k in range(K):
        r = r0 + dr * k
        bump = lobe[:, None]
        if noise > 0:
            bump = bump * (1.0 + noise * np.sin(7.0 * vw[:, :1] + 0.15 * k) * np.cos(5.0 * vw[:, 1:2])
                           + 0.1 * noise * rng.standard_normal((vw.shape[0], 1)))
        vv = (vw * bump * r).astype(np.float32)
        uv = chart_atlas(uv0.astype(np.float64), charts, seed=seed + 17 * k) if charts else uv0.astype(np.float32)
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


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
              "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4",
              "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def load_ply(path, device="cuda"):
    """Stanford PLY -> TensorMesh (the reference accepts `.ply` next to `.obj`,
    utils/mesh_loaders.py:22-31; marching-cubes shells before the xatlas pass are PLY).
    ascii, binary_little_endian and binary_big_endian; vertex x/y/z (+ optional per-vertex
    s/t or u/v or texture_u/texture_v); face `vertex_indices` lists (polygons are
    fan-triangulated) + optional per-face `texcoord` list of 2*n floats (MeshLab wedge uvs).
    Other elements / properties are skipped."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append((tok[1], int(tok[2]), []))
            elif tok[0] == "property":
                if tok[1] == "list":
                    elements[-1][2].append((tok[4], ("list", _PLY_TYPES[tok[2]], _PLY_TYPES[tok[3]])))
                else:
                    elements[-1][2].append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        end = ">" if fmt == "binary_big_endian" else "<"
        data = {}
        if fmt == "ascii":
            toks = iter(f.read().split())
        for name, count, props in elements:
            has_list = any(isinstance(t, tuple) for _, t in props)
            if not has_list:
                if fmt == "ascii":
                    arr = np.array([[float(next(toks)) for _ in props] for _ in range(count)],
                                   np.float64).reshape(count, len(props))
                    data[name] = {pn: arr[:, i] for i, (pn, _) in enumerate(props)}
                else:
                    dt = np.dtype([(pn, end + t) for pn, t in props])
                    arr = np.frombuffer(f.read(dt.itemsize * count), dt, count)
                    data[name] = {pn: arr[pn] for pn, _ in props}
                continue
            rows = {pn: [] for pn, _ in props}
            if fmt != "ascii" and count:
                # fast path: all-triangle faces have fixed-size records (3 indices, 6 wedge uvs)
                guess = {"vertex_indices": 3, "vertex_index": 3, "texcoord": 6}
                fields = []
                for pn, t in props:
                    if isinstance(t, tuple):
                        fields += [(pn + "#n", end + t[1]), (pn, end + t[2], (guess.get(pn, 3),))]
                    else:
                        fields.append((pn, end + t))
                dt = np.dtype(fields)
                pos = f.tell()
                buf = f.read(dt.itemsize * count)
                ok = len(buf) == dt.itemsize * count
                if ok:
                    arr = np.frombuffer(buf, dt, count)
                    ok = all((arr[pn + "#n"] == guess.get(pn, 3)).all() for pn, t in props
                             if isinstance(t, tuple))
                if ok:
                    data[name] = {pn: arr[pn] for pn, _ in props}
                    continue
                f.seek(pos)
            for _ in range(count):
                for pn, t in props:
                    if isinstance(t, tuple):
                        if fmt == "ascii":
                            n = int(next(toks))
                            vals = [float(next(toks)) for _ in range(n)]
                        else:
                            n = int(np.frombuffer(f.read(np.dtype(t[1]).itemsize), end + t[1])[0])
                            vals = np.frombuffer(f.read(np.dtype(t[2]).itemsize * n), end + t[2]).tolist()
                        rows[pn].append(vals)
                    elif fmt == "ascii":
                        rows[pn].append(float(next(toks)))
                    else:
                        rows[pn].append(np.frombuffer(f.read(np.dtype(t).itemsize), end + t)[0])
            data[name] = rows
    if "vertex" not in data or "face" not in data:
        raise ValueError(f"{path}: no geometry")
    vd = data["vertex"]
    verts = np.stack([np.asarray(vd[k], np.float64) for k in ("x", "y", "z")], 1).astype(np.float32)
    fkey = next((k for k in ("vertex_indices", "vertex_index") if k in data["face"]), None)
    if fkey is None:
        raise ValueError(f"{path}: faces carry no vertex_indices")
    vuv = None
    for a, b in (("s", "t"), ("u", "v"), ("texture_u", "texture_v")):
        if a in vd and b in vd:
            vuv = np.stack([np.asarray(vd[a], np.float32), np.asarray(vd[b], np.float32)], 1)
    wedge = data["face"].get("texcoord")
    faces, fuvs = [], []
    for i, poly in enumerate(data["face"][fkey]):
        poly = [int(x) for x in poly]
        for k in range(1, len(poly) - 1):
            tri = [poly[0], poly[k], poly[k + 1]]
            faces.append(tri)
            if wedge is not None and len(wedge[i]) >= 2 * len(poly):
                w = np.asarray(wedge[i], np.float32).reshape(-1, 2)
                fuvs.append([w[0], w[k], w[k + 1]])
            elif vuv is not None:
                fuvs.append(vuv[tri])
    if not len(verts) or not faces:
        raise ValueError(f"{path}: no geometry")
    has_uvs = len(fuvs) == len(faces)
    fuv = np.asarray(fuvs, np.float32).reshape(-1, 3, 2) if has_uvs else np.zeros((len(faces), 3, 2), np.float32)
    mesh = TensorMesh(verts, np.asarray(faces, np.int32), fuv, device=device)
    mesh.has_uvs = has_uvs
    return mesh


def save_ply(path, mesh, binary=True):
    """TensorMesh -> PLY with MeshLab-style per-face `texcoord` wedge uvs (round-trips
    through load_ply)."""
    v = mesh.vertices.detach().cpu().numpy().astype("<f4")
    f = mesh.faces.detach().cpu().numpy().astype("<i4")
    fuv = mesh.get_faces_uvs()
    fuv = None if fuv is None else fuv.detach().cpu().numpy().astype("<f4").reshape(-1, 6)
    hdr = ["ply", "format %s 1.0" % ("binary_little_endian" if binary else "ascii"),
           "comment volsurfs_amd", f"element vertex {len(v)}", "property float x", "property float y",
           "property float z", f"element face {len(f)}", "property list uchar int vertex_indices"]
    if fuv is not None:
        hdr.append("property list uchar float texcoord")
    hdr.append("end_header")
    with open(path, "wb") as out:
        out.write(("\n".join(hdr) + "\n").encode("ascii"))
        if binary:
            out.write(v.tobytes())
            for i in range(len(f)):
                out.write(b"\x03" + f[i].tobytes())
                if fuv is not None:
                    out.write(b"\x06" + fuv[i].tobytes())
        else:
            for p_ in v.tolist():
                out.write(("%.9g %.9g %.9g\n" % tuple(p_)).encode())
            for i in range(len(f)):
                line = "3 %d %d %d" % tuple(f[i].tolist())
                if fuv is not None:
                    line += " 6 " + " ".join("%.9g" % x for x in fuv[i].tolist())
                out.write((line + "\n").encode())


def load_mesh(path, device="cuda"):
    """.obj or .ply by extension (utils/mesh_loaders.py:26)."""
    if path.endswith(".ply"):
        return load_ply(path, device=device)
    if path.endswith(".obj"):
        return load_obj(path, device=device)
    raise ValueError(f"{path}: unsupported mesh format (expected .obj or .ply)")


def load_meshes_indexed_from_path(meshes_indices, meshes_path, require_uvs=False, return_paths=False,
                                  device="cuda"):
    """utils/mesh_loaders.py:35-110: the .obj / .ply files of a directory sorted by the isolevel in
    their name (inner -> outer), optionally a subset by index.  Errors raise (the reference
    prints and exit(1)s)."""
    import os
    if not os.path.exists(meshes_path):
        raise FileNotFoundError(f"mesh path {meshes_path} does not exist")
    names = [n for n in os.listdir(meshes_path) if n.endswith(".obj") or n.endswith(".ply")]
    names.sort(key=lambda x: float(x[:-4]))                       # mesh_loaders.py:22-31
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
    meshes = [load_mesh(p, device=device) for p in paths]
    if require_uvs:
        for n, m in zip(names, meshes):
            if not m.has_uvs:
                raise ValueError(f"mesh {n} does not have UVs")
    return (meshes, paths) if return_paths else meshes


def prepare_run_meshes(meshes_path, meshes_indices, load_checkpoints_path, start_iter_nr=0,
                       require_uvs=True, device="cuda"):
    """methods/volsurfs.py:75-117: at the first iteration the shells are loaded from
    `meshes_path` (optionally a subset by index), and copied to `<checkpoints>/meshes/<nr>.<ext>`
    so that a run is self-contained; a resumed run (`start_iter_nr > 0`) loads them from there.
    Returns the TensorMesh list, inner -> outer."""
    import os
    import shutil
    local = os.path.join(load_checkpoints_path, "meshes")
    if start_iter_nr == 0:
        meshes, paths = load_meshes_indexed_from_path(meshes_indices, meshes_path, require_uvs=require_uvs,
                                                      return_paths=True, device=device)
        os.makedirs(local, exist_ok=True)
        for nr, path in enumerate(paths):
            shutil.copy(path, os.path.join(local, f"{nr}.{os.path.basename(path).split('.')[-1]}"))
        return meshes
    return load_meshes_indexed_from_path(None, local, require_uvs=require_uvs, device=device)
