"""OBJ mesh I/O (host logic, CPU)."""
import os

import pytest
import torch

from volsurfs_amd.mesh import (TensorMesh, icosphere, load_meshes_indexed_from_path, load_obj,
                               octahedral_uv, save_obj)


def _mesh(r):
    import numpy as np
    v, f = icosphere(1, r)
    fuv = octahedral_uv(v / np.linalg.norm(v, axis=1, keepdims=True))[f]
    return TensorMesh(v, f, fuv, device="cpu")


def test_obj_round_trip_and_directory_order(tmp_path):
    for name, r in [("0.01", 0.31), ("-0.01", 0.29), ("0.0", 0.30)]:
        save_obj(os.path.join(tmp_path, name + ".obj"), _mesh(r))
    meshes, paths = load_meshes_indexed_from_path(None, str(tmp_path), require_uvs=True,
                                                  return_paths=True, device="cpu")
    assert [os.path.basename(p) for p in paths] == ["-0.01.obj", "0.0.obj", "0.01.obj"]   # inner -> outer
    radii = [m.vertices.norm(dim=1).mean().item() for m in meshes]
    assert radii == sorted(radii)
    ref = _mesh(0.30)
    got = meshes[1]
    assert torch.allclose(got.vertices, ref.vertices, atol=1e-7) and torch.equal(got.faces, ref.faces)
    assert torch.allclose(got.get_faces_uvs().reshape(-1, 3, 2), ref.get_faces_uvs().reshape(-1, 3, 2), atol=1e-7)
    sub = load_meshes_indexed_from_path(["2", "0"], str(tmp_path), device="cpu")
    assert len(sub) == 2 and sub[0].vertices.norm(dim=1).mean() < sub[1].vertices.norm(dim=1).mean()
    with pytest.raises(IndexError):
        load_meshes_indexed_from_path([5], str(tmp_path), device="cpu")


def test_obj_polygons_negative_indices_and_missing_uvs(tmp_path):
    p = os.path.join(tmp_path, "q.obj")
    open(p, "w").write("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
                       "f -4/-4 -3/-3 -2/-2 -1/-1\nf 1 2 3\n")
    m = load_obj(p, device="cpu")
    assert m.faces.tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 2]]          # quad fan + a uv-less triangle
    assert not m.has_uvs
    fu = m.get_faces_uvs().reshape(-1, 3, 2)
    assert fu[1].tolist() == [[0.0, 0.0], [1.0, 1.0], [0.0, 1.0]] and fu[2].abs().sum() == 0
    with pytest.raises(ValueError):
        load_meshes_indexed_from_path(None, str(tmp_path), require_uvs=True, device="cpu")


@pytest.mark.parametrize("binary", [True, False])
def test_ply_round_trip_mixed_directory_and_polygons(tmp_path, binary):
    """utils/mesh_loaders.py:22-31: `.ply` files are accepted next to `.obj` and sorted by isolevel."""
    import struct
    from volsurfs_amd.mesh import load_mesh, load_ply, save_ply
    save_ply(os.path.join(tmp_path, "0.01.ply"), _mesh(0.31), binary=binary)
    save_obj(os.path.join(tmp_path, "-0.01.obj"), _mesh(0.29))
    save_ply(os.path.join(tmp_path, "0.0.ply"), _mesh(0.30), binary=not binary)
    meshes, paths = load_meshes_indexed_from_path(None, str(tmp_path), require_uvs=True,
                                                  return_paths=True, device="cpu")
    assert [os.path.basename(p) for p in paths] == ["-0.01.obj", "0.0.ply", "0.01.ply"]
    ref = _mesh(0.31)
    got = meshes[2]
    assert torch.allclose(got.vertices, ref.vertices, atol=1e-7) and torch.equal(got.faces, ref.faces)
    assert torch.allclose(got.get_faces_uvs().reshape(-1, 3, 2), ref.get_faces_uvs().reshape(-1, 3, 2), atol=1e-7)
    # a binary big-endian file with a quad + a triangle (slow path), per-vertex s/t, extra properties
    p = os.path.join(tmp_path, "q.ply")
    hdr = ("ply\nformat binary_big_endian 1.0\ncomment t\nelement vertex 4\nproperty double x\n"
           "property double y\nproperty double z\nproperty uchar red\nproperty float s\nproperty float t\n"
           "element face 2\nproperty list uchar uint vertex_indices\nend_header\n").encode()
    body = b""
    for i, (x, y) in enumerate([(0, 0), (1, 0), (1, 1), (0, 1)]):
        body += struct.pack(">dddBff", x, y, 0.0, 7, x * 0.5, y * 0.25)
    body += struct.pack(">BIIII", 4, 0, 1, 2, 3) + struct.pack(">BIII", 3, 0, 1, 2)
    open(p, "wb").write(hdr + body)
    m = load_ply(p, device="cpu")
    assert m.faces.tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 2]] and m.has_uvs
    assert m.get_faces_uvs()[1].tolist() == [[0.0, 0.0], [0.5, 0.25], [0.0, 0.25]]
    with pytest.raises(ValueError):
        load_mesh(os.path.join(tmp_path, "x.stl"), device="cpu")
    # no uvs at all -> has_uvs False
    open(p, "w").write("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\n"
                       "property float z\nelement face 1\nproperty list uchar int vertex_index\n"
                       "end_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n")
    m = load_ply(p, device="cpu")
    assert not m.has_uvs and m.faces.tolist() == [[0, 1, 2]]


def test_run_meshes_are_copied_at_start_and_reloaded_on_resume(tmp_path):
    """methods/volsurfs.py:75-117."""
    from volsurfs_amd.mesh import prepare_run_meshes, save_ply
    src, ckpt = tmp_path / "meshes_simplified_uvs", tmp_path / "run"
    src.mkdir()
    ckpt.mkdir()
    save_obj(os.path.join(src, "0.01.obj"), _mesh(0.31))
    save_ply(os.path.join(src, "-0.01.ply"), _mesh(0.29))
    save_obj(os.path.join(src, "0.0.obj"), _mesh(0.30))
    first = prepare_run_meshes(str(src), ["0", "2"], str(ckpt), start_iter_nr=0, device="cpu")
    assert sorted(os.listdir(ckpt / "meshes")) == ["0.ply", "1.obj"]          # inner -> outer, renumbered
    shutil_free = [m.vertices.norm(dim=1).mean().item() for m in first]
    assert abs(shutil_free[0] - 0.29) < 0.01 and abs(shutil_free[1] - 0.31) < 0.01
    resumed = prepare_run_meshes("/nonexistent", None, str(ckpt), start_iter_nr=100, device="cpu")
    assert len(resumed) == 2
    for a, b in zip(first, resumed):
        assert torch.allclose(a.vertices, b.vertices, atol=1e-7) and torch.equal(a.faces, b.faces)


def test_chart_atlas_fragments_the_parameterisation_deterministically():
    """mesh.chart_atlas (the non-ideal scene of bench.py): every face lands in ONE tile of the G x G
    atlas, uvs stay in [0, 1], neighbouring faces of different charts end up far apart, and the
    same seed gives the same atlas."""
    import numpy as np
    from volsurfs_amd.mesh import chart_atlas, icosphere, octahedral_uv
    v, f = icosphere(3, 1.0)
    uv = octahedral_uv(v.astype(np.float64))[f]
    G = 4
    a = chart_atlas(uv, G, seed=3)
    b = chart_atlas(uv, G, seed=3)
    c = chart_atlas(uv, G, seed=4)
    assert a.shape == uv.shape and a.dtype == np.float32
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert a.min() >= 0.0 and a.max() <= 1.0
    tile = np.floor(np.clip(a.mean(1), 0, 1 - 1e-6) * G).astype(int)
    inside = np.floor(np.clip(a, 0, 1 - 1e-6) * G).astype(int) == tile[:, None, :]
    assert inside.all(axis=(1, 2)).mean() > 0.9        # a face stays inside its tile (fold faces excepted)
    # the packing is a permutation of the charts: every tile that holds faces holds ONE source cell
    src = np.floor(np.clip(uv.mean(1), 0, 1 - 1e-9) * G).astype(int)
    pairs = {(tuple(t_), tuple(s_)) for t_, s_ in zip(tile.tolist(), src.tolist())}
    assert len({p[0] for p in pairs}) == len(pairs)
