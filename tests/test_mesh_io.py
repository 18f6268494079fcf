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
