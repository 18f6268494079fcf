"""Inference renderers (reference class names): BaseRenderer.render, MeshRenderer (one mesh +
baked SH texture, renderers/mesh_renderer.py:16-201), VolsurfsRenderer (K shells from baked
textures)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import neural_texture as ONT


def _texture_ref(tex, uv):
    """Bilinear fetch through the oracle's corner / weight functions (neural_texture.py:107-138)."""
    R = tex.shape[0]
    _, w, corners = ONT.texel_corners(uv.clone(), R)
    i = torch.floor(corners[..., 0]).long().clamp(0, R - 1)
    j = torch.floor(corners[..., 1]).long().clamp(0, R - 1)
    return (tex[j, i] * w).sum(1)


@pytest.mark.gpu
def test_tensor_texture_matches_the_oracle_footprint():
    from volsurfs_amd.renderers import TensorTexture
    g = torch.Generator().manual_seed(0)
    tex = torch.randn(32, 32, 16, generator=g)
    uv = torch.rand(2000, 2, generator=g)
    got = TensorTexture(tex)(uv.cuda()).cpu()
    ref = _texture_ref(tex, uv)
    assert torch.allclose(got, ref, atol=1e-5)
    # a texel centre returns that texel: texel (col i, row j) sits at uv = ((j + .5) / R, 1 - (i + .5) / R)
    i, j = 5, 11
    c = torch.tensor([[(j + 0.5) / 32, 1 - (i + 0.5) / 32]])
    assert torch.allclose(TensorTexture(tex)(c.cuda()).cpu()[0], tex[j, i], atol=1e-5)
    assert torch.allclose(TensorTexture(tex, lerp=False)(c.cuda()).cpu()[0], tex[j, i])


def _scene(deg=2, R=64, seed=3):
    from volsurfs_amd.mesh import nested_shells
    mesh = nested_shells(K=1, subdiv=3, r0=0.4)[0]
    g = torch.Generator().manual_seed(seed)
    tex = torch.randn(R, R, 4 * (deg + 1) ** 2, generator=g) * 2.0
    return mesh, tex


@pytest.mark.gpu
def test_mesh_renderer_buffers_and_values():
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.renderers import MeshRenderer
    from oracle.neural_texture import sh_eval, interp_uv
    mesh, tex = _scene()
    r = MeshRenderer(tensor_mesh=mesh, texture=tex)
    o, d = pinhole_rays(48, 48, focal=70.0)
    out = r.render_rays(o, d)["renders"]["ray_traced"]
    assert set(out) == {"is_hit", "normals", "uvs", "rgb", "alpha", "view_dirs"}
    hit = out["is_hit"][:, 0] > 0
    assert 0.2 < hit.float().mean() < 0.9
    white = torch.ones(3, device="cuda")
    assert torch.equal(out["rgb"][~hit], white.expand((~hit).sum(), 3))          # misses: background
    assert (out["alpha"][~hit] == 0).all() and out["uvs"].shape[1] == 3
    assert torch.equal(out["view_dirs"][hit], ((d + 1) * 0.5)[hit])
    # values of the hit pixels: the same chain on the host through the oracle's pieces
    h = r.raytracer.trace(o, d)
    uv = interp_uv(h["barycentric"].cpu(), mesh.get_faces_uvs().cpu(), h["triangles_id"].cpu())[hit.cpu()]
    sh = _texture_ref(tex, uv).view(-1, 4, 9).half()
    ref = torch.sigmoid(sh_eval(sh, d[hit].cpu(), 2).float()).float()
    got = torch.cat([out["rgb"][hit], out["alpha"][hit]], 1).cpu()
    assert (got - ref).abs().max() < 4e-3                                     # fp16 SH eval both sides
    assert torch.allclose(out["normals"][hit], (h["normals"][hit] + 1) * 0.5)


@pytest.mark.gpu
def test_mesh_renderer_from_scene_directory_and_camera(tmp_path):
    from volsurfs_amd.camera import Camera
    from volsurfs_amd.mesh import save_obj
    from volsurfs_amd.renderers import MeshRenderer
    mesh, tex = _scene(deg=1, R=32)
    save_obj(str(tmp_path / "shell.obj"), mesh)
    np.save(str(tmp_path / "coeffs.npy"), tex.numpy())
    with open(tmp_path / "scene.json", "w") as f:
        json.dump({"meshes": [{"mesh_path": "shell.obj", "textures": [{"texture_path": "coeffs.npy"}]}]}, f)
    a = MeshRenderer(str(tmp_path))
    b = MeshRenderer(tensor_mesh=mesh, texture=tex)
    cam = Camera.look_at((0.0, 0.0, -1.5), focal=60.0, height=40, width=56)
    ra, rb = a.render(cam), b.render(cam)
    assert ra["ray_traced"]["rgb"].shape == (40 * 56, 3) and isinstance(ra["ray_traced"]["rgb"], np.ndarray)
    for k in ra["ray_traced"]:
        assert np.allclose(ra["ray_traced"][k], rb["ray_traced"][k], atol=1e-6), k
    ss = b.render(cam, nr_rays_per_pixel=3)                                   # jittered supersampling, averaged
    assert ss["ray_traced"]["rgb"].shape == (40 * 56, 3)
    inner = ra["ray_traced"]["is_hit"][:, 0] > 0
    assert np.abs(ss["ray_traced"]["rgb"][inner] - ra["ray_traced"]["rgb"][inner]).mean() < 0.1
    with pytest.raises(ValueError):
        MeshRenderer()
    # an 8-bit texture is expanded to +-sh_range
    q = (torch.rand(32, 32, 4) * 255).to(torch.uint8)
    np.save(str(tmp_path / "coeffs.npy"), q.numpy())
    c = MeshRenderer(str(tmp_path), sh_range=15.0)
    want = (q.float() / 255 * 2 - 1) * 15.0
    assert torch.allclose(c.tensor_texture.texture.cpu(), want)


@pytest.mark.gpu
def test_volsurfs_renderer_renders_the_baked_shells():
    from volsurfs_amd.camera import Camera
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    from volsurfs_amd.renderers import VolsurfsRenderer
    torch.manual_seed(0)
    m = VolSurfs(nested_shells(K=2, subdiv=3), max_rays=4096, textures_res=(128, 64, 32, 16))
    r = VolsurfsRenderer(m)                                                   # bakes on construction
    cam = Camera.look_at((0.0, 0.2, -1.5), focal=70.0, height=48, width=48)
    img = r.render(cam)["ray_traced"]
    live = m.render_camera(cam)
    assert np.array_equal(img["rgb"], live["rgb"].reshape(-1, 3).cpu().numpy())   # baked == live, bit for bit
