"""Inference renderers with the reference's class names and call shapes
(volsurfs_py/renderers/{base_renderer,mesh_renderer,volsurfs_renderer}.py).

* BaseRenderer.render(camera, nr_rays_per_pixel) — one-shot full frame: device ray generation
  (jittered when supersampling), render_rays, supersample mean, numpy at the very end
  (base_renderer.py:41-101).
* MeshRenderer — ONE mesh with a baked SH-coefficient texture: closest hit -> uv -> bilinear
  texture fetch -> fp16 SH eval -> sigmoid, and the reference's buffer shading
  (mesh_renderer.py:58-201).  The reference reads mesh + texture through mvdatasets
  (Mesh / TensorTexture, absent): here the mesh is an OBJ (volsurfs_amd.mesh.load_obj) and the
  texture a float array [R, R, 4 * nr_coeffs] of SH coefficients (`.npy`) or an 8-bit image
  expanded to +-sh_range; the fetch uses the texel convention of NeuralTexture.forward
  (models/neural_texture.py:107-138), the one the textures were baked with.
* VolsurfsRenderer — an empty stub in the reference (volsurfs_renderer.py:1-9); here the K-shell
  deploy renderer on VolSurfs.render_baked()."""
import json
import os
from abc import ABC, abstractmethod

import numpy as np
import torch

from .camera import get_camera_rays
from .mesh import load_obj
from .models import sh_eval
from .raytrace import RayTracer


class BaseRenderer(ABC):
    def __init__(self, profiler=None):
        self.profiler = profiler
        self.renders_options = {}
        self.active_shader = "None"
        self.active_render_mode = "None"

    @abstractmethod
    def render_rays(self, rays_o, rays_d, verbose=False) -> dict:
        ...

    def _section(self, name, start):
        if self.profiler is not None:
            (self.profiler.start if start else self.profiler.end)(name)

    @torch.no_grad()
    def render(self, camera, nr_rays_per_pixel=1, verbose=False) -> dict:
        """base_renderer.py:41-101 -> {render_mode: {key: np.ndarray [H*W, C]}}."""
        self._section("ray_gen", True)
        rays_o, rays_d, _ = get_camera_rays(camera, nr_rays_per_pixel=nr_rays_per_pixel,
                                            jitter_pixels=nr_rays_per_pixel > 1, device="cuda")
        self._section("ray_gen", False)
        self._section("render_frame", True)
        res = self.render_rays(rays_o=rays_o, rays_d=rays_d, verbose=verbose)
        out = {}
        for mode, renders in res["renders"].items():
            out[mode] = {}
            for key, r in renders.items():
                if nr_rays_per_pixel > 1:                                   # average supersampling
                    r = r.reshape(r.shape[0] // nr_rays_per_pixel, nr_rays_per_pixel, -1).float().mean(dim=1)
                out[mode][key] = r.cpu().numpy()
        self._section("render_frame", False)
        return out


class TensorTexture:
    """texture [R, R, C] (row = first index) -> values at uv [M, 2], bilinear between the four
    texel centres around the sample, with the texel addressing of NeuralTexture.forward
    (rotate by 90 degrees, align to texel centres; csrc/nt_common.h nt_footprint).  Samples in
    the half-texel border clamp to the edge texel."""

    def __init__(self, texture, lerp=True, device="cuda"):
        t = torch.as_tensor(texture)
        if t.dim() != 3 or t.shape[0] != t.shape[1]:
            raise ValueError("texture must be [R, R, C]")
        self.texture = t.to(device=device, dtype=torch.float32).contiguous()
        self.lerp = lerp

    def __call__(self, uv):
        R = self.texture.shape[0]
        a, b = uv[:, 0] * R, uv[:, 1] * R
        ap, bp = R - b, a
        fl_x, fl_y = torch.floor(ap - 0.5), torch.floor(bp - 0.5)
        fx, fy = ap - (fl_x + 0.5), bp - (fl_y + 0.5)
        i0, j0 = fl_x.long(), fl_y.long()
        if not self.lerp:
            i = (i0 + (fx >= 0.5).long()).clamp(0, R - 1)
            j = (j0 + (fy >= 0.5).long()).clamp(0, R - 1)
            return self.texture[j, i]
        ia, ib = i0.clamp(0, R - 1), (i0 + 1).clamp(0, R - 1)
        ja, jb = j0.clamp(0, R - 1), (j0 + 1).clamp(0, R - 1)
        fx, fy = fx[:, None], fy[:, None]
        return (self.texture[ja, ia] * (1 - fx) * (1 - fy) + self.texture[ja, ib] * fx * (1 - fy)
                + self.texture[jb, ia] * (1 - fx) * fy + self.texture[jb, ib] * fx * fy)


def _load_texture(path, sh_range):
    if path.endswith(".npy"):
        t = np.load(path)
    else:
        from PIL import Image
        t = np.asarray(Image.open(path))
        if t.ndim == 2:
            t = t[..., None]
    if t.dtype == np.uint8:                       # 8-bit texels -> coefficients in +-sh_range
        t = (t.astype(np.float32) / 255.0 * 2.0 - 1.0) * sh_range
    return t.astype(np.float32)


class MeshRenderer(BaseRenderer):
    def __init__(self, scene_path=None, t_near=1e-3, t_far=100, profiler=None, tensor_mesh=None,
                 texture=None, sh_range=15.0):
        """scene_path: a directory with scene.json = {"meshes": [{"mesh_path": ..., "textures":
        [{"texture_path": ...}]}]} (mesh_renderer.py:27-45); or pass tensor_mesh + texture."""
        super().__init__(profiler=profiler)
        if scene_path is not None:
            with open(os.path.join(scene_path, "scene.json")) as f:
                meta = json.load(f)["meshes"][0]
            tensor_mesh = load_obj(os.path.join(scene_path, meta["mesh_path"]))
            texture = _load_texture(os.path.join(scene_path, meta["textures"][0]["texture_path"]), sh_range)
        if tensor_mesh is None or texture is None:
            raise ValueError("MeshRenderer needs a scene_path or a mesh and a texture")
        self.tensor_mesh = tensor_mesh
        self.tensor_texture = TensorTexture(texture, lerp=True)
        if self.tensor_texture.texture.shape[-1] not in (4, 16, 36, 64):
            raise ValueError("texture channels must be 4 * (deg + 1)^2, deg 0..3")
        self.raytracer = RayTracer([tensor_mesh])
        self.t_near, self.t_far = t_near, t_far
        self.active_render_mode, self.active_shader = "ray_traced", "rgb"
        self.default_bg_color = (255, 255, 255)
        self.bg_color = torch.tensor(self.default_bg_color, dtype=torch.float32, device="cuda") / 255.0

    @torch.no_grad()
    def shade(self, buffers_dict) -> dict:
        """mesh_renderer.py:58-104: buffers -> displayable values (misses show the background)."""
        res, is_hit = {}, buffers_dict["is_hit"]
        for name, buf in buffers_dict.items():
            if "normals" in name or "view_dirs" in name:
                v = (buf + 1) * 0.5
                v[~is_hit] = self.bg_color
                res[name] = v
            if "is_hit" in name:
                res[name] = is_hit.float().unsqueeze(-1)
            if "uv" in name:
                v = torch.cat((buf, torch.zeros_like(buf[:, :1])), dim=-1)
                v[~is_hit] = self.bg_color
                res[name] = v
            if "rgb" in name:
                v = buf.clamp(0, 1)
                v[~is_hit] = self.bg_color
                res[name] = v
            if "alpha" in name:
                v = buf.clamp(0, 1)
                v[~is_hit] = 0.0
                res[name] = v
        return res

    @torch.no_grad()
    def render_rays(self, rays_o, rays_d, verbose=False) -> dict:
        """mesh_renderer.py:106-201."""
        N, dev = rays_o.shape[0], rays_o.device
        hits = torch.zeros(N, dtype=torch.bool, device=dev)
        normals, uvs = torch.zeros(N, 3, device=dev), torch.zeros(N, 2, device=dev)
        rgb_fg, alpha_fg = torch.zeros(N, 3, device=dev), torch.zeros(N, 1, device=dev)
        hit = self.raytracer.trace(rays_o, rays_d)
        if hit["any_hit"]:
            hits = hit["is_hit"]
            normals[hits] = hit["normals"][hits]
            corner_uvs = self.tensor_mesh.get_faces_uvs()[hit["triangles_id"].clamp(min=0)]     # [N,3,2]
            uv = torch.sum(hit["barycentric"].unsqueeze(-1) * corner_uvs, dim=1)
            uvs[hits] = uv[hits]
            sh = self.tensor_texture(uv[hits])
            nr_coeffs = sh.shape[-1] // 4
            deg = {1: 0, 4: 1, 9: 2, 16: 3}[nr_coeffs]
            raw = sh_eval(sh.view(-1, 4, nr_coeffs).half(), rays_d[hits], degree=deg)        # fp16, as the reference
            rgba = torch.sigmoid(raw).float()
            rgb_fg[hits], alpha_fg[hits] = rgba[:, :3], rgba[:, 3:4]
        buffers = {"is_hit": hits, "normals": normals, "uvs": uvs, "rgb": rgb_fg, "alpha": alpha_fg,
                   "view_dirs": rays_d.clone()}
        return {"renders": {"ray_traced": self.shade(buffers)}}


class VolsurfsRenderer(BaseRenderer):
    """K nested shells from their baked textures (the deploy format): method = a
    volsurfs_amd.methods.VolSurfs on which bake() has run."""

    def __init__(self, method, profiler=None):
        super().__init__(profiler=profiler)
        if getattr(method, "baked", None) is None:
            method.bake()
        self.method = method
        self.active_render_mode, self.active_shader = "ray_traced", "rgb"

    @torch.no_grad()
    def render_rays(self, rays_o, rays_d, verbose=False) -> dict:
        return {"renders": {"ray_traced": self.method.render_baked(rays_o, rays_d)}}
