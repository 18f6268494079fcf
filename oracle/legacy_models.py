"""Oracle: the legacy appearance models and the background field, restated on torch-CPU
from the reference's Python (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

  MLP          /root/reference/volsurfs_py/models/mlp.py:8-69      Linear(+bias) / exact GELU
  RGB          /root/reference/volsurfs_py/models/rgb.py:104-149   cat(pos enc, dir enc, normals, geom) -> MLP -> sigmoid
  ColorSH      /root/reference/volsurfs_py/models/color_sh.py:82-143
  NerfHash     /root/reference/volsurfs_py/models/nerfhash.py:58-91
  GridHash     /root/reference/volsurfs_py/encodings/gridhash.py:57-85 (window, bb scaling, concat)
  SH basis     /root/reference/volsurfs_py/encodings/sphericalharmonics.py:84-153 — pinned
               by tests/golden/sh_encoder.npz (generated from the reference's SHEncoder).

Pinning: MLP / SHEncoder / FrequencyEncoder are the reference's own torch op sequences
(torch.nn.Linear, torch.nn.GELU, the hardcoded polynomials), checked against fixtures made by
importing the reference classes (tools/make_golden.py gen_legacy).  The hash grid itself is
tiny-cuda-nn's (absent): oracle/tcnn_like.py, PARITY UNPINNED.
"""
import torch

from . import tcnn_like
from .neural_texture import sh_basis_values


def mlp_forward(layers, x):
    """layers: list of (weight [out,in], bias [out] or None); GELU between, last linear."""
    for i, (w, b) in enumerate(layers):
        x = torch.nn.functional.linear(x, w, b)
        if i + 1 < len(layers):
            x = torch.nn.functional.gelu(x)
    return x


def gridhash_encode(geom, table, points, bb_sides, window=None, concat=True):
    if bb_sides is not None:
        points = points * (1 / (bb_sides / 2))
        points = (points + 1) / 2
    enc = tcnn_like.grid_forward_f32(geom, table, points)
    if window is not None:
        enc = enc * window.repeat_interleave(2)
    return torch.cat([enc, points], 1) if concat else enc


def nerfhash_forward(geom, table, mlp_fd, mlp_rgb, points, dirs, bb_sides):
    feats = gridhash_encode(geom, table, points, bb_sides)
    fd = mlp_forward(mlp_fd, feats)
    density, feat_rgb = fd[:, 0:1], fd[:, 1:65]
    dirs_enc = sh_basis_values(dirs, 3)
    rgb = mlp_forward(mlp_rgb, torch.cat([torch.nn.functional.gelu(feat_rgb), dirs_enc], 1))
    return torch.sigmoid(rgb), torch.nn.functional.softplus(density)


def rgb_forward(geom, table, mlp, points, dirs, normals, bb_sides, sh_deg):
    parts = [gridhash_encode(geom, table, points, bb_sides), sh_basis_values(dirs, sh_deg)]
    if normals is not None:
        parts.append(normals)
    return torch.sigmoid(mlp_forward(mlp, torch.cat(parts, 1)))
