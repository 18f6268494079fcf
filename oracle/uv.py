"""Oracle: uv <-> texel helper functions used by NeuralTexture.forward's lerp
branch (/root/reference/volsurfs_py/models/neural_texture.py:107-138).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference imports these from mvdatasets.utils.images (s-esposito/mvdatasets,
branch `volsurfs`, un-vendored submodule: .gitmodules:10-13).  Their source is
absent, so they are restated from their names, the call-site comments ("uv coords
are width, height; res is height, width; flip=True"; "results are non normalized
uv coordinates of the corners") and the requirement that training-time lerp
equals bilinear sampling of the baked texture at texel centres
(neural_texture.py:83-87, 208-223).  PARITY UNPINNED.
"""
import torch


def _wh(res, flip, like):
    r = torch.as_tensor(res).to(like.dtype)
    return r.flip(0) if flip else r


def non_normalize_uv_coord(uv_coords, res, flip=True):
    """[N,2] in [0,1] (u = width axis, v = height axis) -> [N,2] in [0,W]x[0,H]."""
    return uv_coords * _wh(res, flip, uv_coords)


def normalize_uv_coord(uv_coords_nn, res, flip=True):
    return uv_coords_nn / _wh(res, flip, uv_coords_nn)


def non_normalized_uv_coords_to_interp_corners(uv_coords_nn):
    """[N,2] -> [N,4,2]: centres of the 2x2 texels surrounding the point,
    ordered (0,0), (1,0), (0,1), (1,1)."""
    base = torch.floor(uv_coords_nn - 0.5) + 0.5
    offs = torch.tensor([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 1.0]], dtype=uv_coords_nn.dtype)
    return base[:, None, :] + offs[None]


def non_normalized_uv_coords_to_lerp_weights(uv_coords_nn, uv_corners_coords_nn):
    """Bilinear weights [N,4,1] of the four corners."""
    f = uv_coords_nn - uv_corners_coords_nn[:, 0]
    fx, fy = f[:, 0:1], f[:, 1:2]
    w = torch.stack([(1 - fx) * (1 - fy), fx * (1 - fy), (1 - fx) * fy, fx * fy], dim=1)
    return w


def pix_to_texel_center_uv_coord(uv_pix, res, flip=True):
    return (uv_pix.to(torch.float32) + 0.5) / _wh(res, flip, torch.zeros(1))


def uv_coords_to_pix(uv_coords, res, flip=True):
    r = _wh(res, flip, uv_coords)
    return torch.minimum(torch.floor(uv_coords * r), r - 1).long()
