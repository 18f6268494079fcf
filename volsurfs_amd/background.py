"""Background of unbounded scenes and bounding primitives — host mirror of
/root/reference/volsurfs_py/utils/background.py:31-141 (render_contracted_bg),
utils/sampling.py:62-86 and utils/raycasting.py:4-36, on the packed HIP ops.
The radiance model itself (`NerfHash`) is any callable
`model_bg(points [S,3], dirs [S,3], iter_nr) -> (rgb [S,3], density [S,1])`."""
import torch

from . import _lib
from .volsurfs import (CumprodOneMinusAlphaToTransmittanceFunc, IntegrateWithWeights3DFunc,
                       RaySampler, VolumeRendering)


def _intersect(kind, size, rays_o, rays_d):
    N = rays_o.shape[0]
    rays_o = _lib.check_f32(rays_o.contiguous(), N, 3)
    rays_d = _lib.check_f32(rays_d.contiguous(), N, 3)
    dev = rays_o.device
    hit = torch.empty(N, dtype=torch.uint8, device=dev)
    t_near, t_far = torch.empty(N, device=dev), torch.empty(N, device=dev)
    p_near, p_far = torch.empty(N, 3, device=dev), torch.empty(N, 3, device=dev)
    _lib.call("vsa_intersect_primitive", rays_o, rays_d, N, kind, float(size), hit, t_near, t_far,
              p_near, p_far, _lib.stream_ptr())
    return hit.bool(), t_near, t_far, p_near, p_far


class BoundingBox:
    """Axis-aligned cube of side `side` centred at the origin (the reference builds
    it with side 2*scene_radius, utils/volsurfs_utils.py:234-272; the class itself
    lives in the absent mvdatasets).  `intersect` = vsa_intersect_primitive (kind 0)."""

    def __init__(self, side=1.0):
        self.half = 0.5 * float(side)

    def get_radius(self):
        return self.half

    @torch.no_grad()
    def intersect(self, rays_o, rays_d):
        return _intersect(0, self.half, rays_o, rays_d)


class BoundingSphere:
    def __init__(self, radius=0.5):
        self.radius = float(radius)

    def get_radius(self):
        return self.radius

    @torch.no_grad()
    def intersect(self, rays_o, rays_d):
        return _intersect(1, self.radius, rays_o, rays_d)


@torch.no_grad()
def intersect_bounding_primitive(bounding_primitive, rays_o, rays_d):
    """utils/raycasting.py:4-36: same dict keys."""
    is_hit, t_near, t_far, p_near, p_far = bounding_primitive.intersect(rays_o, rays_d)
    return {"rays_o": rays_o, "rays_d": rays_d, "nr_rays": rays_o.shape[0], "points_near": p_near,
            "points_far": p_far, "t_near": t_near.unsqueeze(-1), "t_far": t_far.unsqueeze(-1),
            "is_hit": is_hit}


def render_contracted_bg(model_bg, raycast, nr_samples_bg, jitter_samples=False, iter_nr=None,
                         render_expected_depth=False, render_median_depth=True):
    """background.py:31-141: 32 inverse-depth samples behind t_far, scene contraction,
    NeRF weights, packed composite; returns the same dict."""
    pack = RaySampler.compute_samples_bg(raycast["rays_o"], raycast["rays_d"], raycast["t_far"], 100.0,
                                         nr_samples_bg, jitter_samples)          # sampling.py:62-86
    cpack = RaySampler.contract_samples(pack)                                   # background.py:72
    rgb, density = model_bg(cpack.samples_3d, cpack.samples_dirs, iter_nr)      # :86-90
    alpha = 1.0 - torch.exp(-density.view(-1, 1) * cpack.samples_dt)            # :93-95
    T, _ = CumprodOneMinusAlphaToTransmittanceFunc.apply(cpack, (1 - alpha) + 1e-6)   # :99-104
    weights = alpha * T
    pred_rgb = IntegrateWithWeights3DFunc.apply(cpack, rgb, weights)            # :109-111
    expected = median = None
    if render_expected_depth:
        expected = VolumeRendering.integrate_with_weights_1d(pack, pack.samples_z, weights.detach())
    if render_median_depth:
        median = VolumeRendering.median_depth_over_rays(pack, weights.detach(), 0.5)   # :126-128
    return {"pred_rgb": pred_rgb, "expected_depth": expected, "median_depth": median}
