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


FUSED_COMPOSITE = True      # False: the reference's chain of single ops (tests hold the two together)


class _FusedBgComposite(torch.autograd.Function):
    """background.py:93-111 (alpha from density, transmittance, NeRF weights, weighted sum of the
    sample colours) as ONE launch forward and ONE backward — vsa_packed_composite_fwd / _bwd,
    bit-identical to the chain of single ops below.  Returns (pred_rgb [N,3], weights [S,1])."""

    @staticmethod
    def forward(ctx, pack, rgb, density):
        rgb, density = _lib.check_f32(rgb.contiguous()), _lib.check_f32(density.contiguous())
        S, N = rgb.shape[0], pack.get_nr_rays()
        if density.numel() != S or pack.samples_dt.numel() != S or rgb.shape[1] != 3:
            raise _lib.VolsurfsHipError("fused bg composite: rgb [S,3], density [S,1], pack with S samples")
        pred = torch.empty(N, 3, device=rgb.device)
        weights = torch.empty(S, 1, device=rgb.device)
        _lib.call("vsa_packed_composite_fwd", pack.ray_start_end_idx, density, pack.samples_dt, rgb, pred,
                  weights, N, _lib.stream_ptr())
        ctx.save_for_backward(rgb, density)
        ctx.pack = pack
        ctx.mark_non_differentiable(weights)
        return pred, weights

    @staticmethod
    def backward(ctx, g_pred, _g_weights):
        rgb, density = ctx.saved_tensors
        pack, ctx.pack = ctx.pack, None
        g_rgb, g_density = torch.empty_like(rgb), torch.empty_like(density)
        scratch = torch.empty(2 * rgb.shape[0], device=rgb.device)
        _lib.call("vsa_packed_composite_bwd", pack.ray_start_end_idx, density, pack.samples_dt, rgb,
                  g_pred.contiguous(), g_rgb, g_density, scratch, pack.get_nr_rays(),
                  bool(VolumeRendering.bug_compat), _lib.stream_ptr())
        return None, g_rgb, g_density


def render_contracted_bg(model_bg, raycast, nr_samples_bg, jitter_samples=False, iter_nr=None,
                         render_expected_depth=False, render_median_depth=True):
    """background.py:31-141: 32 inverse-depth samples behind t_far, scene contraction,
    NeRF weights, packed composite; returns the same dict."""
    pack = RaySampler.compute_samples_bg(raycast["rays_o"], raycast["rays_d"], raycast["t_far"], 100.0,
                                         nr_samples_bg, jitter_samples)          # sampling.py:62-86
    cpack = RaySampler.contract_samples(pack)                                   # background.py:72
    rgb, density = model_bg(cpack.samples_3d, cpack.samples_dirs, iter_nr)      # :86-90
    if FUSED_COMPOSITE and rgb.dtype == torch.float32 and rgb.dim() == 2 and rgb.shape[1] == 3:
        pred_rgb, weights = _FusedBgComposite.apply(cpack, rgb, density.view(-1, 1))   # :93-111
    else:
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
