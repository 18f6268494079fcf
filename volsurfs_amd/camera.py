"""Pinhole ray generation for the synthetic bench / tests (the reference takes
its rays from mvdatasets.get_camera_rays, base_method.py:389-394, which is not
in the tree; SURVEY §8f row 2 lists a device ray generator as 'next')."""
import torch


def pinhole_rays(H, W, focal, cam_pos=(0.0, 0.0, -1.5), device="cuda"):
    """Row-major pixel-centre rays of a camera at cam_pos looking down +z."""
    ys, xs = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                            torch.arange(W, device=device, dtype=torch.float32), indexing="ij")
    dx = (xs + 0.5 - 0.5 * W) / focal
    dy = (ys + 0.5 - 0.5 * H) / focal
    d = torch.stack([dx, dy, torch.ones_like(dx)], -1).reshape(-1, 3)
    d = torch.nn.functional.normalize(d, dim=-1).contiguous()
    o = torch.tensor(cam_pos, device=device, dtype=torch.float32).expand(H * W, 3).contiguous()
    return o, d


# ---------------------------------------------------------------------------
# device ray generation + the training-ray sampler.  The call shapes are
# the reference's (mvdatasets `get_camera_rays`, base_method.py:389-394;
# `TensorReel.get_next_rays_batch`, trainer.py:176-190); mvdatasets itself is an empty
# submodule in the reference checkout, so the pinhole arithmetic is this library's own
# definition (csrc/raygen.hip, restated in oracle/raygen.py) — parity unpinned.
import ctypes

from . import _lib
from .volsurfs import _Pcg32State


class Camera:
    """Pinhole camera: `intrinsics` [3,3] (pixels), `pose` = camera-to-world [3,4] or [4,4]
    (columns: camera x right, y down, z forward; last column the centre), `height`, `width`."""

    def __init__(self, intrinsics, pose, height, width, device="cuda"):
        K = torch.as_tensor(intrinsics, dtype=torch.float64).reshape(3, 3)
        c2w = torch.as_tensor(pose, dtype=torch.float64)
        if c2w.shape not in ((3, 4), (4, 4)):
            raise ValueError("pose must be [3,4] or [4,4]")
        self.height, self.width = int(height), int(width)
        self.intrinsics = K.to(torch.float32)
        self.intrinsics_inv = torch.linalg.inv(K).to(torch.float32).contiguous().to(device)
        self.c2w = c2w[:3, :4].to(torch.float32).contiguous().to(device)

    @staticmethod
    def look_at(eye, target=(0.0, 0.0, 0.0), up=(0.0, -1.0, 0.0), focal=800.0, height=800, width=800,
                device="cuda"):
        eye, target, up = (torch.tensor(v, dtype=torch.float64) for v in (eye, target, up))
        z = torch.nn.functional.normalize(target - eye, dim=0)
        x = torch.nn.functional.normalize(torch.linalg.cross(-up, z), dim=0)
        y = torch.linalg.cross(z, x)
        pose = torch.stack([x, y, z, eye], 1)
        K = [[focal, 0, 0.5 * width], [0, focal, 0.5 * height], [0, 0, 1]]
        return Camera(K, pose, height, width, device)


_m_rng = _Pcg32State()   # one process-wide stream, advanced by 2^32 per jittered call like the
                         # reference's own samplers (src/RaySampler.cu:139-142)


def get_camera_rays(camera, nr_rays_per_pixel=1, jitter_pixels=False, device="cuda"):
    """-> rays_o [H*W*R,3], rays_d [H*W*R,3], points_2d [H*W*R,2]; pixel-major, the R rays of a
    pixel consecutive (what BaseMethod.render's supersample mean expects)."""
    n = camera.height * camera.width * int(nr_rays_per_pixel)
    o = torch.empty(n, 3, device=device)
    d = torch.empty(n, 3, device=device)
    p = torch.empty(n, 2, device=device)
    _lib.call("vsa_camera_rays", camera.c2w, camera.intrinsics_inv, camera.height, camera.width,
              int(nr_rays_per_pixel), bool(jitter_pixels), ctypes.c_uint64(_m_rng.state),
              ctypes.c_uint64(_m_rng.inc), o, d, p, _lib.stream_ptr())
    if jitter_pixels:
        _m_rng.advance()
    return o, d, p


class TensorReel:
    """All training views resident in HBM (a 100-view 800x800 fp32 set is 0.77 GB of 288):
    `get_next_rays_batch` draws (camera, pixel) pairs and emits rays + ground truth in one
    launch, with no host round trip inside the training loop."""

    def __init__(self, cameras, rgbs, masks=None, device="cuda"):
        if not cameras:
            raise ValueError("TensorReel needs at least one camera")
        self.height, self.width = cameras[0].height, cameras[0].width
        if any((c.height, c.width) != (self.height, self.width) for c in cameras):
            raise ValueError("all cameras of a reel must have the same resolution")
        self.c2w = torch.stack([c.c2w for c in cameras]).contiguous()
        self.intrinsics_inv = torch.stack([c.intrinsics_inv for c in cameras]).contiguous()
        C = len(cameras)
        self.rgbs = torch.as_tensor(rgbs, dtype=torch.float32).to(device).contiguous()
        if self.rgbs.shape != (C, self.height, self.width, 3):
            raise ValueError("rgbs must be [C,H,W,3]")
        self.masks = None
        if masks is not None:
            self.masks = torch.as_tensor(masks, dtype=torch.float32).to(device).reshape(
                C, self.height, self.width).contiguous()
        self.nr_cameras = C
        self.rng = _Pcg32State()

    def get_next_rays_batch(self, batch_size=512, jitter_pixels=False, nr_rays_per_pixel=1):
        """-> (camera_idx [B], rays_o [B*R,3], rays_d [B*R,3], {"rgb": [B,3], "mask": [B,1]},
        points_2d [B*R,2])."""
        B, R = int(batch_size), int(nr_rays_per_pixel)
        dev = self.rgbs.device
        cam = torch.empty(B, dtype=torch.int32, device=dev)
        o = torch.empty(B * R, 3, device=dev)
        d = torch.empty(B * R, 3, device=dev)
        p = torch.empty(B * R, 2, device=dev)
        vals = {"rgb": torch.empty(B, 3, device=dev)}
        if self.masks is not None:
            vals["mask"] = torch.empty(B, 1, device=dev)
        _lib.call("vsa_reel_next_rays_batch", self.c2w, self.intrinsics_inv, self.rgbs, self.masks,
                  self.nr_cameras, self.height, self.width, B, R, bool(jitter_pixels),
                  ctypes.c_uint64(self.rng.state), ctypes.c_uint64(self.rng.inc), cam, o, d,
                  vals["rgb"], vals.get("mask"), p, _lib.stream_ptr())
        self.rng.advance()
        return cam, o, d, vals, p
