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
