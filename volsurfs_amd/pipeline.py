"""KShellPipeline — the fused host path of VolSurfs.render_rays + L1 loss +
backward, one stream, no host syncs.

Stage order mirrors /root/reference/volsurfs_py/methods/volsurfs.py:423-761:
trace (K shells, one launch) -> shade (neural textures) -> dense composite ->
loss_l1 (utils/losses.py:14-19) -> backward of the same chain.
"""
import torch

from . import _lib
from .camera import pinhole_rays
from .mesh import nested_shells
from .raytrace import RayTracer


class StageTimer:
    def __init__(self):
        self.records = []          # (name, start_evt, end_evt)
        self.meta = {}             # name -> dict(bytes=, flops=, bound=)

    def run(self, name, fn, record, **meta):
        if not record:
            return fn()
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn()
        e.record()
        self.records.append((name, s, e))
        self.meta[name] = meta
        return out

    def report(self):
        torch.cuda.synchronize()
        acc, cnt = {}, {}
        for name, s, e in self.records:
            acc[name] = acc.get(name, 0.0) + s.elapsed_time(e)
            cnt[name] = cnt.get(name, 0) + 1
        return {k: dict(ms=acc[k] / cnt[k], **self.meta[k]) for k in acc}


class KShellPipeline:
    dtype_desc = "f16 (composite / MLP compute as in the reference), f32 I/O"

    def __init__(self, meshes, rays_o, rays_d, gt_rgb, bg_color=(1.0, 1.0, 1.0)):
        self.meshes = meshes
        self.K = len(meshes)
        self.tracer = RayTracer(meshes)
        self.rays_o, self.rays_d, self.gt = rays_o, rays_d, gt_rgb
        self.nr_rays = rays_o.shape[0]
        dev = rays_o.device
        self.bg = torch.tensor([bg_color], device=dev, dtype=torch.float32)
        self.timer = StageTimer()
        N, K = self.nr_rays, self.K
        # PLACEHOLDER appearance until the neural-texture kernels land: fixed
        # per-(ray,shell) colours/opacities, masked by the hit.  Flagged in
        # config_desc() as a missing stage.
        g = torch.Generator(device=dev).manual_seed(7)
        self._rgb_raw = torch.rand(N, K, 3, device=dev, generator=g)
        self._alpha_raw = torch.rand(N, K, device=dev, generator=g)
        self.shading = "placeholder"

    @classmethod
    def synthetic(cls, K=5, subdiv=6, res=800, device="cuda", seed=42):
        meshes = nested_shells(K=K, subdiv=subdiv, device=device)
        o, d = pinhole_rays(res, res, focal=1111.1 * res / 800.0, cam_pos=(0.0, 0.0, -1.5),
                            device=device)
        g = torch.Generator(device=device).manual_seed(seed)
        gt = torch.rand(o.shape[0], 3, device=device, generator=g)
        p = cls(meshes, o, d, gt)
        p.res = res
        p.subdiv = subdiv
        return p

    def reset_stage_timers(self):
        self.timer = StageTimer()

    def stage_report(self):
        return self.timer.report()

    @staticmethod
    def stage_roofline(st, hbm_peak, mfma_peak):
        sec = st["ms"] * 1e-3
        out = {}
        if st.get("bytes"):
            out["GB/s"] = round(st["bytes"] / sec / 1e9, 1)
            out["hbm_frac"] = round(st["bytes"] / sec / 1e9 / hbm_peak, 4)
        if st.get("flops"):
            out["TFLOP/s"] = round(st["flops"] / sec / 1e12, 2)
            out["mfma_frac"] = round(st["flops"] / sec / 1e12 / mfma_peak, 4)
        return out

    def config_desc(self, world):
        return {
            "workload": f"synthetic kitten-like: {self.res}x{self.res} rays, K={self.K} nested "
                        f"icospheres subdiv {self.subdiv} ({self.tracer.mesh_nr_tris[0]} tris each), "
                        "white bg, L1 loss",
            "rays_per_gpu": self.nr_rays, "global_rays": self.nr_rays * world,
            "parallelism": f"tile-parallel x{world}",
            "stages": ["trace", "shade:" + self.shading, "composite_fwd", "loss_l1",
                       "composite_bwd"],
            "missing_stages": (["neural-texture shading fwd/bwd (placeholder colours used)"]
                               if self.shading == "placeholder" else []),
        }

    def step(self, record=False):
        from .composite import composite_dense
        N, K = self.nr_rays, self.K
        T = self.timer
        nodes_b = self.tracer.nodes.numel() * 4 + self.tracer.tris.numel() * 4
        hit_t, hit_slot, hit_uv = T.run(
            "trace", lambda: self.tracer.trace_all(self.rays_o, self.rays_d), record,
            bytes=N * (24 + 16 * K) + nodes_b, bound="hbm")

        def shade():
            hit = (hit_slot >= 0).t()                       # [N,K]
            rgb = (self._rgb_raw * hit[..., None]).requires_grad_(True)
            alpha = (self._alpha_raw * hit).requires_grad_(True)
            return rgb, alpha
        rgb, alpha = T.run("shade_placeholder", shade, record, bytes=N * K * 40, bound="hbm")
        out = T.run("composite_fwd", lambda: composite_dense(rgb, alpha, self.bg), record,
                    bytes=N * (16 * K + 12 + 16 * K + 16 + 4 * K), bound="hbm")

        def loss_fn():
            return (self.gt - out["rgb"]).abs().mean()
        loss = T.run("loss_l1", loss_fn, record, bytes=N * 24, bound="hbm")
        T.run("backward(loss+composite_bwd)", lambda: loss.backward(), record,
              bytes=N * (12 + 16 * K + 16 * K + 36), bound="hbm")
        return loss
