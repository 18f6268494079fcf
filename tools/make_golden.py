#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference Python in place.

Runs only in the build container (needs /root/reference).  The arrays it
writes are the committed fixtures; the reference source itself is never copied.
Usage:  python tools/make_golden.py [composite] [sh] [nt] [glue] [misc]
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)

import ref_import  # noqa: E402


def _save(name, **arrs):
    path = os.path.join(GOLD, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print("wrote", path, os.path.getsize(path), "bytes")


# --------------------------------------------------------------------------
# 1. dense K-shell composite through the reference's own VolSurfs.render_rays
#    (volsurfs_py/methods/volsurfs.py:423-761) with fake tracer / models.
# --------------------------------------------------------------------------
def gen_composite():
    ref_import.install_placeholders()
    from volsurfs_py.methods.volsurfs import VolSurfs

    for K, N, seed, decay, bg_mode in [
        (1, 256, 1, False, "white"),
        (5, 256, 2, True, "white"),
        (7, 256, 3, True, "perray"),
        (5, 64, 4, False, "black"),
    ]:
        g = torch.Generator().manual_seed(seed)
        rays_o = torch.zeros(N, 3)
        rays_d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
        hit_p = torch.rand(N, K, generator=g)
        is_hit = hit_p < 0.7
        is_hit[:4] = False          # rays that miss every shell
        is_hit[4:8] = True          # rays that hit every shell
        normals = torch.nn.functional.normalize(torch.randn(N, K, 3, generator=g), dim=-1)
        positions = torch.randn(N, K, 3, generator=g)
        rgb_in = torch.rand(N, K, 3, generator=g)
        alpha_in = torch.rand(N, K, 1, generator=g)
        alpha_in[8:12] = 1.0        # opaque shells
        alpha_in[12:16] = 0.0       # fully transparent shells
        rgb_leaf = rgb_in.clone().requires_grad_(True)
        alpha_leaf = alpha_in.clone().requires_grad_(True)

        class Tracer:
            def trace(self, o, d, mesh_id):
                h = is_hit[:, mesh_id]
                return {
                    "any_hit": bool(h.any()),
                    "triangles_id": torch.zeros(N, dtype=torch.long),
                    "depth": torch.ones(N),
                    "is_hit": h,
                    "positions": positions[:, mesh_id],
                    "normals": normals[:, mesh_id],
                    "barycentric": torch.full((N, 3), 1.0 / 3.0),
                }

        def mk_model(leaf, i):
            idx_of = {}

            def model(points=None, samples_dirs=None, normals=None, iter_nr=None):
                h = is_hit[:, i]
                return leaf[h, i]

            return model

        models = {"bg": None}
        for i in range(K):
            models[f"rgb_{i}"] = mk_model(rgb_leaf, i)
            models[f"alpha_{i}"] = mk_model(alpha_leaf, i)

        if bg_mode == "white":
            bg_color = torch.ones(1, 3)
        elif bg_mode == "black":
            bg_color = torch.zeros(1, 3)
        else:
            bg_color = torch.rand(N, 3, generator=g)
        bg_leaf = bg_color.clone().requires_grad_(True)

        class BP:
            def intersect(self, o, d):
                n = o.shape[0]
                return (torch.ones(n, dtype=torch.bool), torch.zeros(n), torch.ones(n),
                        o, o + d)

        fake = SimpleNamespace(
            bounding_primitive=BP(), nr_meshes=K,
            hyper_params=SimpleNamespace(using_neural_textures=False,
                                         are_volsurfs_colors_indep=True,
                                         are_volsurfs_alphas_indep=True,
                                         nr_samples_bg=0),
            profiler=None, raytracer=Tracer(), models=models,
            with_alpha_decay=decay, bg_color=bg_leaf, is_training=True,
            tensor_meshes=[None] * K)

        res = VolSurfs.render_rays(fake, rays_o, rays_d, iter_nr=0)
        rt = res["renders"]["ray_traced"]
        gt = torch.rand(N, 3, generator=g)
        # reference loss: utils/losses.py:14-19 via volsurfs.py:806
        from volsurfs_py.utils.losses import loss_l1
        loss = loss_l1(gt, rt["rgb"])
        loss.backward()
        out = {k: v.detach().numpy() for k, v in rt.items()}
        _save(f"composite_K{K}_N{N}_{bg_mode}.npz",
              rays_d=rays_d.numpy(), is_hit=is_hit.numpy(), normals=normals.numpy(),
              positions=positions.numpy(), rgb_in=rgb_in.numpy(), alpha_in=alpha_in.numpy(),
              bg_color=bg_color.numpy(), with_alpha_decay=np.array(decay), gt=gt.numpy(),
              loss=loss.detach().numpy(),
              g_rgb_in=rgb_leaf.grad.numpy(), g_alpha_in=alpha_leaf.grad.numpy(),
              g_bg=bg_leaf.grad.numpy(),
              samples_3d=res["samples_3d"].detach().numpy(),
              samples_grad=res["samples_grad"].detach().numpy(),
              **{"out_" + k: v for k, v in out.items()})


GENS = {"composite": gen_composite}

if __name__ == "__main__":
    which = sys.argv[1:] or list(GENS)
    for w in which:
        GENS[w]()
