"""Oracle: the whole K-shell render step the reference's way:
trace every shell (brute force) -> per shell, per model SHNeuralTextures on the
hits (4 network evaluations per hit and degree) -> alpha decay -> scatter ->
dense fp16 composite -> L1 loss -> backward.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows
/root/reference/volsurfs_py/methods/volsurfs.py:423-761 (render_rays) and
:789-806 (loss).  Used by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.
"""
import numpy as np
import torch

from . import composite as ocomp
from . import neural_texture as ONT
from . import raytrace as ort


def render_step(meshes, tables, weights, tex_index, tex_res, rays_o, rays_d, gt, bg=(1.0, 1.0, 1.0),
                sh_range=15.0, with_alpha_decay=True, backward=True, loss_scale=1.0, tex_flags=None,
                has_alpha=None):
    """meshes: list of (verts [V,3], faces [F,3], faces_uvs [F,3,2] torch);
    tables [n_tex,E,2], weights [n_tex,8192] torch (fp16-representable values);
    tex_index(shell, type, deg) -> row.  rays_*: numpy [N,3]; gt torch [N,3].
    loss_scale multiplies the loss before the (fp16) autograd and is divided out of
    the returned gradients — what the reference's GradScaler / tiny-cuda-nn's loss
    scale (128) do against fp16 underflow (base_method.py:255-262).
    tex_flags: NeuralTexture's switches (anchor / lerp / quantize_output / squeeze_output; default: the shipped
    config's).  tex_index may map several shells to ONE row (are_volsurfs_colors_indep / _alphas_indep = 0,
    volsurfs.py:159-165, 200-206, 524-527, 553-556): that model's parameters are then one set of leaves and the
    shells' gradients accumulate in it, as in the reference's single module.  has_alpha(shell) -> False: the
    shell's alpha model is None (volsurfs.py:558-561: alpha = 1, no decay).
    Returns dict(rgb [N,3] np, surfs_rgb, surfs_alpha, hit [N,K], grads {tex: (g_table, g_weights)})."""
    n, K = rays_o.shape[0], len(meshes)
    surfs_rgb = torch.zeros(n, K, 3)
    surfs_alpha = torch.zeros(n, K)
    leaves = {}
    dirs_all = torch.from_numpy(rays_d)
    hits = np.zeros((n, K), bool)
    for s, (v, f, fuv) in enumerate(meshes):
        h = ort.trace_bruteforce(v, f, rays_o, rays_d)
        att = ort.hit_attributes(v, f, rays_o, rays_d, h)
        hit = torch.from_numpy(att["is_hit"])
        hits[:, s] = att["is_hit"]
        if not hit.any():
            continue
        uv = ONT.interp_uv(torch.from_numpy(att["barycentric"])[hit], fuv,
                           torch.from_numpy(att["triangles_id"]).long()[hit])
        dirs = dirs_all[hit]
        rows = hit.nonzero()[:, 0]
        for typ, C in ((0, 3), (1, 1)):
            if typ == 1 and has_alpha is not None and not has_alpha(s):
                surfs_alpha = surfs_alpha.index_put((rows, torch.tensor(s)), torch.ones(rows.shape[0]))
                continue
            texs = []
            for deg in range(4):
                x = tex_index(s, typ, deg)
                if x not in leaves:
                    w = weights[x]
                    leaves[x] = [t.clone().requires_grad_(True) for t in
                                 (tables[x], w[:2048].view(64, 32), w[2048:6144].view(64, 64), w[6144:].view(32, 64))]
                texs.append(ONT.NeuralTextureOracle(tex_res[deg], C * (2 * deg + 1),
                                                    (-sh_range, sh_range), *leaves[x], **(tex_flags or {})))
            out = ONT.sh_neural_textures_forward(texs, uv, dirs, C, 3)
            if typ == 0:
                surfs_rgb = surfs_rgb.index_put((rows, torch.tensor(s)), out)
            else:
                a = out[:, 0]
                if with_alpha_decay:
                    a = a * ONT.alpha_decay(dirs, torch.from_numpy(att["normals"])[hit])[:, 0]
                surfs_alpha = surfs_alpha.index_put((rows, torch.tensor(s)), a)
    c_np, a_np = surfs_rgb.detach().numpy(), surfs_alpha.detach().numpy()
    bg_np = np.asarray([bg], np.float32)
    fwd = ocomp.composite_dense_fwd(c_np, a_np, bg_np)
    res = {"rgb": fwd["rgb"], "surfs_rgb": c_np, "surfs_alpha": a_np, "hit": hits, "grads": {}}
    if backward:
        g = np.sign(fwd["rgb"] - gt.numpy()).astype(np.float32) / (n * 3)   # d mean|gt-pred|
        gc, ga, _ = ocomp.composite_dense_bwd(c_np, a_np, bg_np, g)
        loss = ((surfs_rgb * torch.from_numpy(gc)).sum() + (surfs_alpha * torch.from_numpy(ga)).sum()) * loss_scale
        if loss.requires_grad:
            loss.backward()
        for x, (t, w1, w2, w3) in leaves.items():
            if t.grad is not None:
                res["grads"][x] = (t.grad / loss_scale,
                                   torch.cat([w1.grad.flatten(), w2.grad.flatten(),
                                              w3.grad.flatten()]) / loss_scale)
    return res
