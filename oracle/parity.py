"""Checker: one whole K-shell step of the HIP path against oracle.pipeline.render_step on the SAME rays.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): called by tests/ (test_fullsize_parity.py,
test_parity_report.py) and by bench.py's cpu_baseline leg, which evaluates the oracle on a sample of the
full-size frame anyway and reports this comparison as `parity_sample`.  Nothing here imports the product:
`pipe` is whatever object carries the step's buffers (volsurfs_amd.pipeline.KShellPipeline by duck typing —
.bank, .tracer, .meshes, .rays_o / .rays_d / .gt, .face_uvs, .surfs_rgb / .surfs_alpha, .to_ray_order()).

What is compared follows north_star's statement of parity (BASELINE.json): hits bit-exact (integer work),
per-shell colours / alphas and the composited RGB within 1e-4 except on rays that read an 8-bit texel whose
quantised value differs from the oracle's by one step (the MLP's fp32 summation order: MFMA vs torch-CPU),
gradients of every hash table and MLP relative to each tensor's largest entry.
Reference: /root/reference/volsurfs_py/methods/volsurfs.py:423-761, 789-806 through oracle/pipeline.py.
"""
import time

import numpy as np
import torch

from . import neural_texture as ONT
from . import pipeline as opipe
from . import tcnn_like


def _unpack_weights(w):
    return w[:2048].view(64, 32), w[2048:6144].view(64, 64), w[6144:].view(32, 64)


def oracle_step(pipe, loss_scale=128.0, **kw):
    """oracle.pipeline.render_step on pipe's meshes, f16 parameter copies, rays and target.  Returns (result, seconds)."""
    bank = pipe.bank
    meshes = [(m.vertices.cpu().numpy(), m.faces.cpu().numpy(), m.faces_uvs.cpu()) for m in pipe.meshes]
    tabs, wts = bank.tables_h.cpu().float(), bank.weights_h.cpu().float()
    o, d, gt = pipe.rays_o.cpu().numpy(), pipe.rays_d.cpu().numpy(), pipe.gt.cpu()
    t0 = time.perf_counter()
    ref = opipe.render_step(meshes, tabs, wts, bank.tex_index, bank.tex_res, o, d, gt, loss_scale=loss_scale, **kw)
    return ref, time.perf_counter() - t0


def flipped_texel_rays(pipe):
    """Which rays read a texel whose 8-bit row differs from the oracle MLP + quantiser evaluated on the kernel's own
    (bit-exact) features.  Runs encode + MLP on the frame the bank last compacted (no backward may follow without a
    new forward).  Returns (ray_flip bool [N] in the caller's ray order, flipped channels, channels, max |dq|)."""
    bank, K = pipe.bank, pipe.K
    bank.encode()
    feats = bank.features_level_major()
    texels, _ = bank.mlp(want_pre=True)
    seg = bank.seg_start.cpu().numpy()
    flipped_slot = torch.zeros(bank.slot_capacity, dtype=torch.bool)
    flips = total = dq_max = 0
    for s in range(K):
        for typ in range(2):
            for d in range(4):
                x = bank.tex_index(s, typ, d)
                C = bank.tex_channels(x)
                a, b = seg[s * 4 + d], seg[s * 4 + d + 1]
                if C == 0 or b <= a:
                    continue
                f = feats[typ, :, a:b].cpu().permute(1, 0, 2).reshape(-1, 32)
                w1, w2, w3 = _unpack_weights(bank.weights_h[x].cpu())
                _, q_ref = ONT.quantise(tcnn_like.mlp_forward(w1, w2, w3, f, C))
                base = 0 if typ == 0 else 24
                dq = (texels[a:b, base:base + C].cpu().int() - q_ref.int()).abs()
                flipped_slot[a:b] |= (dq > 0).any(dim=1)
                flips += int((dq > 0).sum())
                total += dq.numel()
                dq_max = max(dq_max, int(dq.max()))
    _, hs, hit_uv = pipe.tracer.trace_all(pipe.rays_o, pipe.rays_d)
    tex_uv = bank.tex_uv_only(hs, hit_uv, pipe.face_uvs).cpu()
    hs, slot_of = hs.cpu(), bank.slot_of.cpu()
    ray_flip = torch.zeros(pipe.rays_o.shape[0], dtype=torch.bool)
    for s in range(K):
        hit = hs[s] >= 0
        rows = hit.nonzero()[:, 0]
        for d in range(4):
            R = bank.tex_res[d]
            W = R + 2
            _, _, corners = ONT.texel_corners(tex_uv[s][hit].clone(), R)
            ij = torch.floor(corners).long() + 1
            sl = slot_of[int(bank.plan.dom_off[s * 4 + d]) + ij[..., 1] * W + ij[..., 0]].long()     # [M,4]
            ray_flip[rows[flipped_slot[sl].any(dim=1)]] = True
    return ray_flip, flips, total, dq_max


def compare_step(pipe, rgb, ref, attribute_flips=True):
    """pipe has just run one step() that returned `rgb` (caller's ray order); ref = oracle_step(pipe)[0].
    Returns a JSON-able summary.  Gradients are read BEFORE the flip attribution re-runs encode / MLP."""
    bank, K = pipe.bank, pipe.K
    gw, gt = bank.weights.grad.detach().cpu().clone(), bank.tables.grad.detach().cpu().clone()
    rgb = rgb.detach().cpu().numpy()
    hit = (pipe.to_ray_order(pipe._hit_slot, dim=1) >= 0).t().cpu().numpy()
    s_rgb = pipe.to_ray_order(pipe.surfs_rgb).cpu().numpy()
    s_a = pipe.to_ray_order(pipe.surfs_alpha).cpu().numpy()
    e = np.abs(rgb - ref["rgb"])
    e_rgb, e_a = np.abs(s_rgb - ref["surfs_rgb"]), np.abs(s_a - ref["surfs_alpha"])
    out = {"rays": int(rgb.shape[0]), "shells": int(K), "hits": int(ref["hit"].sum()),
           "hit_mismatches": int((hit != ref["hit"]).sum()),
           "rgb_max_err": float(e.max()), "rgb_frac_over_1e-4": float((e > 1e-4).mean()),
           "rgb_median_err": float(np.median(e)),
           "surfs_rgb_max_err": float(e_rgb.max()), "surfs_rgb_frac_over_1e-5": float((e_rgb > 1e-5).mean()),
           "surfs_alpha_max_err": float(e_a.max()), "surfs_alpha_frac_over_1e-5": float((e_a > 1e-5).mean())}
    worst_w = worst_t = 0.0
    cos_min = 1.0
    over = 0
    n_el = 0
    for x, (g_t, g_w) in ref["grads"].items():
        cos_min = min(cos_min, float(torch.nn.functional.cosine_similarity(gw[x], g_w, dim=0)),
                      float(torch.nn.functional.cosine_similarity(gt[x].flatten(), g_t.flatten(), dim=0)))
        rw = (gw[x] - g_w).abs() / g_w.abs().max()
        rt = (gt[x] - g_t).abs() / g_t.abs().max()
        worst_w, worst_t = max(worst_w, float(rw.max())), max(worst_t, float(rt.max()))
        over += int((rw > 1e-3).sum()) + int((rt > 1e-3).sum())
        n_el += rw.numel() + rt.numel()
    out.update({"grad_tensors": len(ref["grads"]), "grad_cos_min": cos_min,
                "grad_weights_rel_max": worst_w, "grad_tables_rel_max": worst_t,
                "grad_frac_over_1e-3_of_tensor_max": over / max(1, n_el),
                "grad_note": "relative to each tensor's largest entry; the oracle's gradients are the reference's own "
                             "fp16 autograd under its loss scale of 128 (noisy itself: DESIGN.md section 6)"})
    if attribute_flips:
        ray_flip, flips, total, dq_max = flipped_texel_rays(pipe)
        over_rays = torch.from_numpy(e > 1e-4).any(dim=1)
        unexplained = over_rays & ~ray_flip
        out.update({"texel_flip_rate": flips / max(1, total), "texel_channels": total, "texel_max_step": dq_max,
                    "rays_over_1e-4": int(over_rays.sum()), "rays_touching_a_flipped_texel": int(ray_flip.sum()),
                    "rays_over_1e-4_without_a_flipped_texel": int(unexplained.sum()),
                    "max_err_without_a_flipped_texel": float(e[(~ray_flip).numpy()].max()) if (~ray_flip).any() else 0.0})
    return out
