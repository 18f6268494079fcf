"""Measured parity of the whole step (VERDICT r1 "next" item 1): error percentiles instead of
loose pass/fail bars, and two order-matched checks that remove the one legitimately
implementation-dependent step (fp32 summation order inside the MLP) from the comparison.

  A. independent oracle (oracle/pipeline.py, the reference's algorithm on torch-CPU):
     texel-flip rate, RGB error percentiles, gradient error percentiles — reported
     (printed as `PARITY_REPORT {json}` and written to gpurun_out/parity_report.json).
  B. order-matched: the oracle's post-processing (quantise -> fp16 expand -> lerp -> SH ->
     sigmoid -> alpha decay -> fp16 composite) driven by the KERNEL's own network outputs
     (`pre_out`).  Everything after the MLP must then agree: texels bit-exact, per-shell
     colours <= 2e-6, composited RGB within 1e-4 (north_star) except where a 1e-6 difference of
     an fp32 sigmoid flips the fp16 rounding of a composite input (one fp16 ulp = 4.9e-4 in
     [0.5, 1)); the number of such pixels is measured and bounded.
  C. gradients against an fp32 oracle AT the kernel's quantised texels: the reference's maths
     with fp32 autograd (straight-through at every fp16 rounding point, forward values pinned
     to the kernel's), so what is compared is the kernel's fp16 gradient chain (f16 MFMA
     operands, f16 gradient rows, fixed-point table accumulation) against exact arithmetic:
     north_star's 1e-3 on grads, relative to each tensor's largest gradient.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import composite as OC
from oracle import neural_texture as ONT
from oracle import tcnn_like

from test_nt_mlp import unpack_weights

PCT = (50, 90, 99, 100)


def _pcts(e):
    e = np.asarray(e, np.float64).reshape(-1)
    return {("p%d" % p if p < 100 else "max"): float(np.percentile(e, p)) for p in PCT}


def _emit(tag, rep):
    line = "PARITY_REPORT " + json.dumps({tag: rep})
    print(line)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "parity_report.json")
        cur = json.load(open(path)) if os.path.exists(path) else {}
        cur[tag] = rep
        json.dump(cur, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


# (K, subdiv, res, noise): the two small frames of rounds 1-2 and BASELINE's shell counts
# (K = 5: configs[1]-[3], K = 7: configs[4]) on NOISY shells at >= 16 k rays
CASES = [(3, 3, 56, 0.0), (1, 3, 64, 0.0), (5, 4, 128, 0.05), (7, 4, 128, 0.05)]


INDEP_GRAD_MAX = {1: 8.1e-4, 3: 4.0e-3, 5: 1.7e-2, 7: 7.8e-2}      # 2x measured, per shell count


def _pipe(K, subdiv, res, seed=5, noise=0.0):
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.pipeline import KShellPipeline
    meshes = nested_shells(K=K, subdiv=subdiv, noise=noise)
    o, d = pinhole_rays(res, res, focal=1.6 * res, cam_pos=(0.0, 0.0, -1.5))
    gt = torch.rand(o.shape[0], 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(seed))
    pipe = KShellPipeline(meshes, o, d, gt, seed=seed, init="spread")      # ray order = caller's order
    pipe.grad_scale *= float(os.environ.get("VSA_TEST_GRAD_SCALE_MULT", "1"))   # experiments
    return pipe


def _independent_oracle(pipe):
    from oracle import pipeline as opipe
    bank = pipe.bank
    meshes = [(m.vertices.cpu().numpy(), m.faces.cpu().numpy(), m.faces_uvs.cpu()) for m in pipe.meshes]
    return opipe.render_step(meshes, bank.tables_h.cpu().float(), bank.weights_h.cpu().float(),
                             bank.tex_index, bank.tex_res, pipe.rays_o.cpu().numpy(),
                             pipe.rays_d.cpu().numpy(), pipe.gt.cpu(), loss_scale=128.0)


@pytest.mark.gpu
@pytest.mark.parametrize("K,subdiv,res,noise", CASES)
def test_parity_report_vs_independent_oracle(K, subdiv, res, noise):
    pipe = _pipe(K, subdiv, res, noise=noise)
    bank = pipe.bank
    rgb = pipe.step().cpu().numpy()
    gw, gt = bank.weights.grad.cpu().clone(), bank.tables.grad.cpu().clone()
    ref = _independent_oracle(pipe)
    # texel flips: the kernel's texels vs the oracle MLP + quantise on the same (bit-exact) features
    bank.encode()
    feats = bank.features_level_major()
    texels, pre = bank.mlp(want_pre=True)
    seg = bank.seg_start.cpu().numpy()
    flips = total = 0
    # ... and between the accumulation models of the absent tiny-cuda-nn (oracle/tcnn_like.py): the
    # published half FMA chain of the grid (= the kernel) against the fp32 sum of rounds 1-2, and the
    # fp32 MLP accumulator (= MFMA) against a model of FullyFusedMLP's half accumulator fragments —
    # the attainable floor of "matches the reference CUDA path" 
    flips_grid = flips_mlp = 0
    geom = tcnn_like.GridGeometry()
    slot_xy = bank.slot_xy.cpu()
    for s in range(K):
        for typ in range(2):
            for d in range(4):
                x = bank.tex_index(s, typ, d)
                C = bank.tex_channels(x)
                a, b = seg[s * 4 + d], seg[s * 4 + d + 1]
                f = feats[typ, :, a:b].cpu().permute(1, 0, 2).reshape(-1, 32)
                w1, w2, w3 = unpack_weights(bank.weights_h[x].cpu())
                _, q_ref = ONT.quantise(tcnn_like.mlp_forward(w1, w2, w3, f, C))
                base = 0 if typ == 0 else 24
                dq = (texels[a:b, base:base + C].cpu().int() - q_ref.int()).abs()
                assert dq.max() <= 1
                flips += int((dq > 0).sum())
                total += dq.numel()
                tab = bank.tables_h[x].cpu().float()
                f16 = tcnn_like.hashgrid_forward(geom, tab, slot_xy[a:b], accumulate="f16")
                assert torch.equal(f16.view(torch.int16), f.view(torch.int16))      # the kernel IS the f16 chain
                f32 = tcnn_like.hashgrid_forward(geom, tab, slot_xy[a:b], accumulate="f32")
                _, q_g = ONT.quantise(tcnn_like.mlp_forward(w1, w2, w3, f32, C))
                _, q_m = ONT.quantise(tcnn_like.mlp_forward(w1, w2, w3, f, C, accumulate="f16"))
                flips_grid += int((q_g.int() != q_ref.int()).sum())
                flips_mlp += int((q_m.int() != q_ref.int()).sum())
    e = np.abs(rgb - ref["rgb"])
    # ---- which rays touch a flipped texel: "RGB within 1e-4 except where an 8-bit texel
    # flipped" as a TESTED statement — every value over 1e-4 must belong to a ray one of whose hits reads a
    # slot whose quantised row differs from the oracle's
    flipped_slot = torch.zeros(bank.slot_capacity, dtype=torch.bool)
    for s in range(K):
        for typ in range(2):
            for d in range(4):
                x = bank.tex_index(s, typ, d)
                C = bank.tex_channels(x)
                a, b = seg[s * 4 + d], seg[s * 4 + d + 1]
                f = feats[typ, :, a:b].cpu().permute(1, 0, 2).reshape(-1, 32)
                w1, w2, w3 = unpack_weights(bank.weights_h[x].cpu())
                _, q_ref = ONT.quantise(tcnn_like.mlp_forward(w1, w2, w3, f, C))
                base = 0 if typ == 0 else 24
                flipped_slot[a:b] |= (texels[a:b, base:base + C].cpu() != q_ref).any(dim=1)
    _, hs2, hit_uv = pipe.tracer.trace_all(pipe.rays_o, pipe.rays_d)
    tex_uv = bank.tex_uv_only(hs2, hit_uv, pipe.face_uvs).cpu()
    hs2, slot_of = hs2.cpu(), bank.slot_of.cpu()
    ray_flip = torch.zeros(rgb.shape[0], dtype=torch.bool)
    for s in range(K):
        hit = hs2[s] >= 0
        rows = hit.nonzero()[:, 0]
        for d in range(4):
            R = bank.tex_res[d]
            W = R + 2
            _, _, corners = ONT.texel_corners(tex_uv[s][hit].clone(), R)
            ij = torch.floor(corners).long() + 1
            sl = slot_of[int(bank.plan.dom_off[s * 4 + d]) + ij[..., 1] * W + ij[..., 0]].long()     # [M,4]
            ray_flip[rows[flipped_slot[sl].any(dim=1)]] = True
    over = torch.from_numpy(e > 1e-4).any(dim=1)
    unexplained = over & ~ray_flip
    g_rel = []
    for x, (g_t, g_w) in ref["grads"].items():
        g_rel.append(((gw[x] - g_w).abs() / g_w.abs().max()).numpy())
        g_rel.append(((gt[x] - g_t).abs() / g_t.abs().max()).flatten().numpy())
    g_rel = np.concatenate(g_rel)
    rep = {"rays": int(rgb.shape[0]), "hits": int(ref["hit"].sum()),
           "texel_flip_rate": flips / total, "texel_channels": total,
           "texel_flip_rate_grid_f16_chain_vs_f32_sum": flips_grid / total,
           "texel_flip_rate_mlp_f32_acc_vs_f16_acc_model": flips_mlp / total,
           "rgb_abs_err": _pcts(e), "rgb_frac_over_1e-4": float((e > 1e-4).mean()),
           "rays_over_1e-4": int(over.sum()), "rays_touching_a_flipped_texel": int(ray_flip.sum()),
           "rays_over_1e-4_without_a_flipped_texel": int(unexplained.sum()),
           "max_err_without_a_flipped_texel": float(e[(~ray_flip).numpy()].max()) if (~ray_flip).any() else 0.0,
           "grad_err_rel_to_tensor_max": _pcts(g_rel),
           "note": "oracle gradients are the reference's fp16 autograd (itself noisy)"}
    _emit(f"independent_K{K}_res{res}", rep)
    # a texel flip moves one SH coefficient by 30/255: bounded, and rare (measured 1.2-1.5e-5)
    assert rep["texel_flip_rate"] < 3e-5            # measured 0.6-1.7e-5 (profiles/r0[2-5]/parity_report.json)
    # RGB is within north_star's 1e-4 on EVERY value of every ray that does not read a flipped texel — up to
    # the one-fp16-ulp moves the order-matched test below bounds (an fp32 sigmoid's last bit straddling an
    # fp16 rounding boundary of a composite input: measured 0 rays at K <= 5, <= 2 at K = 7)
    assert rep["rays_over_1e-4_without_a_flipped_texel"] <= 2, rep
    assert rep["max_err_without_a_flipped_texel"] <= 1e-3, rep
    # where no texel flipped the pixel is bit-identical; a flip shows as one or two fp16 ulps
    assert rep["rgb_abs_err"]["p99"] == 0.0 and rep["rgb_abs_err"]["max"] <= 1e-2      # measured <= 4.9e-3
    assert rep["rgb_frac_over_1e-4"] <= 2e-3
    assert rep["grad_err_rel_to_tensor_max"]["p99"] < 1e-2
    # ... and the MAXIMUM, at 2x the measured value of each case (VERDICT r3 next #1c).  These bounds are
    # loose because THIS oracle's gradients are the reference's own fp16 autograd (with the reference's
    # loss scale of 128), noisy itself: measured max 4.0e-4 / 2.0e-3 / 8.5e-3 / 3.9e-2 at K = 1 / 3 / 5 /
    # 7 (profiles/r03/parity_report.json).  The tight gradient bound (1e-3 of each tensor's largest entry,
    # every element) is asserted against the fp32 oracle in test_order_matched_rgb_and_f32_gradients.
    assert rep["grad_err_rel_to_tensor_max"]["max"] <= INDEP_GRAD_MAX[K], rep["grad_err_rel_to_tensor_max"]


def _ste_half(x):
    """Forward: the fp16 rounding of x; backward: identity (fp32 gradient oracle)."""
    return x + (x.half().float() - x).detach()


def _pin(x, value):
    """Forward: `value` (the kernel's number); backward: identity through x."""
    return x + (value - x).detach()


def _grid_f32(geom, table, xy):
    """tcnn_like.hashgrid_forward in differentiable fp32 (table values are fp16-representable)."""
    outs = []
    for l in range(geom.n_levels):
        pos = xy * np.float32(geom.scale[l]) + np.float32(0.5)
        cell = torch.floor(pos)
        frac = pos - cell
        c = cell.to(torch.int64) & 0xFFFFFFFF
        feat = 0.0
        for corner in range(4):
            dx, dy = corner & 1, (corner >> 1) & 1
            wx = frac[:, 0] if dx else 1 - frac[:, 0]
            wy = frac[:, 1] if dy else 1 - frac[:, 1]
            idx = geom.index(l, (c[:, 0] + dx) & 0xFFFFFFFF, (c[:, 1] + dy) & 0xFFFFFFFF)
            feat = feat + (wx * wy)[:, None] * table[geom.offset[l] + idx]
        outs.append(feat)
    return torch.cat(outs, dim=1)


# (r5: + K = 9, the largest shell count the reference configures — config/volsurfs/base_9.cfg — on noisy
#  shells; 72 textures, i.e. the scalar work split of nt_for_each_piece_scalar)
@pytest.mark.gpu
@pytest.mark.parametrize("K,subdiv,res,noise", CASES + [(9, 3, 96, 0.05)])
def test_order_matched_rgb_and_f32_gradients(K, subdiv, res, noise):
    from volsurfs_amd.composite import composite_fwd_bwd_l1_raw
    pipe = _pipe(K, subdiv, res, noise=noise)
    bank = pipe.bank
    N = pipe.nr_rays
    rgb = pipe.step().cpu()
    gw, gt = bank.weights.grad.cpu().clone(), bank.tables.grad.cpu().clone()
    dF = bank.features_level_major().cpu()      # the backward left the f16 feature gradients (x grad_scale) here
    hit_slot = pipe._hit_slot.cpu()
    surfs_rgb_k, surfs_alpha_k = pipe.surfs_rgb.cpu(), pipe.surfs_alpha.cpu()
    # the step's backward overwrote the feature planes with dF: recompute the forward state
    bank.encode()
    feats = bank.features_level_major().cpu()
    texels, pre = bank.mlp(want_pre=True)
    texels, pre = texels.cpu(), pre.cpu()
    _, hs2, hit_uv = pipe.tracer.trace_all(pipe.rays_o, pipe.rays_d)
    assert torch.equal(hs2.cpu(), hit_slot)
    tex_uv = bank.tex_uv_only(pipe._hit_slot, hit_uv, pipe.face_uvs).cpu()
    _, _, normals, _ = bank.shade(pipe._hit_slot, tex_uv.cuda(), pipe.rays_d, pipe.tracer.tris,
                                  want_normals=True)
    normals = normals.cpu()
    rgb2, g_c, g_a = composite_fwd_bwd_l1_raw(pipe.surfs_rgb, pipe.surfs_alpha, pipe.bg, pipe.gt,
                                              1.0 / (3.0 * N))
    assert torch.equal(rgb2.cpu(), rgb)
    g_c, g_a = g_c.cpu(), g_a.cpu()
    slot_of, slot_xy = bank.slot_of.cpu(), bank.slot_xy.cpu()
    seg = bank.seg_start.cpu().numpy()
    dirs_all = pipe.rays_d.cpu()
    geom = tcnn_like.GridGeometry()

    # ---- B1: texels == the oracle's quantisation of the kernel's own network outputs, bit-exact
    leaves = {}
    q_of = {}
    for s in range(K):
        for typ in range(2):
            for d in range(4):
                x = bank.tex_index(s, typ, d)
                C = bank.tex_channels(x)
                a, b = seg[s * 4 + d], seg[s * 4 + d + 1]
                base = 0 if typ == 0 else 24
                _, q_ref = ONT.quantise(pre[a:b, base:base + C])
                assert torch.equal(texels[a:b, base:base + C], q_ref), (s, typ, d)
                q_of[x] = q_ref

    # ---- B2 + C: the reference's op sequence from there on, fp32 autograd, forward values pinned
    net_state = {}
    surfs_rgb = torch.zeros(N, K, 3)
    surfs_alpha = torch.zeros(N, K)
    for s in range(K):
        hit = hit_slot[s] >= 0
        rows = hit.nonzero()[:, 0]
        uv, dirs = tex_uv[s][hit], dirs_all[hit]
        M = uv.shape[0]
        for typ, C1 in ((0, 3), (1, 1)):
            coeffs = []
            for d in range(4):
                x = bank.tex_index(s, typ, d)
                C, n = bank.tex_channels(x), 2 * d + 1
                a, b = seg[s * 4 + d], seg[s * 4 + d + 1]
                w = bank.weights_h[x].cpu().float()
                ps = [t.clone().requires_grad_(True) for t in
                      (bank.tables_h[x].cpu().float(), *unpack_weights(w))]
                leaves[x] = ps
                table, w1, w2, w3 = ps
                # network at the touched texel centres (once per slot == 4x per hit, same function)
                F = _pin(_grid_f32(geom, table, slot_xy[a:b]),
                         feats[typ, :, a:b].permute(1, 0, 2).reshape(-1, 32).float())
                h1 = _ste_half(torch.relu(F @ w1.t()))
                h = _ste_half(torch.relu(h1 @ w2.t()))
                base = 0 if typ == 0 else 24
                p = _pin((h @ w3.t())[:, :C], pre[a:b, base:base + C].float())
                p.retain_grad()
                net_state[x] = (h1.detach(), h.detach(), p, C, a, b)
                o = torch.sigmoid(p)                                     # neural_texture.py:162
                o = _pin(o, q_of[x].float() / 255.0)                     # x255, round_ste, /255 (:166-169)
                o = _ste_half(o)                                         # :177
                lo, span = float(bank.plan.sh_lo[d]), float(bank.plan.sh_span[d])
                e16 = (lo + span * o.detach().half())                    # :183-185 in fp16
                e = _pin(lo + span * o, e16.float())
                # lerp of the 4 corner texels (:188-191)
                R = bank.tex_res[d]
                W = R + 2
                _, lw, corners = ONT.texel_corners(uv.clone(), R)
                ij = torch.floor(corners).long() + 1
                dom = int(bank.plan.dom_off[s * 4 + d])
                sl = slot_of[dom + ij[..., 1] * W + ij[..., 0]].long() - a              # [M,4] local
                assert (sl >= 0).all() and (sl < b - a).all()
                r = (e[sl] * lw).sum(dim=1)                                             # [M,C] fp32
                coeffs.append(r.reshape(M, C1, n))
            sh = _ste_half(torch.cat(coeffs, dim=2))                     # sh_neural_textures.py:88
            out = torch.sigmoid(ONT.sh_eval(sh, dirs, 3, round0=_ste_half))   # :89-95 (C0 * half -> half)
            if typ == 0:
                surfs_rgb = surfs_rgb.index_put((rows, torch.tensor(s)), out)
            else:
                aa = out[:, 0] * ONT.alpha_decay(dirs, normals[:, s][hit])[:, 0]   # volsurfs.py:583-594
                surfs_alpha = surfs_alpha.index_put((rows, torch.tensor(s)), aa)
    c_np, a_np = surfs_rgb.detach().numpy(), surfs_alpha.detach().numpy()
    e_c = np.abs(c_np - surfs_rgb_k.numpy())
    e_a = np.abs(a_np - surfs_alpha_k.numpy())
    assert e_c.max() <= 2e-6 and e_a.max() <= 2e-6          # every stage after the MLP, every hit
    ref_rgb = OC.composite_dense_fwd(c_np, a_np, pipe.bg.cpu().numpy())["rgb"]
    e = np.abs(rgb.numpy() - ref_rgb)
    # the kernel's composite of ITS OWN per-shell colours is bit-exact against the oracle's
    own = OC.composite_dense_fwd(surfs_rgb_k.numpy(), surfs_alpha_k.numpy(), pipe.bg.cpu().numpy())["rgb"]
    assert np.array_equal(own, rgb.numpy())
    over = float((e > 1e-4).mean())
    rep = {"rays": N, "hits": int((hit_slot >= 0).sum()), "texels_bit_exact": True,
           "surfs_rgb_max_err": float(e_c.max()), "surfs_alpha_max_err": float(e_a.max()),
           "rgb_abs_err": _pcts(e), "rgb_frac_over_1e-4": over,
           "rgb_cause_of_any_excess": "an fp32 difference <= 2e-6 in a per-shell colour flips its fp16 "
                                      "cast in the composite (1 ulp = 4.9e-4 in [0.5,1))"}
    # north_star: 1e-4 on RGB, EVERY pixel (measured on MI355X: bit-identical, max error 0.0 — the
    # <= 2.4e-7 differences of the per-shell colours never straddle an fp16 rounding boundary in
    # these frames; profiles/r02/parity_report.json)
    # K = 5 / 7 at 16 k rays: 0 and 2 of 49 152 values move by ONE fp16 ulp (4.9e-4) — the per-shell
    # colours above agree to 4e-7, but an fp32 sigmoid that differs in its last bit between the
    # hardware exp2 / rcp and torch-CPU can straddle an fp16 rounding boundary of the composite's
    # inputs; the reference's own CUDA sigmoid has the same freedom.  Bounded at 2x measured.
    assert over <= 1e-4 and e.max() <= 1e-3

    # ---- C: gradients, fp32 oracle at the kernel's texels vs the kernel's fp16 chain
    loss = (surfs_rgb * g_c).sum() + (surfs_alpha * g_a).sum()
    loss.backward()
    rel_w, rel_t, worst = [], [], 0.0
    lvl_worst = np.zeros(geom.n_levels)
    # where the table-gradient error comes from.  The table gradient is a scatter
    # grad[idx_c] += w_c * dF of the feature gradients dF that the MLP backward stored as f16.  The SAME
    # scatter of THOSE dF in float64 separates the accumulation's own error (fixed-point LDS sums,
    # vsa_nt_encode_bwd) from what is already in dF (the fp16 gradient chain: f16 MFMA operands dOut,
    # dH2, dH1 and the f16 store, each 2^-11 relative — tiny-cuda-nn back-propagates in half as well)
    acc_worst = 0.0
    for s in range(K):
        for typ in range(2):
            for d in range(4):
                x = bank.tex_index(s, typ, d)
                if not bank.tex_channels(x):
                    continue
                a, b = seg[s * 4 + d], seg[s * 4 + d + 1]
                exact = torch.zeros(geom.offset[-1], 2, dtype=torch.float64)
                for l in range(geom.n_levels):
                    idx, w = tcnn_like._grid_cells(geom, l, slot_xy[a:b])
                    g = dF[typ, l, a:b].double() / pipe.grad_scale
                    for c in range(4):
                        exact.index_add_(0, geom.offset[l] + idx[c], w[c].double()[:, None] * g)
                acc_worst = max(acc_worst, float(((gt[x].double() - exact).abs() / exact.abs().max()).max()))
    for x, (table, w1, w2, w3) in leaves.items():
        ref_w = torch.cat([w1.grad.flatten(), w2.grad.flatten(), w3.grad.flatten()])
        rw = ((gw[x] - ref_w).abs() / ref_w.abs().max()).numpy()
        rt = ((gt[x] - table.grad).abs() / table.grad.abs().max()).flatten().numpy()
        rel_w.append(rw)
        rel_t.append(rt)
        rt2 = rt.reshape(-1, 2).max(1)
        for l in range(geom.n_levels):
            lvl_worst[l] = max(lvl_worst[l], rt2[geom.offset[l]:geom.offset[l + 1]].max())
        worst = max(worst, float(rw.max()), float(rt.max()))
        cw = torch.nn.functional.cosine_similarity(gw[x], ref_w, dim=0)
        ct = torch.nn.functional.cosine_similarity(gt[x].flatten(), table.grad.flatten(), dim=0)
        assert cw > 0.9999 and ct > 0.9999, (x, float(cw), float(ct))
    # ---- the same comparison for a MODEL of the reference CUDA path's own arithmetic (tiny-cuda-nn
    # back-propagates in half and accumulates the table gradient with half2 atomics:
    # oracle/tcnn_like.py mlp_backward_half / hashgrid_backward_half_atomics, a conservative model).
    # north_star's "1e-3 on grads vs the reference CUDA path" cannot be tighter than the distance of
    # that path's own arithmetic from exact arithmetic: where the kernel's tail exceeds 1e-3 (K = 7),
    # the bound is justified by showing the model's tail is larger (VERDICT r3 next #1b).
    rel_m = []
    for x, (table, w1, w2, w3) in leaves.items():
        h1, h2, p, C, a, b = net_state[x]
        dX = tcnn_like.mlp_backward_half(w1.detach(), w2.detach(), w3.detach(), h1, h2,
                                         p.grad * pipe.grad_scale, C)
        gm = tcnn_like.hashgrid_backward_half_atomics(geom, dX, slot_xy[a:b]) / pipe.grad_scale
        rel_m.append(((gm - table.grad).abs() / table.grad.abs().max()).flatten().numpy())
    rel_m = np.concatenate(rel_m)
    rep["tcnn_half_backward_model_tables_err_rel_to_tensor_max"] = _pcts(rel_m)
    rep["tcnn_half_backward_model_tables_frac_over_1e-3"] = float((rel_m > 1e-3).mean())
    rep["grad_weights_err_rel_to_tensor_max"] = _pcts(np.concatenate(rel_w))
    rep["grad_tables_err_rel_to_tensor_max"] = _pcts(np.concatenate(rel_t))
    rep["grad_tables_worst_err_per_level"] = [float(x) for x in lvl_worst]
    rep["grad_tables_accumulation_err_vs_exact_scatter_of_kernel_dF"] = acc_worst
    rep["grad_tables_frac_over_1e-3"] = float((np.concatenate(rel_t) > 1e-3).mean())
    _emit(f"order_matched_K{K}_res{res}", rep)
    # the accumulation itself is exact to fp32 noise (measured 2.5e-6 of the tensor's largest entry):
    # every outlier against the fp32 oracle is already in the f16 feature gradients
    assert acc_worst <= 3e-5
    # north_star: 1e-3 on grads, relative to each tensor's largest gradient.  Measured
    # (profiles/r03/parity_report.json): EVERY element of every MLP weight gradient <= 5.5e-4; every
    # hash-table entry <= 7.7e-4 at K = 1, 3, 5 and all but 7 of 11 M entries (6e-7 of them, max
    # 2.2e-3) at K = 7.  Those are the tail of the fp16 chain's roundings (dOut, the f16 gradient
    # rows, dH2, dH1, dF: five roundings of 2^-11 each — independent of grad_scale, VSA_TEST_GRAD_SCALE_MULT
    # = 16 / 256 give the same numbers, so no underflow), not of the accumulation (asserted above);
    # tiny-cuda-nn's backward rounds to half at the same places.  Bounds: north_star's where it is
    # met, 2x the measured value where it is not.
    # (r5: K = 9 added.  At the round-4 scale of the f16 chain — 1 x N — its MLP-weight gradients were within
    #  3.6e-3 only and 2.7e-5 of the table entries over 1e-3: the shells behind eight others underflowed f16;
    #  pipeline.GRAD_CHAIN_GAIN = 16 brings the weights to 3.1e-4 and the entries over 1e-3 to 1.3e-6.)
    assert rep["grad_weights_err_rel_to_tensor_max"]["max"] <= 1e-3             # north_star, every element, every K
    model = rep["tcnn_half_backward_model_tables_err_rel_to_tensor_max"]
    if K <= 5:
        assert rep["grad_tables_err_rel_to_tensor_max"]["max"] <= 1e-3          # north_star, every entry
    else:
        # K = 7: 7 of 11 M entries exceed 1e-3 (max 2.2e-3); K = 9: 1.3e-6 of them (max 4.6e-3).  Justified by
        # the reference path's own arithmetic, not by "2x measured" alone: the half-atomics model's worst entry
        # and its share of entries over 1e-3 must both be LARGER than the kernel's (profiles/r05/parity_report.json)
        assert rep["grad_tables_err_rel_to_tensor_max"]["max"] <= min({7: 4.4e-3, 9: 9.2e-3}[K], model["max"])
        assert rep["grad_tables_frac_over_1e-3"] <= min({7: 2e-6, 9: 2.7e-6}[K],
                                                        rep["tcnn_half_backward_model_tables_frac_over_1e-3"])
    assert model["max"] > rep["grad_tables_err_rel_to_tensor_max"]["max"]
    assert model["p99"] > rep["grad_tables_err_rel_to_tensor_max"]["p99"]
    assert rep["grad_tables_err_rel_to_tensor_max"]["p99"] <= 1e-4
