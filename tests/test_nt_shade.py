"""Neural-texture step 5 (per-hit shading) vs the oracle; and the whole shade
stage (mark -> encode -> MLP -> shade) vs the oracle's SHNeuralTextures model."""
import numpy as np
import pytest
import torch

from oracle import neural_texture as ONT

from test_nt_mlp import _bank, unpack_weights


def _scene(K, N, seed, **kw):
    """Random hits on random triangles (geometry only matters for normals)."""
    bank, face_uvs, hit_slot, hit_uv = _bank(K=K, N=N, seed=seed, **kw)
    g = torch.Generator().manual_seed(seed + 100)
    nr_tris = face_uvs.shape[0]
    tris = torch.randn(nr_tris, 12, generator=g).cuda()
    rays_d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).cuda()
    return bank, face_uvs, hit_slot, hit_uv, tris, rays_d


def _oracle_from_texels(bank, s, typ, uv, dirs, texels, slot_of, degrees):
    """Reference op sequence from quantised texels on (neural_texture.py:177-192,
    sh_neural_textures.py:69-95)."""
    C = 3 if typ == 0 else 1
    M = uv.shape[0]
    out = torch.zeros(M, C, degrees ** 2)
    written = 0
    for d in range(degrees):
        R, n = bank.tex_res[d], 2 * d + 1
        W = R + 2
        _, w, corners = ONT.texel_corners(uv.clone(), R)
        ij = torch.floor(corners).long() + 1
        dom = int(bank.plan.dom_off[s * 4 + d])
        slots = slot_of[dom + ij[..., 1] * W + ij[..., 0]].long()          # [M,4]
        assert (slots >= 0).all()
        base = 0 if typ == 0 else 24
        q = texels[slots][..., base:base + C * n]                          # [M,4,C*n] u8
        o = (q.float() / 255.0).half()
        lo, span = float(bank.plan.sh_lo[d]), float(bank.plan.sh_span[d])
        e = lo + span * o
        r = (e * w).sum(dim=1).float().reshape(M, C, n)
        out[:, :, written:written + n] = r
        written += n
    sh = out.half()
    raw = ONT.sh_eval(sh, dirs, degrees - 1)
    return sh.float(), torch.sigmoid(raw).float()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(), dict(alpha_sh_degree=0), dict(inner_solid=True),
                                 dict(with_alpha_decay=False, sh_degree=2, alpha_sh_degree=1)])
def test_shade_fwd_from_texels(cfg):
    bank, face_uvs, hit_slot, hit_uv, tris, rays_d = _scene(2, 3000, 1, **cfg)
    tex_uv = bank.mark_and_compact(hit_slot, hit_uv, face_uvs)
    bank.encode()
    bank.mlp()
    rgb, alpha, normals, coeffs = bank.shade(hit_slot, tex_uv, rays_d, tris, True, True)
    torch.cuda.synchronize()
    texels, slot_of = bank.rows_dense(bank.texels).cpu(), bank.slot_of.cpu()
    K, N = hit_slot.shape
    for s in range(K):
        hit = (hit_slot[s] >= 0).cpu()
        uv = tex_uv[s].cpu()[hit]
        dirs = rays_d.cpu()[hit]
        sh_ref, rgb_ref = _oracle_from_texels(bank, s, 0, uv, dirs, texels, slot_of, bank.rgb_degrees)
        nd = bank.rgb_degrees ** 2
        got = coeffs[s].cpu()[hit][:, :48].reshape(-1, 3, 16)[:, :, :nd]
        assert torch.equal(got, sh_ref)
        np.testing.assert_allclose(rgb[:, s].cpu()[hit].numpy(), rgb_ref.numpy(), atol=2e-6, rtol=0)
        # normals + alpha (volsurfs.py:583-596)
        t = tris.cpu()[hit_slot[s].cpu()[hit].long()]
        nrm = torch.nn.functional.normalize(torch.cross(t[:, 4:7], t[:, 8:11], dim=1), dim=1)
        np.testing.assert_allclose(normals[:, s].cpu()[hit].numpy(), nrm.numpy(), atol=1e-6)
        if cfg.get("inner_solid") and s == 0:
            a_ref = torch.ones(uv.shape[0])
        else:
            sh_a, a_ref = _oracle_from_texels(bank, s, 1, uv, dirs, texels, slot_of, bank.alpha_degrees)
            na = bank.alpha_degrees ** 2
            assert torch.equal(coeffs[s].cpu()[hit][:, 48:48 + na], sh_a[:, 0])
            a_ref = a_ref[:, 0]
            if cfg.get("with_alpha_decay", True):
                a_ref = a_ref * ONT.alpha_decay(dirs, normals[:, s].cpu()[hit])[:, 0]
        np.testing.assert_allclose(alpha[:, s].cpu()[hit].numpy(), a_ref.numpy(), atol=2e-6, rtol=0)
        # misses are zero (volsurfs.py:455-456)
        assert (rgb[:, s].cpu()[~hit] == 0).all() and (alpha[:, s].cpu()[~hit] == 0).all()


@pytest.mark.gpu
def test_shade_stage_end_to_end_vs_oracle_model():
    """Whole stage vs the oracle's SHNeuralTextures forward evaluated per hit (the
    reference's algorithm: 4 network evaluations per hit and degree)."""
    bank, face_uvs, hit_slot, hit_uv, tris, rays_d = _scene(1, 1500, 2)
    tex_uv = bank.mark_and_compact(hit_slot, hit_uv, face_uvs)
    bank.encode()
    bank.mlp()
    rgb, alpha, normals, coeffs = bank.shade(hit_slot, tex_uv, rays_d, tris, True, True)
    torch.cuda.synchronize()
    hit = (hit_slot[0] >= 0).cpu()
    uv, dirs = tex_uv[0].cpu()[hit], rays_d.cpu()[hit]
    for typ, C in ((0, 3), (1, 1)):
        texs = []
        for d in range(4):
            x = bank.tex_index(0, typ, d)
            w1, w2, w3 = unpack_weights(bank.weights_h[x].cpu().float())
            texs.append(ONT.NeuralTextureOracle(bank.tex_res[d], C * (2 * d + 1), (-15, 15),
                                                bank.tables_h[x].cpu().float(), w1, w2, w3))
        ref = ONT.sh_neural_textures_forward(texs, uv, dirs, C, 3)
        if typ == 0:
            got = rgb[:, 0].cpu()[hit]
        else:
            got = alpha[:, 0].cpu()[hit][:, None]
            ref = ref * ONT.alpha_decay(dirs, normals[:, 0].cpu()[hit])
        err = (got - ref).abs()
        # differences come only from 8-bit quantisation flips (|dq| = 1 on ~1e-3 of
        # the texels, each worth 30/255 in one SH coefficient); none elsewhere
        assert (err > 1e-5).float().mean() < 0.05
        assert err.max() < 0.05
        assert err.mean() < 1e-4
