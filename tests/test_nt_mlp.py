"""Neural-texture step 4 (MFMA MLP + sigmoid/quantise) vs the oracle."""
import numpy as np
import pytest
import torch

from oracle import neural_texture as ONT
from oracle import tcnn_like


def _bank(K=2, N=2000, seed=0, res=(2048, 1024, 512, 256), **kw):
    from volsurfs_amd.neural_textures import NeuralTextureBank
    g = torch.Generator().manual_seed(seed)
    nr_tris = 300
    face_uvs = torch.rand(nr_tris, 6, generator=g)
    hit_slot = torch.randint(0, nr_tris, (K, N), generator=g, dtype=torch.int32)
    hit_slot[torch.rand(K, N, generator=g) < 0.2] = -1
    bu = torch.rand(K, N, generator=g)
    bv = torch.rand(K, N, generator=g) * (1 - bu)
    hit_uv = torch.stack([bu, bv], -1)
    bank = NeuralTextureBank(K, N, device="cuda", seed=seed + 1, textures_res=res, **kw)
    with torch.no_grad():
        bank.tables.copy_(torch.rand(bank.tables.shape, generator=g) * 2 - 1)
        w = (torch.rand(bank.weights.shape, generator=g) * 2 - 1) * 0.45
        bank.weights.copy_(w.cuda() * (bank.weights != 0))      # keep the zero padding rows of W3
    bank.refresh_half_params()
    return bank, face_uvs.cuda(), hit_slot.cuda(), hit_uv.cuda()


def unpack_weights(w):
    return w[:2048].view(64, 32), w[2048:6144].view(64, 64), w[6144:].view(32, 64)


@pytest.mark.gpu
def test_mlp_fwd_matches_oracle():
    bank, face_uvs, hit_slot, hit_uv = _bank()
    bank.mark_and_compact(hit_slot, hit_uv, face_uvs)
    bank.encode()
    feats = bank.features_level_major()
    texels, pre = bank.mlp(want_pre=True)
    torch.cuda.synchronize()
    seg = bank.seg_start.cpu().numpy()
    n_checked = 0
    for s in range(bank.K):
        for typ in range(2):
            for d in range(4):
                x = bank.tex_index(s, typ, d)
                C = bank.tex_channels(x)
                a, b = seg[s * 4 + d], seg[s * 4 + d + 1]
                assert b > a
                f = feats[typ, :, a:b].cpu().permute(1, 0, 2).reshape(-1, 32)
                w1, w2, w3 = unpack_weights(bank.weights_h[x].cpu())
                ref = tcnn_like.mlp_forward(w1, w2, w3, f, C).float()
                base = 0 if typ == 0 else 24
                got = pre[a:b, base:base + C].cpu().float()
                # fp32-accumulate MFMA vs fp32 torch matmul: identical up to the
                # summation order -> at most a couple of fp16 ulps after 3 layers
                err = (got - ref).abs()
                tol = 4e-3 * ref.abs() + 2e-3
                assert (err <= tol).all(), (s, typ, d, err.max())
                assert (err > 0).float().mean() < 0.05
                # quantisation of the kernel's own pre-activation: sigmoid, x255, round
                _, q_ref = ONT.quantise(pre[a:b, base:base + C].cpu())
                q = texels[a:b, base:base + C].cpu()
                dq = (q.int() - q_ref.int()).abs()
                assert dq.max() <= 1 and (dq > 0).float().mean() < 1e-3, (s, typ, d)
                n_checked += (b - a) * C
    assert n_checked > 100000


@pytest.mark.gpu
def test_mlp_fwd_asymmetric_weights_exact():
    """Integer-valued asymmetric weights/features: every product and sum is exact
    in fp16/fp32, so any fragment-layout mistake shows up as a hard mismatch."""
    bank, face_uvs, hit_slot, hit_uv = _bank(K=1, N=500, seed=5, res=(64, 32, 16, 8))
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        bank.tables.copy_(torch.randint(-2, 3, bank.tables.shape, generator=g).float())
        w = torch.randint(-2, 3, bank.weights.shape, generator=g).float() * 0.25
        bank.weights.copy_(w.cuda() * (bank.weights != 0))
    bank.refresh_half_params()
    bank.mark_and_compact(hit_slot, hit_uv, face_uvs)
    bank.encode()
    feats = bank.features_level_major()
    # make the features integers too (overwrite with small ints)
    bank.features.copy_(torch.randint(-3, 4, bank.features.shape, generator=g).half())
    feats = bank.features_level_major()
    texels, pre = bank.mlp(want_pre=True)
    seg = bank.seg_start.cpu().numpy()
    for typ in range(2):
        for d in range(4):
            x = bank.tex_index(0, typ, d)
            C = bank.tex_channels(x)
            a, b = seg[d], seg[d + 1]
            f = feats[typ, :, a:b].cpu().permute(1, 0, 2).reshape(-1, 32)
            w1, w2, w3 = unpack_weights(bank.weights_h[x].cpu())
            ref = tcnn_like.mlp_forward(w1, w2, w3, f, C)
            base = 0 if typ == 0 else 24
            assert torch.equal(pre[a:b, base:base + C].cpu(), ref), (typ, d)
