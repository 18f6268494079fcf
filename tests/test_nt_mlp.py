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


def test_quant_table_header_is_the_generated_one():
    """volsurfs_amd/csrc/nt_quant_table.h == tools/gen_quant_table.py's output on this machine
    (the reference's sigmoid -> x255 -> round evaluated by torch-CPU for every fp16)."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_quant_table.py"), "--check"])
    assert r.returncode == 0


@pytest.mark.gpu
def test_quantise_exhaustive():
    """The kernel's 8-bit texel for EVERY finite fp16 network output equals the reference's
    round(sigmoid(float(x)) * 255) (neural_texture.py:155-169, torch-CPU): the network is set
    to the identity (x -> relu(x) - relu(-x)) and the 63 488 finite fp16 patterns are fed as
    features."""
    bank, face_uvs, hit_slot, hit_uv = _bank(K=1, N=30000, seed=11)
    bank.mark_and_compact(hit_slot, hit_uv, face_uvs)
    seg = bank.seg_start.cpu().numpy()
    a, b = int(seg[0]), int(seg[1])
    bits = np.arange(65536, dtype=np.uint16)
    vals = bits.view(np.float16)
    pat = torch.from_numpy(vals[np.isfinite(vals)].copy())
    n = pat.numel()
    assert n == 63488 and b - a >= n
    with torch.no_grad():
        bank.weights.zero_()
        for typ, C in ((0, 3), (1, 1)):
            w = bank.weights[bank.tex_index(0, typ, 0)]
            w1, w2, w3 = unpack_weights(w)
            w1[0, 0], w1[1, 0] = 1.0, -1.0
            w2[0, 0] = w2[1, 1] = 1.0
            w3[:C, 0], w3[:C, 1] = 1.0, -1.0
    bank.refresh_half_params()
    bank.encode()
    f = bank.features_level_major()                       # [type, level, slot, 2] (a copy)
    f.zero_()
    f[:, 0, a:a + n, 0] = pat.cuda()
    bank.features.copy_(f.reshape(2, 16, -1, 256, 2).permute(0, 2, 1, 3, 4))
    texels, pre = bank.mlp(want_pre=True)
    torch.cuda.synchronize()
    x = pat.float()
    q_ref = torch.round(torch.sigmoid(x) * 255.0).to(torch.uint8)
    for base, C in ((0, 3), (24, 1)):
        for c in range(C):
            got_pre = pre[a:a + n, base + c].cpu()
            assert torch.equal(got_pre.float(), x), "identity network must reproduce its input"
            q = texels[a:a + n, base + c].cpu()
            bad = (q != q_ref).nonzero()[:, 0]
            assert bad.numel() == 0, [(float(x[i]), int(q[i]), int(q_ref[i])) for i in bad[:10]]
