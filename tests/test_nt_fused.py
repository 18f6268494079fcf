"""The fused encode + MLP launch (csrc/nt_fused.hip, vsa_nt_encode_mlp_fwd) against the two-kernel
path it replaces (vsa_nt_encode_fwd + vsa_nt_mlp_fwd): features, pre-activations and quantised
texel rows must be bit-identical — same arithmetic, operation for operation — for every texture,
ragged segment lengths, empty segments, an inner-solid bank and a 9-shell bank (72 textures: the
work split's second batch of 64)."""
import pytest
import torch

from test_nt_mlp import _bank


def _both(bank, face_uvs, hit_slot, hit_uv):
    bank.mark_and_compact(hit_slot, hit_uv, face_uvs)
    bank.features.zero_()
    bank.texels.zero_()
    bank.encode()
    texels_ref, pre_ref = bank.mlp(want_pre=True)
    feats_ref = bank.features.clone()
    raw_ref = bank.texels.clone()
    torch.cuda.synchronize()
    return feats_ref, raw_ref, texels_ref, pre_ref


def _used_feature_mask(bank):
    """[2, cap] bool: (type, slot) pairs a texture of that type really owns."""
    seg = bank.seg_start.cpu().tolist()
    m = torch.zeros(2, bank.slot_capacity, dtype=torch.bool)
    for s in range(bank.K):
        for d in range(4):
            for typ in range(2):
                if bank.tex_channels(bank.tex_index(s, typ, d)):
                    m[typ, seg[s * 4 + d]:seg[s * 4 + d + 1]] = True
    return m.cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("K,N,res,kw", [
    (2, 2000, (2048, 1024, 512, 256), {}),
    (1, 37, (64, 32, 16, 8), {}),                                   # a handful of ragged tiles
    (3, 5000, (256, 128, 64, 32), {"inner_solid": True}),           # shell 0 has no alpha textures
    (9, 700, (128, 64, 32, 16), {}),                                # 72 textures
    (2, 3000, (1024, 512, 256, 128), {"sh_degree": 2, "alpha_sh_degree": 1}),
])
def test_fused_forward_is_bit_identical_to_encode_then_mlp(K, N, res, kw):
    bank, face_uvs, hit_slot, hit_uv = _bank(K=K, N=N, seed=3, res=res, **kw)
    feats_ref, raw_ref, texels_ref, pre_ref = _both(bank, face_uvs, hit_slot, hit_uv)
    used = _used_feature_mask(bank)
    for write_features in (True, False):
        bank.features.fill_(7.0)
        bank.texels.zero_()
        texels, pre = bank.encode_mlp(write_features=write_features, want_pre=True)
        torch.cuda.synchronize()
        assert torch.equal(bank.texels, raw_ref), "quantised texel rows differ"
        assert torch.equal(texels, texels_ref)
        assert torch.equal(pre.view(torch.int16), pre_ref.view(torch.int16)), "pre-activations differ"
        f = bank.features.permute(0, 2, 1, 3, 4).reshape(2, 16, bank.slot_capacity, 2)
        fr = feats_ref.permute(0, 2, 1, 3, 4).reshape(2, 16, bank.slot_capacity, 2)
        if write_features:
            a = f.view(torch.int16).permute(0, 2, 1, 3)[used]
            b = fr.view(torch.int16).permute(0, 2, 1, 3)[used]
            assert torch.equal(a, b), "feature planes differ"
            # nothing outside the owned (type, slot) pairs is written
            assert (f.permute(0, 2, 1, 3)[~used] == 7.0).all()
        else:
            assert (bank.features == 7.0).all(), "write_features=False must not touch the planes"


@pytest.mark.gpu
def test_fused_forward_no_hits_is_a_no_op():
    bank, face_uvs, hit_slot, hit_uv = _bank(K=2, N=100, seed=1, res=(64, 32, 16, 8))
    hit_slot.fill_(-1)
    bank.mark_and_compact(hit_slot, hit_uv, face_uvs)
    bank.texels.fill_(3)
    bank.encode_mlp()
    torch.cuda.synchronize()
    assert int(bank.seg_start[-1]) == 0 and (bank.texels == 3).all()


@pytest.mark.gpu
def test_fused_forward_full_frame_equals_two_kernel_path():
    """BASELINE configs[1]'s frame (800x800, K=5): 5.7 M unique texels through both paths."""
    from volsurfs_amd.pipeline import KShellPipeline
    pipe = KShellPipeline.synthetic(K=5, subdiv=6, res=800, device="cuda", init="spread")
    bank = pipe.bank
    hit_t, hit_slot, hit_uv = pipe.tracer.trace_all(pipe.rays_o, pipe.rays_d)
    bank.mark_and_compact(hit_slot, hit_uv, pipe.face_uvs)
    bank.encode()
    bank.mlp()
    feats_ref, raw_ref = bank.features.clone(), bank.texels.clone()
    bank.features.zero_()
    bank.texels.zero_()
    bank.encode_mlp(write_features=True)
    torch.cuda.synchronize()
    assert int(bank.seg_start[-1]) > 5_000_000
    assert torch.equal(bank.texels, raw_ref)
    used = _used_feature_mask(bank)
    f = bank.features.permute(0, 2, 1, 3, 4).reshape(2, 16, bank.slot_capacity, 2).view(torch.int16)
    fr = feats_ref.permute(0, 2, 1, 3, 4).reshape(2, 16, bank.slot_capacity, 2).view(torch.int16)
    assert torch.equal(f.permute(0, 2, 1, 3)[used], fr.permute(0, 2, 1, 3)[used])
