"""Reference-format checkpoints (per-model state_dicts with tiny-cuda-nn `params` vectors,
base_method.py:118-211) <-> the stacked tables / weights (volsurfs_amd/checkpoint.py)."""
import os

import numpy as np
import pytest
import torch


def _ref_sd(C_per_coeff, degrees, seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for d in range(degrees):
        C = C_per_coeff * (2 * d + 1)
        enc = (torch.rand(354184 * 2, generator=g) * 2 - 1) * 0.5
        net = (torch.rand(6144 + 64 * ((C + 15) // 16 * 16), generator=g) * 2 - 1) * 0.3
        for k, v in (("encoding", enc), ("network", net)):
            sd[f"neural_textures.{d}.{k}.params"] = v
        sd[f"neural_textures.{d}.model.0.params"] = enc          # Sequential alias, as torch saves it
        sd[f"neural_textures.{d}.model.1.params"] = net
    return sd


def test_reference_state_dict_round_trip_cpu():
    from volsurfs_amd.checkpoint import (is_reference_state_dict, load_reference_state_dict,
                                         to_reference_state_dict)
    from volsurfs_amd.neural_textures import NeuralTextureBank
    bank = NeuralTextureBank(2, 64, device="cpu", textures_res=(64, 32, 16, 8), alpha_sh_degree=1)
    before = bank.tables.detach().clone()
    sd_rgb, sd_a = _ref_sd(3, 4, 1), _ref_sd(1, 2, 2)
    assert is_reference_state_dict(sd_rgb) and not is_reference_state_dict({"tables": 0, "weights": 0})
    assert load_reference_state_dict(bank, 1, 0, sd_rgb) == [0, 1, 2, 3]
    assert load_reference_state_dict(bank, 1, 1, sd_a) == [0, 1]
    assert torch.equal(bank.tables[:8], before[:8])                              # shell 0 untouched
    for d in range(4):
        x = bank.tex_index(1, 0, d)
        C = 3 * (2 * d + 1)
        assert torch.equal(bank.tables[x].reshape(-1), sd_rgb[f"neural_textures.{d}.encoding.params"])
        net = sd_rgb[f"neural_textures.{d}.network.params"]
        assert torch.equal(bank.weights[x][:6144 + 64 * C], net[:6144 + 64 * C])
        assert bank.weights[x][6144 + 64 * C:].abs().sum() == 0                   # padding rows cleared
        # layer views: W1[64,32] | W2[64,64] | W3[C,64] row-major [out, in]
        assert torch.equal(bank.weights[x][:2048].view(64, 32)[5], net[5 * 32:6 * 32])
    back = to_reference_state_dict(bank, 1, 0)
    assert set(back) == set(sd_rgb)
    for d in range(4):
        C = 3 * (2 * d + 1)
        assert torch.equal(back[f"neural_textures.{d}.encoding.params"], sd_rgb[f"neural_textures.{d}.encoding.params"])
        assert torch.equal(back[f"neural_textures.{d}.network.params"][:6144 + 64 * C],
                           sd_rgb[f"neural_textures.{d}.network.params"][:6144 + 64 * C])
    # wrong sizes / missing degrees are errors, not silent skips
    bad = dict(sd_rgb)
    bad["neural_textures.2.network.params"] = torch.zeros(10)
    with pytest.raises(ValueError):
        load_reference_state_dict(bank, 0, 0, bad)
    with pytest.raises(KeyError):
        load_reference_state_dict(bank, 0, 0, {k: v for k, v in sd_rgb.items() if ".3." not in k})
    with pytest.raises(ValueError):
        load_reference_state_dict(bank, 0, 1, sd_rgb)       # 4 degrees into an alpha model that has 2


@pytest.mark.gpu
def test_reference_checkpoint_renders_like_the_oracle_with_the_same_params(tmp_path):
    """A checkpoint directory in the reference's layout (<iter>/models/rgb_i.pt, alpha_i.pt as
    SHNeuralTextures state_dicts) loaded through VolSurfs.load: the HIP render equals the oracle
    evaluating tcnn-ordered parameters (oracle/tcnn_like.py: the stated layout assumption)."""
    from oracle import pipeline as opipe
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    K = 2
    meshes = nested_shells(K=K, subdiv=2)
    m = VolSurfs(meshes, max_rays=4096, textures_res=(128, 64, 32, 16))
    path = os.path.join(tmp_path, format(7, "07d"), "models")
    os.makedirs(path)
    sds = {}
    for i in range(K):
        sds[f"rgb_{i}"], sds[f"alpha_{i}"] = _ref_sd(3, 4, 10 + i), _ref_sd(1, 4, 20 + i)
        torch.save(sds[f"rgb_{i}"], os.path.join(path, f"rgb_{i}.pt"))
        torch.save(sds[f"alpha_{i}"], os.path.join(path, f"alpha_{i}.pt"))
    m.load_checkpoints_path = str(tmp_path)
    m.load(7)
    o, d = pinhole_rays(40, 40, focal=64.0)
    with torch.no_grad():
        rgb = m.render_rays(o, d, return_samples=False)["renders"]["ray_traced"]["rgb"].cpu().numpy()
    # oracle from the raw reference-format vectors (f16-rounded, as tcnn computes)
    tables = torch.zeros(K * 8, 354184, 2)
    weights = torch.zeros(K * 8, 8192)
    for i in range(K):
        for typ, name in ((0, "rgb"), (1, "alpha")):
            for dg in range(4):
                x = (i * 2 + typ) * 4 + dg
                C = (3, 1)[typ] * (2 * dg + 1)
                sd = sds[f"{name}_{i}"]
                tables[x] = sd[f"neural_textures.{dg}.encoding.params"].view(-1, 2).half().float()
                net = sd[f"neural_textures.{dg}.network.params"]
                weights[x, :6144 + 64 * C] = net[:6144 + 64 * C].half().float()
    ms = [(q.vertices.cpu().numpy(), q.faces.cpu().numpy(), q.faces_uvs.cpu()) for q in meshes]
    ref = opipe.render_step(ms, tables, weights, m.bank.tex_index, (128, 64, 32, 16), o.cpu().numpy(),
                            d.cpu().numpy(), torch.zeros(1600, 3), backward=False)
    e = np.abs(rgb - ref["rgb"])
    assert ref["hit"].sum() > 400
    assert np.median(e) == 0.0 and (e <= 1e-4).mean() > 0.9 and e.max() < 0.05
