"""Oracle restatement of the SH neural-texture model vs fixtures produced by the
reference's own SHNeuralTextures / NeuralTexture / SHEncoder classes."""
import os

import numpy as np
import pytest
import torch

from oracle import neural_texture as NT


def _model(z):
    C, sh_deg, seed = int(z["nr_channels"]), int(z["sh_deg"]), int(z["seed"])
    params = NT.make_test_params(seed, C, sh_deg)
    res = [2048, 1024, 512, 256]
    texs = []
    leaves = []
    for deg, (table, w1, w2, w3) in enumerate(params):
        # guard against RNG drift between the generating and the testing torch
        assert abs(table.double().sum().item() - z[f"param_sum_{deg}"][0]) < 1e-6
        assert abs(w3.double().sum().item() - z[f"param_sum_{deg}"][1]) < 1e-6
        ps = [t.clone().requires_grad_(True) for t in (table, w1, w2, w3)]
        leaves.append(ps)
        flags = dict(zip(("anchor", "lerp", "quantize_output", "squeeze_output"), z["flags"].tolist())) \
            if "flags" in z.files else {}
        texs.append(NT.NeuralTextureOracle(res[deg], C * (2 * deg + 1), (-15, 15), *ps, **flags))
    return texs, leaves, C, sh_deg


# (r5: the reference classes were also run with anchor=True, with squeeze but no quantisation, and with neither)
@pytest.mark.parametrize("name", ["rgb", "alpha", "alpha_deg0", "rgb_anchor", "alpha_anchor", "rgb_noquant", "rgb_raw"])
def test_restatement_matches_reference_classes(golden_dir, name):
    z = np.load(os.path.join(golden_dir, f"sh_neural_textures_{name}.npz"))
    texs, leaves, C, sh_deg = _model(z)
    uv, dirs = torch.from_numpy(z["uv"]), torch.from_numpy(z["dirs"])
    coeffs = NT.sh_neural_textures_forward(texs, uv, None, C, sh_deg)
    assert np.array_equal(coeffs.detach().numpy(), z["coeffs"])
    out = NT.sh_neural_textures_forward(texs, uv, dirs, C, sh_deg)
    assert np.array_equal(out.detach().numpy(), z["out"])
    loss = (torch.from_numpy(z["gt"]) - out).abs().mean()
    loss.backward()
    for deg in range(sh_deg + 1):
        table, w1, w2, w3 = leaves[deg]
        for wn, w in (("w1", w1), ("w2", w2), ("w3", w3)):
            assert np.array_equal(w.grad.numpy(), z[f"g_{wn}_{deg}"]), (deg, wn)
        idx = torch.from_numpy(z[f"g_table_top_idx_{deg}"])
        assert np.array_equal(table.grad[idx].numpy(), z[f"g_table_top_val_{deg}"])
        sums = z[f"g_table_sums_{deg}"]
        assert abs(table.grad.double().sum().item() - sums[0]) <= 1e-9 + 1e-6 * abs(sums[1])


def test_sh_eval_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "sh_encoder.npz"))
    dirs = torch.from_numpy(z["dirs"])
    for deg in range(4):
        sh = torch.from_numpy(z[f"sh_{deg}"]).half()
        got = NT.sh_eval(sh, dirs, deg).float().numpy()
        assert np.array_equal(got, z[f"eval_{deg}"])


def test_expand_lut_is_the_fp16_expansion():
    lut = NT.expand_lut((-15, 15))
    assert lut.dtype == torch.float16 and lut.shape == (256,)
    assert lut[0].item() == -15.0 and lut[255].item() == 15.0
    # monotone; 8-bit steps of 30/255
    assert (lut[1:].float() >= lut[:-1].float()).all()
