"""A5 / A10: the fused fp32 matrix-core MLP (csrc/mlp_f32.hip) vs the reference's own op
sequence — torch.nn.Linear + exact GELU (models/mlp.py:8-69), which tests/golden/legacy_models.npz
pins to the reference classes."""
import ctypes

import numpy as np
import pytest
import torch


def test_mlp_plan_layout_and_argument_checks():
    import re
    from volsurfs_amd import _lib
    from volsurfs_amd.models import MlpGrads, MlpPlan, fused_mlp_supported
    hdr = open(_lib.HEADER_PATH).read()
    assert "#define VSA_MLP_MAX_LAYERS 6" in hdr
    # vsa_mlp_grads: 6 + 6 pointers, then the int32 `accumulate` flag (padded to 8)
    assert ctypes.sizeof(MlpPlan) == 4 + 7 * 4 + 6 * 8 + 6 * 8 and ctypes.sizeof(MlpGrads) == 96 + 8
    assert MlpGrads.accumulate.offset == 96 and "int32_t accumulate;" in hdr
    body = re.search(r"typedef struct vsa_mlp_plan \{(.*?)\} vsa_mlp_plan;", hdr, re.S).group(1)
    assert re.findall(r"(\w+)(?:\[[^\]]*\])?;", body) == [f[0] for f in MlpPlan._fields_]
    L = _lib.lib()
    p = MlpPlan()
    p.n_layers = 2
    p.dims[0], p.dims[1], p.dims[2] = 66, 48, 3            # hidden width not a multiple of 32
    p.w[0] = p.w[1] = 1
    sz = ctypes.c_longlong()
    assert L.vsa_mlp_workspace(ctypes.byref(p), ctypes.c_longlong(10), ctypes.byref(sz), None, None) == -2
    p.dims[1] = 256                                         # wider than 128
    assert L.vsa_mlp_workspace(ctypes.byref(p), ctypes.c_longlong(10), ctypes.byref(sz), None, None) == -2
    p.n_layers = 7
    assert L.vsa_mlp_workspace(ctypes.byref(p), ctypes.c_longlong(10), ctypes.byref(sz), None, None) == -1
    assert L.vsa_mlp_fwd(None, None, 0, 0, None, 0, None, None, None) == -1
    x = torch.zeros(4, 66)
    assert not fused_mlp_supported([66, 128, 3], x)         # CPU tensors take the torch path
    assert not fused_mlp_supported([66, 48, 3], x.to(torch.float64))


CONFIGS = [([66, 128, 128, 64, 3], True),      # RGB (models/rgb.py:139), rgb head
           ([66, 128, 128, 64, 1], True),      # alpha head
           ([51, 64, 64, 64, 65], True),       # NerfHash feat + density (models/nerfhash.py:44-50)
           ([80, 64, 64, 3], True),            # NerfHash rgb
           ([53, 128, 128, 64, 48], True),     # ColorSH: 3 x 16 SH coefficients
           ([55, 32, 32, 3], True),            # BASELINE configs[0]: frequency(39) + SH(16) -> [32, 32]
           ([50, 3], True),                    # a single linear layer
           ([37, 96, 5], False)]               # no bias


@pytest.mark.gpu
@pytest.mark.parametrize("dims,bias", CONFIGS)
@pytest.mark.parametrize("M", [1, 33, 70001])
def test_fused_mlp_matches_torch_forward_and_backward(dims, bias, M):
    from volsurfs_amd.models import MLP
    torch.manual_seed(len(dims) * 1000 + M)
    m = MLP(dims[0], dims[1:], last_layer_linear=True, bias=bias).cuda()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(2.0)                                     # activations well inside GELU's curved part
    x = torch.randn(M, dims[0], device="cuda", requires_grad=True)
    gy = torch.randn(M, dims[-1], device="cuda")
    res = {}
    for fused in (True, False):
        MLP.fused = fused
        try:
            for p in m.parameters():
                p.grad = None
            x.grad = None
            y = m(x)
            y.backward(gy)
        finally:
            MLP.fused = True
        res[fused] = (y.detach().cpu().numpy(), x.grad.cpu().numpy(),
                      [p.grad.cpu().numpy() for p in m.parameters()])
    yf, yt = res[True][0], res[False][0]
    np.testing.assert_allclose(yf, yt, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(yt).max())))
    np.testing.assert_allclose(res[True][1], res[False][1], rtol=1e-4,
                               atol=1e-5 * float(np.abs(res[False][1]).max() + 1e-30))
    for a, b in zip(res[True][2], res[False][2]):
        assert a.shape == b.shape
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=2e-5 * float(np.abs(b).max() + 1e-30))


@pytest.mark.gpu
def test_fused_mlp_inference_and_frozen_inputs():
    """no_grad forward saves nothing; inputs without grad (the SH-encoded directions) get none."""
    from volsurfs_amd.models import MLP
    torch.manual_seed(0)
    m = MLP(80, [64, 64, 3], last_layer_linear=True).cuda()
    x = torch.randn(5000, 80, device="cuda")
    with torch.no_grad():
        y0 = m(x)
    y1 = m(x)
    assert torch.equal(y0, y1) and y1.requires_grad
    y1.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    MLP.fused = False
    try:
        y2 = m(x)
    finally:
        MLP.fused = True
    assert (y1 - y2).abs().max() < 1e-5
    assert m(torch.zeros(0, 80, device="cuda")).shape == (0, 3)
