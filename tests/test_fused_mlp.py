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
    assert L.vsa_mlp_fwd(None, None, 0, 0, None, 0, None, None, None, None) == -1
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


@pytest.mark.gpu
@pytest.mark.parametrize("sizes", [[9000, 0, 12001, 700, 10000], [300] * 9 + [0, 41]])
def test_grouped_launch_equals_the_networks_one_by_one(sizes):
    """vsa_mlp_fwd_grouped / vsa_mlp_bwd_grouped (group = blockIdx.y, up to 8 per launch; 11 groups
    = two launches) against one vsa_mlp_fwd / vsa_mlp_bwd per network: outputs bit for bit (the
    tiles and their arithmetic are the same), gradients to 1e-5 of their scale (the weight-gradient
    workgroups share the co-resident budget between the groups, so the partial sums are grouped
    differently)."""
    from volsurfs_amd.models import MLP, fused_mlp_grouped
    torch.manual_seed(11)
    mlps = [MLP(66, [128, 128, 64, 3], last_layer_linear=True).cuda() for _ in sizes]
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn(sum(sizes), 66, device="cuda", generator=g)
    gy = torch.randn(sum(sizes), 3, device="cuda", generator=g)

    xa = x.clone().requires_grad_(True)
    ya = fused_mlp_grouped(mlps, xa, sizes)
    ya.backward(gy)
    ga = [p_.grad.clone() if p_.grad is not None else None for m in mlps for p_ in m.parameters()]
    for m in mlps:
        m.zero_grad(set_to_none=True)

    xb = x.clone().requires_grad_(True)
    outs, a = [], 0
    for m, n in zip(mlps, sizes):
        outs.append(m(xb[a:a + n]))
        a += n
    yb = torch.cat(outs, 0)
    yb.backward(gy)
    gb = [p_.grad if p_.grad is not None else None for m in mlps for p_ in m.parameters()]

    assert torch.equal(ya, yb)
    np.testing.assert_allclose(xa.grad.cpu().numpy(), xb.grad.cpu().numpy(), rtol=1e-5, atol=1e-6)
    for (m, n) in zip(mlps, sizes):
        pass
    k = 0
    for m, n in zip(mlps, sizes):
        for p_ in m.parameters():
            a_, b_ = ga[k], gb[k]
            k += 1
            if n == 0:
                assert a_ is None or float(a_.abs().max()) == 0.0
                assert b_ is None or float(b_.abs().max()) == 0.0
                continue
            scale = float(b_.abs().max())
            np.testing.assert_allclose(a_.cpu().numpy(), b_.cpu().numpy(), rtol=1e-4, atol=1e-5 * scale)


@pytest.mark.gpu
def test_mlp_outside_the_fused_shapes_raises_unless_the_torch_path_is_asked_for():
    """no silent library fallback on the product path."""
    from volsurfs_amd import _lib
    from volsurfs_amd.models import MLP
    m = MLP(40, [200, 3], last_layer_linear=True).cuda()          # wider than the kernel's 128
    x = torch.randn(64, 40, device="cuda")
    with pytest.raises(_lib.VolsurfsHipError, match="fused = False"):
        m(x)
    m.fused = False                                                 # explicit, per instance
    assert m(x).shape == (64, 3)
    g = MLP(40, [64, 3], last_layer_linear=False).cuda()            # GELU after the last layer
    with pytest.raises(_lib.VolsurfsHipError):
        g(x)
