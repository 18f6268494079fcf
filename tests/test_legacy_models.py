"""GridHashEncoder, SHEncoder, MLP / RGB / ColorSH / NerfHash.

* oracle vs fixtures (CPU): oracle/legacy_models.py + oracle/tcnn_like.grid_forward_f32
  reproduce the outputs of the REFERENCE classes (tests/golden/legacy_models.npz was made by
  running volsurfs_py.models.{nerfhash,rgb,color_sh} themselves, tools/make_golden.py
  gen_legacy; the hash tables are re-created here from the generator seed).
* HIP vs oracle (GPU): grid encode fwd bit-exact, bwd vs autograd, SH basis bit-exact vs the
  reference fixture, and the mirror models against the same reference outputs.
"""
import os

import numpy as np
import pytest
import torch

from oracle import legacy_models as OL
from oracle import tcnn_like
from oracle.neural_texture import sh_basis_values

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _fixture():
    d = np.load(os.path.join(GOLD, "legacy_models.npz"))
    g = torch.Generator().manual_seed(6)          # the generator's call order (gen_legacy)
    M = d["points"].shape[0]
    pts = (torch.rand(M, 3, generator=g) * 2 - 1) * 0.9
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1)
    nrm = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1)
    assert np.array_equal(pts.numpy(), d["points"]) and np.array_equal(dirs.numpy(), d["dirs"])
    geom = tcnn_like.GridGeometryND(3, 24, 18, 16, 2)
    tables = [torch.rand(geom.offset[-1], 2, generator=g) * 2 - 1 for _ in range(3)]
    return d, pts, dirs, nrm, geom, tables


def _layers(d, prefix):
    out, i = [], 0
    while f"{prefix}.layers.{i}.weight" in d.files or i < 12:
        k = f"{prefix}.layers.{i}.weight"
        if k in d.files:
            out.append((torch.from_numpy(d[k]), torch.from_numpy(d[f"{prefix}.layers.{i}.bias"])))
        i += 1
    return out


def test_oracle_reproduces_reference_models():
    d, pts, dirs, nrm, geom, (t_nh, t_rgb, t_csh) = _fixture()
    bb2 = torch.tensor([2.0, 2.0, 2.0])
    rgb, dens = OL.nerfhash_forward(geom, t_nh, _layers(d, "nerfhash/mlp_feat_and_density"),
                                    _layers(d, "nerfhash/mlp_rgb"), pts, dirs, bb2)
    np.testing.assert_allclose(rgb.numpy(), d["nerfhash_rgb"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(dens.numpy(), d["nerfhash_density"], rtol=1e-5, atol=2e-6)
    bb1 = torch.tensor([1.0, 1.0, 1.0])
    out = OL.rgb_forward(geom, t_rgb, _layers(d, "rgb/mlp"), pts * 0.5, dirs, nrm, bb1, 3)
    np.testing.assert_allclose(out.numpy(), d["rgb_out"], rtol=0, atol=2e-6)
    # SH basis against the reference's SHEncoder.__call__ fixture
    s = np.load(os.path.join(GOLD, "sh_encoder.npz"))
    for deg in range(4):
        assert np.array_equal(sh_basis_values(torch.from_numpy(s["dirs"]), deg).numpy(), s[f"enc_{deg}"])


def _load_mlp(mlp, d, prefix):
    sd = {k[len(prefix) + 1:]: torch.from_numpy(d[k]) for k in d.files if k.startswith(prefix + ".")}
    mlp.load_state_dict(sd)


@pytest.mark.gpu
@pytest.mark.parametrize("n_dims,levels,log2,growth", [(3, 24, 18, 2.0), (2, 16, 15, 1.5), (3, 6, 10, 1.7)])
def test_grid_encode_matches_oracle(n_dims, levels, log2, growth):
    from volsurfs_amd.encodings import HashGrid
    cfg = {"otype": "Grid", "type": "Hash", "n_levels": levels, "n_features_per_level": 2,
           "log2_hashmap_size": log2, "base_resolution": 16, "per_level_scale": growth}
    enc = HashGrid(n_dims, cfg)
    geom = tcnn_like.GridGeometryND(n_dims, levels, log2, 16, growth)
    assert [enc.plan.level_size[l] for l in range(levels)] == geom.size
    assert [enc.plan.level_res[l] for l in range(levels)] == geom.res
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        enc.params.copy_((torch.rand(enc.params.shape, generator=g) * 2 - 1).cuda())
    x = torch.rand(3000, n_dims, generator=g)
    x[:4] = torch.tensor([0.0, 1.0, 0.5, 0.999999])[:, None]        # edges of the unit cube
    out = enc(x.cuda())
    ref = tcnn_like.grid_forward_f32(geom, enc.params.detach().cpu(), x)
    assert torch.equal(out.cpu(), ref)                               # same fp32 ops, same order
    # backward: transpose of the interpolation (float atomics: order differs)
    go = torch.randn(out.shape, generator=g)
    out.backward(go.cuda())
    t = enc.params.detach().cpu().clone().requires_grad_(True)
    tcnn_like.grid_forward_f32(geom, t, x).backward(go)
    np.testing.assert_allclose(enc.params.grad.cpu().numpy(), t.grad.numpy(), rtol=1e-4,
                               atol=1e-5 * t.grad.abs().max().item())


@pytest.mark.gpu
def test_sh_encoder_matches_reference_fixture():
    from volsurfs_amd.encodings import SHEncoder
    from volsurfs_amd.models import sh_eval
    s = np.load(os.path.join(GOLD, "sh_encoder.npz"))
    dirs = torch.from_numpy(s["dirs"]).cuda()
    for deg in range(4):
        assert np.array_equal(SHEncoder(3, deg)(dirs).cpu().numpy(), s[f"enc_{deg}"])
        # fp32 evaluation (ColorSH): against the oracle's restatement of SHEncoder.eval on
        # fp32 inputs (the fixture's eval_k is the fp16 evaluation of the texture path)
        sh = torch.from_numpy(s[f"sh_{deg}"]).float()
        got = sh_eval(sh.cuda(), dirs, deg).cpu().numpy()
        from oracle.neural_texture import sh_eval as o_sh_eval
        np.testing.assert_allclose(got, o_sh_eval(sh, dirs.cpu(), deg).numpy(), rtol=1e-6, atol=1e-6)
    e4 = SHEncoder(3, 4)(dirs)
    assert e4.shape == (dirs.shape[0], 25) and torch.isfinite(e4).all()


@pytest.mark.gpu
def test_models_match_reference_outputs():
    from volsurfs_amd.models import ColorSH, NerfHash, RGB
    d, pts, dirs, nrm, geom, (t_nh, t_rgb, t_csh) = _fixture()
    pts, dirs, nrm = pts.cuda(), dirs.cuda(), nrm.cuda()
    nh = NerfHash(3, "gridhash", "spherical_harmonics")
    _load_mlp(nh.mlp_feat_and_density, d, "nerfhash/mlp_feat_and_density")
    _load_mlp(nh.mlp_rgb, d, "nerfhash/mlp_rgb")
    with torch.no_grad():
        nh.pos_encoder.encoder.params.copy_(t_nh.cuda())
    rgb, dens = nh(pts, dirs, iter_nr=None)
    np.testing.assert_allclose(rgb.detach().cpu().numpy(), d["nerfhash_rgb"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(dens.detach().cpu().numpy(), d["nerfhash_density"], rtol=1e-5, atol=5e-6)
    # gradients of the fixture's loss: hash table (sparse) and one MLP weight
    loss = (rgb * torch.linspace(0.5, 1.5, 3).cuda()).sum() + 0.3 * dens.sum()
    loss.backward()
    gt = nh.pos_encoder.encoder.params.grad.cpu()
    idx = torch.from_numpy(d["nerfhash_grad/table_idx"])
    ref = torch.from_numpy(d["nerfhash_grad/table_val"])
    np.testing.assert_allclose(gt[idx].numpy(), ref.numpy(), rtol=1e-3, atol=1e-5 * ref.abs().max().item())
    mask = torch.ones(gt.shape[0], dtype=torch.bool)
    mask[idx] = False
    assert gt[mask].abs().max() == 0
    np.testing.assert_allclose(nh.mlp_rgb.layers[0].weight.grad.cpu().numpy(),
                               d["nerfhash_grad/mlp_rgb0"], rtol=1e-3, atol=1e-5)

    m = RGB(3, [128, 128, 64], "gridhash", "spherical_harmonics", sh_deg=3, normal_dep=True, bb_sides=1.0)
    _load_mlp(m.mlp, d, "rgb/mlp")
    with torch.no_grad():
        m.pos_encoder.encoder.params.copy_(t_rgb.cuda())
    out = m(points=pts * 0.5, samples_dirs=dirs, normals=nrm, iter_nr=None)
    np.testing.assert_allclose(out.detach().cpu().numpy(), d["rgb_out"], rtol=0, atol=5e-6)

    c = ColorSH(3, [128, 128, 64], "gridhash", sh_deg=3, bb_sides=1.0)
    _load_mlp(c.mlp, d, "colorsh/mlp")
    with torch.no_grad():
        c.pos_encoder.encoder.params.copy_(t_csh.cuda())
    np.testing.assert_allclose(c(pts * 0.45).detach().cpu().numpy(), d["colorsh_coeffs"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(c(pts * 0.45, samples_dirs=dirs).detach().cpu().numpy(), d["colorsh_out"],
                               rtol=0, atol=2e-5)


@pytest.mark.gpu
def test_nerfhash_drives_the_background_path():
    """VolSurfs(bg_color=None, bg_model=NerfHash): the packed bg ops + the field, fwd + bwd."""
    from volsurfs_amd.background import render_contracted_bg
    from volsurfs_amd.models import NerfHash
    nh = NerfHash(3, "gridhash", "spherical_harmonics")
    g = torch.Generator().manual_seed(0)
    N = 500
    raycast = {"rays_o": torch.zeros(N, 3).cuda(),
               "rays_d": torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).cuda(),
               "t_far": torch.full((N, 1), 0.5).cuda()}
    out = render_contracted_bg(nh, raycast, 16, jitter_samples=False, iter_nr=None)
    rgb = out["pred_rgb"]
    assert rgb.shape == (N, 3) and torch.isfinite(rgb).all()
    rgb.sum().backward()
    assert nh.pos_encoder.encoder.params.grad.abs().sum() > 0
    assert nh.mlp_rgb.layers[0].weight.grad.abs().sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1000, 300_000])
def test_nerfhash_fused_glue_equals_the_torch_op_sequence(n):
    """nerfhash.py:72-91 as the reference writes it (slice, gelu, cat, softplus; GridHashEncoder's
    torch.cat of the points) against the fused kernels (csrc/field_head.hip, vsa_grid_encode_*_ld):
    outputs and every parameter gradient.  300 k samples take the sliced table-gradient path."""
    from volsurfs_amd.encodings import GridHashEncoder
    from volsurfs_amd.models import NerfHash
    torch.manual_seed(3)
    nh = NerfHash(3, "gridhash", "spherical_harmonics")
    with torch.no_grad():      # tables at a scale where the features matter
        nh.pos_encoder.encoder.params.mul_(3e3)
    g = torch.Generator().manual_seed(1)
    pts = (torch.rand(n, 3, generator=g) * 1.6 - 0.8).cuda()
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
    w_rgb = torch.rand(n, 3, generator=g).cuda()
    w_den = torch.rand(n, 1, generator=g).cuda()

    def run(fused):
        NerfHash.fused_head = GridHashEncoder.fused_concat = fused
        try:
            nh.zero_grad(set_to_none=True)
            rgb, dens = nh(pts, dirs, iter_nr=None)
            ((rgb * w_rgb).sum() + (dens * w_den).sum()).backward()
            return rgb.detach(), dens.detach(), [p_.grad.clone() for p_ in nh.parameters()]
        finally:
            NerfHash.fused_head = GridHashEncoder.fused_concat = True

    rgb_a, den_a, g_a = run(True)
    rgb_b, den_b, g_b = run(False)
    # the same arithmetic term for term; torch's own GELU / softplus kernels may contract differently
    np.testing.assert_allclose(rgb_a.cpu().numpy(), rgb_b.cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(den_a.cpu().numpy(), den_b.cpu().numpy(), rtol=2e-6, atol=1e-7)
    for (name, _), ga, gb in zip(nh.named_parameters(), g_a, g_b):
        scale = gb.abs().max().item()
        assert scale > 0, name
        np.testing.assert_allclose(ga.cpu().numpy(), gb.cpu().numpy(), rtol=2e-4, atol=2e-6 * scale, err_msg=name)


@pytest.mark.gpu
@pytest.mark.parametrize("n_dims,levels,log2,growth", [(3, 24, 18, 2.0), (2, 16, 15, 1.5)])
def test_grid_encode_backward_sliced_and_binned_equal_atomic_scatter(n_dims, levels, log2, growth):
    """vsa_grid_encode_bwd_sliced (LDS-resident table slices, large batches) accumulates the
    same table gradients as vsa_grid_encode_bwd (memory-side float atomics), itself pinned to
    the oracle above; exercised through the size switch of encodings._GridEncode."""
    from volsurfs_amd import encodings as E
    cfg = {"otype": "Grid", "type": "Hash", "n_levels": levels, "n_features_per_level": 2,
           "log2_hashmap_size": log2, "base_resolution": 16, "per_level_scale": growth}
    enc = E.HashGrid(n_dims, cfg)
    g = torch.Generator().manual_seed(2)
    B = 300001
    x = torch.rand(B, n_dims, generator=g).cuda()
    go = torch.randn(B, 2 * levels, generator=g).cuda()
    go[::7] = 0                                             # rows the kernels skip
    grads = {}
    keep = (E.SLICED_BWD_MIN_POINTS, E.BINNED_BWD_MIN_POINTS)
    for name, thr in (("sliced", (1, 1 << 30)), ("binned", (1, 1)), ("atomic", (1 << 30, 1 << 30))):
        E.SLICED_BWD_MIN_POINTS, E.BINNED_BWD_MIN_POINTS = thr
        try:
            enc.params.grad = None
            enc(x).backward(go)
        finally:
            E.SLICED_BWD_MIN_POINTS, E.BINNED_BWD_MIN_POINTS = keep
        grads[name] = enc.params.grad.clone()
    b = grads["atomic"]
    assert b.abs().max() > 0
    for name in ("sliced", "binned"):
        a = grads[name]
        assert (a != 0).sum() == (b != 0).sum(), name
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-5 * b.abs().max().item())
    # both LDS paths accumulate in 64-bit fixed point: independent of the summation order, i.e. equal
    # to each other bit for bit up to the final per-chunk float adds of the sliced path
    np.testing.assert_allclose(grads["sliced"].cpu().numpy(), grads["binned"].cpu().numpy(), rtol=1e-6,
                               atol=1e-7 * b.abs().max().item())
