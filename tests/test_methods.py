"""Reference-shaped host API (VolSurfs.render_rays / forward / render)."""
import numpy as np
import pytest
import torch

CONFIG0_GRAD_REL_MAX = 2e-6      # 2x measured: 7.7e-7 (fp32 models; profiles/r03/measured_bounds.txt)


def _method(K=2, max_rays=4096):
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    meshes = nested_shells(K=K, subdiv=3)
    m = VolSurfs(meshes, max_rays=max_rays, textures_res=(256, 128, 64, 32))
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        m.bank.tables.copy_((torch.rand(m.bank.tables.shape, generator=g) * 2 - 1).cuda())
    m.bank.refresh_half_params()
    return m


@pytest.mark.gpu
def test_render_rays_dict_matches_reference_contract():
    from volsurfs_amd.camera import pinhole_rays
    m = _method()
    o, d = pinhole_rays(48, 48, focal=80.0)
    res = m.render_rays(o, d, iter_nr=0)
    rt = res["renders"]["ray_traced"]
    N, K = 48 * 48, 2
    shapes = {"rgb": (N, 3), "rgb_fg": (N, 3), "rgb_bg": (N, 3), "surfs_alpha": (N, K, 1),
              "surfs_rgb": (N, K, 3), "surfs_normals": (N, K, 3),
              "surfs_blending_weights": (N, K, 1), "bg_transmittance": (N, 1), "surfs_uvs": (N, K, 2)}
    assert set(rt) == set(shapes)                       # volsurfs.py:738-751
    for k, s in shapes.items():
        assert tuple(rt[k].shape) == s and rt[k].dtype == torch.float32, k
    nh = int((rt["surfs_normals"].abs().sum(-1) > 0).sum())
    assert res["samples_3d"].shape == (nh, 3) and res["samples_grad"].shape == (nh, 3)
    # rays missing everything show the background
    miss = (rt["surfs_alpha"].sum((1, 2)) == 0)
    assert miss.any() and torch.equal(rt["rgb"][miss], torch.ones_like(rt["rgb"][miss]))


@pytest.mark.gpu
def test_forward_backward_and_adam_step_reduce_the_loss():
    from volsurfs_amd.camera import pinhole_rays
    m = _method()
    opt = m.init_optim()
    o, d = pinhole_rays(64, 64, focal=110.0)
    gt = torch.rand(64 * 64, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) * 0.2
    m.grad_scale = float(64 * 64)
    losses = []
    for it in range(12):
        opt.zero_grad()
        l, _, pts = m(o, d, gt, None, it)
        l["loss"].backward()
        assert m.bank.tables.grad.abs().sum() > 0 and m.bank.weights.grad.abs().sum() > 0
        m.optim_step()
        losses.append(l["loss"].item())
    assert losses[-1] < losses[0] - 1e-3, losses


@pytest.mark.gpu
def test_render_chunks_equal_one_shot():
    from volsurfs_amd.camera import pinhole_rays
    m = _method(max_rays=4096)
    o, d = pinhole_rays(64, 64, focal=110.0)
    full = m.render(o, d, chunk=4096)
    part = m.render(o, d, chunk=1024)
    for k in full:
        assert torch.equal(full[k], part[k]), k
    with pytest.raises(Exception):
        m.render_rays(torch.cat([o, o]), torch.cat([d, d]))    # more rays than the buffers hold


@pytest.mark.gpu
def test_learned_background_path():
    """bg_color None -> render_contracted_bg through the packed ops (volsurfs.py:690-702)."""
    from volsurfs_amd.background import BoundingBox
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4)).cuda()

    def bg(p, d, it):
        y = net(torch.cat([p, d], 1))
        return torch.sigmoid(y[:, :3]), torch.nn.functional.softplus(y[:, 3:])
    m = VolSurfs(nested_shells(K=2, subdiv=2), max_rays=4096, textures_res=(128, 64, 32, 16),
                 bg_color=None, bg_model=bg, bounding_primitive=BoundingBox(1.0))
    m.is_training = False
    o, d = pinhole_rays(32, 32, focal=50.0)
    rt = m.render_rays(o, d)["renders"]["ray_traced"]
    assert rt["rgb_bg"].shape == (1024, 3) and rt["rgb"].shape == (1024, 3)
    miss = rt["surfs_alpha"].sum((1, 2)) == 0
    # where no shell is hit the pixel is the (fp16-rounded) background colour
    assert torch.equal(rt["rgb"][miss], rt["rgb_bg"][miss])
    rt["rgb"].sum().backward()
    assert all(p.grad is not None and p.grad.abs().sum() > 0 for p in net.parameters())
    with pytest.raises(Exception):
        VolSurfs(nested_shells(K=1, subdiv=1), bg_color=None)


@pytest.mark.gpu
@pytest.mark.parametrize("sh_coeffs", [False, True])
def test_legacy_appearance_branch_matches_oracle_and_trains(sh_coeffs):
    """using_neural_textures=False (config C3's branch, volsurfs.py:208-300, 548-580): per-shell
    RGB / ColorSH models on hit points, dirs, normals.  Checked against the oracle's
    restatement (hash grid + MLP on torch-CPU + oracle composite) from the same hits."""
    from oracle import composite as OC
    from oracle import legacy_models as OL
    from oracle import tcnn_like
    from oracle.neural_texture import sh_basis_values, sh_eval
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    K = 2
    m = VolSurfs(nested_shells(K=K, subdiv=3), max_rays=4096, using_neural_textures=False,
                 appearance_predict_sh_coeffs=sh_coeffs, rgb_normal_dep=not sh_coeffs,
                 rgb_mlp_layers_dims=(64, 32), bb_sides=1.0, sh_degree=3)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for mod in m.models.values():
            p = mod.pos_encoder.encoder.params
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1).cuda())
    o, d = pinhole_rays(40, 40, focal=70.0)
    res = m.render_rays(o, d, iter_nr=None)
    rt = res["renders"]["ray_traced"]
    assert rt["surfs_uvs"] is None and torch.isfinite(rt["rgb"]).all()
    # ---- oracle from the same hits
    hit_t, hit_slot, _ = m.raytracer.trace_all(o, d)
    N = o.shape[0]
    geom = tcnn_like.GridGeometryND(3, 24, 18, 16, 2)
    s_rgb, s_a = torch.zeros(N, K, 3), torch.zeros(N, K)
    for i in range(K):
        hits = (hit_slot[i] >= 0).cpu()
        tri = m.raytracer.tris[hit_slot[i][hit_slot[i] >= 0].long()].cpu()
        nrm = torch.nn.functional.normalize(torch.cross(tri[:, 4:7], tri[:, 8:11], dim=1), dim=1)
        dd = d.cpu()[hits]
        pts = o.cpu()[hits] + hit_t[i].cpu()[hits][:, None] * dd
        outs = []
        for key in (f"rgb_{i}", f"alpha_{i}"):
            mod = m.models[key]
            layers = [(l.weight.detach().cpu(), l.bias.detach().cpu()) for l in mod.mlp.layers
                      if isinstance(l, torch.nn.Linear)]
            table = mod.pos_encoder.encoder.params.detach().cpu()
            if sh_coeffs:       # ColorSH: the encoder gets no bounding box (color_sh.py:62-69)
                feats = OL.gridhash_encode(geom, table, pts, None)
                coeffs = OL.mlp_forward(layers, feats).reshape(pts.shape[0], -1, 16)
                outs.append(torch.sigmoid(sh_eval(coeffs, dd, 3)))
            else:
                bb = torch.tensor([1.0, 1.0, 1.0])
                nrm_in = nrm if mod.normal_dep else None
                outs.append(OL.rgb_forward(geom, table, layers, pts, dd, nrm_in, bb, 3))
        dot = torch.sum(-dd * nrm, dim=1).clamp(0.0, 1.0)
        s_rgb[hits, i] = outs[0]
        s_a[hits, i] = outs[1][:, 0] * (torch.sigmoid(10.0 * dot) * 2.0 - 1.0)
    ref = OC.composite_dense_fwd(s_rgb.numpy(), s_a.numpy(), np.ones((1, 3), np.float32))
    ref_rgb = ref["rgb"] if isinstance(ref, dict) else ref[0]
    err = np.abs(rt["rgb"].detach().cpu().numpy() - np.asarray(ref_rgb))
    assert err.max() < 2e-3 and np.median(err) < 1e-4, (err.max(), np.median(err))   # fp16 composite
    # ---- trains
    opt = m.init_optim()
    gt = torch.rand(N, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) * 0.2
    losses = []
    for it in range(8):
        opt.zero_grad()
        l, _, _ = m(o, d, gt, None, it)
        l["loss"].backward()
        m.optim_step()
        losses.append(l["loss"].item())
    assert losses[-1] < losses[0], losses


@pytest.mark.gpu
def test_baked_textures_render_identically_and_match_the_oracle():
    """bake every texel once, render from the 8-bit textures only.  The
    baked render equals the live render bit for bit (same texel values), and the baked texels
    are the oracle's network output at the texel centres."""
    from oracle import neural_texture as ONT
    from oracle import tcnn_like
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    from test_nt_mlp import unpack_weights
    res = (64, 32, 16, 8)
    m = VolSurfs(nested_shells(K=2, subdiv=3), max_rays=4096, textures_res=res)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        m.bank.tables.copy_((torch.rand(m.bank.tables.shape, generator=g) * 2 - 1).cuda())
    m.bank.refresh_half_params()
    o, d = pinhole_rays(56, 56, focal=90.0)
    live = m.render(o, d)
    baked = m.bake()
    out = m.render_baked(o, d)
    assert (live["surfs_alpha"].sum((1, 2)) > 0).sum() > 200
    assert torch.equal(out["rgb"], live["rgb"]) and torch.equal(out["surfs_rgb"], live["surfs_rgb"])
    assert torch.equal(out["surfs_alpha"], live["surfs_alpha"])
    tex = baked.baked_textures()
    assert set(tex) == {(s, t, dg) for s in range(2) for t in range(2) for dg in range(4)}
    geom = tcnn_like.GridGeometry()
    for (s, typ, dg) in [(0, 0, 3), (1, 1, 2), (1, 0, 0)]:
        R, n = res[dg], 2 * dg + 1
        C = (3 if typ == 0 else 1) * n
        img = tex[(s, typ, dg)]
        assert img.shape == (R, R, C) and img.dtype == torch.uint8
        iy, ix = torch.meshgrid(torch.arange(R), torch.arange(R), indexing="ij")
        xy = torch.stack([(ix.flatten() + 0.5) / R, (iy.flatten() + 0.5) / R], 1).float()
        x = baked.tex_index(s, typ, dg)
        feats = tcnn_like.hashgrid_forward(geom, baked.tables_h[x].cpu(), xy)
        w1, w2, w3 = unpack_weights(baked.weights_h[x].cpu())
        _, q_ref = ONT.quantise(tcnn_like.mlp_forward(w1, w2, w3, feats, C))
        dq = (img.cpu().reshape(-1, C).int() - q_ref.int()).abs()
        assert dq.max() <= 1 and (dq > 0).float().mean() < 2e-3, (s, typ, dg, dq.max())


@pytest.mark.gpu
def test_gradients_do_not_depend_on_the_f16_gradient_scale():
    """The gradient chain runs in f16 behind a power-of-two scale (tcnn's loss scale; chosen per
    call from the incoming gradients when `grad_scale` is None).  A mean-reduced and a
    sum-reduced loss, whose per-ray gradients differ by the ray count, must give the same
    gradients (up to that factor), and so must explicit scales around the automatic one."""
    from volsurfs_amd.camera import pinhole_rays
    m = _method()
    o, d = pinhole_rays(64, 64, focal=110.0)
    gt = torch.rand(o.shape[0], 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))

    def grads(scale, reduce):
        m.grad_scale = scale
        for p in (m.bank.tables, m.bank.weights):
            p.grad = None
        rgb = m.render_rays(o, d)["renders"]["ray_traced"]["rgb"].float()
        loss = (rgb - gt).abs().mean() if reduce == "mean" else (rgb - gt).abs().sum()
        loss.backward()
        return m.bank.tables.grad.clone(), m.bank.weights.grad.clone()

    ref_t, ref_w = grads(None, "mean")
    assert ref_t.abs().max() > 0 and torch.isfinite(ref_t).all() and torch.isfinite(ref_w).all()
    n = gt.numel()
    for scale, reduce, factor in ((2.0 ** 12, "mean", 1.0), (2.0 ** 16, "mean", 1.0), (None, "sum", float(n))):
        g_t, g_w = grads(scale, reduce)
        assert torch.isfinite(g_t).all() and torch.isfinite(g_w).all()
        for a, b in ((g_t / factor, ref_t), (g_w / factor, ref_w)):
            cos = torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0)
            assert cos > 0.999, (scale, reduce, cos)
            assert (a - b).abs().max() <= 2e-2 * b.abs().max()


@pytest.mark.gpu
def test_checkpoint_save_load_resumes_bit_exactly(tmp_path):
    """base_method.py:118-264 layout (<run>/<iter:07d>/models/{key}.pt + optimiser state): a fresh
    method that loads the checkpoint renders the same image and takes the same next step."""
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.trainer import save_checkpoints, get_last_checkpoint_in_path
    o, d = pinhole_rays(48, 48, focal=80.0)
    gt = torch.rand(48 * 48, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))

    def step(m, it):
        m.optimizer.zero_grad()
        l, _, _ = m(o, d, gt, None, it)
        l["loss"].backward()
        m.optim_step()

    a = _method()
    a.init_optim()
    a.save_checkpoints_path = str(tmp_path)
    for it in range(3):
        step(a, it)
    save_checkpoints(a, 2, remove_previous=True)
    save_checkpoints(a, 3, remove_previous=True)                         # rotation: only the newest stays
    assert get_last_checkpoint_in_path(str(tmp_path)) == "0000003" and len(list(tmp_path.iterdir())) == 1
    files = sorted(p.name for p in (tmp_path / "0000003" / "models").iterdir())
    # the optimiser state is saved under its lower-cased class name (base_method.py:255-262): the
    # reference's is apex FusedAdam -> fusedadam.pt, and so is this one
    assert files == ["alpha_0.pt", "alpha_1.pt", "fusedadam.pt", "rgb_0.pt", "rgb_1.pt"]
    b = _method()
    with torch.no_grad():
        b.bank.tables.normal_()                                           # something else than a's state
    b.bank.refresh_half_params()
    b.init_optim()
    b.load_checkpoints_path = str(tmp_path)
    b.load(3)
    ra = a.render(o, d)["rgb"]
    assert torch.equal(ra, b.render(o, d)["rgb"])
    # the restored Adam moments make the next step identical up to the backward's atomics
    step(a, 3)
    step(b, 3)
    assert (a.bank.tables - b.bank.tables).abs().max() <= 1e-3 * a.bank.tables.abs().max()
    sa, sb = a.optimizer.state_dict(), b.optimizer.state_dict()
    assert sb["param_groups"][0]["step"] == sa["param_groups"][0]["step"] == 4
    assert torch.allclose(sa["state"][0]["exp_avg"], sb["state"][0]["exp_avg"], atol=1e-3 * float(sa["state"][0]["exp_avg"].abs().max()))


@pytest.mark.gpu
def test_profiler_sections_of_the_reference():
    """the method reports the reference's hot-path sections to a Profiler."""
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.trainer import Profiler
    m = _method()
    m.profiler = Profiler()
    o, d = pinhole_rays(64, 64, focal=110.0)
    for _ in range(3):
        m.render_rays(o, d)
    t = m.profiler.get_avg_times()
    assert set(t) == {"meshes_raytracing", "ray_color_inference", "render_fg"}
    assert all(0 < v < 1.0 for v in t.values())
    m.profiler.reset()
    assert m.profiler.get_avg_times() == {}


@pytest.mark.gpu
def test_dtu_config_full_size_learned_background():
    """BASELINE configs[3] at full size: 1600x1200 rays, K=5 shells, `bg_color=None` ->
    NerfHash background with 32 contracted samples per ray through the packed ops
    (volsurfs.py:686-702, utils/background.py:31-141).  Size-independent properties of the
    full frame, then one training step (fwd + bwd + Adam) on a 65 536-ray batch."""
    from volsurfs_amd.background import BoundingBox
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    from volsurfs_amd.models import NerfHash
    from volsurfs_amd.trainer import train_step
    torch.manual_seed(0)
    H, W, K = 1200, 1600, 5
    bg = NerfHash(3, "gridhash", "spherical_harmonics")
    m = VolSurfs(nested_shells(K=K, subdiv=6), max_rays=65536, bg_color=None, bg_model=bg,
                 bounding_primitive=BoundingBox(1.0), nr_samples_bg=32)
    o, d = pinhole_rays(H, W, focal=1111.1 * H / 800.0, cam_pos=(0.0, 0.0, -1.5))
    assert o.shape[0] == 1920000
    m.is_training = False                              # no sample jitter: renders are repeatable
    full = m.render(o, d, chunk=65536)
    rgb, rgb_bg, bgT = full["rgb"], full["rgb_bg"], full["bg_transmittance"]
    assert rgb.shape == (H * W, 3) and torch.isfinite(rgb).all() and rgb.min() >= 0 and rgb.max() <= 1.001
    miss = full["surfs_alpha"].sum((1, 2)) == 0
    assert miss.any() and (~miss).sum() > H * W // 8
    assert torch.equal(rgb[miss], rgb_bg[miss])        # a ray that misses every shell shows the field
    assert (bgT[miss] == 1).all()
    # rgb = rgb_fg + bg_T * rgb_bg in fp16 (volsurfs.py:704-708), everywhere
    recon = (full["rgb_fg"].half() + bgT.half() * rgb_bg.half()).float()
    assert (rgb - recon).abs().max() <= 1e-3
    # partition of unity of the blending weights
    tot = full["surfs_blending_weights"].sum(1) + bgT
    assert (tot - 1).abs().max() < 4e-3
    # the background is a function of the ray only: a second render of a slice is identical
    again = m.render(o[:131072], d[:131072], chunk=65536)
    assert torch.equal(again["rgb"], rgb[:131072])
    # one training step on a 65 536-ray batch spread over the frame
    m.init_optim()
    idx = torch.linspace(0, H * W - 1, 65536, device="cuda").long()
    gt = torch.rand(65536, 3, device="cuda")
    w0 = bg.mlp_rgb.layers[0].weight.detach().clone()
    t0 = m.bank.tables.detach().clone()
    losses, _ = train_step(m, o[idx].contiguous(), d[idx].contiguous(), gt, iter_nr=0, is_first_iter=True)
    assert np.isfinite(losses["loss"])
    g0 = bg.pos_encoder.encoder.params.detach().clone()
    losses2, _ = train_step(m, o[idx].contiguous(), d[idx].contiguous(), gt, iter_nr=1)
    assert not torch.equal(bg.mlp_rgb.layers[0].weight, w0) and not torch.equal(m.bank.tables, t0)
    assert not torch.equal(bg.pos_encoder.encoder.params, g0)         # the field's hash grid trains
    assert float(bg.pos_encoder.encoder.params.grad.abs().sum()) == 0  # cleared by the fused step


@pytest.mark.gpu
def test_train_step_chunks_a_batch_larger_than_max_rays():
    """the dynamic ray count grows past max_rays (trainer.py:288-304); the step then
    runs as chunks whose gradients accumulate and equals the one-shot step of a larger bank."""
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.trainer import train_step
    o, d = pinhole_rays(64, 64, focal=110.0)
    gt = torch.rand(4096, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) * 0.3
    big, small = _method(max_rays=4096), _method(max_rays=1024)
    seen = {}
    for name, m in (("big", big), ("small", small)):
        m.init_optim()
        m.grad_scale = 4096.0

        def snap(m=m, name=name, inner=m.optim_step):     # the gradients the optimiser is about to consume
            seen[name] = (m.bank.weights.grad.clone(), m.bank.tables.grad.clone())
            inner()
        m.optim_step = snap
    l1, n1 = train_step(big, o, d, gt, iter_nr=0, is_first_iter=True, nr_rays=4096,
                        target_nr_of_training_samples=3000)
    l2, n2 = train_step(small, o, d, gt, iter_nr=0, is_first_iter=True, nr_rays=4096,
                        target_nr_of_training_samples=3000)
    assert abs(l1["loss"] - l2["loss"]) < 1e-6 and n1 == n2 and n1 != 4096
    # the same sums, accumulated chunk by chunk (fp16 gradient chain with a per-chunk rounding pattern)
    for a, b in zip(seen["big"], seen["small"]):
        s_ = a.abs().max().item()
        assert s_ > 0 and (a - b).abs().max().item() <= 2e-2 * s_
        assert torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0) > 0.9995


@pytest.mark.gpu
def test_permutohash_legacy_branch_matches_oracle_and_trains():
    """BASELINE configs[2]'s appearance: `using_neural_textures=False`,
    `rgb_pos_encoder_type="permutohash"` (24 levels x 2, capacity 2^18) + SH-3 view encoding ->
    MLP [128,128,64] -> sigmoid (models/rgb.py:104-149), per shell, rgb + alpha models."""
    from oracle import composite as OC
    from oracle import legacy_models as OL
    from oracle import permuto as OP
    from oracle.neural_texture import sh_basis_values
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    K = 2
    m = VolSurfs(nested_shells(K=K, subdiv=3), max_rays=4096, using_neural_textures=False,
                 rgb_pos_encoder_type="permutohash", rgb_mlp_layers_dims=(128, 128, 64), bb_sides=1.0)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for mod in m.models.values():
            p = mod.pos_encoder.encoder.lattice_values
            assert tuple(p.shape) == (24, 1 << 18, 2) and mod.pos_encoder.output_dim == 50
            assert mod.mlp.layers[0].in_features == 50 + 16
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1).cuda())
    o, d = pinhole_rays(40, 40, focal=70.0)
    rt = m.render_rays(o, d, iter_nr=None)["renders"]["ray_traced"]
    hit_t, hit_slot, _ = m.raytracer.trace_all(o, d)
    N = o.shape[0]
    s_rgb, s_a = torch.zeros(N, K, 3), torch.zeros(N, K)
    for i in range(K):
        hits = (hit_slot[i] >= 0).cpu()
        tri = m.raytracer.tris[hit_slot[i][hit_slot[i] >= 0].long()].cpu()
        nrm = torch.nn.functional.normalize(torch.cross(tri[:, 4:7], tri[:, 8:11], dim=1), dim=1)
        dd = d.cpu()[hits]
        pts = o.cpu()[hits] + hit_t[i].cpu()[hits][:, None] * dd
        outs = []
        for key in (f"rgb_{i}", f"alpha_{i}"):
            mod = m.models[key]
            layers = [(l.weight.detach().cpu(), l.bias.detach().cpu()) for l in mod.mlp.layers
                      if isinstance(l, torch.nn.Linear)]
            enc, _ = OP.permuto_hash_encoder(mod.pos_encoder.encoder.lattice_values.detach().cpu().numpy(),
                                             pts.numpy(), mod.pos_encoder.encoder.random_shift_per_level.cpu().numpy(),
                                             bb_sides=1.0)
            x = torch.cat([torch.from_numpy(enc), sh_basis_values(dd, 3)], 1)
            outs.append(torch.sigmoid(OL.mlp_forward(layers, x)))
        dot = torch.sum(-dd * nrm, dim=1).clamp(0.0, 1.0)
        s_rgb[hits, i] = outs[0]
        s_a[hits, i] = outs[1][:, 0] * (torch.sigmoid(10.0 * dot) * 2.0 - 1.0)
    ref = OC.composite_dense_fwd(s_rgb.numpy(), s_a.numpy(), np.ones((1, 3), np.float32))["rgb"]
    err = np.abs(rt["rgb"].detach().cpu().numpy() - ref)
    assert err.max() < 2e-3 and np.median(err) < 1e-4, (err.max(), np.median(err))   # fp16 composite
    # full-frame evaluation of the legacy configuration (render() crashed on None keys)
    full = m.render(o, d, chunk=1024)
    assert full["surfs_uvs"] is None and torch.allclose(full["rgb"], rt["rgb"].detach(), atol=1e-6)
    ss = m.render(torch.cat([o, o]).reshape(2, -1, 3).transpose(0, 1).reshape(-1, 3).contiguous(),
                  torch.cat([d, d]).reshape(2, -1, 3).transpose(0, 1).reshape(-1, 3).contiguous(),
                  nr_rays_per_pixel=2, chunk=2048)
    assert ss["surfs_uvs"] is None and torch.allclose(ss["rgb"], full["rgb"], atol=1e-6)
    opt = m.init_optim()
    gt = torch.rand(N, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) * 0.2
    losses = []
    for it in range(8):
        opt.zero_grad()
        l, _, _ = m(o, d, gt, None, it)
        l["loss"].backward()
        m.optim_step()
        losses.append(l["loss"].item())
    assert losses[-1] < losses[0], losses


@pytest.mark.gpu
def test_from_meshes_path_builds_the_method_and_resumes(tmp_path):
    """methods/volsurfs.py:62-128: shells from a directory of .obj / .ply files (sorted by isolevel),
    copied into the run's checkpoints at the first iteration, loaded from there on resume."""
    import os
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import nested_shells, save_obj, save_ply
    from volsurfs_amd.methods import VolSurfs
    src, run = tmp_path / "meshes_simplified_uvs", tmp_path / "checkpoints"
    src.mkdir()
    run.mkdir()
    shells = nested_shells(K=3, subdiv=3)
    save_obj(os.path.join(src, "-0.01.obj"), shells[0])
    save_ply(os.path.join(src, "0.0.ply"), shells[1])
    save_obj(os.path.join(src, "0.01.obj"), shells[2])
    kw = dict(max_rays=4096, textures_res=(128, 64, 32, 16))
    a = VolSurfs.from_meshes_path(str(src), str(run), start_iter_nr=0, **kw)
    assert a.nr_meshes == 3 and sorted(os.listdir(run / "meshes")) == ["0.obj", "1.ply", "2.obj"]
    b = VolSurfs.from_meshes_path("/nonexistent", str(run), start_iter_nr=10, **kw)
    ref = VolSurfs(shells, **kw)
    o, d = pinhole_rays(32, 32, focal=50.0)
    with torch.no_grad():
        ra, rb, rr = (m.render_rays(o, d, return_samples=False)["renders"]["ray_traced"]["rgb"] for m in (a, b, ref))
    assert torch.equal(ra, rb) and torch.equal(ra, rr)          # same seed, same shells -> same image
    sub = VolSurfs.from_meshes_path(str(src), str(tmp_path / "run2"), meshes_indices=["2", "0"], **kw)
    assert sub.nr_meshes == 2


@pytest.mark.gpu
def test_baseline_config0_plumbing_case_matches_cpu_restatement():
    """BASELINE configs[0] / SURVEY §8d C1: one 64x64 pinhole view (focal 70 px, camera at (0,0,-1.5)),
    K=1 icosphere (subdiv 4, r = 0.3), legacy `RGB` appearance with the frequency position encoder
    (39 dims) + SH-3 view encoding and a [32, 32] MLP, white background — the reference's own
    CPU-runnable case.  The HIP path (trace, fused fp32 MLP, fp16 composite) against the same
    models evaluated with plain torch on the CPU from the same hits, forward and gradients."""
    from oracle import composite as OC
    from oracle import legacy_models as OL
    from oracle.neural_texture import sh_basis_values
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import TensorMesh, icosphere, octahedral_uv
    from volsurfs_amd.methods import VolSurfs
    v, f = icosphere(4, 0.3)
    fuv = octahedral_uv(v.astype(np.float64) / 0.3)[f].astype(np.float32)
    m = VolSurfs([TensorMesh(v, f, fuv)], max_rays=4096, using_neural_textures=False,
                 rgb_pos_encoder_type="frequency", rgb_mlp_layers_dims=(32, 32), bb_sides=1.0)
    assert m.models["rgb_0"].mlp.layers[0].in_features == 39 + 16
    o, d = pinhole_rays(64, 64, focal=70.0, cam_pos=(0.0, 0.0, -1.5))
    gt = torch.rand(4096, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    losses, _, _ = m(o, d, gt, None, 0)
    losses["loss"].backward()
    rgb = m.render_rays(o, d, return_samples=False)["renders"]["ray_traced"]["rgb"].detach().cpu().numpy()
    # CPU restatement from the same hits
    hit_t, hit_slot, _ = m.raytracer.trace_all(o, d)
    hits = (hit_slot[0] >= 0).cpu()
    assert 400 < int(hits.sum()) < 2000
    tri = m.raytracer.tris[hit_slot[0][hit_slot[0] >= 0].long()].cpu()
    nrm = torch.nn.functional.normalize(torch.cross(tri[:, 4:7], tri[:, 8:11], dim=1), dim=1)
    dd = d.cpu()[hits]
    pts = o.cpu()[hits] + hit_t[0].cpu()[hits][:, None] * dd
    outs, leaves = [], []
    for key in ("rgb_0", "alpha_0"):
        mod = m.models[key]
        layers = [(l.weight.detach().cpu().clone().requires_grad_(True), l.bias.detach().cpu().clone().requires_grad_(True))
                  for l in mod.mlp.layers if isinstance(l, torch.nn.Linear)]
        leaves.append(layers)
        parts = [pts] + [fn(pts * 2.0 ** l) for l in range(6) for fn in (torch.sin, torch.cos)]
        x = torch.cat(parts + [sh_basis_values(dd, 3)], 1)
        outs.append(torch.sigmoid(OL.mlp_forward(layers, x)))
    dot = torch.sum(-dd * nrm, dim=1).clamp(0.0, 1.0)
    s_rgb = torch.zeros(4096, 1, 3).index_put((hits.nonzero()[:, 0], torch.tensor(0)), outs[0])
    a = outs[1][:, 0] * (torch.sigmoid(10.0 * dot) * 2.0 - 1.0)
    s_a = torch.zeros(4096, 1).index_put((hits.nonzero()[:, 0], torch.tensor(0)), a)
    bg = np.ones((1, 3), np.float32)
    ref = OC.composite_dense_fwd(s_rgb.detach().numpy(), s_a.detach().numpy(), bg)["rgb"]
    e = np.abs(rgb - ref)
    assert e.max() < 1e-3 and (e <= 1e-4).mean() > 0.99          # fp16 composite of fp32 models
    g = np.sign(ref - gt.cpu().numpy()).astype(np.float32) / (4096 * 3)
    gc, ga, _ = OC.composite_dense_bwd(s_rgb.detach().numpy(), s_a.detach().numpy(), bg, g)
    ((s_rgb * torch.from_numpy(gc)).sum() + (s_a * torch.from_numpy(ga)).sum()).backward()
    for key, layers in zip(("rgb_0", "alpha_0"), leaves):
        lin = [l for l in m.models[key].mlp.layers if isinstance(l, torch.nn.Linear)]
        for l, (w, b) in zip(lin, layers):
            for got, want in ((l.weight.grad.cpu(), w.grad), (l.bias.grad.cpu(), b.grad)):
                assert want.abs().max() > 0
                rel0 = float((got - want).abs().max() / want.abs().max())
                print(f"MEASURED config0 {key} grad_rel_max={rel0:.3e}")
                assert rel0 <= CONFIG0_GRAD_REL_MAX                               # fp16 composite backward
                assert torch.nn.functional.cosine_similarity(got.flatten(), want.flatten(), dim=0) > 0.999


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(), dict(is_inner_mesh_solid=True, rgb_normal_dep=True),
                                 dict(are_volsurfs_colors_indep=False, are_volsurfs_alphas_indep=False,
                                      rgb_pos_encoder_type="gridhash")])
def test_legacy_grouped_shading_equals_the_per_shell_loop(cfg):
    """methods._shade_legacy_grouped (all shells' hits prepared at once, one grouped MLP op per
    model type) against the per-shell loop it replaces: identical forward, equal gradients."""
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    kw = dict(max_rays=4096, using_neural_textures=False, rgb_pos_encoder_type="permutohash",
              rgb_mlp_layers_dims=(64, 32), bb_sides=1.0)
    kw.update(cfg)
    torch.manual_seed(21)          # (the MLPs draw from the global generator: the same model alone or in the full suite)
    m = VolSurfs(nested_shells(K=3, subdiv=3), **kw)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for mod in m.models.values():
            enc = mod.pos_encoder.encoder
            p = enc.lattice_values if hasattr(enc, "lattice_values") else enc.params
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1).cuda())
    o, d = pinhole_rays(40, 40, focal=70.0)
    gt = torch.rand(1600, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    res = {}
    glue0 = VolSurfs.legacy_fused_glue
    # "glue": the grouped path with its glue fused (hit preparation, sigmoid / decay / scatter, composite + L1 as one
    # launch each — the default); True / False: the grouped path on torch expressions / the per-shell loop
    for mode, grouped, glue, samples in (("glue", True, True, False), (True, True, False, True), (False, False, False, True)):
        VolSurfs.legacy_grouped, VolSurfs.legacy_fused_glue = grouped, glue
        try:
            assert m._legacy_groupable(o)
            for p in m.parameters():
                p.grad = None
            losses, _, _ = m(o, d, gt, None, None, return_samples=samples)
            losses["loss"].backward()
            rt = m.render_rays(o, d, return_samples=False)["renders"]["ray_traced"]
        finally:
            VolSurfs.legacy_grouped, VolSurfs.legacy_fused_glue = True, glue0
        res[mode] = ({k: v.detach().clone() for k, v in rt.items() if v is not None},
                     [None if p.grad is None else p.grad.clone() for p in m.parameters()],
                     losses["loss"].item())
    for k in res[True][0]:
        assert torch.equal(res[True][0][k], res[False][0][k]), k
        # the fused glue takes the same steps in fp32; expf / the norm may round the last bit differently — and a last
        # fp32 bit of a per-shell colour can flip its fp16 cast in the composite (the reference composites in fp16):
        # one fp16 ulp (9.8e-4 below 2) on a handful of values, 2e-6 everywhere else
        diff = (res["glue"][0][k].float() - res[False][0][k].float()).abs()
        assert diff.max().item() <= 1e-3 and (diff > 2e-6).float().mean().item() <= 2e-3, k
    assert res[True][2] == res[False][2]
    assert abs(res["glue"][2] - res[False][2]) <= 2e-6 * abs(res[False][2])
    for mode, tol in ((True, 1e-5), ("glue", 5e-5)):
        n_with_grad = 0
        for a, b in zip(res[mode][1], res[False][1]):
            assert (a is None) == (b is None)
            if a is not None:
                n_with_grad += 1
                s_ = b.abs().max().item()
                assert (a - b).abs().max().item() <= tol * s_ + 1e-12
        assert n_with_grad >= 6


@pytest.mark.gpu
def test_inference_takes_the_fused_forward_and_equals_the_two_kernel_path(monkeypatch):
    """`render_rays` under no_grad needs no feature planes, so `NeuralTextureBank.evaluate` takes the
    one-launch encode + MLP there (neural_textures.FUSED_FORWARD = "auto"); a differentiated call takes
    the two kernels.  Same pixels either way, bit for bit."""
    from volsurfs_amd import neural_textures as NT
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    m = VolSurfs(nested_shells(K=3, subdiv=3, noise=0.05, atlas_charts=4), max_rays=4096, seed=11)
    with torch.no_grad():
        m.bank.tables.uniform_(-1.0, 1.0)
    m.bank.refresh_half_params()
    o, d = pinhole_rays(64, 64, focal=90.0, cam_pos=(0.0, 0.0, -1.5))
    calls = []
    real = NT.NeuralTextureBank.encode_mlp
    monkeypatch.setattr(NT.NeuralTextureBank, "encode_mlp",
                        lambda self, **kw: (calls.append(kw.get("write_features")), real(self, **kw))[1])
    out = {}
    for mode in ("auto", False, True):
        monkeypatch.setattr(NT, "FUSED_FORWARD", mode)
        calls.clear()
        with torch.no_grad():
            out[mode] = m.render_rays(o, d, return_samples=False)["renders"]["ray_traced"]["rgb"].clone()
        assert calls == ([] if mode is False else [False])
    assert torch.equal(out["auto"], out[False]) and torch.equal(out[True], out[False])
    assert float(out["auto"].std()) > 0.01
    monkeypatch.setattr(NT, "FUSED_FORWARD", "auto")
    calls.clear()
    gt = torch.rand(4096, 3, device="cuda")
    losses, _, _ = m(o, d, gt, None, 0)         # a differentiated call: the feature planes are needed
    losses["loss"].backward()
    assert calls == []


# configs[2] at its shell count: bounds = 2x the values measured on MI355X (printed as MEASURED config2_K5 ...)
# measured: rgb bit-identical to the oracle (max error 0.0 on 49 152 values); lattice gradients 4.3e-7,
# MLP gradients 3.0e-6 of each tensor's largest entry (fp32 path; fp64 oracle accumulation).  The rgb bound
# leaves room for ONE fp16 ulp (4.9e-4 in [0.5, 1)) on 1e-4 of the values: an fp32 difference in the last
# bit of a per-shell colour can flip its fp16 cast in the composite on another box / compiler.
CONFIG2_RGB_MAX, CONFIG2_RGB_FRAC_OVER_1E4 = 1e-3, 1e-4
# (r6: the MLPs are initialised from torch's GLOBAL generator, so what this test measured depended on which tests ran
#  before it — alone instead of in the full suite the lattice figure was 3.1e-6 ... 5.8e-6 against a bound of 1e-6.  The
#  test now seeds the generator itself; the bounds are 4x the largest value seen over seeds, far inside north_star's 1e-3)
CONFIG2_GRAD_REL_MAX = {"lattice": 2.4e-5, "mlp": 2.4e-5}


@pytest.mark.gpu
def test_config2_permutohash_K5_noisy_shells_oracle_parity_gradients_and_training():
    """VERDICT r3 next #1a / configs_untested: BASELINE configs[2] (NeRF-Synthetic 'lego', K = 5,
    permutohedral encoding, legacy appearance branch: models/rgb.py:104-149,
    encodings/permutohash.py:68-96) AT ITS SHELL COUNT on noisy shells, 128 x 128 rays:

      * forward vs the oracle chain (oracle/permuto.py -> SH-3 view encoding -> oracle/legacy_models.py
        MLP [128,128,64] -> sigmoid -> alpha decay -> oracle/composite.py fp16 composite);
      * EVERY parameter gradient (10 lattices, 10 MLPs) of the L1 loss vs the oracle's
        (fp32 autograd through the MLPs, oracle.permuto.encode_backward in fp64 for the lattices,
        oracle.composite.composite_dense_bwd), relative to each tensor's largest entry;
      * 24 iterations of trainer.train_step (trainer.py:118-308 order): the loss falls.
    """
    torch.manual_seed(5)
    from oracle import composite as OC
    from oracle import legacy_models as OL
    from oracle import permuto as OP
    from oracle.neural_texture import sh_basis_values
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    from volsurfs_amd.trainer import train_step
    K, res = 5, 128
    m = VolSurfs(nested_shells(K=K, subdiv=4, noise=0.05), max_rays=res * res, using_neural_textures=False,
                 rgb_pos_encoder_type="permutohash", rgb_mlp_layers_dims=(128, 128, 64), bb_sides=1.0,
                 nr_warmup_iters=0, seed=11)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for mod in m.models.values():
            p = mod.pos_encoder.encoder.lattice_values
            assert tuple(p.shape) == (24, 1 << 18, 2) and mod.pos_encoder.output_dim == 50
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1).cuda())
    assert len(m.models) == 2 * K
    o, d = pinhole_rays(res, res, focal=1.6 * res)
    N = o.shape[0]
    gt = torch.rand(N, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    for p in m.parameters():
        p.grad = None
    losses, _, _ = m(o, d, gt, None, None)
    losses["loss"].backward()
    pred = m.render_rays(o, d, iter_nr=None, return_samples=False)["renders"]["ray_traced"]["rgb"].detach().cpu()
    hit_t, hit_slot, _ = m.raytracer.trace_all(o, d)
    # ---- oracle: forward with autograd leaves
    s_rgb, s_a = torch.zeros(N, K, 3), torch.zeros(N, K)
    leaves = {}
    for i in range(K):
        hits = (hit_slot[i] >= 0).cpu()
        rows = hits.nonzero()[:, 0]
        tri = m.raytracer.tris[hit_slot[i][hit_slot[i] >= 0].long()].cpu()
        nrm = torch.nn.functional.normalize(torch.cross(tri[:, 4:7], tri[:, 8:11], dim=1), dim=1)
        dd = d.cpu()[hits]
        pts = (o.cpu()[hits] + hit_t[i].cpu()[hits][:, None] * dd).numpy()
        outs = []
        for key in (f"rgb_{i}", f"alpha_{i}"):
            mod = m.models[key]
            layers = [(l.weight.detach().cpu().clone().requires_grad_(True), l.bias.detach().cpu().clone().requires_grad_(True))
                      for l in mod.mlp.layers if isinstance(l, torch.nn.Linear)]
            shift = mod.pos_encoder.encoder.random_shift_per_level.cpu().numpy()
            enc, _ = OP.permuto_hash_encoder(mod.pos_encoder.encoder.lattice_values.detach().cpu().numpy(), pts, shift,
                                             bb_sides=1.0)
            enc = torch.from_numpy(enc).requires_grad_(True)
            x = torch.cat([enc, sh_basis_values(dd, 3)], 1)
            outs.append(torch.sigmoid(OL.mlp_forward(layers, x)))
            leaves[key] = (layers, enc, pts, shift)
        dot = torch.sum(-dd * nrm, dim=1).clamp(0.0, 1.0)
        s_rgb = s_rgb.index_put((rows, torch.tensor(i)), outs[0])
        s_a = s_a.index_put((rows, torch.tensor(i)), outs[1][:, 0] * (torch.sigmoid(10.0 * dot) * 2.0 - 1.0))
    bg = np.ones((1, 3), np.float32)
    ref = OC.composite_dense_fwd(s_rgb.detach().numpy(), s_a.detach().numpy(), bg)["rgb"]
    err = np.abs(pred.numpy() - ref)
    over = float((err > 1e-4).mean())
    print(f"MEASURED config2_K5 rgb_max_err={err.max():.3e} rgb_frac_over_1e-4={over:.3e} hits={int((hit_slot >= 0).sum())}")
    assert err.max() <= CONFIG2_RGB_MAX and over <= CONFIG2_RGB_FRAC_OVER_1E4      # fp16 composite of fp32 colours
    # ---- oracle: backward of mean |gt - pred|
    g_rgb = (np.sign(ref - gt.cpu().numpy()) / (3.0 * N)).astype(np.float32)
    g_c, g_a, _ = OC.composite_dense_bwd(s_rgb.detach().numpy(), s_a.detach().numpy(), bg, g_rgb)
    ((s_rgb * torch.from_numpy(g_c)).sum() + (s_a * torch.from_numpy(g_a)).sum()).backward()
    worst = {"lattice": 0.0, "mlp": 0.0}
    for key, (layers, enc, pts, shift) in leaves.items():
        mod = m.models[key]
        half = np.float32(0.5)
        p = (((pts * (np.float32(1) / half)).astype(np.float32) + np.float32(1)) / np.float32(2)).astype(np.float32)
        want = OP.encode_backward(enc.grad[:, :48].numpy(), p, np.geomspace(1.0, 1e-4, num=24), shift, 1 << 18)
        got = mod.pos_encoder.encoder.lattice_values.grad.cpu().numpy().astype(np.float64)
        assert np.abs(want).max() > 0
        worst["lattice"] = max(worst["lattice"], float(np.abs(got - want).max() / np.abs(want).max()))
        lin = [l for l in mod.mlp.layers if isinstance(l, torch.nn.Linear)]
        for l, (w, b) in zip(lin, layers):
            for got_t, want_t in ((l.weight.grad, w.grad), (l.bias.grad, b.grad)):
                s_ = float(want_t.abs().max())
                assert s_ > 0
                worst["mlp"] = max(worst["mlp"], float((got_t.cpu() - want_t).abs().max()) / s_)
                assert torch.nn.functional.cosine_similarity(got_t.cpu().flatten(), want_t.flatten(), dim=0) > 0.999, key
    print(f"MEASURED config2_K5 grad_rel_max lattice={worst['lattice']:.3e} mlp={worst['mlp']:.3e}")
    assert worst["lattice"] <= CONFIG2_GRAD_REL_MAX["lattice"] and worst["mlp"] <= CONFIG2_GRAD_REL_MAX["mlp"]
    # ---- the training loop's step order on this configuration
    m.init_optim()
    gt2 = gt * 0.2
    hist = []
    for it in range(24):
        l, _ = train_step(m, o, d, gt2, iter_nr=it, is_first_iter=it == 0, sync_losses=True)
        hist.append(float(l["loss"]))
    assert all(np.isfinite(hist)) and hist[-1] < 0.9 * hist[0], hist      # measured 0.701 -> 0.568


@pytest.mark.gpu
def test_config2_full_size_training_loop_properties():
    """configs[2] at FULL size (800 x 800 views, K = 5 subdiv-6 shells, permutohedral appearance, the
    reference's dynamic batch steering the hit count to 49 152): the loop of `bench.py --workload
    train-permuto` — size-independent properties: the hit count converges to the target, the ray count
    follows nr_rays <- nr_rays * target / hits (trainer.py:289-304), the loss is finite, every
    parameter tensor (10 lattices of 2^18 x 24 x 2, 10 MLPs) receives updates."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "train-permuto", "--res", "800",
                        "--shells", "5", "--subdiv", "6", "--views", "8", "--steps", "40", "--warmup", "40",
                        "--target-hits", "49152"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    dline = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    print("MEASURED config2_full", {k: dline[k] for k in ("value", "hits_per_iter", "rays_per_iter", "fixed_share")})
    assert dline["unit"] == "it/s" and dline["value"] > 0
    assert abs(dline["hits_per_iter"] - 49152) < 0.15 * 49152                # the dynamic batch found its target
    assert 49152 / 5 < dline["rays_per_iter"] < 49152 * 4                     # h = 0.29 per (ray, shell), K = 5
    assert dline["config"]["parameters"] > 10 * 24 * (1 << 18) * 2


@pytest.mark.gpu
def test_reference_texture_switches_reach_the_renderer():
    """VERDICT r4 missing #2: VolSurfs takes the reference's four hyper-parameters
    (config/volsurfs/base_5.cfg:16-19 -> volsurfs.py:149-153).  anchor renders (and differs from lerp: one
    texel per hit instead of a blend of four, trains too); un-quantised f16 rows and raw (un-squeezed) rows render."""
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    o, d = pinhole_rays(48, 48, focal=80.0)
    outs = {}
    for name, kw in (("lerp", {}), ("anchor", dict(using_neural_textures_anchor=1, using_neural_textures_lerp=0))):
        m = VolSurfs(nested_shells(K=2, subdiv=3), max_rays=4096, textures_res=(256, 128, 64, 32), seed=5, **kw)
        g = torch.Generator().manual_seed(0)
        with torch.no_grad():
            m.bank.tables.copy_((torch.rand(m.bank.tables.shape, generator=g) * 2 - 1).cuda())
        m.bank.refresh_half_params()
        assert m.bank.anchor == (name == "anchor") and int(m.bank.plan.anchor) == int(name == "anchor")
        outs[name] = m.render_rays(o, d, iter_nr=0)["renders"]["ray_traced"]["rgb"].clone()
        if name == "anchor":
            m.init_optim()
            gt = torch.rand(48 * 48, 3, device="cuda")
            m.grad_scale = float(48 * 48)
            l, _, _ = m(o, d, gt, None, 0)
            l["loss"].backward()
            assert m.bank.tables.grad.abs().sum() > 0
    assert torch.isfinite(outs["anchor"]).all() and not torch.equal(outs["anchor"], outs["lerp"])
    # un-quantised rows render (continuous texel values: differs from the 8-bit default, finite)
    m = VolSurfs(nested_shells(K=2, subdiv=3), max_rays=4096, textures_res=(256, 128, 64, 32), seed=5, using_sh_quantization=0)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        m.bank.tables.copy_((torch.rand(m.bank.tables.shape, generator=g) * 2 - 1).cuda())
    m.bank.refresh_half_params()
    nq = m.render_rays(o, d, iter_nr=0)["renders"]["ray_traced"]["rgb"]
    assert torch.isfinite(nq).all() and not torch.equal(nq, outs["lerp"]) and (nq - outs["lerp"]).abs().max() < 0.1
    # raw (un-squeezed) rows render too: the network output itself is the SH coefficient
    m = VolSurfs(nested_shells(K=2, subdiv=3), max_rays=4096, textures_res=(256, 128, 64, 32), seed=5,
                 using_sh_quantization=0, using_sh_squeezing=0)
    assert m.bank.row_format == 2
    raw = m.render_rays(o, d, iter_nr=0)["renders"]["ray_traced"]["rgb"]
    assert torch.isfinite(raw).all() and not torch.equal(raw, outs["lerp"])
    with pytest.raises(ValueError):
        VolSurfs(nested_shells(K=1, subdiv=2), max_rays=1024, using_neural_textures_anchor=1)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(), dict(is_inner_mesh_solid=True, rgb_normal_dep=True)])
def test_fused_legacy_step_equals_the_autograd_step(cfg):
    """VolSurfs.fused_legacy_forward / _backward (the legacy training step with every launch called directly: no
    autograd graph, no engine pass) against forward() + loss.backward(): the same kernels in the same order — the
    same loss and the same gradients, with and without persistent .grad buffers (optim.FusedAdam's)."""
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    kw = dict(max_rays=4096, using_neural_textures=False, rgb_pos_encoder_type="permutohash",
              rgb_mlp_layers_dims=(64, 32), bb_sides=1.0)
    kw.update(cfg)
    m = VolSurfs(nested_shells(K=3, subdiv=3), **kw)
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for mod in m.models.values():
            p = mod.pos_encoder.encoder.lattice_values
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1).cuda())
    o, d = pinhole_rays(40, 40, focal=70.0)
    gt = torch.rand(1600, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    assert m.supports_fused_legacy_step(o)
    for persistent in (False, True):
        for p in m.parameters():
            p.grad = torch.zeros_like(p) if persistent else None
        losses, _, _ = m(o, d, gt, None, 3, return_samples=False)
        (losses["loss"] * 0.5).backward()
        want = [None if p.grad is None else p.grad.clone() for p in m.parameters()]
        want_loss = losses["loss"].item()
        for p in m.parameters():
            p.grad = torch.zeros_like(p) if persistent else None
        loss, state = m.fused_legacy_forward(o, d, gt, iter_nr=3, loss_weight=0.5)
        m.fused_legacy_backward(state)
        assert loss.item() == want_loss
        n = 0
        for p, w in zip(m.parameters(), want):
            assert (p.grad is None) == (w is None)
            if w is not None and w.abs().max() > 0:
                n += 1
                assert (p.grad - w).abs().max().item() <= 2e-6 * w.abs().max().item()
        assert n >= 6


# measured on MI355X (printed as MEASURED shared ...): bounds = 2x measured
SHARED_GRAD_REL_MAX = {"weights": 4e-3, "tables": 3.1e-2}      # measured: 1.6e-3 / 1.5e-2 (the raw-row case; the others 1e-3 / 2.6e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(),        # (independent models: the same scene's baseline for the bounds)
                                 dict(are_volsurfs_colors_indep=0, are_volsurfs_alphas_indep=0),
                                 dict(are_volsurfs_colors_indep=0, are_volsurfs_alphas_indep=0, is_inner_mesh_solid=True),
                                 dict(are_volsurfs_alphas_indep=0, using_sh_quantization=0, using_sh_squeezing=0)])
def test_shared_appearance_models_on_the_neural_texture_branch_match_the_oracle(cfg, tmp_path):
    """VERDICT r5 missing #2 / #1: are_volsurfs_colors_indep = 0 / are_volsurfs_alphas_indep = 0 on the NEURAL-TEXTURE
    branch (methods/volsurfs.py:159-165, 200-206, 524-527, 553-556) — one models["rgb"] / models["alpha"] for all shells
    — and the un-squeezed rows (models/neural_texture.py:157-187), through VolSurfs.forward + backward against
    oracle.pipeline.render_step, whose tex_index maps every shell to the one model (its leaves collect all shells'
    gradients, as the reference's single module does).  A solid inner mesh + a shared alpha model = no alpha model on
    any shell (the reference's loop leaves at i = 0 with None).  Checkpoints carry the reference's keys."""
    import os
    from oracle import pipeline as opipe
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    K, res = 3, 56
    meshes = nested_shells(K=K, subdiv=3)
    # (full-resolution textures: on 256 .. 32 the 3 136 rays pile hundreds of hits on each texel of the coarse
    #  degrees and the f16 gradient rows — the kernel's and the oracle's — carry 5-15 % noise, shared or not)
    m = VolSurfs(meshes, max_rays=4096, seed=7, **cfg)
    b = m.bank
    sr, sa = not cfg.get("are_volsurfs_colors_indep", 1), not cfg.get("are_volsurfs_alphas_indep", 1)
    solid = bool(cfg.get("is_inner_mesh_solid"))
    raw = not cfg.get("using_sh_squeezing", 1)
    assert (b.shared_rgb, b.shared_alpha) == (sr, sa) and (int(b.plan.shared_rgb), int(b.plan.shared_alpha)) == (int(sr), int(sa))
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        own = torch.tensor([float(b.param_tex(x) == x and b.tex_channels(x) > 0) for x in range(b.n_tex)]).view(-1, 1, 1)
        b.tables.copy_(((torch.rand(b.tables.shape, generator=g) * 2 - 1) * own).cuda())
        if raw:
            b.weights.mul_(0.3)       # raw SH coefficients: keep the sums out of the sigmoid's saturation
    b.refresh_half_params()
    o, d = pinhole_rays(res, res, focal=1.6 * res, cam_pos=(0.0, 0.0, -1.5))
    gt = torch.rand(res * res, 3, generator=torch.Generator().manual_seed(1)).cuda()
    m.init_optim()
    m.grad_scale = 16.0 * res * res
    loss, _, _ = m(o, d, gt, None, 0)
    loss["loss"].backward()
    with torch.no_grad():
        rgb = m.render_rays(o, d, iter_nr=0)["renders"]["ray_traced"]["rgb"].detach()
    torch.cuda.synchronize()
    flags = dict(quantize_output=False, squeeze_output=False) if raw else None
    ref = opipe.render_step([(x.vertices.cpu().numpy(), x.faces.cpu().numpy(), x.get_faces_uvs().cpu()) for x in meshes],
                            b.tables_h.cpu().float(), b.weights_h.cpu().float(),
                            lambda s, t, dg: b.param_tex(b.tex_index(s, t, dg)), b.tex_res, o.cpu().numpy(),
                            d.cpu().numpy(), gt.cpu(), loss_scale=128.0 * (64.0 if raw else 1.0), tex_flags=flags,
                            has_alpha=lambda s: not (solid and (s == 0 or sa)))
    e = np.abs(rgb.cpu().numpy() - ref["rgb"])
    # (one flipped 8-bit texel moves a pixel by a few fp16 ulps: measured <= 3.4e-3 here, <= 4.9e-3 in tests/test_parity_report.py)
    assert np.median(e) == 0.0 and (e <= 1e-4).mean() > 0.998 and e.max() < 1e-2, (e.max(), (e > 1e-4).mean())
    owners = [x for x in range(b.n_tex) if b.tex_channels(x) and b.param_tex(x) == x]
    assert sorted(ref["grads"]) == owners
    assert len(owners) == (4 if sr else 4 * K) + (0 if (sa and solid) else 4 if sa else 4 * (K - solid))
    gw, gtb = b.weights.grad.cpu(), b.tables.grad.cpu()
    worst_w = worst_t = 0.0
    for x, (g_t, g_w) in ref["grads"].items():
        assert torch.nn.functional.cosine_similarity(gw[x], g_w, dim=0) > 0.995
        assert torch.nn.functional.cosine_similarity(gtb[x].flatten(), g_t.flatten(), dim=0) > 0.995
        worst_w = max(worst_w, float((gw[x] - g_w).abs().max() / g_w.abs().max()))
        worst_t = max(worst_t, float((gtb[x] - g_t).abs().max() / g_t.abs().max()))
        print(f"  tex {x}: w {float((gw[x] - g_w).abs().max() / g_w.abs().max()):.2e} t {float((gtb[x] - g_t).abs().max() / g_t.abs().max()):.2e} "
              f"|g_t|max {float(g_t.abs().max()):.2e}")
    print(f"MEASURED shared cfg={cfg} rgb_max={e.max():.3e} frac_over_1e-4={(e > 1e-4).mean():.3e} "
          f"gw_rel_max={worst_w:.3e} gt_rel_max={worst_t:.3e}")
    assert worst_w <= SHARED_GRAD_REL_MAX["weights"] and worst_t <= SHARED_GRAD_REL_MAX["tables"]
    for x in range(b.n_tex):          # a shell that reads the shared model owns nothing and receives nothing
        if b.param_tex(x) != x:
            assert gtb[x].abs().max() == 0 and gw[x].abs().max() == 0
    # one Adam step moves the shared parameters only; the checkpoint holds the reference's model keys
    m.optim_step()
    m.save_checkpoints_path = m.load_checkpoints_path = str(tmp_path)
    path = m.save(1)
    names = sorted(f[:-3] for f in os.listdir(path) if f.startswith(("rgb", "alpha")))
    want = (["rgb"] if sr else [f"rgb_{i}" for i in range(K)]) + \
           ([] if (sa and solid) else ["alpha"] if sa else [f"alpha_{i}" for i in range(int(solid), K)])
    assert names == sorted(want), names
    before = (b.tables.detach().clone(), b.weights.detach().clone())
    with torch.no_grad():
        b.tables.zero_()
    m.load(1)
    assert torch.equal(b.tables, before[0]) and torch.equal(b.weights, before[1])
