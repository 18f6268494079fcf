"""Packed ops of the background path (A8/A9/A11): oracle known answers (CPU) and
HIP kernels vs oracle + the reference-glue fixture (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import packed as OP


def _ragged(n_rays, seed, max_n=70):
    g = np.random.default_rng(seed)
    counts = g.integers(0, max_n, n_rays)
    counts[:4] = [0, 1, 32, 33]
    ends = np.cumsum(counts)
    return np.stack([ends - counts, ends], 1).astype(np.int32), int(ends[-1])


def test_oracle_worked_example_and_pcg32():
    se = np.array([[0, 3]], np.int32)
    T, b = OP.cumprod_fwd(se, np.array([0.9, 0.5, 0.1], np.float32))
    # kernels/volsurfs/VolumeRenderingGPU.cuh:60-62
    np.testing.assert_allclose(T, [1.0, 0.9, 0.45], rtol=1e-6)
    np.testing.assert_allclose(b, [0.45], rtol=1e-6)
    # PCG32 reference stream (pcg-c demo, seed (42, 54)): published first outputs
    r = OP.Pcg32(0, (54 << 1) | 1)
    r.next_uint(); r.state = (r.state + 42) & r.M64; r.next_uint()
    assert [r.next_uint() for _ in range(3)] == [0xa15c02b7, 0x7b47f409, 0xba1d3330]
    # advance(k) == k steps
    a, b2 = OP.Pcg32(), OP.Pcg32()
    for _ in range(37):
        a.next_uint()
    b2.advance(37)
    assert a.state == b2.state


def test_oracle_cumprod_backward_is_the_gradient():
    se, S = _ragged(20, 1, 12)
    g = np.random.default_rng(2)
    a = g.uniform(0.2, 0.99, S).astype(np.float32)
    gT = g.standard_normal(S).astype(np.float32)
    gb = g.standard_normal(20).astype(np.float32)
    T, bgT = OP.cumprod_fwd(se, a)
    lv = OP.cumsum(se, gT * T, True)
    ga = OP.cumprod_bwd(se, gb, a, bgT, lv)
    at = torch.tensor(a, dtype=torch.float64, requires_grad=True)
    loss = 0
    for r in range(20):
        i0, i1 = se[r]
        if i1 > i0:
            seg = at[i0:i1]
            Tt = torch.cat([torch.ones(1, dtype=torch.float64), torch.cumprod(seg[:-1], 0)])
            loss = loss + (Tt * torch.tensor(gT[i0:i1], dtype=torch.float64)).sum() + Tt[-1] * float(gb[r])
    loss.backward()
    np.testing.assert_allclose(ga, at.grad.numpy(), rtol=2e-4, atol=1e-5)


def test_oracle_sampler_properties():
    o = np.zeros((3, 3), np.float32)
    d = np.tile(np.array([[0, 0, 1.0]], np.float32), (3, 1))
    ts = np.array([0.5, 1.0, 2.0], np.float32)
    s = OP.sample_bg(o, d, ts, 100.0, 32)
    z = s["samples_z"].reshape(3, 32)
    assert np.allclose(z[:, 0], ts) and np.all(np.diff(z, axis=1) >= 0) and np.all(z <= 100.0)
    assert np.all(z[:, -1] == 100.0)             # s -> 0: 1/(0+1e-6) - 1 clamps to t_far
    sj = OP.sample_bg(o, d, ts, 100.0, 32, jitter=True, rng=OP.Pcg32())
    zj = sj["samples_z"].reshape(3, 32)
    assert np.all(zj[:, 0] == z[:, 0]) and np.all(zj[:, -1] == z[:, -1]) and np.all(zj <= z + 1e-6)
    c3d, cz = OP.contract(o, s["ray_start_end_idx"], s["samples_3d"], s["samples_z"])
    assert np.all(np.linalg.norm(c3d * 2, axis=1) <= 2.0 + 1e-5)     # contracted into radius 1 (scale 2)


# ------------------------------- GPU ---------------------------------------
def _pack(se):
    from volsurfs_amd.volsurfs import RaySamplesPacked
    p = RaySamplesPacked(se.shape[0], int(se[-1, 1]))
    p.ray_start_end_idx = torch.from_numpy(se).cuda()
    return p


@pytest.mark.gpu
def test_hip_packed_ops_vs_oracle():
    from volsurfs_amd.volsurfs import VolumeRendering as VR
    se, S = _ragged(3000, 3)
    g = np.random.default_rng(4)
    a = g.uniform(0.0, 1.0, (S, 1)).astype(np.float32)
    v3 = g.standard_normal((S, 3)).astype(np.float32)
    w = g.uniform(0, 0.1, (S, 1)).astype(np.float32)
    z = np.sort(g.uniform(0, 5, (S, 1)).astype(np.float32), axis=0)
    p = _pack(se)
    p.samples_z = torch.from_numpy(z).cuda()
    cu = lambda x: torch.from_numpy(x).cuda()
    T, bgT = VR.cumprod_one_minus_alpha_to_transmittance(p, cu(a))
    Tr, br = OP.cumprod_fwd(se, a)
    np.testing.assert_allclose(T.cpu().numpy()[:, 0], Tr, rtol=2e-6, atol=1e-30)
    np.testing.assert_allclose(bgT.cpu().numpy()[:, 0], br, rtol=2e-6, atol=1e-30)
    for inv in (False, True):
        c = VR.cumsum_over_rays(p, cu(w), inv)
        np.testing.assert_allclose(c.cpu().numpy()[:, 0], OP.cumsum(se, w, inv), rtol=2e-6, atol=1e-7)
    o3 = VR.integrate_with_weights_3d(p, cu(v3), cu(w))
    np.testing.assert_allclose(o3.cpu().numpy(), OP.integrate_fwd(se, v3, w), rtol=1e-5, atol=1e-6)
    o1 = VR.integrate_with_weights_1d(p, cu(z), cu(w))
    np.testing.assert_allclose(o1.cpu().numpy(), OP.integrate_fwd(se, z, w), rtol=1e-5, atol=1e-6)
    gr = g.standard_normal((3000, 3)).astype(np.float32)
    for compat in (False, True):
        VR.bug_compat = compat
        gv, gw = VR.integrate_with_weights_3d_backward(cu(gr), p, cu(v3), cu(w), o3)
        rv, rw = OP.integrate_bwd(se, gr, v3, w, bug_compat=compat)
        np.testing.assert_allclose(gv.cpu().numpy(), rv, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(gw.cpu().numpy()[:, 0], rw, rtol=1e-5, atol=1e-6)
        md = VR.median_depth_over_rays(p, cu(w), 0.5)
        # identical decisions except when the scanned cumsum straddles the threshold by 1 ulp
        ref = OP.median_depth(se, z, w, 0.5, fallback_compat=compat)
        assert (md.cpu().numpy()[:, 0] != ref).mean() < 2e-3
    VR.bug_compat = True            # the default: what the reference computes
    gb = g.standard_normal((3000, 1)).astype(np.float32)
    lv = OP.cumsum(se, (gr[:1].sum() * 0 + g.standard_normal(S).astype(np.float32)) * Tr, True)
    ga = VR.cumprod_one_minus_alpha_to_transmittance_backward(cu(a), cu(gb), p, cu(a), T, bgT,
                                                              cu(lv[:, None]))
    np.testing.assert_allclose(ga.cpu().numpy()[:, 0], OP.cumprod_bwd(se, gb[:, 0], a, br, lv),
                               rtol=1e-5, atol=1e-6)
    # rays without samples: bg transmittance 1, zero colour
    assert bgT[0].item() == 1.0 and (o3[0] == 0).all()


@pytest.mark.gpu
def test_hip_sampler_contract_update_dt_vs_oracle():
    from volsurfs_amd.volsurfs import RaySampler
    g = np.random.default_rng(5)
    N = 777
    o = (g.standard_normal((N, 3)) * 0.2).astype(np.float32)
    d = g.standard_normal((N, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ts = g.uniform(0.1, 2.0, (N, 1)).astype(np.float32)
    for jitter in (False, True):
        RaySampler.m_rng.__init__()
        p = RaySampler.compute_samples_bg(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(),
                                          torch.from_numpy(ts).cuda(), 100.0, 32, jitter)
        ref = OP.sample_bg(o, d, ts[:, 0], 100.0, 32, jitter=jitter, rng=OP.Pcg32())
        assert np.array_equal(p.samples_z.cpu().numpy()[:, 0], ref["samples_z"])
        assert np.array_equal(p.ray_start_end_idx.cpu().numpy(), ref["ray_start_end_idx"])
        np.testing.assert_allclose(p.samples_3d.cpu().numpy(), ref["samples_3d"], rtol=0, atol=1e-6)
        assert np.array_equal(p.samples_dirs.cpu().numpy(), ref["samples_dirs"])
        np.testing.assert_allclose(p.ray_max_dt.cpu().numpy()[:, 0], ref["ray_max_dt"], rtol=1e-6)
        assert p.get_total_nr_samples() == N * 32 and not p.is_empty()
        c = RaySampler.contract_samples(p)
        r3d, rz = OP.contract(o, ref["ray_start_end_idx"], p.samples_3d.cpu().numpy(),
                              p.samples_z.cpu().numpy())
        np.testing.assert_allclose(c.samples_3d.cpu().numpy(), r3d, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(c.samples_z.cpu().numpy()[:, 0], rz, rtol=1e-6, atol=1e-7)
        rdt = OP.update_dt(ref["ray_start_end_idx"], p.ray_max_dt.cpu().numpy()[:, 0],
                           np.full(N, 100.0, np.float32), c.samples_z.cpu().numpy(), True)
        np.testing.assert_allclose(c.samples_dt.cpu().numpy()[:, 0], rdt, rtol=1e-6, atol=1e-7)
    # after a jittered call the global RNG moved on by 2^32 (RaySampler.cu:139-142)
    r = OP.Pcg32(); r.advance()
    assert RaySampler.m_rng.state == r.state


@pytest.mark.gpu
def test_hip_glue_matches_reference_fixture(golden_dir):
    from volsurfs_amd import volsurfs as V
    z = np.load(os.path.join(golden_dir, "packed_glue.npz"))
    p = _pack(z["start_end"])
    assert V.VolumeRendering.bug_compat is True  # default: the fixture carries the reference's :1021 behaviour
    try:
        density = torch.from_numpy(z["density"]).cuda().requires_grad_(True)
        rgb = torch.from_numpy(z["rgb"]).cuda().requires_grad_(True)
        dt = torch.from_numpy(z["dt"]).cuda()
        alpha = 1.0 - torch.exp(-density * dt)
        T, bgT = V.CumprodOneMinusAlphaToTransmittanceFunc.apply(p, (1 - alpha) + 1e-6)
        pred = V.IntegrateWithWeights3DFunc.apply(p, rgb, alpha * T)
        loss = (torch.from_numpy(z["gt"]).cuda() - pred).abs().mean() + 0.1 * bgT.mean()
        loss.backward()
    finally:
        V.VolumeRendering.bug_compat = True
    np.testing.assert_allclose(T.detach().cpu().numpy(), z["T"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(bgT.detach().cpu().numpy(), z["bgT"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(pred.detach().cpu().numpy(), z["pred"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rgb.grad.cpu().numpy(), z["g_rgb"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(density.grad.cpu().numpy(), z["g_density"], rtol=2e-3, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("bug_compat", [True, False])
def test_fused_bg_composite_equals_the_op_sequence(golden_dir, bug_compat):
    """vsa_packed_composite_fwd / _bwd (one launch each way) against the chain of single ops it
    replaces in render_contracted_bg (background.py:93-111), on ragged rays (0, 1, 32, 33 .. 69
    samples) and on the reference's own fixture: same fp32 operations in the same order, so the
    colours, the weights and both gradients are bit-identical."""
    from volsurfs_amd import volsurfs as V
    from volsurfs_amd.background import _FusedBgComposite
    z = np.load(os.path.join(golden_dir, "packed_glue.npz"))
    se_r, S_r = _ragged(2000, 11)
    g = np.random.default_rng(12)
    cases = [(z["start_end"], z["density"], z["rgb"], z["dt"]),
             (se_r, g.uniform(0.0, 30.0, (S_r, 1)).astype(np.float32), g.uniform(0, 1, (S_r, 3)).astype(np.float32),
              g.uniform(1e-3, 0.2, (S_r, 1)).astype(np.float32))]
    V.VolumeRendering.bug_compat = bug_compat
    try:
        for se, dens, col, dts in cases:
            p = _pack(se)
            p.samples_dt = torch.from_numpy(dts).cuda()
            gp = torch.from_numpy(g.standard_normal((se.shape[0], 3)).astype(np.float32)).cuda()
            outs = []
            for fused in (False, True):
                density = torch.from_numpy(dens).cuda().requires_grad_(True)
                rgb = torch.from_numpy(col).cuda().requires_grad_(True)
                if fused:
                    pred, w = _FusedBgComposite.apply(p, rgb, density)
                else:
                    alpha = 1.0 - torch.exp(-density.view(-1, 1) * p.samples_dt)
                    T, _ = V.CumprodOneMinusAlphaToTransmittanceFunc.apply(p, (1 - alpha) + 1e-6)
                    w = alpha * T
                    pred = V.IntegrateWithWeights3DFunc.apply(p, rgb, w)
                (pred * gp).sum().backward()
                outs.append((pred.detach(), w.detach(), rgb.grad, density.grad))
            for a, b, name in zip(outs[0], outs[1], ("pred_rgb", "weights", "g_rgb", "g_density")):
                assert torch.equal(a, b), (name, float((a - b).abs().max()))
        # and against the fixture of the reference's own glue (bug_compat on: what the reference computes)
        if bug_compat:
            p = _pack(z["start_end"])
            p.samples_dt = torch.from_numpy(z["dt"]).cuda()
            density = torch.from_numpy(z["density"]).cuda().requires_grad_(True)
            rgb = torch.from_numpy(z["rgb"]).cuda().requires_grad_(True)
            pred, _ = _FusedBgComposite.apply(p, rgb, density)
            np.testing.assert_allclose(pred.detach().cpu().numpy(), z["pred"], rtol=1e-5, atol=1e-6)
    finally:
        V.VolumeRendering.bug_compat = True


@pytest.mark.gpu
def test_render_contracted_bg_end_to_end_vs_oracle():
    """background.py:31-141 with a tiny stand-in radiance model: HIP packed path vs the
    oracle's serial restatement, values and gradients."""
    from volsurfs_amd.background import BoundingBox, intersect_bounding_primitive, render_contracted_bg
    torch.manual_seed(0)
    N = 500
    o = torch.zeros(N, 3, device="cuda") + torch.tensor([0.0, 0.0, -0.2], device="cuda")
    d = torch.nn.functional.normalize(torch.randn(N, 3, device="cuda"), dim=-1)
    rc = intersect_bounding_primitive(BoundingBox(1.0), o, d)
    assert rc["is_hit"].all() and (rc["t_far"] > 0).all()
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4)).cuda()

    def model_bg(p, dirs, it):
        y = net(torch.cat([p, dirs], 1))
        return torch.sigmoid(y[:, :3]), torch.nn.functional.softplus(y[:, 3:])
    # the checker here is torch autograd over the intended maths, so the reference's :1021 slip
    # (on by default) is switched off for this one call
    from volsurfs_amd.volsurfs import VolumeRendering as _VR
    _VR.bug_compat = False
    try:
        res = render_contracted_bg(model_bg, rc, 32)
        res["pred_rgb"].sum().backward()
    finally:
        _VR.bug_compat = True
    g_hip = [p.grad.clone() for p in net.parameters()]
    # oracle: same model on the CPU, packed ops from oracle/packed.py, autograd by torch
    s = OP.sample_bg(o.cpu().numpy(), d.cpu().numpy(), rc["t_far"].cpu().numpy()[:, 0], 100.0, 32)
    c3d, cz = OP.contract(o.cpu().numpy(), s["ray_start_end_idx"], s["samples_3d"], s["samples_z"])
    dt = OP.update_dt(s["ray_start_end_idx"], s["ray_max_dt"], np.full(N, 100.0, np.float32), cz, True)
    netc = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4))
    netc.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
    y = netc(torch.cat([torch.from_numpy(c3d), torch.from_numpy(s["samples_dirs"])], 1))
    rgb, dens = torch.sigmoid(y[:, :3]), torch.nn.functional.softplus(y[:, 3:])
    alpha = 1.0 - torch.exp(-dens * torch.from_numpy(dt)[:, None])
    a = ((1 - alpha) + 1e-6).view(N, 32)
    T = torch.cat([torch.ones(N, 1), torch.cumprod(a[:, :-1], 1)], 1).reshape(-1, 1)
    pred = ((alpha * T) * rgb).view(N, 32, 3).sum(1)
    np.testing.assert_allclose(res["pred_rgb"].detach().cpu().numpy(), pred.detach().numpy(),
                               rtol=1e-4, atol=1e-5)
    pred.sum().backward()
    for gh, pc in zip(g_hip, netc.parameters()):
        np.testing.assert_allclose(gh.cpu().numpy(), pc.grad.numpy(), rtol=2e-3, atol=2e-4)
    md = OP.median_depth(s["ray_start_end_idx"], s["samples_z"], (alpha * T).detach().numpy(), 0.5)
    assert (res["median_depth"].cpu().numpy()[:, 0] != md).mean() < 0.01


def test_oracle_sibling_ops_known_answers():
    """SURVEY §8f row 4 ops: hand-checkable cases of the restated kernels."""
    se = np.array([[0, 3], [3, 3], [3, 5], [5, 6]], np.int32)
    v = np.arange(12, dtype=np.float32).reshape(6, 2)
    per_ray, per_sample = OP.sum_over_rays(se, v)
    assert per_ray.tolist() == [[6.0, 9.0], [0.0, 0.0], [14.0, 16.0], [10.0, 11.0]]
    assert per_sample[:3].tolist() == [[6.0, 9.0]] * 3 and per_sample[5].tolist() == [10.0, 11.0]
    g = OP.sum_over_rays_bwd(se, np.ones((4, 2), np.float32), v)
    assert np.array_equal(g, v + 1)
    # cdf: exclusive running sum; a ray whose weights sum to 1 keeps cdf[-1] = 1 - w_last,
    # snapped to exactly 1 only if it is more than 1e-3 away
    w = np.array([0.25, 0.25, 0.5, 0.3, 0.7, 1.0], np.float32)
    cdf = OP.compute_cdf(se, w)
    assert cdf.tolist() == [0.0, 0.25, 1.0, 0.0, 1.0, 0.0]      # last entries snapped; 1-sample ray untouched
    # sdf2alpha: a surface crossing gives alpha in (0, 1), the last sample of each ray stays 0
    a = OP.sdf2alpha(se, np.full(6, 0.1, np.float32), np.array([0.2, 0.1, -0.1, 0.3, 0.25, 0.0], np.float32),
                     np.full(6, 50.0, np.float32))
    assert a[2] == 0 and a[4] == 0 and a[5] == 0 and 0 < a[0] < a[1] < 1


@pytest.mark.gpu
def test_hip_sibling_ops_vs_oracle():
    from volsurfs_amd.volsurfs import SumOverRaysFunc, VolumeRendering as VR
    se, S = _ragged(2500, 11)
    g = np.random.default_rng(12)
    cu = lambda x: torch.from_numpy(x).cuda()
    p = _pack(se)
    for D in (1, 2, 3, 32):
        v = g.standard_normal((S, D)).astype(np.float32)
        per_ray, per_sample = VR.sum_over_rays(p, cu(v))
        rr, rs = OP.sum_over_rays(se, v)
        np.testing.assert_allclose(per_ray.cpu().numpy(), rr, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(per_sample.cpu().numpy(), rs, rtol=1e-5, atol=1e-5)
        gr, gs = g.standard_normal((2500, D)).astype(np.float32), g.standard_normal((S, D)).astype(np.float32)
        gv = VR.sum_over_rays_backward(cu(gr), cu(gs), p, cu(v))
        assert np.array_equal(gv.cpu().numpy(), OP.sum_over_rays_bwd(se, gr, gs))
    # autograd glue (volume_rendering_funcs.py:244-272)
    vt = cu(g.standard_normal((S, 3)).astype(np.float32)).requires_grad_(True)
    pr, ps = SumOverRaysFunc.apply(p, vt)
    (pr.sum() + (ps * 2).sum()).backward()
    # the reference's backward (VolumeRenderingGPU.cuh:1036-1077) adds the ray's gradient to the
    # sample's OWN per-sample gradient (it does not sum the per-sample gradients over the ray)
    torch.testing.assert_close(vt.grad, torch.full_like(vt.grad, 3.0))
    # cdf
    w = g.uniform(0, 0.1, (S, 1)).astype(np.float32)
    for r in range(0, 2500, 7):                       # some rays normalised to sum 1
        i0, i1 = se[r]
        if i1 - i0 >= 2:
            w[i0:i1] /= w[i0:i1].sum()
    cdf = VR.compute_cdf(p, cu(w)).cpu().numpy()[:, 0]
    ref = OP.compute_cdf(se, w)
    np.testing.assert_allclose(cdf, ref, rtol=1e-5, atol=1e-6)
    assert (cdf == 1.0).sum() == (ref == 1.0).sum() > 50
    # sdf2alpha (needs dt on the pack)
    p.samples_dt = cu(g.uniform(0.01, 0.2, (S, 1)).astype(np.float32))
    p.has_dt = True
    sdf = g.standard_normal((S, 1)).astype(np.float32) * 0.2
    beta = g.uniform(5, 200, (S, 1)).astype(np.float32)
    al = VR.sdf2alpha(p, cu(sdf), cu(beta)).cpu().numpy()[:, 0]
    ref = OP.sdf2alpha(se, p.samples_dt.cpu().numpy(), sdf, beta)
    np.testing.assert_allclose(al, ref, rtol=2e-5, atol=2e-6)   # expf implementations differ by an ulp
    last = se[se[:, 1] > se[:, 0], 1] - 1
    assert (al[last] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("jitter", [False, True])
def test_hip_fg_sampler_compaction_importance_sampling_vs_oracle(jitter):
    """RaySampler.compute_samples_fg -> compact_to_valid_samples -> compute_cdf ->
    VolumeRendering.importance_sample (the NeRF / NeuS sampling chain of the sibling methods)."""
    from volsurfs_amd.volsurfs import RaySampler, VolumeRendering as VR, _Pcg32State
    g = np.random.default_rng(21)
    N = 700
    o = g.standard_normal((N, 3)).astype(np.float32) * 0.1
    d = g.standard_normal((N, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t0 = g.uniform(0.1, 0.5, (N, 1)).astype(np.float32)
    t1 = (t0 + g.uniform(-0.1, 1.2, (N, 1))).astype(np.float32)     # some rays exit before they enter
    t1[:3] = t0[:3] + np.array([[0.0], [0.004], [5.0]], np.float32)    # empty, one sample, clamped to max_n
    cu = lambda x: torch.from_numpy(x).cuda()
    RaySampler.m_rng = _Pcg32State()
    VR.m_rng = _Pcg32State()
    rng0 = OP.Pcg32(RaySampler.m_rng.state, RaySampler.m_rng.inc)
    pack = RaySampler.compute_samples_fg(cu(o), cu(d), cu(t0), cu(t1), 0.01, 1, 64, jitter, 0)
    ref = OP.sample_fg(o, d, t0[:, 0], t1[:, 0], 0.01, 1, 64, jitter=jitter, rng=rng0)
    assert pack.is_compacted and pack.get_total_nr_samples() == ref["samples_z"].shape[0] > 5000
    assert np.array_equal(pack.ray_start_end_idx.cpu().numpy(), ref["ray_start_end_idx"])
    assert np.array_equal(pack.samples_z.cpu().numpy()[:, 0], ref["samples_z"])
    np.testing.assert_allclose(pack.samples_3d.cpu().numpy(), ref["samples_3d"], rtol=0, atol=2e-7)
    assert np.array_equal(pack.samples_dirs.cpu().numpy(), ref["samples_dirs"])
    has = ref["ray_max_dt"] >= 0
    assert np.array_equal(pack.ray_max_dt.cpu().numpy()[has, 0], ref["ray_max_dt"][has])
    se = ref["ray_start_end_idx"]
    assert se[0, 1] - se[0, 0] == 0 and se[1, 1] - se[1, 0] == 1 and se[2, 1] - se[2, 0] == 64
    # importance sampling from a cdf over those samples
    S = pack.get_total_nr_samples()
    w = g.uniform(0, 1, (S, 1)).astype(np.float32)
    for r in range(N):
        a, b = se[r]
        if b - a >= 2:
            w[a:b] /= w[a:b].sum()
    cdf = VR.compute_cdf(pack, cu(w))
    multi = (se[:, 1] - se[:, 0]) >= 2                 # the reference never importance-samples 1-sample rays
    keep = np.repeat(multi, se[:, 1] - se[:, 0])
    rng1 = OP.Pcg32(VR.m_rng.state, VR.m_rng.inc)
    imp = VR.importance_sample(pack, cdf, 16, jitter)
    ref_i = OP.importance_sample(o, d, se, ref["samples_z"], cdf.cpu().numpy(), 16, jitter=jitter, rng=rng1)
    got_se, ref_se = imp.ray_start_end_idx.cpu().numpy(), ref_i["ray_start_end_idx"]
    assert np.array_equal(got_se[:, 1] - got_se[:, 0], ref_se[:, 1] - ref_se[:, 0])
    zi, zr = imp.samples_z.cpu().numpy()[:, 0], ref_i["samples_z"]
    rows = np.repeat(multi, ref_se[:, 1] - ref_se[:, 0])
    assert rows.sum() > 5000 and np.array_equal(zi[rows], zr[rows])
    assert (np.diff(zi.reshape(-1, 16)[multi[(ref_se[:, 1] - ref_se[:, 0]) > 0]], axis=1) >= 0).all()   # sorted per ray
    if jitter:      # the static generators were advanced like the reference's m_rng
        assert RaySampler.m_rng.state != rng0.state and VR.m_rng.state != rng1.state


@pytest.mark.gpu
def test_hip_uncontract_inverts_contract_and_one_sample_packs():
    from volsurfs_amd.volsurfs import RaySampler
    g = np.random.default_rng(31)
    N = 400
    o = g.standard_normal((N, 3)).astype(np.float32) * 0.05
    d = g.standard_normal((N, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    cu = lambda x: torch.from_numpy(x).cuda()
    pack = RaySampler.compute_samples_bg(cu(o), cu(d), cu(np.full((N, 1), 0.4, np.float32)), 100.0, 32, False)
    cp = RaySampler.contract_samples(pack)
    up = RaySampler.uncontract_samples(cp)
    ref3, refz = OP.uncontract(o, cp.ray_start_end_idx.cpu().numpy(), cp.samples_3d.cpu().numpy(),
                               cp.samples_z.cpu().numpy())
    np.testing.assert_allclose(up.samples_3d.cpu().numpy(), ref3, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(up.samples_z.cpu().numpy()[:, 0], refz, rtol=2e-6, atol=1e-7)
    # un-contracting the contracted samples gives the original positions back (up to fp32
    # conditioning near the boundary of the contracted ball, where 1 / (2 - norm) blows up)
    p0, p1 = pack.samples_3d.cpu().numpy(), up.samples_3d.cpu().numpy()
    near = np.linalg.norm(p0, axis=1) < 20
    np.testing.assert_allclose(p1[near], p0[near], rtol=2e-3, atol=1e-4)
    assert up.has_dt and up.is_compacted
    one = RaySampler.init_with_one_sample_per_ray(cu(o), cu(d))
    assert one.get_total_nr_samples() == N and one.get_values_dim() == 1
    assert torch.equal(one.samples_3d.cpu(), torch.from_numpy(o)) and (one.samples_z == 0).all()
    assert one.ray_start_end_idx[5].tolist() == [5, 6]


@pytest.mark.gpu
def test_hip_combine_ray_samples_packets_vs_oracle():
    """Uniform + importance packs merged in depth order with a minimum spacing (the NeuS
    up-sampling step of the sibling methods)."""
    from volsurfs_amd.volsurfs import RaySampler, VolumeRendering as VR, _Pcg32State
    g = np.random.default_rng(41)
    N = 500
    o = g.standard_normal((N, 3)).astype(np.float32) * 0.1
    d = g.standard_normal((N, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t0 = g.uniform(0.1, 0.5, (N, 1)).astype(np.float32)
    t1 = (t0 + g.uniform(0.05, 1.0, (N, 1))).astype(np.float32)
    cu = lambda x: torch.from_numpy(x).cuda()
    RaySampler.m_rng, VR.m_rng = _Pcg32State(), _Pcg32State()
    uni = RaySampler.compute_samples_fg(cu(o), cu(d), cu(t0), cu(t1), 0.02, 2, 48, False, 0)
    S = uni.get_total_nr_samples()
    w = g.uniform(0, 1, (S, 1)).astype(np.float32)
    se_u = uni.ray_start_end_idx.cpu().numpy()
    for r in range(N):
        a, b = se_u[r]
        if b - a >= 2:
            w[a:b] /= w[a:b].sum()
    imp = VR.importance_sample(uni, VR.compute_cdf(uni, cu(w)), 8, False)
    comb = VR.combine_ray_samples_packets(uni, imp, 0.004)
    se_i = imp.ray_start_end_idx.cpu().numpy()
    ref_se, ref_z, src = OP.combine_packs(se_u, uni.samples_z.cpu().numpy(), se_i, imp.samples_z.cpu().numpy(), 0.004)
    assert comb.is_compacted and not comb.has_dt
    assert np.array_equal(comb.ray_start_end_idx.cpu().numpy(), ref_se)
    assert np.array_equal(comb.samples_z.cpu().numpy()[:, 0], ref_z)
    pos = [uni.samples_3d.cpu().numpy(), imp.samples_3d.cpu().numpy()]
    ref_pos = np.stack([pos[k][i] for k, i in src])
    assert np.array_equal(comb.samples_3d.cpu().numpy(), ref_pos)
    z = comb.samples_z.cpu().numpy()[:, 0]
    for r in range(0, N, 37):                     # per ray: sorted, spaced by at least min_dist
        a, b = ref_se[r]
        if b - a > 1:
            assert (np.diff(z[a:b]) >= 0.004 - 1e-7).all()
    assert S < comb.get_total_nr_samples() < S + imp.get_total_nr_samples()
