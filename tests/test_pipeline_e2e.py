"""End-to-end: the HIP hot path (trace -> neural textures -> composite -> L1 ->
backward) vs the oracle evaluating the reference's algorithm on the same scene."""
import numpy as np
import pytest
import torch

# per K: (MLP weights, hash tables) max gradient error relative to the tensor's largest entry against
# the independent oracle: 2x the values measured on MI355X (profiles/r03/measured_bounds.txt)
GRAD_BOUND = {1: (8e-4, 1.4e-3), 3: (9e-4, 4e-3)}
# K = 9 (r5): THIS oracle's gradients are the reference's fp16 autograd, whose own rounding noise behind eight
# shells is the larger term — rounds 2-4 asserted (0.28, 0.12) here, a bound nothing could fail.  Now the
# oracle calibrates itself: the same oracle back-propagated under another loss scale (1024 instead of the
# reference's 128: different fp16 roundings, same mathematics) gives the oracle's own noise per tensor, and the
# kernel's worst element must be within K9_NOISE_FACTOR x the oracle's own worst element (measured: oracle
# against itself 0.129 / 0.055 of a tensor's largest entry for weights / tables, kernel against the oracle
# 0.142 / 0.057, i.e. 1.10x / 1.04x; bound 2x).  The tight K = 9 bound — every element against exact
# arithmetic: 3.1e-4 for the weights — is tests/test_parity_report.py's order-matched case.
K9_NOISE_FACTOR = 2.0


def _oracle(pipe, loss_scale=128.0):
    from oracle import pipeline as opipe
    bank = pipe.bank
    meshes = [(m.vertices.cpu().numpy(), m.faces.cpu().numpy(), m.faces_uvs.cpu()) for m in pipe.meshes]
    return opipe.render_step(meshes, bank.tables_h.cpu().float(), bank.weights_h.cpu().float(),
                             bank.tex_index, bank.tex_res, pipe.rays_o.cpu().numpy(),
                             pipe.rays_d.cpu().numpy(), pipe.gt.cpu(), loss_scale=loss_scale)


@pytest.mark.gpu
# (9 shells = 72 textures: more than the 64 the one-texture-per-lane work split handles, i.e. the
# scalar walk of nt_for_each_piece_scalar)
@pytest.mark.parametrize("K,subdiv,res", [(1, 2, 40), (3, 3, 56), (1, 3, 64), (9, 2, 40)])   # third: the shape of BASELINE configs[0] (one 64x64 view, K=1)
def test_pipeline_matches_oracle(K, subdiv, res):
    from volsurfs_amd.pipeline import KShellPipeline
    pipe = KShellPipeline.synthetic(K=K, subdiv=subdiv, res=res, init="spread", seed=5)
    # zoom the camera so that the shells fill the tiny frame
    from volsurfs_amd.camera import pinhole_rays
    pipe.rays_o, pipe.rays_d = pinhole_rays(res, res, focal=1.6 * res, cam_pos=(0.0, 0.0, -1.5))
    rgb = pipe.step()
    torch.cuda.synchronize()
    ref = _oracle(pipe)
    hits, slots = pipe.stats()
    assert hits == int(ref["hit"].sum()) and hits > res * res // 4
    # per-shell appearance before compositing
    e_rgb = np.abs(pipe.to_ray_order(pipe.surfs_rgb).cpu().numpy() - ref["surfs_rgb"])
    e_a = np.abs(pipe.to_ray_order(pipe.surfs_alpha).cpu().numpy() - ref["surfs_alpha"])
    # identical except where an 8-bit texel flipped by one step (fp32 summation
    # order inside the MLP: MFMA vs torch-CPU) — see tests/test_nt_shade.py
    # (measured: <= 2.6e-4 of the values differ, by <= 3.7e-3; bounds = 2x)
    assert (e_rgb > 1e-5).mean() < 6e-4 and e_rgb.max() < 8e-3
    assert (e_a > 1e-5).mean() < 6e-4 and e_a.max() < 8e-3
    # composited colour: BASELINE north_star asks 1e-4 on RGB; fp16 composite => the
    # bulk is bit-identical, the flipped-texel pixels move by a few fp16 ulps
    # (measured, profiles/r02/parity_report.json: 8e-5 .. 7e-4 of the pixels above 1e-4, max 1.5e-3 —
    # one or two fp16 ulps where an 8-bit texel flipped (rate 1.4e-5); the order-matched check of
    # tests/test_parity_report.py, which removes the MLP's summation order, is bit-exact)
    e = np.abs(rgb.cpu().numpy() - ref["rgb"])
    assert np.median(e) == 0.0
    assert (e <= 1e-4).mean() > 0.999        # measured: <= 4.3e-4 of the values above 1e-4, max 1.5e-3
    assert e.max() < 3e-3
    # gradients w.r.t. every hash table and MLP (north_star: 1e-3 on grads, here as a
    # relative bound on each tensor plus direction)
    gw, gt = pipe.bank.weights.grad.cpu(), pipe.bank.tables.grad.cpu()
    assert len(ref["grads"]) == K * 8
    worst_w = worst_t = 0.0
    for x, (g_t, g_w) in ref["grads"].items():
        cw = torch.nn.functional.cosine_similarity(gw[x], g_w, dim=0)
        ct = torch.nn.functional.cosine_similarity(gt[x].flatten(), g_t.flatten(), dim=0)
        # (K = 9: the innermost shells sit behind eight others; their gradients are a few f16 ulps of
        # the gradient rows, so the direction is noisier — 0.987 for the innermost table, the same
        # with the scalar and the per-lane work split)
        cos_min = 0.995 if K <= 3 else 0.98
        assert cw > cos_min and ct > cos_min, (x, cw, ct)
        worst_w = max(worst_w, float((gw[x] - g_w).abs().max() / g_w.abs().max()))
        worst_t = max(worst_t, float((gt[x] - g_t).abs().max() / g_t.abs().max()))
    print(f"MEASURED pipeline_e2e K={K} res={res} rgb_max={e.max():.3e} rgb_frac_over_1e-4={(e > 1e-4).mean():.3e} "
          f"surfs_rgb_max={e_rgb.max():.3e} surfs_frac={(e_rgb > 1e-5).mean():.3e} gw_rel_max={worst_w:.3e} gt_rel_max={worst_t:.3e}")
    # 2x the measured error against the independent oracle (whose gradients are the reference's own
    # fp16 autograd, loss scale 128: itself noisy, and K = 9 puts the inner shells behind eight others)
    if K in GRAD_BOUND:
        assert worst_w <= GRAD_BOUND[K][0] and worst_t <= GRAD_BOUND[K][1]
    else:
        ref2 = _oracle(pipe, loss_scale=1024.0)
        ratio_w = ratio_t = 0.0
        noise_w = noise_t = 0.0
        for x, (g_t, g_w) in ref["grads"].items():
            g_t2, g_w2 = ref2["grads"][x]
            nw = float((g_w2 - g_w).abs().max() / g_w.abs().max())
            nt = float((g_t2 - g_t).abs().max() / g_t.abs().max())
            kw = float((gw[x] - g_w).abs().max() / g_w.abs().max())
            kt = float((gt[x] - g_t).abs().max() / g_t.abs().max())
            noise_w, noise_t = max(noise_w, nw), max(noise_t, nt)
            ratio_w, ratio_t = max(ratio_w, kw / max(nw, 1e-4)), max(ratio_t, kt / max(nt, 1e-4))
        print(f"MEASURED pipeline_e2e K={K} oracle_self_noise weights={noise_w:.3e} tables={noise_t:.3e} "
              f"kernel_over_noise weights={ratio_w:.2f} tables={ratio_t:.2f}")
        assert noise_w > 1e-3 and noise_t > 1e-3          # (the two oracle runs really differ: the premise)
        assert worst_w <= K9_NOISE_FACTOR * noise_w and worst_t <= K9_NOISE_FACTOR * noise_t


# measured on MI355X (printed as MEASURED stress ...): rgb max 1.95e-3 (fp16 ulps where an 8-bit texel flipped),
# 4.1e-4 of the values above 1e-4; gradients against the oracle's fp16 autograd (loss scale 128): weights
# 5.4e-4, tables 2.8e-3 of each tensor's largest entry.  Bounds = 2x measured.
STRESS_GRAD_REL_MAX = {"weights": 1.1e-3, "tables": 5.7e-3}


@pytest.mark.gpu
def test_stress_scene_matches_oracle():
    """the stress shells (mesh.stress_shells: non-convex lobes across the view axis
    so that rays cross a shell up to six times, 12x triangle-area spread, 256 randomly packed uv
    charts) through the whole step against the oracle pipeline (brute-force closest hit, the
    reference's per-hit evaluation) at 128 x 128 rays, K = 3.  Closest hit must pick the right one
    of several crossings, and the fragmented atlas scatters a pixel neighbourhood's texels."""
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import stress_shells
    from volsurfs_amd.pipeline import KShellPipeline
    from oracle import raytrace as ort
    res, K = 128, 3
    meshes = stress_shells(K=K, subdiv=4)
    o, d = pinhole_rays(res, res, focal=1.39 * res, cam_pos=(0.0, 0.0, -1.5))
    gt = torch.rand(o.shape[0], 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    pipe = KShellPipeline(meshes, o, d, gt, seed=3, init="spread")
    rgb = pipe.step().cpu().numpy()
    torch.cuda.synchronize()
    ref = _oracle(pipe)
    # the scene really is what it claims: a share of the rays crosses a shell more than twice
    v, f = meshes[K - 1].vertices.cpu().numpy(), meshes[K - 1].faces.cpu().numpy()
    sub = np.arange(0, o.shape[0], 37)
    crossings = ort.count_crossings(v, f, o.cpu().numpy()[sub], d.cpu().numpy()[sub])
    assert (crossings >= 4).mean() > 0.02, np.bincount(crossings)
    hit = pipe.to_ray_order(pipe._hit_slot, dim=1).cpu().numpy() >= 0
    assert np.array_equal(hit.T, ref["hit"])                      # closest hit among several crossings
    e = np.abs(rgb - ref["rgb"])
    print(f"MEASURED stress rgb_max={e.max():.3e} frac_over_1e-4={(e > 1e-4).mean():.3e} hits={int(hit.sum())}")
    assert (e > 1e-4).mean() <= 8.2e-4 and e.max() < 4e-3         # an 8-bit texel flip shows as fp16 ulps
    bank = pipe.bank
    worst = {"weights": 0.0, "tables": 0.0}
    for x, (g_t, g_w) in ref["grads"].items():
        worst["weights"] = max(worst["weights"], float((bank.weights.grad[x].cpu() - g_w).abs().max() / g_w.abs().max()))
        worst["tables"] = max(worst["tables"], float((bank.tables.grad[x].cpu() - g_t).abs().max() / g_t.abs().max()))
        assert torch.nn.functional.cosine_similarity(bank.weights.grad[x].cpu(), g_w, dim=0) > 0.9995
    print(f"MEASURED stress grad_rel_max weights={worst['weights']:.3e} tables={worst['tables']:.3e}")
    assert worst["weights"] <= STRESS_GRAD_REL_MAX["weights"] and worst["tables"] <= STRESS_GRAD_REL_MAX["tables"]


@pytest.mark.gpu
def test_pipeline_is_deterministic_in_forward_and_chunk_independent():
    from volsurfs_amd.pipeline import KShellPipeline
    pipe = KShellPipeline.synthetic(K=2, subdiv=3, res=64, init="spread", seed=1)
    a = pipe.step().clone()
    b = pipe.step().clone()
    assert torch.equal(a, b)
    # the same rays rendered as two half-frames give the same pixels
    N = pipe.nr_rays
    h1 = KShellPipeline(pipe.meshes, pipe.rays_o[:N // 2].contiguous(), pipe.rays_d[:N // 2].contiguous(),
                        pipe.gt[:N // 2].contiguous(), seed=1, init="spread")
    r1 = h1.step()
    assert torch.equal(r1, a[:N // 2])


@pytest.mark.gpu
def test_data_parallel_step_is_the_plain_step_plus_device_flags():
    """pipe.step(dp=signals): the same launches as the plain step (graph-capturable) with the hash-grid
    backward walking the shells phase by phase; gradients as the plain step's, and a graph replay of it
    advances the device epoch the host counts."""
    from volsurfs_amd.parallel import OverlappedStep
    from volsurfs_amd.pipeline import KShellPipeline
    pipe = KShellPipeline.synthetic(K=3, subdiv=3, res=64, init="spread", seed=2)
    ref_rgb = pipe.step().clone()
    gw, gt = pipe.bank.weights.grad.clone(), pipe.bank.tables.grad.clone()
    o = OverlappedStep(pipe, 1)                  # one rank, no group: the flags are published, nothing waits
    rgb = o.run()
    torch.cuda.synchronize()
    assert torch.equal(rgb, ref_rgb) and o.signals.epoch_host == 1
    np.testing.assert_allclose(pipe.bank.weights.grad.cpu().numpy(), gw.cpu().numpy(), rtol=0,
                               atol=2e-3 * gw.abs().max().item())
    np.testing.assert_allclose(pipe.bank.tables.grad.cpu().numpy(), gt.cpu().numpy(), rtol=0,
                               atol=2e-3 * gt.abs().max().item())
    pipe.capture_graph(dp=o.signals)
    for _ in range(3):
        rgb = o.run(pipe.replay)
    torch.cuda.synchronize()
    assert torch.equal(rgb, ref_rgb)
    flags, dev_epoch, counters = o.signals.read()
    assert flags == [6] * (o.signals.n + 1) and dev_epoch == 6 == o.signals.epoch_host and not any(counters)
    np.testing.assert_allclose(pipe.bank.tables.grad.cpu().numpy(), gt.cpu().numpy(), rtol=0,
                               atol=2e-3 * gt.abs().max().item())


@pytest.mark.gpu
def test_step_with_gradient_callbacks_matches_plain_step():
    """The multi-GPU schedule (weights.grad, then one tables.grad slice per shell handed
    to a callback as soon as it is final) computes the same gradients as the plain step
    and hands out every slice exactly once."""
    from volsurfs_amd.pipeline import KShellPipeline
    pipe = KShellPipeline.synthetic(K=3, subdiv=3, res=64, init="spread", seed=2)
    pipe.step()
    gw, gt = pipe.bank.weights.grad.clone(), pipe.bank.tables.grad.clone()
    seen = []
    pipe.step(grad_ready=lambda t: seen.append((t.data_ptr(), tuple(t.shape))))
    torch.cuda.synchronize()
    tg = pipe.bank.tables.grad
    assert seen[0] == (pipe.bank.weights.grad.data_ptr(), tuple(gw.shape))
    assert [p for p, _ in seen[1:]] == [tg[s * 8:(s + 1) * 8].data_ptr() for s in range(3)]
    assert all(sh == (8,) + tuple(tg.shape[1:]) for _, sh in seen[1:])
    assert gt.abs().max() > 0
    # two backward passes differ in the last bits (unordered float atomics in shade_bwd,
    # then f16 rounding): compare at the scale of each tensor
    np.testing.assert_allclose(pipe.bank.weights.grad.cpu().numpy(), gw.cpu().numpy(), rtol=0,
                               atol=2e-3 * gw.abs().max().item())
    np.testing.assert_allclose(tg.cpu().numpy(), gt.cpu().numpy(), rtol=0,
                               atol=2e-3 * gt.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("K_,res,n_rays,subdiv", [
    (5, 800, 640000, 6),             # BASELINE configs[1] (the bench workload)
    (7, (1080, 1920), 2073600, 8),   # configs[4]: K=7 HIGH-POLY shells (subdiv 8: 1 310 720 triangles each) at 1080p
    (5, (1200, 1600), 1920000, 6),   # configs[3]'s frame with a constant background (learned background: test_methods.py)
    (-5, 800, 640000, 6)])           # K = 5 STRESS shells (non-convex, 256 charts, 12x triangle spread) at the bench size
def test_full_size_frame_properties(K_, res, n_rays, subdiv):
    """BASELINE configurations at their full sizes (SURVEY §8d geometry, full-resolution textures):
    size-independent properties of the whole step.
      * forward is bit-deterministic; HIP-graph replay equals the eager step
      * every ray that misses all shells shows the background, hit rays do not exceed [0, 1]
      * sum_k w_k + bg_T = 1 per ray (composite partition of unity, fp16 tolerance)
      * the unique-texel structure is consistent: slots <= 4 corners x 4 bands x hits, segments
        sorted, every hit's 16 corner texels own a slot
      * backward is linear in the loss weight: two steps with the target moved give gradients
        whose difference matches the change of sign(pred - gt) (checked through a doubled
        grad_scale: same gradients after unscaling)"""
    from volsurfs_amd.composite import composite_dense
    from volsurfs_amd.pipeline import KShellPipeline
    stress, K_ = K_ < 0, abs(K_)
    pipe = KShellPipeline.synthetic(K=K_, res=res, subdiv=subdiv, stress=stress, init="spread" if stress else "tcnn")
    N, K = pipe.nr_rays, pipe.K
    assert N == n_rays and K == K_ and pipe.tracer.mesh_nr_tris[0] == 20 * 4 ** subdiv
    a = pipe.step().clone()
    gw1, gt1 = pipe.bank.weights.grad.clone(), pipe.bank.tables.grad.clone()
    b = pipe.step().clone()
    assert torch.equal(a, b)
    hits, slots = pipe.stats()
    assert hits > N * K // 8 and 16 * hits >= slots > hits     # 4 corners x 4 bands per hit
    seg = pipe.bank.seg_start.cpu()
    assert (seg[1:] >= seg[:-1]).all() and int(seg[-1]) == slots
    hit_slot = pipe.to_ray_order(pipe._hit_slot, dim=1)
    miss_all = (hit_slot < 0).all(0)
    assert miss_all.any() and torch.equal(a[miss_all], torch.ones_like(a[miss_all]))
    assert a.min() >= 0 and a.max() <= 1.0005
    out = composite_dense(pipe.surfs_rgb, pipe.surfs_alpha, pipe.bg)
    total = out["surfs_blending_weights"].sum(1) + out["bg_transmittance"]
    assert (total - 1).abs().max() < 4e-3
    # graph replay == eager
    pipe.capture_graph()
    c = pipe.replay().clone()
    torch.cuda.synchronize()
    assert torch.equal(c, a)
    # gradient scale invariance (grad_scale only conditions the fp16 intermediates)
    pipe.grad_scale *= 2.0
    pipe.step()
    gw2, gt2 = pipe.bank.weights.grad, pipe.bank.tables.grad
    for g1, g2 in ((gw1, gw2), (gt1, gt2)):
        s = g1.abs().max().item()
        assert s > 0 and (g1 - g2).abs().max().item() < 2e-2 * s
        cos = torch.nn.functional.cosine_similarity(g1.flatten(), g2.flatten(), dim=0)
        assert cos > 0.999


@pytest.mark.gpu
def test_degenerate_frames():
    """Edge cases of the whole step: a frame in which every ray misses every shell (zero
    slots: the persistent kernels have no work), and a one-ray frame."""
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.pipeline import KShellPipeline
    pipe = KShellPipeline.synthetic(K=2, subdiv=2, res=16, init="spread", seed=1)
    # look away from the shells: no hits at all
    pipe.rays_d = (-pipe.rays_d).contiguous()
    rgb = pipe.step()
    torch.cuda.synchronize()
    hits, slots = pipe.stats()
    assert hits == 0 and slots == 0
    assert torch.equal(rgb, torch.ones_like(rgb))                     # white background
    assert pipe.bank.tables.grad.abs().sum() == 0 and pipe.bank.weights.grad.abs().sum() == 0
    # the same pipeline renders normally afterwards
    pipe.rays_d = (-pipe.rays_d).contiguous()
    rgb = pipe.step()
    hits, slots = pipe.stats()
    assert hits > 0 and slots > 0 and pipe.bank.tables.grad.abs().sum() > 0
    # a single ray through the centre
    one = KShellPipeline(pipe.meshes, torch.tensor([[0.0, 0.0, -1.5]]).cuda(),
                         torch.tensor([[0.0, 0.0, 1.0]]).cuda(), torch.rand(1, 3).cuda(), seed=1,
                         init="spread")
    r = one.step()
    torch.cuda.synchronize()
    h, sl = one.stats()
    assert r.shape == (1, 3) and h == 2 and 8 <= sl <= 32 and torch.isfinite(r).all()
    assert one.bank.weights.grad.abs().sum() > 0


@pytest.mark.gpu
def test_split_graphs_equal_the_one_graph_step():
    """capture_graph_split: prefix (ray order, traversal, mark / compact) + rest replayed back to back =
    the one-graph step, bit for bit in the colours; OverlappedStep.run_split drives them."""
    from volsurfs_amd.parallel import OverlappedStep
    from volsurfs_amd.pipeline import KShellPipeline
    pipe = KShellPipeline.synthetic(K=3, subdiv=3, res=64, init="spread", seed=2)
    ref_rgb = pipe.step().clone()
    gt = pipe.bank.tables.grad.clone()
    o = OverlappedStep(pipe, 1)
    pipe.capture_graph_split(dp=o.signals)
    for _ in range(3):
        rgb = o.run_split(pipe.replay_prefix, pipe.replay_mid, pipe.replay_tail)
    o.finish()
    torch.cuda.synchronize()
    assert torch.equal(rgb, ref_rgb)
    assert o.signals.read()[0] == [5] * (o.signals.n + 1)
    np.testing.assert_allclose(pipe.bank.tables.grad.cpu().numpy(), gt.cpu().numpy(), rtol=0,
                               atol=2e-3 * gt.abs().max().item())
