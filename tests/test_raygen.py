"""Ray generation + training-ray sampler.  mvdatasets is absent from the
reference checkout, so parity is UNPINNED here: the HIP kernels are checked bit for bit against
oracle/raygen.py (this library's own pinhole definition) and through geometric properties."""
import numpy as np
import pytest
import torch

from oracle import raygen as OR
from oracle.packed import Pcg32


def _cam_np(seed=0, H=24, W=32):
    g = np.random.default_rng(seed)
    q, _ = np.linalg.qr(g.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    t = g.standard_normal(3)
    K = np.array([[40.0 + seed, 0.3, W / 2 + 0.7], [0, 41.5, H / 2 - 0.2], [0, 0, 1]])
    return K, np.concatenate([q, t[:, None]], 1), H, W


def test_oracle_rays_reproject_to_their_pixels():
    K, pose, H, W = _cam_np(3)
    kinv = np.linalg.inv(K).astype(np.float32)
    o, d, p = OR.camera_rays(pose.astype(np.float32), kinv, H, W, 2, True, Pcg32())
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-6)
    assert np.allclose(o, pose[:, 3].astype(np.float32))
    cam = (d.astype(np.float64) @ pose[:, :3])            # world -> camera (R^T d)
    uv = (cam / cam[:, 2:]) @ K.T
    assert np.abs(uv[:, :2] - p).max() < 1e-3
    px = np.floor(p).astype(int).reshape(H, W, 2, 2)
    assert (px[..., 0] == np.arange(W)[None, :, None]).all() and (px[..., 1] == np.arange(H)[:, None, None]).all()
    frac = p - np.floor(p)
    assert 0.4 < frac.mean() < 0.6 and frac.std() > 0.2   # jitter spreads over the pixel


@pytest.mark.gpu
@pytest.mark.parametrize("R,jitter", [(1, False), (3, True)])
def test_hip_camera_rays_vs_oracle(R, jitter):
    from volsurfs_amd import camera as C
    K, pose, H, W = _cam_np(1)
    cam = C.Camera(K, pose, H, W)
    C._m_rng.__init__()
    o, d, p = C.get_camera_rays(cam, R, jitter)
    ro, rd, rp = OR.camera_rays(cam.c2w.cpu().numpy(), cam.intrinsics_inv.cpu().numpy(), H, W, R, jitter, Pcg32())
    assert np.array_equal(o.cpu().numpy(), ro)
    assert np.array_equal(d.cpu().numpy(), rd)
    assert np.array_equal(p.cpu().numpy(), rp)
    if jitter:                                            # the stream moved on: a second call differs
        _, d2, _ = C.get_camera_rays(cam, R, jitter)
        assert not torch.equal(d, d2)


@pytest.mark.gpu
def test_hip_camera_rays_match_the_bench_pinhole_at_full_size():
    """800x800 look-at camera == the torch pinhole the bench has used so far (to rounding)."""
    from volsurfs_amd import camera as C
    cam = C.Camera.look_at((0.0, 0.0, -1.5), focal=1000.0, height=800, width=800)
    o, d, p = C.get_camera_rays(cam)
    o2, d2 = C.pinhole_rays(800, 800, 1000.0)
    assert torch.equal(o, o2)
    assert (d - d2).abs().max().item() < 2e-7
    assert torch.equal(p.reshape(800, 800, 2)[5, 7], torch.tensor([7.5, 5.5], device="cuda"))


@pytest.mark.gpu
@pytest.mark.parametrize("R,jitter,with_mask", [(1, False, True), (2, True, False)])
def test_hip_reel_batch_vs_oracle(R, jitter, with_mask):
    from volsurfs_amd import camera as C
    g = np.random.default_rng(5)
    cams = []
    for s in range(5):
        K, pose, H, W = _cam_np(10 + s)
        cams.append(C.Camera(K, pose, H, W))
    rgbs = g.uniform(0, 1, (5, H, W, 3)).astype(np.float32)
    masks = (g.uniform(0, 1, (5, H, W)) > 0.5).astype(np.float32) if with_mask else None
    reel = C.TensorReel(cams, rgbs, masks)
    B = 700
    cam, o, d, vals, p = reel.get_next_rays_batch(B, jitter, R)
    rc, ro, rd, rgt, rgm, rp = OR.reel_batch(reel.c2w.cpu().numpy(), reel.intrinsics_inv.cpu().numpy(),
                                            rgbs, masks, B, R, jitter, Pcg32())
    assert np.array_equal(cam.cpu().numpy(), rc)
    assert np.array_equal(o.cpu().numpy(), ro) and np.array_equal(d.cpu().numpy(), rd)
    assert np.array_equal(p.cpu().numpy(), rp)
    assert np.array_equal(vals["rgb"].cpu().numpy(), rgt)
    assert ("mask" in vals) == with_mask
    if with_mask:
        assert np.array_equal(vals["mask"].cpu().numpy(), rgm)
    # every camera and a good share of the pixels are visited; the next batch is a new draw
    assert set(rc.tolist()) == set(range(5))
    cam2, *_ = reel.get_next_rays_batch(B, jitter, R)
    assert not torch.equal(cam, cam2)


@pytest.mark.gpu
def test_reel_feeds_a_training_step():
    """Reel -> K-shell render -> fused L1 backward: the loop of trainer.py:176-235 without a
    host round trip for rays or ground truth."""
    from volsurfs_amd import camera as C
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    torch.manual_seed(0)
    H = W = 64
    cams = [C.Camera.look_at((1.5 * np.sin(a), 0.3, -1.5 * np.cos(a)), focal=80.0, height=H, width=W)
            for a in (0.0, 0.7, 1.9)]
    reel = C.TensorReel(cams, torch.rand(3, H, W, 3), torch.ones(3, H, W))
    m = VolSurfs(nested_shells(K=3, subdiv=3), bg_color=(1.0, 1.0, 1.0))
    cam, o, d, vals, _ = reel.get_next_rays_batch(2048, True, 1)
    losses, _, _ = m.forward(o, d, vals["rgb"], gt_mask=vals["mask"], is_training_masked=True)
    assert torch.isfinite(losses["loss"]).item() and losses["loss"].item() > 0
    losses["loss"].backward()


@pytest.mark.gpu
def test_render_camera_and_reel_training_loop():
    """Camera -> image (BaseMethod.render from the camera down) and a few optimiser steps drawn
    from a reel whose ground truth is the model's own first render: the loss falls."""
    from volsurfs_amd import camera as C
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    from volsurfs_amd.trainer import train_step_from_reel
    torch.manual_seed(0)
    H = W = 48
    cams = [C.Camera.look_at((1.5 * np.sin(a), 0.2, -1.5 * np.cos(a)), focal=60.0, height=H, width=W)
            for a in (0.0, 1.0)]
    m = VolSurfs(nested_shells(K=3, subdiv=3), bg_color=(1.0, 1.0, 1.0), lr=1e-2, nr_warmup_iters=0)
    img = m.render_camera(cams[0])
    assert img["rgb"].shape == (H, W, 3) and torch.isfinite(img["rgb"]).all()
    ss = m.render_camera(cams[0], nr_rays_per_pixel=2, jitter_pixels=True)
    assert ss["rgb"].shape == (H, W, 3)
    centre, corner = img["rgb"][H // 2, W // 2], img["rgb"][0, 0]
    assert torch.allclose(corner.float(), torch.ones(3, device="cuda"))      # misses: background
    assert not torch.allclose(centre.float(), torch.ones(3, device="cuda"))  # hits the shells
    target = torch.stack([torch.full((H, W, 3), 0.25, device="cuda")] * 2)
    reel = C.TensorReel(cams, target)
    m.init_optim()
    losses = [train_step_from_reel(m, reel, 1024, iter_nr=i)[0]["loss"] for i in range(12)]
    assert losses[-1] < losses[0]


@pytest.mark.gpu
def test_hip_tile_order_round_trip_and_pipeline_equivalence():
    """vsa_tile_order is a permutation (8x8 tiles, tile-major, boustrophedon inside a tile) with an exact inverse, and the
    step in tile order returns what the step in the caller's order returns."""
    import ctypes
    from volsurfs_amd import _lib
    from volsurfs_amd.pipeline import KShellPipeline
    H, W = 24, 40
    x = torch.arange(H * W * 3, dtype=torch.float32, device="cuda").reshape(H * W, 3)
    t, back = torch.empty_like(x), torch.empty_like(x)
    _lib.call("vsa_tile_order", x, t, H, W, 3, 0, _lib.stream_ptr())
    _lib.call("vsa_tile_order", t, back, H, W, 3, 1, _lib.stream_ptr())
    assert torch.equal(back, x)
    ref = x.reshape(H // 8, 8, W // 8, 8, 3).permute(0, 2, 1, 3, 4).clone()
    ref[:, :, 1::2] = ref[:, :, 1::2].flip(3)                   # odd pixel rows of a tile run right to left
    ref = ref.reshape(-1, 3)
    assert torch.equal(t, ref)
    with pytest.raises(_lib.VolsurfsHipError):
        _lib.call("vsa_tile_order", x, t, 20, 48, 3, 0, _lib.stream_ptr())      # 20 is not a multiple of 8
    a = KShellPipeline.synthetic(K=2, subdiv=3, res=64, init="spread", seed=4)
    assert a.image_hw == (64, 64)
    b = KShellPipeline(a.meshes, a.rays_o, a.rays_d, a.gt, seed=4, init="spread")   # no hint: caller's order
    ra, rb = a.step().clone(), b.step().clone()
    assert torch.equal(ra, rb)
    assert torch.equal(a.to_ray_order(a.surfs_alpha), b.surfs_alpha)
    ga, gb = a.bank.tables.grad, b.bank.tables.grad
    assert (ga - gb).abs().max() <= 2e-3 * gb.abs().max()       # same sums, other atomic order


@pytest.mark.gpu
def test_train_loop_with_callbacks():
    """trainer.train: the reference loop's hook order and iteration bookkeeping."""
    from volsurfs_amd import camera as C
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    from volsurfs_amd.trainer import train
    torch.manual_seed(0)
    H = W = 32
    cams = [C.Camera.look_at((0.0, 0.2, -1.5), focal=45.0, height=H, width=W)]
    m = VolSurfs(nested_shells(K=2, subdiv=2), max_rays=4096, textures_res=(64, 32, 16, 8), lr=1e-2,
                 nr_warmup_iters=0)
    reel = C.TensorReel(cams, torch.full((1, H, W, 3), 0.3, device="cuda"))
    log = []

    class Recorder:
        def training_started(self, **kw): log.append("start")
        def iter_started(self, phase, **kw): log.append(("it", phase.iter_nr, phase.is_first_iter))
        def iter_ended(self, phase, losses, **kw): log.append(("end", phase.iter_nr, losses["loss"]))
        def training_ended(self, **kw): log.append("stop")

    done = train(reel, m, start_iter_nr=5, iter_finish_nr=13, callbacks=[Recorder()], nr_training_rays=512)
    assert done == 13 and log[0] == "start" and log[-1] == "stop"
    its = [e for e in log if isinstance(e, tuple) and e[0] == "it"]
    assert [e[1] for e in its] == list(range(5, 13)) and its[0][2] and not its[1][2]
    ends = [e for e in log if isinstance(e, tuple) and e[0] == "end"]
    assert len(ends) == 8 and ends[-1][2] < ends[0][2]


def test_oracle_primitive_intersection_known_answers():
    """A1 (utils/raycasting.py:4-36): hand-checkable cases of the restated slab / sphere tests."""
    from oracle.raygen import intersect_primitive
    o = np.array([[0, 0, -2], [0, 0, 0], [0, 0, -2], [0, 2, -2], [0, 0, 2]], np.float32)
    d = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 0, 1]], np.float32)
    hit, tn, tf, pn, pf = intersect_primitive(o, d, 0, 0.5)
    assert hit.tolist() == [True, True, False, False, False]       # through, from inside, parallel, beside, behind
    assert tn[:2].tolist() == [1.5, 0.0] and tf[:2].tolist() == [2.5, 0.5]
    assert pf[0].tolist() == [0.0, 0.0, 0.5] and tn[2] == tf[2] == 0.0 and pn[2].tolist() == [0.0, 0.0, -2.0]
    hit, tn, tf, pn, pf = intersect_primitive(o, d, 1, 0.5)
    assert hit.tolist() == [True, True, False, False, False]
    assert tn[:2].tolist() == [1.5, 0.0] and tf[:2].tolist() == [2.5, 0.5]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [0, 1])
def test_hip_primitive_intersection_bit_exact_vs_oracle(kind):
    """vsa_intersect_primitive through BoundingBox / BoundingSphere and
    intersect_bounding_primitive (same dict keys as utils/raycasting.py:4-36)."""
    from oracle.raygen import intersect_primitive
    from volsurfs_amd.background import BoundingBox, BoundingSphere, intersect_bounding_primitive
    g = np.random.default_rng(kind)
    n = 20000
    o = (g.standard_normal((n, 3)) * 0.6).astype(np.float32)
    d = g.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:50, 0] = 0.0                                   # axis-parallel rays: infinite reciprocals
    d[50:100, 1:] = 0.0
    o[100:150] = 0.0
    prim = BoundingBox(0.9) if kind == 0 else BoundingSphere(0.45)
    rc = intersect_bounding_primitive(prim, torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda())
    assert set(rc) == {"rays_o", "rays_d", "nr_rays", "points_near", "points_far", "t_near", "t_far", "is_hit"}
    assert rc["t_near"].shape == (n, 1) and rc["t_far"].shape == (n, 1) and rc["nr_rays"] == n
    hit, tn, tf, pn, pf = intersect_primitive(o, d, kind, 0.45)
    assert 0.1 < hit.mean() < 0.95
    assert np.array_equal(rc["is_hit"].cpu().numpy(), hit)
    for got, ref in ((rc["t_near"][:, 0], tn), (rc["t_far"][:, 0], tf), (rc["points_near"], pn), (rc["points_far"], pf)):
        assert np.array_equal(got.cpu().numpy(), ref, equal_nan=True)
    empty = intersect_bounding_primitive(prim, torch.zeros(0, 3).cuda(), torch.zeros(0, 3).cuda())
    assert empty["t_far"].shape == (0, 1)
