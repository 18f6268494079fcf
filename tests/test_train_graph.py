"""The training iteration as one replayed HIP graph (volsurfs_amd.trainer.GraphTrainLoop; include/volsurfs_hip.h:
vsa_train_ctl): every `_ctl` entry point against the host-driven launch it stands for, the device tick against the
host's rules (trainer.dynamic_nr_rays, schedulers.lr_at, the sampler's stream), and the graph loop against the eager loop
of /root/reference/volsurfs_py/trainer.py:118-308 as volsurfs_amd.trainer.train_step_from_reel runs it."""
import ctypes

import numpy as np
import pytest
import torch


def _setup(seed=3, K=2, res=96, views=4, max_rays=8192, warm=6, milestones=(9, 14)):
    import bench
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    torch.manual_seed(seed)
    m = VolSurfs(nested_shells(K=K, subdiv=3), max_rays=max_rays, textures_res=(256, 128, 64, 32), seed=seed,
                 nr_warmup_iters=warm, lr_milestones=list(milestones))
    m.init_optim()
    reel = bench.synthetic_reel(views, res, "cuda", seed=seed)
    return m, reel


def _ctl_of(loop):
    from volsurfs_amd.trainer import TrainCtl
    return TrainCtl.from_buffer_copy(loop.ctl.cpu().numpy().tobytes())


def _write_ctl(loop, c):
    loop.ctl.copy_(torch.frombuffer(bytearray(bytes(c)), dtype=torch.uint8).to(loop.ctl.device))


@pytest.mark.gpu
def test_ctl_entry_points_equal_the_host_driven_launches():
    from volsurfs_amd import _lib
    from volsurfs_amd.composite import composite_fwd_bwd_l1_raw, l1_mean
    from volsurfs_amd.trainer import GraphTrainLoop
    m, reel = _setup()
    n, cap = 700, 1024
    loop = GraphTrainLoop(m, reel, n, 4096, iter_nr=0, capacity=cap)
    dev = loop.ctl.device
    # ---- the sampler: the first n rays are the eager sampler's, the rest the dummy ray
    cam = torch.empty(cap, dtype=torch.int32, device=dev)
    o, d, gt = (torch.empty(cap, 3, device=dev) for _ in range(3))
    _lib.call("vsa_reel_next_rays_batch_ctl", reel.c2w, reel.intrinsics_inv, reel.rgbs, None, reel.nr_cameras,
              reel.height, reel.width, cap, 1, True, loop.ctl, *loop._dummy, cam, o, d, gt, None, None, _lib.stream_ptr())
    cam_r, o_r, d_r, vals, _ = reel.get_next_rays_batch(n, True, 1)        # (same stream state: the loop copied it)
    assert torch.equal(o[:n], o_r) and torch.equal(d[:n], d_r) and torch.equal(gt[:n], vals["rgb"]) and torch.equal(cam[:n], cam_r)
    dm = [torch.tensor(list(x), device=dev) for x in loop._dummy]
    assert torch.equal(o[n:], dm[0].expand(cap - n, 3)) and torch.equal(d[n:], dm[1].expand(cap - n, 3))
    assert torch.equal(gt[n:], dm[2].expand(cap - n, 3))
    hit = m.raytracer.trace_all(o, d)[1]
    assert (hit[:, n:] < 0).all() and (hit[:, :n] >= 0).any()              # dummy rays miss every shell
    # ---- composite + L1: gradients of the first n rays bit-equal to the n-ray launch, none beyond
    K = m.nr_meshes
    g = torch.Generator(device=dev).manual_seed(1)
    rgb_k, alpha_k = torch.rand(cap, K, 3, device=dev, generator=g), torch.rand(cap, K, device=dev, generator=g)
    rgb = torch.empty(cap, 3, device=dev)
    g_c, g_a = torch.empty_like(rgb_k), torch.empty_like(alpha_k)
    _lib.call("vsa_composite_dense_fwd_bwd_l1_ctl", rgb_k, alpha_k, m.bg_color, True, gt, loop.ctl, rgb, g_c, g_a, cap, K, 0,
              _lib.stream_ptr())
    rgb_r, gc_r, ga_r = composite_fwd_bwd_l1_raw(rgb_k[:n].contiguous(), alpha_k[:n].contiguous(), m.bg_color,
                                                 gt[:n].contiguous(), 1.0 / (3.0 * n))
    assert torch.equal(rgb[:n], rgb_r) and torch.equal(g_c[:n], gc_r) and torch.equal(g_a[:n], ga_r)
    assert (g_c[n:] == 0).all() and (g_a[n:] == 0).all() and torch.isfinite(rgb[n:]).all()
    # ---- the logged loss
    _lib.call("vsa_l1_mean_ctl", rgb, gt, cap, loop._scr_loss, loop.ctl, _lib.stream_ptr())
    assert _ctl_of(loop).loss == float(l1_mean(rgb[:n].contiguous(), gt[:n].contiguous()))
    # ---- Adam from the control block against the host-driven step
    opt = m.optimizer
    grp = opt.param_groups[0]
    for p in grp["params"]:
        p.grad = torch.randn(p.shape, device=dev, generator=g) * 1e-3
    opt.mark_grads_dirty()
    keep = [p.detach().clone() for p in grp["params"]]
    grads = [p.grad.clone() for p in grp["params"]]
    _, desc, ck, nck, _ = opt._plan(0, grp)
    c = _ctl_of(loop)
    c.adam_step, c.adam_pending, c.adam_lr = 7, 1, 3.5e-4
    _write_ctl(loop, c)
    _lib.call("vsa_adam_step_ctl", desc, ck, nck, 0.9, 0.99, 1e-15, 1.0, 1, 0, loop.ctl, _lib.stream_ptr())
    got = [p.detach().clone() for p in grp["params"]]
    assert all(float(p.grad.abs().max()) == 0 for p in grp["params"])          # the fused zero_grad
    with torch.no_grad():
        for p, k_, g_ in zip(grp["params"], keep, grads):
            p.copy_(k_)
            p.grad.copy_(g_)
            st = opt.state[p]
            st["exp_avg"].zero_()
            st["exp_avg_sq"].zero_()
    grp["step"], grp["lr"] = 6, 3.5e-4
    opt.mark_grads_dirty()
    # (the moments: both runs start from the state the first run left?  No — the first run wrote them: reset above)
    opt.step()
    for a, p in zip(got, grp["params"]):
        torch.testing.assert_close(a, p.detach(), rtol=2e-6, atol=1e-9)
    # a launch with nothing pending leaves everything alone
    c.adam_pending = 0
    _write_ctl(loop, c)
    before = [p.detach().clone() for p in grp["params"]]
    _lib.call("vsa_adam_step_ctl", desc, ck, nck, 0.9, 0.99, 1e-15, 1.0, 1, 0, loop.ctl, _lib.stream_ptr())
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, grp["params"]))


@pytest.mark.gpu
def test_device_tick_applies_the_hosts_rules():
    """Dynamic ray count (trainer.py:288-304), warm-up + MultiStepLR (schedulers/warmup.py, base_method.py:71-76), Adam's
    step count and the sampler's stream, stepped on the device against the host's own functions."""
    from volsurfs_amd import _lib
    from volsurfs_amd.schedulers import lr_at
    from volsurfs_amd.trainer import GraphTrainLoop, dynamic_nr_rays
    from volsurfs_amd.volsurfs import _Pcg32State
    m, reel = _setup(warm=5, milestones=(3, 6))
    cap, target = 4096, 3000
    loop = GraphTrainLoop(m, reel, 512, target, iter_nr=0, capacity=cap)
    rng = _Pcg32State()
    rng.state, rng.inc = reel.rng.state, reel.rng.inc
    n, it, step = 512, 0, 0
    hits_seq = [700, 1400, 2900, 3100, 2950, 3300, 0, 10, 2999, 3001, 3000, 2800, 3100, 1, 3000, 3000]
    base = float(m.optimizer.param_groups[0].get("initial_lr", m.lr))
    for hits in hits_seq:
        c = _ctl_of(loop)
        c.nr_hits = hits
        _write_ctl(loop, c)
        _lib.call("vsa_train_ctl_tick", loop.ctl, _lib.stream_ptr())
        c = _ctl_of(loop)
        step += 1
        want_lr = ctypes.c_float(lr_at(it, base, 5, [3, 6], 0.3)).value
        n_next = dynamic_nr_rays(n, hits, target) if hits else n
        clamp = n_next > cap
        n_next = max(1, min(cap, n_next))
        it += 1
        rng.advance()
        assert (c.iter, c.adam_step, c.adam_pending) == (it, step, 1)
        assert c.adam_lr == want_lr, (it, c.adam_lr, want_lr)
        assert c.nr_rays == n_next, (it, c.nr_rays, n_next)
        assert c.loss_scale == ctypes.c_float(1.0 / (3.0 * n_next)).value
        assert c.rng_state == rng.state and c.rng_inc == rng.inc
        n = n_next
    assert _ctl_of(loop).clamped >= 1            # (hits = 1 / 10 asked for more rays than the capacity)


@pytest.mark.gpu
def test_graph_loop_follows_the_eager_loop_and_hands_back_its_state():
    from volsurfs_amd.trainer import GraphTrainLoop, train_step_from_reel
    W, T, target = 12, 14, 1500
    seqs = {}
    for mode in ("eager", "graph"):
        m, reel = _setup(seed=5, warm=6, milestones=(9, 14))
        n, rec = 256, []
        for it in range(W):
            m.grad_scale = 16.0 * n
            losses, n2 = train_step_from_reel(m, reel, n, jitter_pixels=True, iter_nr=it, is_first_iter=it == 0,
                                              target_nr_of_training_samples=target, sync_losses=False)
            rec.append((n, int(m.last_nr_samples)))
            n = n2
        if mode == "eager":
            for it in range(W, W + T):
                m.grad_scale = 16.0 * n
                losses, n2 = train_step_from_reel(m, reel, n, jitter_pixels=True, iter_nr=it, is_first_iter=False,
                                                  target_nr_of_training_samples=target, sync_losses=False)
                rec.append((n, int(m.last_nr_samples), float(losses["loss"])))
                n = n2
        else:
            m.grad_scale = None
            loop = GraphTrainLoop(m, reel, n, target, iter_nr=W).capture(warm_iterations=2)
            st = loop.read()
            assert st["iter"] == W + 2 and st["adam_step"] == W + 2
            tail = []
            for it in range(W + 2, W + T):
                nr = loop.read()["nr_rays"]
                loop.step()
                st = loop.read()
                tail.append((nr, st["nr_hits"], st["loss"]))
                assert st["iter"] == it + 1 and st["clamped"] == 0
            rec += [None, None] + tail
            st = loop.finish()
            g = m.optimizer.param_groups[0]
            assert g["step"] == W + T and m.lr_scheduler.it == W + T
            from volsurfs_amd.schedulers import lr_at
            assert g["lr"] == lr_at(W + T, g["initial_lr"], 6, [9, 14], 0.3)
            # ... and the eager loop goes on from there
            m.grad_scale = 16.0 * st["nr_rays"]
            losses, _ = train_step_from_reel(m, reel, st["nr_rays"], jitter_pixels=True, iter_nr=W + T,
                                             target_nr_of_training_samples=target, sync_losses=False)
            assert np.isfinite(float(losses["loss"]))
        seqs[mode] = rec
    # the same batches: ray counts and hit counts depend on the rays only (exact); the losses on parameters that two runs
    # of ONE loop already change in the last bits of (atomics' order)
    for a, b in zip(seqs["eager"][W + 2:], seqs["graph"][W + 2:]):
        assert a[0] == b[0] and a[1] == b[1], (a, b)
        assert abs(a[2] - b[2]) < 5e-3, (a, b)
