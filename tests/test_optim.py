"""A13: the fused HIP Adam step vs torch.optim.Adam (the arithmetic apex FusedAdam implements,
base_method.py:87-94: betas (0.9, 0.99), eps 1e-15, no weight decay)."""
import numpy as np
import pytest
import torch


def test_adam_descriptor_layout_matches_header():
    import ctypes
    import re
    from volsurfs_amd import _lib
    from volsurfs_amd.optim import AdamTensor
    hdr = open(_lib.HEADER_PATH).read()
    body = re.search(r"typedef struct vsa_adam_tensor \{(.*?)\} vsa_adam_tensor;", hdr, re.S).group(1)
    names = re.findall(r"(\w+);", body)
    assert names == [f[0] for f in AdamTensor._fields_] and ctypes.sizeof(AdamTensor) == 48
    L = _lib.lib()
    assert L.vsa_adam_chunk_elems() == 4096
    null = ctypes.c_void_p(0)
    f = ctypes.c_float
    assert L.vsa_adam_step(null, null, 3, f(1e-3), f(0.9), f(0.99), f(1e-15), 1, f(1.0), 1, null) == -1
    assert L.vsa_adam_step(null, null, 0, f(1e-3), f(0.9), f(0.99), f(1e-15), 1, f(1.0), 1, null) == 0
    assert L.vsa_adam_step(null, null, 0, f(1e-3), f(0.9), f(0.99), f(1e-15), 0, f(1.0), 1, null) == -1   # step < 1


@pytest.mark.gpu
def test_fused_adam_matches_torch_adam():
    from volsurfs_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(40, 354184, 2)[1:], (3, 8192), (4097,), (1,), (5, 7, 3), (64, 66)]
    a = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    halves = {a[0]: torch.empty(shapes[0], dtype=torch.float16, device="cuda"),
              a[2]: torch.empty(shapes[2], dtype=torch.float16, device="cuda")}
    opt = FusedAdam(a, lr=1e-3, betas=(0.9, 0.99), eps=1e-15, half_copies=halves)
    ref = torch.optim.Adam(b, lr=1e-3, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.0)
    for it in range(25):
        lr = 1e-3 * (0.5 if it >= 15 else 1.0) * min(1.0, it / 5.0)       # schedulers write group["lr"]
        opt.param_groups[0]["lr"] = ref.param_groups[0]["lr"] = lr
        opt.zero_grad()
        ref.zero_grad()
        for p, q in zip(a, b):
            gr = torch.randn(p.shape, generator=g).cuda() * (10.0 ** float(torch.randint(-6, 2, (1,), generator=g)))
            if it % 7 == 3:
                gr[..., ::2] = 0                                              # sparse gradients
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0         # cleared by the step kernel
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            p.grad += gr                                                       # accumulate, as autograd does
            q.grad = gr.clone()
        opt.step()
        ref.step()
        for p, q in zip(a, b):
            np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=1e-6, atol=1e-6)
    for p, q in zip(a, b):
        sa, sb = opt.state[p], ref.state[q]
        for k in ("exp_avg", "exp_avg_sq"):      # 1e-6 of each moment tensor's scale
            x, y = sa[k].cpu().numpy(), sb[k].cpu().numpy()
            np.testing.assert_allclose(x, y, rtol=1e-5, atol=1e-6 * float(np.abs(y).max()))
    assert torch.equal(halves[a[0]], a[0].detach().half()) and torch.equal(halves[a[2]], a[2].detach().half())
    # the optimiser state survives a state_dict round trip (checkpoints, base_method.py:118-264)
    sd = opt.state_dict()
    opt2 = FusedAdam(a, lr=1e-3, betas=(0.9, 0.99), eps=1e-15, half_copies=halves)
    opt2.load_state_dict(sd)
    assert opt2.param_groups[0]["step"] == 25
    for p in a:
        assert torch.equal(opt2.state[p]["exp_avg"], opt.state[p]["exp_avg"])
    for p, q in zip(a, b):
        gr = torch.randn(p.shape, generator=g).cuda()
        p.grad.copy_(gr)
        q.grad = gr.clone()
    opt2.step()
    ref.step()
    for p, q in zip(a, b):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_volsurfs_trains_with_the_fused_optimiser_and_direct_gradient_accumulation():
    """VolSurfs.init_optim() -> FusedAdam over the bank's stacked tables / weights with their f16
    copies; the texture backward accumulates straight into the persistent .grad buffers."""
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    from volsurfs_amd.optim import FusedAdam
    from volsurfs_amd.trainer import train_step
    m = VolSurfs(nested_shells(K=2, subdiv=3), max_rays=4096, textures_res=(256, 128, 64, 32),
                 nr_warmup_iters=0, lr=2e-3)
    opt = m.init_optim()
    assert isinstance(opt, FusedAdam)
    o, d = pinhole_rays(48, 48, focal=80.0)
    gt = torch.rand(48 * 48, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) * 0.2
    m.grad_scale = float(48 * 48)
    gptr = None
    losses = []
    for it in range(10):
        l, _ = train_step(m, o, d, gt, None, iter_nr=it, is_first_iter=(it == 0))
        losses.append(l["loss"])
        if gptr is None:
            gptr = m.bank.tables.grad.data_ptr()
        assert m.bank.tables.grad.data_ptr() == gptr                       # persistent buffer
        assert float(m.bank.tables.grad.abs().sum()) == 0.0                # cleared by the step kernel
        assert torch.equal(m.bank.tables_h, m.bank.tables.detach().half())  # f16 copy refreshed in the kernel
        assert torch.equal(m.bank.weights_h, m.bank.weights.detach().half())
    assert losses[-1] < losses[0] - 1e-3, losses


def _rand_grads(ps, seed):
    g = torch.Generator().manual_seed(seed)
    for p in ps:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        p.grad.copy_(torch.randn(p.shape, generator=g).cuda())


@pytest.mark.gpu
@pytest.mark.parametrize("cls_name", ["FusedAdam", "ShardedFusedAdam"])
def test_optimiser_save_load_step_uses_the_loaded_moments(cls_name):
    """`load_state_dict` replaces every state tensor; the cached kernel
    descriptors must follow (they hold raw addresses).  An optimiser that loads a checkpoint and
    steps must equal the one that never stopped, bit for bit, and a state dict written by either
    class must load into the other (full-size moments in both)."""
    import copy
    from volsurfs_amd import optim
    shapes = [(16, 1000, 2), (16, 8192), (1001,)]
    g0 = torch.Generator().manual_seed(1)
    init = [torch.randn(s, generator=g0).cuda() for s in shapes]

    def make(name):
        ps = [torch.nn.Parameter(x.clone()) for x in init]
        hs = {ps[0]: ps[0].detach().half(), ps[1]: ps[1].detach().half()}
        kw = dict(lr=1e-2, betas=(0.9, 0.99), eps=1e-15, half_copies=hs)
        opt = optim.FusedAdam(ps, **kw) if name == "FusedAdam" else optim.ShardedFusedAdam(ps, 1, 0, **kw)
        return ps, hs, opt

    ps, hs, opt = make(cls_name)
    for it in range(3):
        _rand_grads(ps, 10 + it)
        opt.mark_grads_dirty()
        opt.step()
    sd = copy.deepcopy(opt.state_dict())
    saved = [p.detach().clone() for p in ps]
    _rand_grads(ps, 99)
    opt.mark_grads_dirty()
    opt.step()                                   # the uninterrupted run's 4th step
    torch.cuda.synchronize()
    for k in ("exp_avg", "exp_avg_sq"):          # full-size moments whatever the class
        assert all(sd["state"][i][k].shape == ps[i].shape for i in range(3))
    for other in ("FusedAdam", "ShardedFusedAdam"):
        qs, hq, opt2 = make(other)
        _rand_grads(qs, 5)
        opt2.mark_grads_dirty()
        opt2.step()                              # a step BEFORE the load: descriptors cached on the old state
        with torch.no_grad():
            for q, s in zip(qs, saved):
                q.copy_(s)
        opt2.load_state_dict(copy.deepcopy(sd))
        garbage = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(4)]   # reuse freed blocks
        _rand_grads(qs, 99)
        opt2.mark_grads_dirty()
        opt2.step()
        torch.cuda.synchronize()
        del garbage
        for p, q in zip(ps, qs):
            assert torch.equal(p.detach(), q.detach()), (cls_name, other)
        for p, q in zip(ps, qs):
            for k in ("exp_avg", "exp_avg_sq"):
                assert torch.equal(opt.state[p][k].view(-1), opt2.state[q][k].view(-1)), (cls_name, other, k)
        assert torch.equal(hq[qs[0]], qs[0].detach().half())
