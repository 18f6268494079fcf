"""vsa_nt_rebalance (csrc/nt_texels.hip, nt_common.h): the persistent kernels' work split corrected by the
previous frame's measured workgroup times.  Only who does which piece changes, never the pieces: texels and
colours must be bit-identical to the equal split, gradients equal up to their run-to-run atomics noise."""
import ctypes

import numpy as np
import pytest
import torch

BAL_KERNELS, BAL_MAX_WG, BAL_ONE = 6, 1024, 1 << 24


def _balance_views(bank):
    raw = bank.balance.cpu().numpy()
    n_frac = BAL_KERNELS * (BAL_MAX_WG + 1)
    n_ticks = BAL_KERNELS * BAL_MAX_WG
    frac = raw[:4 * n_frac].view(np.uint32).reshape(BAL_KERNELS, BAL_MAX_WG + 1)
    ticks = raw[4 * n_frac:4 * (n_frac + n_ticks)].view(np.uint32).reshape(BAL_KERNELS, BAL_MAX_WG)
    wgs = raw[4 * (n_frac + n_ticks):4 * (n_frac + n_ticks) + 8 * BAL_KERNELS].view(np.int32).reshape(2, BAL_KERNELS)
    return frac, ticks, wgs[0], wgs[1]


@pytest.mark.gpu
def test_rebalanced_split_is_bit_identical_and_shares_follow_the_times():
    from volsurfs_amd import neural_textures as NT
    from volsurfs_amd.pipeline import KShellPipeline
    assert NT.REBALANCE
    torch.manual_seed(0)
    p = KShellPipeline.synthetic(K=3, subdiv=4, res=256, init="spread")
    bank = p.bank
    assert bank.plan.balance
    outs = []
    for _ in range(4):                       # frames 2.. run on shares derived from the frame before
        rgb = p.step().clone()
        outs.append((rgb, bank.texels.clone(), bank.tables.grad.clone(), bank.weights.grad.clone()))
    torch.cuda.synchronize()
    frac, ticks, frac_wgs, tick_wgs = _balance_views(bank)
    for k in (4, 5):                         # the two MLP kernels are the ones whose shares are corrected
        G = int(tick_wgs[k])
        assert 1 < G <= BAL_MAX_WG and frac_wgs[k] == G
        assert frac[k][0] == 0 and frac[k][G] == BAL_ONE
        w = np.diff(frac[k][:G + 1].astype(np.int64))
        assert (w > 0).all() and w.min() >= 0.2 * BAL_ONE / G
        assert (ticks[k][:G] > 0).all()
    assert (np.diff(frac[5][:int(tick_wgs[5]) + 1].astype(np.int64)) != BAL_ONE // int(tick_wgs[5])).any()
    for k in (0, 1, 2, 3):                   # stamped, not corrected (measured slower: profiles/r03/rebalance.txt)
        assert tick_wgs[k] > 0 and frac_wgs[k] == 0
    # the same frame with the equal split
    bank.plan.balance = None
    rgb = p.step()
    ref = (rgb, bank.texels, bank.tables.grad, bank.weights.grad)
    for o in outs:
        assert torch.equal(o[0], ref[0])                      # colours
        assert torch.equal(o[1], ref[1])                      # quantised texels
        # gradients: the per-slot gradient rows are packed-f16 atomics of the shading backward and the
        # weight gradients float atomics - order-dependent from run to run in EITHER split
        for a, b in ((o[2], ref[2]), (o[3], ref[3])):
            assert (a - b).abs().max() <= 2e-3 * b.abs().max()


@pytest.mark.gpu
def test_rebalance_without_a_buffer_or_before_any_launch_is_a_noop():
    from volsurfs_amd import _lib
    from volsurfs_amd.neural_textures import NeuralTextureBank
    bank = NeuralTextureBank(1, 64, device="cuda", textures_res=(64, 32, 16, 8))
    st = _lib.stream_ptr()
    _lib.call("vsa_nt_rebalance", ctypes.byref(bank.plan), st)     # nothing stamped yet
    torch.cuda.synchronize()
    assert not bank.balance.any()
    keep = bank.plan.balance
    bank.plan.balance = None
    _lib.call("vsa_nt_rebalance", ctypes.byref(bank.plan), st)
    bank.plan.balance = keep


@pytest.mark.gpu
def test_any_measured_times_give_a_valid_split():
    """Adversarial times in the balance buffer (zeros, huge values, random) must still yield shares that
    partition every cost axis: the frame's texels and colours stay bit-identical to the equal split."""
    from volsurfs_amd.pipeline import KShellPipeline
    torch.manual_seed(1)
    p = KShellPipeline.synthetic(K=2, subdiv=4, res=192, init="spread")
    bank = p.bank
    keep = bank.plan.balance
    bank.plan.balance = None
    ref_rgb = p.step().clone()
    ref_texels = bank.texels.clone()
    bank.plan.balance = keep
    p.step()                                                  # stamps tick_wgs for every kernel
    n_frac = BAL_KERNELS * (BAL_MAX_WG + 1)
    g = torch.Generator(device="cuda").manual_seed(3)
    for trial in range(6):
        ticks = bank.balance[4 * n_frac:4 * (n_frac + BAL_KERNELS * BAL_MAX_WG)].view(torch.int32)
        if trial == 0:
            ticks.zero_()
        elif trial == 1:
            ticks.fill_(0x7fffffff)
        else:
            r = torch.randint(0, 1 << 30, ticks.shape, generator=g, device="cuda", dtype=torch.int32)
            ticks.copy_(torch.where(torch.rand(ticks.shape, generator=g, device="cuda") < 0.3, torch.zeros_like(r), r))
        rgb = p.step()          # rebalance runs on the planted times, then the kernels restamp them
        assert torch.equal(rgb, ref_rgb), trial
        assert torch.equal(bank.texels, ref_texels), trial
        frac, _, frac_wgs, tick_wgs = _balance_views(bank)
        for k in (4, 5):
            G = int(tick_wgs[k])
            w = np.diff(frac[k][:G + 1].astype(np.int64))
            assert frac_wgs[k] == G and frac[k][0] == 0 and frac[k][G] == BAL_ONE and (w > 0).all(), (trial, k)
