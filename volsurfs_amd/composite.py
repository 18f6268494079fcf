"""Dense K-shell composite op (host side of SURVEY §8a row A7).

Mirrors the tail of VolSurfs.render_rays
(/root/reference/volsurfs_py/methods/volsurfs.py:601-640, 704-708, 728-748):
same inputs (per-shell rgb / alpha, inner->outer; background colour), same
outputs (keys of the `ray_traced` dict).  One HIP kernel forward, one backward.
"""
import torch

from . import _lib


class _CompositeDense(torch.autograd.Function):
    @staticmethod
    def forward(ctx, surfs_rgb, surfs_alpha, rgb_bg, carry_f16):
        N, K, _ = surfs_rgb.shape
        surfs_rgb = _lib.check_f32(surfs_rgb.contiguous(), N, K, 3)
        surfs_alpha = _lib.check_f32(surfs_alpha.contiguous().view(N, K), N, K)
        rgb_bg = rgb_bg.contiguous()
        bcast = rgb_bg.shape[0] == 1 and N != 1
        _lib.check_f32(rgb_bg, 1 if bcast else N, 3)
        dev = surfs_rgb.device
        out_rgb = torch.empty(N, 3, device=dev)
        out_fg = torch.empty(N, 3, device=dev)
        out_bgT = torch.empty(N, 1, device=dev)
        out_w = torch.empty(N, K, 1, device=dev)
        out_rgb_h = torch.empty(N, K, 3, device=dev)
        out_alpha_h = torch.empty(N, K, 1, device=dev)
        _lib.call("vsa_composite_dense_fwd", surfs_rgb, surfs_alpha, rgb_bg, bcast, out_rgb,
                  out_fg, out_bgT, out_w, out_rgb_h, out_alpha_h, N, K, int(carry_f16),
                  _lib.stream_ptr())
        ctx.save_for_backward(surfs_rgb, surfs_alpha, rgb_bg)
        ctx.bcast = bcast
        ctx.carry_f16 = int(carry_f16)
        ctx.mark_non_differentiable(out_fg, out_bgT, out_w, out_rgb_h, out_alpha_h)
        return out_rgb, out_fg, out_bgT, out_w, out_rgb_h, out_alpha_h

    @staticmethod
    def backward(ctx, g_rgb, *unused):
        surfs_rgb, surfs_alpha, rgb_bg = ctx.saved_tensors
        N, K, _ = surfs_rgb.shape
        g_rgb = _lib.check_f32(g_rgb.contiguous(), N, 3)
        g_c = torch.empty_like(surfs_rgb)
        g_a = torch.empty_like(surfs_alpha)
        need_bg = ctx.needs_input_grad[2]
        g_bg = torch.empty(N, 3, device=g_rgb.device) if need_bg else None
        _lib.call("vsa_composite_dense_bwd", surfs_rgb, surfs_alpha, rgb_bg, ctx.bcast, g_rgb,
                  g_c, g_a, g_bg, N, K, ctx.carry_f16, _lib.stream_ptr())
        if need_bg and ctx.bcast:
            g_bg = g_bg.sum(0, keepdim=True)
        return g_c, g_a, g_bg, None


def composite_dense(surfs_rgb, surfs_alpha, rgb_bg, carry_f16=False):
    """surfs_rgb [N,K,3], surfs_alpha [N,K] or [N,K,1] (inner->outer, zero on
    miss), rgb_bg [N,3] or [1,3].  Returns the dict of volsurfs.py:738-748
    (minus the pass-through buffers).  Only `rgb` carries gradient, as in the
    reference's training loss (volsurfs.py:791, 806)."""
    N, K = surfs_rgb.shape[:2]
    alpha2d = surfs_alpha.reshape(N, K)
    rgb, fg, bgT, w, rgb_h, alpha_h = _CompositeDense.apply(surfs_rgb, alpha2d, rgb_bg, carry_f16)
    return {
        "rgb": rgb,
        "rgb_fg": fg,
        "rgb_bg": rgb_bg.expand(N, 3).half().float(),
        "surfs_alpha": alpha_h,
        "surfs_rgb": rgb_h,
        "surfs_blending_weights": w,
        "bg_transmittance": bgT,
    }


class _CompositeL1(torch.autograd.Function):
    """Composite + mean-L1 loss of a training step as ONE autograd node and one launch
    (vsa_composite_dense_fwd_bwd_l1): the forward pass already leaves d loss / d surfs_rgb, d loss / d surfs_alpha of
    the unit-weight loss; backward scales them by the incoming scalar.  Replaces composite_dense + (gt - pred).abs()
    .mean() — 6 launches forward, 16 backward in the legacy training loop (volsurfs.py:601-640, 704-708, 791-806;
    utils/losses.py:14-19).  Returns (loss [], rgb [N,3] without gradient)."""

    @staticmethod
    def forward(ctx, surfs_rgb, surfs_alpha, rgb_bg, gt_rgb):
        N, K = surfs_rgb.shape[:2]
        rgb, g_c, g_a = composite_fwd_bwd_l1_raw(surfs_rgb.contiguous(), surfs_alpha.reshape(N, K).contiguous(),
                                                 rgb_bg.contiguous(), gt_rgb.contiguous(), 1.0 / (3.0 * N))
        ctx.save_for_backward(g_c, g_a)
        ctx.alpha_shape = surfs_alpha.shape
        loss = l1_mean(rgb, gt_rgb)
        ctx.mark_non_differentiable(rgb)
        return loss, rgb

    @staticmethod
    def backward(ctx, g_loss, _g_rgb):
        g_c, g_a = ctx.saved_tensors
        return g_c * g_loss, (g_a * g_loss).reshape(ctx.alpha_shape), None, None


def composite_l1(surfs_rgb, surfs_alpha, rgb_bg, gt_rgb):
    """(mean |gt - composite|, composite rgb): see _CompositeL1."""
    return _CompositeL1.apply(surfs_rgb, surfs_alpha, rgb_bg, gt_rgb)


def composite_fwd_raw(surfs_rgb, surfs_alpha, rgb_bg, carry_f16=False):
    """Forward only, rgb [N,3] only (the fused pipeline's call)."""
    N, K, _ = surfs_rgb.shape
    out = torch.empty(N, 3, device=surfs_rgb.device)
    bcast = rgb_bg.shape[0] == 1 and N != 1
    _lib.call("vsa_composite_dense_fwd", surfs_rgb, surfs_alpha, rgb_bg, bcast, out, None, None,
              None, None, None, N, K, int(carry_f16), _lib.stream_ptr())
    return out


def composite_bwd_raw(surfs_rgb, surfs_alpha, rgb_bg, g_rgb, carry_f16=False):
    N, K, _ = surfs_rgb.shape
    g_c = torch.empty_like(surfs_rgb)
    g_a = torch.empty_like(surfs_alpha)
    bcast = rgb_bg.shape[0] == 1 and N != 1
    _lib.call("vsa_composite_dense_bwd", surfs_rgb, surfs_alpha, rgb_bg, bcast, g_rgb, g_c, g_a,
              None, N, K, int(carry_f16), _lib.stream_ptr())
    return g_c, g_a


def composite_bwd_l1_raw(surfs_rgb, surfs_alpha, rgb_bg, pred_rgb, gt_rgb, loss_scale, carry_f16=False):
    """composite_bwd_raw with d mean|gt - pred| / d pred formed inside the kernel
    (loss_scale = 1 / (3 * N_global))."""
    N, K, _ = surfs_rgb.shape
    g_c = torch.empty_like(surfs_rgb)
    g_a = torch.empty_like(surfs_alpha)
    bcast = rgb_bg.shape[0] == 1 and N != 1
    _lib.call("vsa_composite_dense_bwd_l1", surfs_rgb, surfs_alpha, rgb_bg, bcast, pred_rgb, gt_rgb,
              float(loss_scale), g_c, g_a, N, K, int(carry_f16), _lib.stream_ptr())
    return g_c, g_a


def composite_fwd_bwd_l1_raw(surfs_rgb, surfs_alpha, rgb_bg, gt_rgb, loss_scale, carry_f16=False):
    """One launch: the composited colour [N,3] and the gradients of the mean-L1 loss w.r.t. the
    per-shell colours / alphas (composite_fwd_raw + composite_bwd_l1_raw)."""
    N, K, _ = surfs_rgb.shape
    rgb = torch.empty(N, 3, device=surfs_rgb.device)
    g_c = torch.empty_like(surfs_rgb)
    g_a = torch.empty_like(surfs_alpha)
    bcast = rgb_bg.shape[0] == 1 and N != 1
    _lib.call("vsa_composite_dense_fwd_bwd_l1", surfs_rgb, surfs_alpha, rgb_bg, bcast, gt_rgb,
              float(loss_scale), rgb, g_c, g_a, N, K, int(carry_f16), _lib.stream_ptr())
    return rgb, g_c, g_a


_reduce_scratch = {}


def reduce_scratch(device):
    """The scratch of vsa_count_hits / vsa_l1_mean (zeroed once; every call leaves it ready): ticket + partials, a
    last-ticket protocol that is only valid for calls SERIALISED on one stream — so there is one per (device, stream):
    two reductions on different streams (look-ahead traversal, the alpha chain, the optimiser's side stream) never
    share a ticket."""
    import ctypes
    dev = torch.device(device)
    dev = torch.device("cuda", torch.cuda.current_device()) if dev.index is None else dev
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    sc = _reduce_scratch.get(key)
    if sc is None:
        fn = _lib.lib().vsa_reduce_scratch_bytes
        fn.restype = ctypes.c_longlong
        sc = _reduce_scratch[key] = torch.zeros(int(fn()), dtype=torch.uint8, device=dev)
    return sc


def l1_mean(pred, gt):
    """mean |pred - gt| as a [] device tensor, one launch (vsa_l1_mean; utils/losses.py:14-19 without a mask)."""
    import ctypes
    pred, gt = _lib.check_f32(pred.contiguous()), _lib.check_f32(gt.contiguous(), *pred.shape)
    pred, gt = (x if x.data_ptr() % 16 == 0 else x.clone() for x in (pred, gt))      # (a slice at an odd row)
    out = torch.empty((), device=pred.device)
    _lib.call("vsa_l1_mean", pred, gt, ctypes.c_longlong(pred.numel()), reduce_scratch(pred.device), out,
              _lib.stream_ptr())
    return out


def count_hits(hit_slot):
    """Number of entries >= 0 as a [] int64 device tensor, one launch (vsa_count_hits)."""
    import ctypes
    if hit_slot.dtype != torch.int32 or not hit_slot.is_contiguous() or not hit_slot.is_cuda:
        raise _lib.VolsurfsHipError("count_hits: contiguous CUDA int32 tensor expected")
    out = torch.empty((), dtype=torch.int64, device=hit_slot.device)
    _lib.call("vsa_count_hits", hit_slot, ctypes.c_longlong(hit_slot.numel()), reduce_scratch(hit_slot.device), out,
              _lib.stream_ptr())
    return out

