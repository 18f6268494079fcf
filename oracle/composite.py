"""Oracle: dense K-shell alpha composite.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows /root/reference/volsurfs_py/methods/volsurfs.py:601-640 (flip to
outer->inner, cast to fp16, cumprod, blending weights, weighted sum) and
:704-708 (background blend), :728-748 (cast back to fp32).

Rounding points restated explicitly (they are what the HIP kernel mirrors):
  a_h   = fp16(alpha)                    volsurfs.py:607
  c_h   = fp16(rgb)                      volsurfs.py:606
  om_h  = fp16(1 - a_h)                  volsurfs.py:612 (argument of cumprod)
  cum_k = running product of om_h.  ATen's CPU kernel carries the running
          product in fp32 and rounds only what it stores
          (aten/src/ATen/native/cpu/ReduceOpsKernel.cpp cumprod: acc_type),
          ATen's CUDA scan carries it in fp16.  `carry` selects: "f32"
          (CPU ATen; what the golden fixtures pin) or "f16" (CUDA ATen).
  T_k   = fp16(cum_{k-1}), T_0 = 1       volsurfs.py:614-623
  w_k   = fp16(T_k * a_h_k)              volsurfs.py:631
  p_kc  = fp16(c_h_kc * w_k)             volsurfs.py:634 (product)
  fg_c  = fp16(sum_k fp32(p_kc))         volsurfs.py:634 (ATen half sum: fp32 accumulate)
  bg_h  = fp16(rgb_bg)                   volsurfs.py:705
  rgb_c = fp16(fg_c + fp16(bgT * bg_h_c))  volsurfs.py:708
All outputs are then widened to fp32 (volsurfs.py:729-736).

Shell index convention: input/outputs are inner->outer (mesh 0 innermost,
utils/mesh_loaders.py:28-30); the composite itself runs outer->inner.
"""
import numpy as np

f16 = np.float16
f32 = np.float32


def composite_dense_fwd(surfs_rgb, surfs_alpha, rgb_bg, carry="f32"):
    """surfs_rgb [N,K,3] f32, surfs_alpha [N,K] f32 (inner->outer), rgb_bg [N,3]
    or [1,3] f32.  Returns dict of fp32 arrays with the reference's keys."""
    surfs_rgb = np.asarray(surfs_rgb, f32)
    surfs_alpha = np.asarray(surfs_alpha, f32)
    N, K, _ = surfs_rgb.shape
    rgb_bg = np.broadcast_to(np.asarray(rgb_bg, f32), (N, 3))

    c_h = surfs_rgb[:, ::-1].astype(f16)          # outer -> inner
    a_h = surfs_alpha[:, ::-1].astype(f16)
    om_h = (f32(1.0) - a_h.astype(f32)).astype(f16)

    T = np.empty((N, K), f16)
    if carry == "f32":
        acc = np.ones(N, f32)
    else:
        acc = np.ones(N, f16)
    T_prev = np.ones(N, f16)
    for k in range(K):
        T[:, k] = T_prev
        if carry == "f32":
            acc = acc * om_h[:, k].astype(f32)
            T_prev = acc.astype(f16)
        else:
            acc = (acc.astype(f32) * om_h[:, k].astype(f32)).astype(f16)
            T_prev = acc
    bgT = T_prev                                   # [N] fp16

    w = (T.astype(f32) * a_h.astype(f32)).astype(f16)             # [N,K]
    p = (c_h.astype(f32) * w.astype(f32)[:, :, None]).astype(f16)  # [N,K,3]
    fg = np.zeros((N, 3), f32)
    for k in range(K):
        fg = fg + p[:, k].astype(f32)
    fg_h = fg.astype(f16)
    bg_h = rgb_bg.astype(f16)
    t = (bgT.astype(f32)[:, None] * bg_h.astype(f32)).astype(f16)
    rgb_h = (fg_h.astype(f32) + t.astype(f32)).astype(f16)

    return {
        "rgb": rgb_h.astype(f32),
        "rgb_fg": fg_h.astype(f32),
        "rgb_bg": bg_h.astype(f32),
        "surfs_alpha": a_h[:, ::-1].astype(f32)[:, :, None],
        "surfs_rgb": c_h[:, ::-1].astype(f32),
        "surfs_blending_weights": w[:, ::-1].astype(f32)[:, :, None],
        "bg_transmittance": bgT.astype(f32)[:, None],
    }


def composite_dense_bwd(surfs_rgb, surfs_alpha, rgb_bg, g_rgb, carry="f32"):
    """Analytic gradient of `rgb` w.r.t. the fp32 inputs, evaluated at the
    fp16-rounded forward operating point in fp32 (the reference back-propagates
    through the same graph with fp16 autograd; agreement is to fp16 noise,
    tests/ state the tolerance).

      rgb_c   = sum_k T_k a_k c_kc + T_K bg_c        (outer->inner, T_K = bgT)
      g_c_kc  = g_c * w_k
      S_{K-1} = g . bg ;  S_{k-1} = a_k (g . c_k) + om_k S_k   (om_k = fp16(1 - a_k))
      g_a_k   = T_k * (g . c_k - S_k)
      g_bg_c  = g_c * bgT
    Returns (g_surfs_rgb [N,K,3], g_surfs_alpha [N,K], g_rgb_bg [N,3]) in
    inner->outer order."""
    surfs_rgb = np.asarray(surfs_rgb, f32)
    surfs_alpha = np.asarray(surfs_alpha, f32)
    g_rgb = np.asarray(g_rgb, f32)
    N, K, _ = surfs_rgb.shape
    rgb_bg = np.broadcast_to(np.asarray(rgb_bg, f32), (N, 3))
    fwd = composite_dense_fwd(surfs_rgb, surfs_alpha, rgb_bg, carry)
    c = fwd["surfs_rgb"][:, ::-1]                  # outer->inner, rounded
    a = fwd["surfs_alpha"][:, ::-1, 0]
    w = fwd["surfs_blending_weights"][:, ::-1, 0]
    bgT = fwd["bg_transmittance"][:, 0]
    bg = fwd["rgb_bg"]
    # T_k rebuilt exactly like the forward (w/a would lose T where a == 0)
    om_h = (f32(1.0) - a).astype(f16)
    acc = np.ones(N, f32)
    Tp = np.ones(N, f16)
    T = np.empty((N, K), f32)
    for k in range(K):
        T[:, k] = Tp.astype(f32)
        if carry == "f32":
            acc = acc * om_h[:, k].astype(f32)
            Tp = acc.astype(f16)
        else:
            Tp = (Tp.astype(f32) * om_h[:, k].astype(f32)).astype(f16)

    g_c = g_rgb[:, None, :] * w[:, :, None]
    r = (g_rgb[:, None, :] * c).sum(-1)            # [N,K]
    S = (g_rgb * bg).sum(-1)                       # S_{K-1}
    g_a = np.empty((N, K), f32)
    for k in range(K - 1, -1, -1):
        g_a[:, k] = T[:, k] * (r[:, k] - S)
        S = a[:, k] * r[:, k] + om_h[:, k].astype(f32) * S
    g_bg = g_rgb * bgT[:, None]
    return g_c[:, ::-1].copy(), g_a[:, ::-1].copy(), g_bg
