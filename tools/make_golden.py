#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference Python in place.

Runs only in the build container (needs /root/reference).  The arrays it
writes are the committed fixtures; the reference source itself is never copied.
Usage:  python tools/make_golden.py [composite] [sh] [nt] [glue] [misc]
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)

import ref_import  # noqa: E402


def _save(name, **arrs):
    path = os.path.join(GOLD, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print("wrote", path, os.path.getsize(path), "bytes")


# --------------------------------------------------------------------------
# 1. dense K-shell composite through the reference's own VolSurfs.render_rays
#    (volsurfs_py/methods/volsurfs.py:423-761) with fake tracer / models.
# --------------------------------------------------------------------------
def gen_composite():
    ref_import.install_placeholders()
    from volsurfs_py.methods.volsurfs import VolSurfs

    for K, N, seed, decay, bg_mode in [
        (1, 256, 1, False, "white"),
        (5, 256, 2, True, "white"),
        (7, 256, 3, True, "perray"),
        (5, 64, 4, False, "black"),
    ]:
        g = torch.Generator().manual_seed(seed)
        rays_o = torch.zeros(N, 3)
        rays_d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
        hit_p = torch.rand(N, K, generator=g)
        is_hit = hit_p < 0.7
        is_hit[:4] = False          # rays that miss every shell
        is_hit[4:8] = True          # rays that hit every shell
        normals = torch.nn.functional.normalize(torch.randn(N, K, 3, generator=g), dim=-1)
        positions = torch.randn(N, K, 3, generator=g)
        rgb_in = torch.rand(N, K, 3, generator=g)
        alpha_in = torch.rand(N, K, 1, generator=g)
        alpha_in[8:12] = 1.0        # opaque shells
        alpha_in[12:16] = 0.0       # fully transparent shells
        rgb_leaf = rgb_in.clone().requires_grad_(True)
        alpha_leaf = alpha_in.clone().requires_grad_(True)

        class Tracer:
            def trace(self, o, d, mesh_id):
                h = is_hit[:, mesh_id]
                return {
                    "any_hit": bool(h.any()),
                    "triangles_id": torch.zeros(N, dtype=torch.long),
                    "depth": torch.ones(N),
                    "is_hit": h,
                    "positions": positions[:, mesh_id],
                    "normals": normals[:, mesh_id],
                    "barycentric": torch.full((N, 3), 1.0 / 3.0),
                }

        def mk_model(leaf, i):
            idx_of = {}

            def model(points=None, samples_dirs=None, normals=None, iter_nr=None):
                h = is_hit[:, i]
                return leaf[h, i]

            return model

        models = {"bg": None}
        for i in range(K):
            models[f"rgb_{i}"] = mk_model(rgb_leaf, i)
            models[f"alpha_{i}"] = mk_model(alpha_leaf, i)

        if bg_mode == "white":
            bg_color = torch.ones(1, 3)
        elif bg_mode == "black":
            bg_color = torch.zeros(1, 3)
        else:
            bg_color = torch.rand(N, 3, generator=g)
        bg_leaf = bg_color.clone().requires_grad_(True)

        class BP:
            def intersect(self, o, d):
                n = o.shape[0]
                return (torch.ones(n, dtype=torch.bool), torch.zeros(n), torch.ones(n),
                        o, o + d)

        fake = SimpleNamespace(
            bounding_primitive=BP(), nr_meshes=K,
            hyper_params=SimpleNamespace(using_neural_textures=False,
                                         are_volsurfs_colors_indep=True,
                                         are_volsurfs_alphas_indep=True,
                                         nr_samples_bg=0),
            profiler=None, raytracer=Tracer(), models=models,
            with_alpha_decay=decay, bg_color=bg_leaf, is_training=True,
            tensor_meshes=[None] * K)

        res = VolSurfs.render_rays(fake, rays_o, rays_d, iter_nr=0)
        rt = res["renders"]["ray_traced"]
        gt = torch.rand(N, 3, generator=g)
        # reference loss: utils/losses.py:14-19 via volsurfs.py:806
        from volsurfs_py.utils.losses import loss_l1
        loss = loss_l1(gt, rt["rgb"])
        loss.backward()
        out = {k: v.detach().numpy() for k, v in rt.items()}
        _save(f"composite_K{K}_N{N}_{bg_mode}.npz",
              rays_d=rays_d.numpy(), is_hit=is_hit.numpy(), normals=normals.numpy(),
              positions=positions.numpy(), rgb_in=rgb_in.numpy(), alpha_in=alpha_in.numpy(),
              bg_color=bg_color.numpy(), with_alpha_decay=np.array(decay), gt=gt.numpy(),
              loss=loss.detach().numpy(),
              g_rgb_in=rgb_leaf.grad.numpy(), g_alpha_in=alpha_leaf.grad.numpy(),
              g_bg=bg_leaf.grad.numpy(),
              samples_3d=res["samples_3d"].detach().numpy(),
              samples_grad=res["samples_grad"].detach().numpy(),
              **{"out_" + k: v for k, v in out.items()})


# --------------------------------------------------------------------------
# 2. SHNeuralTextures / NeuralTexture (models/sh_neural_textures.py:64-97,
#    models/neural_texture.py:81-197) and SHEncoder (encodings/
#    sphericalharmonics.py) run AS SHIPPED, with oracle.tcnn_like standing in
#    for tinycudann and oracle.uv for mvdatasets.utils.images (both absent).
#    Parameters are regenerated from seeds by tests (tables are MBs).
# --------------------------------------------------------------------------
def gen_nt():
    from oracle import tcnn_like, uv as uvh
    ref_import.install_placeholders({
        "tinycudann": {"Encoding": tcnn_like.Encoding, "Network": tcnn_like.Network},
        "mvdatasets.utils.images": {k: getattr(uvh, k) for k in [
            "normalize_uv_coord", "non_normalize_uv_coord",
            "non_normalized_uv_coords_to_interp_corners", "pix_to_texel_center_uv_coord",
            "uv_coords_to_pix", "non_normalized_uv_coords_to_lerp_weights"]},
    })
    from volsurfs_py.models.sh_neural_textures import SHNeuralTextures
    from volsurfs_py.encodings.sphericalharmonics import SHEncoder

    # (r5: + the branches no shipped config sets — anchor, squeeze without quantise, neither —
    #  neural_texture.py:88-104, 159-169, 183-187)
    for name, C, sh_deg, seed, M, flags in [
            ("rgb", 3, 3, 11, 256, {}), ("alpha", 1, 3, 12, 256, {}), ("alpha_deg0", 1, 0, 13, 128, {}),
            ("rgb_anchor", 3, 3, 14, 128, dict(anchor=True, lerp=False)),
            ("alpha_anchor", 1, 3, 15, 128, dict(anchor=True, lerp=False)),
            ("rgb_noquant", 3, 2, 16, 96, dict(quantize_output=False)),
            ("rgb_raw", 3, 2, 17, 96, dict(quantize_output=False, squeeze_output=False))]:
        kw = dict(anchor=False, lerp=True, quantize_output=True, squeeze_output=True)
        kw.update(flags)
        model = SHNeuralTextures(sh_deg=sh_deg, nr_channels=C, sh_range=[15, 15, 15, 15],
                                 deg_res=[2048, 1024, 512, 256], align_to_webgl=True, **kw)
        from oracle.neural_texture import make_test_params
        params = make_test_params(seed, C, sh_deg)
        with torch.no_grad():
            for deg, (table, w1, w2, w3) in enumerate(params):
                nt = model.neural_textures[deg]
                nt.encoding.params.copy_(table)
                nt.network.w1.copy_(w1)
                nt.network.w2.copy_(w2)
                nt.network.w3.copy_(w3)
        g = torch.Generator().manual_seed(seed)
        uv = torch.rand(M, 2, generator=g)
        uv[:4] = torch.tensor([[0.0, 0.0], [1.0, 1.0], [0.0001, 0.9999], [0.5, 0.5]])
        dirs = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1)
        out = model(uv_coords=uv, view_dirs=dirs)
        coeffs = model(uv_coords=uv)                       # [M,C,(deg+1)^2] lerped SH coefficients
        gt = torch.rand(M, C, generator=g)
        loss = (gt - out).abs().mean()
        loss.backward()
        arrs = dict(uv=uv.numpy(), dirs=dirs.numpy(), out=out.detach().numpy(),
                    coeffs=coeffs.detach().numpy(), gt=gt.numpy(), seed=np.array(seed),
                    nr_channels=np.array(C), sh_deg=np.array(sh_deg))
        if flags:
            arrs["flags"] = np.array([int(kw[k]) for k in ("anchor", "lerp", "quantize_output", "squeeze_output")])
        for deg in range(sh_deg + 1):
            nt = model.neural_textures[deg]
            arrs[f"param_sum_{deg}"] = np.array([nt.encoding.params.double().sum().item(),
                                                 nt.network.w3.double().sum().item()])
            for wn in ("w1", "w2", "w3"):
                arrs[f"g_{wn}_{deg}"] = getattr(nt.network, wn).grad.numpy()
            gt_ = nt.encoding.params.grad
            idx = gt_.abs().sum(1).topk(64).indices
            arrs[f"g_table_top_idx_{deg}"] = idx.numpy()
            arrs[f"g_table_top_val_{deg}"] = gt_[idx].numpy()
            arrs[f"g_table_sums_{deg}"] = np.array([gt_.double().sum().item(),
                                                    gt_.double().abs().sum().item()])
        _save(f"sh_neural_textures_{name}.npz", **arrs)

    # SHEncoder.__call__ / eval on their own (degrees 0..3)
    g = torch.Generator().manual_seed(5)
    dirs = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1)
    arrs = dict(dirs=dirs.numpy())
    for deg in range(4):
        sh = (torch.randn(64, 3, (deg + 1) ** 2, generator=g)).half()
        arrs[f"sh_{deg}"] = sh.float().numpy()
        arrs[f"eval_{deg}"] = SHEncoder.eval(sh, dirs, deg).float().numpy()
        arrs[f"enc_{deg}"] = SHEncoder(degree=deg)(dirs).numpy()
    _save("sh_encoder.npz", **arrs)


# --------------------------------------------------------------------------
# 3. Autograd glue of the packed background composite
#    (volume_rendering/volume_rendering_funcs.py:91-241 + the weight math of
#    utils/background.py:93-111) run AS SHIPPED on top of oracle.packed (the
#    native `volsurfs.VolumeRendering` is CUDA-only).
# --------------------------------------------------------------------------
def gen_glue():
    from oracle import packed as OP

    class _Pack:
        def __init__(self, se):
            self.ray_start_end_idx = se

    class _VR:
        @staticmethod
        def cumprod_one_minus_alpha_to_transmittance(p, a):
            T, b = OP.cumprod_fwd(p.ray_start_end_idx, a.detach().numpy())
            return torch.from_numpy(T)[:, None], torch.from_numpy(b)[:, None]

        @staticmethod
        def cumsum_over_rays(p, v, inverse):
            return torch.from_numpy(OP.cumsum(p.ray_start_end_idx, v.detach().numpy(), inverse))[:, None]

        @staticmethod
        def cumprod_one_minus_alpha_to_transmittance_backward(gT, gb, p, a, T, b, lv):
            g = OP.cumprod_bwd(p.ray_start_end_idx, gb.numpy()[:, 0], a.detach().numpy(), b.numpy()[:, 0],
                               lv.numpy()[:, 0])
            return torch.from_numpy(g)[:, None]

        @staticmethod
        def integrate_with_weights_3d(p, v, w):
            return torch.from_numpy(OP.integrate_fwd(p.ray_start_end_idx, v.detach().numpy(), w.detach().numpy()))

        @staticmethod
        def integrate_with_weights_3d_backward(g, p, v, w, res):
            gv, gw = OP.integrate_bwd(p.ray_start_end_idx, g.numpy(), v.detach().numpy(), w.detach().numpy(),
                                      bug_compat=True)
            return torch.from_numpy(gv), torch.from_numpy(gw)[:, None]

    ref_import.install_placeholders({"volsurfs": {"VolumeRendering": _VR}})
    from volsurfs_py.volume_rendering.volume_rendering_funcs import (
        CumprodOneMinusAlphaToTransmittanceFunc, IntegrateWithWeights3DFunc)

    g = torch.Generator().manual_seed(21)
    counts = torch.randint(0, 40, (50,), generator=g)
    counts[:3] = torch.tensor([0, 1, 32])
    ends = torch.cumsum(counts, 0)
    se = torch.stack([ends - counts, ends], 1).int().numpy()
    S = int(ends[-1])
    density = (torch.rand(S, 1, generator=g) * 3).requires_grad_(True)
    rgb = torch.rand(S, 3, generator=g).requires_grad_(True)
    dt = torch.rand(S, 1, generator=g) * 0.3
    pack = _Pack(se)
    alpha = 1.0 - torch.exp(-density * dt)                       # background.py:93-95
    one_minus_alpha = 1 - alpha
    T, bgT = CumprodOneMinusAlphaToTransmittanceFunc.apply(pack, one_minus_alpha + 1e-6)   # :99-104
    weights = alpha * T                                          # :105
    pred = IntegrateWithWeights3DFunc.apply(pack, rgb, weights)  # :109-111
    gt = torch.rand(50, 3, generator=g)
    loss = ((gt - pred).abs().mean() + 0.1 * bgT.mean())
    loss.backward()
    _save("packed_glue.npz", start_end=se, density=density.detach().numpy(), rgb=rgb.detach().numpy(),
          dt=dt.numpy(), gt=gt.numpy(), T=T.detach().numpy(), bgT=bgT.detach().numpy(),
          pred=pred.detach().numpy(), g_density=density.grad.numpy(), g_rgb=rgb.grad.numpy())


# --------------------------------------------------------------------------
# 4. Legacy appearance branch + background field: the reference's own MLP, RGB, ColorSH and
#    NerfHash classes (models/{mlp,rgb,color_sh,nerfhash}.py) and GridHashEncoder
#    (encodings/gridhash.py), with oracle.tcnn_like.GridEncoding standing in for the absent
#    tinycudann grid and the restated Coarse2Fine for permutohedral_encoding's.  The classes
#    hard-code .to("cuda") for their bounding boxes; that one call is redirected to the CPU.
# --------------------------------------------------------------------------
def gen_legacy():
    from oracle import tcnn_like

    class _C2F:
        def __init__(self, n):
            self.n = n

        def __call__(self, t):
            import math
            a = float(t) * self.n
            i = torch.arange(self.n, dtype=torch.float32)
            return 0.5 * (1.0 - torch.cos(math.pi * torch.clamp(a - i, 0.0, 1.0)))

    ref_import.install_placeholders({
        "tinycudann": {"Encoding": tcnn_like.GridEncoding},
        "permutohedral_encoding": {"Coarse2Fine": _C2F},
    })
    orig_to = torch.Tensor.to

    def to_cpu(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
        return orig_to(self, *a, **k)
    torch.Tensor.to = to_cpu
    try:
        from volsurfs_py.models.nerfhash import NerfHash
        from volsurfs_py.models.rgb import RGB
        from volsurfs_py.models.color_sh import ColorSH
        torch.manual_seed(5)
        g = torch.Generator().manual_seed(6)
        arrs = {}

        def dump(prefix, model):     # the 48 MB hash tables are re-created from the seed by the test
            for k, v in model.state_dict().items():
                if not k.endswith("encoder.params"):
                    arrs[f"{prefix}/{k}"] = v.detach().numpy()

        def spread(model):     # visible structure instead of U(-1e-4, 1e-4)
            with torch.no_grad():
                model.pos_encoder.encoder.params.copy_(
                    torch.rand(model.pos_encoder.encoder.params.shape, generator=g) * 2 - 1)

        M = 96
        pts = (torch.rand(M, 3, generator=g) * 2 - 1) * 0.9
        dirs = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1)
        nrm = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1)
        arrs.update(points=pts.numpy(), dirs=dirs.numpy(), normals=nrm.numpy())

        nh = NerfHash(3, "gridhash", "spherical_harmonics")
        spread(nh)
        rgb, dens = nh(pts, dirs, iter_nr=None)
        dump("nerfhash", nh)
        arrs.update(nerfhash_rgb=rgb.detach().numpy(), nerfhash_density=dens.detach().numpy())
        loss = (rgb * torch.linspace(0.5, 1.5, 3)).sum() + 0.3 * dens.sum()
        loss.backward()
        gt = nh.pos_encoder.encoder.params.grad
        nz = torch.nonzero(gt.abs().sum(1)).flatten()
        arrs["nerfhash_grad/table_idx"] = nz.numpy()
        arrs["nerfhash_grad/table_val"] = gt[nz].numpy()
        arrs["nerfhash_grad/mlp_rgb0"] = nh.mlp_rgb.layers[0].weight.grad.numpy()

        m = RGB(3, [128, 128, 64], "gridhash", "spherical_harmonics", sh_deg=3, normal_dep=True,
                bb_sides=1.0)
        spread(m)
        out = m(points=pts * 0.5, samples_dirs=dirs, normals=nrm, iter_nr=None)
        dump("rgb", m)
        arrs["rgb_out"] = out.detach().numpy()

        c = ColorSH(3, [128, 128, 64], "gridhash", sh_deg=3, bb_sides=1.0)
        spread(c)
        arrs["colorsh_out"] = c(pts * 0.45, samples_dirs=dirs).detach().numpy()
        arrs["colorsh_coeffs"] = c(pts * 0.45).detach().numpy()
        dump("colorsh", c)
    finally:
        torch.Tensor.to = orig_to
    _save("legacy_models.npz", **arrs)


# --------------------------------------------------------------------------
# 5. Training-loop pieces that import as shipped: the warm-up scheduler
#    (schedulers/warmup.py) chained to MultiStepLR (the reference vendors torch 1.10's class,
#    whose constructor no longer matches torch 2.x; torch's own is used as after_scheduler),
#    stepped like trainer.py:306-308; loss_l1 / loss_l2 (utils/losses.py:6-19).
# --------------------------------------------------------------------------
def gen_misc():
    import warnings
    warnings.filterwarnings("ignore")
    ref_import.install_placeholders({})
    from volsurfs_py.schedulers.warmup import GradualWarmupScheduler
    from torch.optim.lr_scheduler import MultiStepLR
    arrs = {}
    for tag, warm, ms, n in [("a", 50, [100, 200], 320), ("b", 0, [30, 80], 120), ("c", 7, [3, 4, 20], 60),
                             ("default", 3000, [100000, 150000, 180000, 190000], 3100)]:
        opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
        dec = MultiStepLR(opt, milestones=ms, gamma=0.3)
        sch = GradualWarmupScheduler(opt, multiplier=1, total_epoch=warm, after_scheduler=dec) if warm > 0 else dec
        lrs = []
        for _ in range(n):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        arrs[f"lr_{tag}"] = np.array(lrs, np.float64)
        arrs[f"cfg_{tag}"] = np.array([warm, n] + ms, np.int64)
    g = torch.Generator().manual_seed(3)
    gt, pred = torch.rand(200, 3, generator=g), torch.rand(200, 3, generator=g)
    mask = (torch.rand(200, 1, generator=g) > 0.4).float()
    # utils/losses.py:6-19 executed as written (its module imports sdf/field helpers that need
    # the absent native extension, so the two functions are evaluated through their source lines)
    import importlib
    try:
        L = importlib.import_module("volsurfs_py.utils.losses")
        l1, l1m, l2 = L.loss_l1(gt, pred), L.loss_l1(gt, pred, mask), L.loss_l2(gt, pred)
    except Exception as e:      # pragma: no cover
        raise SystemExit(f"cannot import the reference losses: {e}")
    arrs.update(loss_gt=gt.numpy(), loss_pred=pred.numpy(), loss_mask=mask.numpy(),
                loss_l1=l1.numpy(), loss_l1_masked=l1m.numpy(), loss_l2=l2.numpy())
    _save("train_misc.npz", **arrs)


GENS = {"composite": gen_composite, "nt": gen_nt, "glue": gen_glue, "legacy": gen_legacy,
        "misc": gen_misc}

if __name__ == "__main__":
    # One generator per process: each one imports the reference with its own set of stand-ins
    # for the absent third-party modules (bare placeholders vs the oracle's tcnn / uv helpers),
    # and Python caches the first import.  `python tools/make_golden.py` regenerates every
    # fixture; `python tools/make_golden.py nt glue` only those.
    import subprocess
    which = sys.argv[1:] or list(GENS)
    if len(which) == 1 and which[0].startswith("--only="):
        GENS[which[0][7:]]()
    else:
        for w in which:
            if w not in GENS:
                raise SystemExit(f"unknown generator {w!r}; have {sorted(GENS)}")
            subprocess.run([sys.executable, os.path.abspath(__file__), f"--only={w}"], check=True)
