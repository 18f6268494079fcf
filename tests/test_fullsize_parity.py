"""Oracle parity AT the BASELINE frame sizes (VERDICT r5 missing #4 / next #1c).

The full-size frames were property-checked only (tests/test_pipeline_e2e.py::test_full_size_frame_properties);
oracle comparisons stopped at 128 x 128 rays / subdiv <= 4.  Here every BASELINE frame configuration — full-size
meshes (subdiv 6: 81 920 triangles per shell; subdiv 8: 1 310 720), full-resolution textures (2048 / 1024 / 512 /
256), the frame's camera — is sampled: rays spread evenly over the whole frame go through the HIP path and through
oracle.pipeline.render_step (brute-force closest hit over ALL triangles, the reference's per-hit evaluation of
methods/volsurfs.py:423-761, its fp16 autograd under loss scale 128), and hits (bit-exact), per-shell values, RGB
and every gradient are compared (oracle/parity.py).  bench.py reports the same comparison as `parity_sample`.
"""
import json
import os

import pytest
import torch

# (tag, K, subdiv, (H, W), sampled rays): configs[1], configs[4], configs[3]'s frame (its learned background is
# tests/test_methods.py::test_dtu_config_full_size_learned_background).  The sample of configs[4] is smaller: the
# oracle's closest hit is brute force over 7 x 1.31 M triangles.
CASES = [("config1_800x800_K5_subdiv6", 5, 6, (800, 800), 16384),
         ("config4_1920x1080_K7_subdiv8", 7, 8, (1080, 1920), 4096),
         ("config3_1600x1200_K5_subdiv6", 5, 6, (1200, 1600), 16384)]

# measured on MI355X (printed as FULLSIZE_PARITY ...; profiles/r06/fullsize_parity.json): weights 5.7e-3 / 2.8e-3 / 6.8e-3,
# tables 7.3e-3 / 9.1e-3 / 5.6e-3 for the three cases, against the oracle's fp16 autograd (loss scale 128: noisy itself);
# 4.6-7.8e-5 of all gradient entries are over north_star's 1e-3.  Bounds = 2x measured
GRAD_REL_MAX = {5: (1.2e-2, 1.9e-2), 7: (1.4e-2, 1.2e-2)}      # (MLP weights, hash tables) relative to a tensor's largest entry


@pytest.mark.gpu
@pytest.mark.parametrize("tag,K,subdiv,hw,n", CASES, ids=[c[0] for c in CASES])
def test_sampled_rays_of_the_full_size_frame_match_the_oracle(tag, K, subdiv, hw, n):
    from oracle import parity as opar
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.pipeline import KShellPipeline
    H, W = hw
    meshes = nested_shells(K=K, subdiv=subdiv, noise=0.05)
    o, d = pinhole_rays(H, W, focal=1111.1 * min(H, W) / 800.0, cam_pos=(0.0, 0.0, -1.5))
    idx = torch.linspace(0, H * W - 1, n, device=o.device).long()
    gt = torch.rand(n, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    pipe = KShellPipeline(meshes, o[idx].contiguous(), d[idx].contiguous(), gt, seed=5, init="spread")
    assert pipe.tracer.mesh_nr_tris[0] == 20 * 4 ** subdiv and pipe.bank.tex_res == (2048, 1024, 512, 256)
    rgb = pipe.step()
    torch.cuda.synchronize()
    ref, sec = opar.oracle_step(pipe, loss_scale=128.0)
    rep = opar.compare_step(pipe, rgb, ref)
    rep["oracle_seconds"] = round(sec, 1)
    print("FULLSIZE_PARITY " + json.dumps({tag: rep}))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "fullsize_parity.json")
        cur = json.load(open(path)) if os.path.exists(path) else {}
        cur[tag] = rep
        json.dump(cur, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    # integer work: the closest hit of every (ray, shell) against brute force over all triangles
    assert rep["hit_mismatches"] == 0 and rep["hits"] > n // 2
    # per-shell values and RGB: identical except where an 8-bit texel flipped by one step
    assert rep["texel_max_step"] <= 1 and rep["texel_flip_rate"] < 3e-5
    assert rep["rgb_median_err"] == 0.0 and rep["rgb_frac_over_1e-4"] <= 2e-3 and rep["rgb_max_err"] <= 1e-2
    assert rep["rays_over_1e-4_without_a_flipped_texel"] <= 2 and rep["max_err_without_a_flipped_texel"] <= 1e-3, rep
    assert rep["surfs_rgb_frac_over_1e-5"] < 1e-3 and rep["surfs_alpha_frac_over_1e-5"] < 1e-3
    # gradients of all 8 K textures' tables and MLPs
    assert rep["grad_tensors"] == 8 * K and rep["grad_cos_min"] > 0.98
    assert rep["grad_weights_rel_max"] <= GRAD_REL_MAX[K][0] and rep["grad_tables_rel_max"] <= GRAD_REL_MAX[K][1], rep
