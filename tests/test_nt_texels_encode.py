"""Neural-texture steps 1-3 on the GPU vs the oracle: per-hit uv, unique texel
slots, hash-grid features (bit-exact)."""
import numpy as np
import pytest
import torch

from oracle import neural_texture as ONT
from oracle import tcnn_like


def test_product_grid_geometry_equals_oracle():
    from volsurfs_amd.neural_textures import grid_geometry
    scale, res, size, offset = grid_geometry()
    g = tcnn_like.GridGeometry()
    assert scale == g.scale and res == g.res and size == g.size and offset == g.offset


def test_plan_struct_matches_header_layout():
    """sizeof/offsets of the ctypes mirror vs the C struct, via a tiny C probe."""
    import ctypes, os, subprocess, tempfile
    from volsurfs_amd.neural_textures import Plan
    from volsurfs_amd import _lib
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "volsurfs_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",sizeof(vsa_nt_plan),offsetof(vsa_nt_plan,level_scale),offsetof(vsa_nt_plan,dom_off),offsetof(vsa_nt_plan,slot_capacity),offsetof(vsa_nt_plan,max_rays),offsetof(vsa_nt_plan,row_base));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "p.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "p")
        subprocess.run(["gcc", "-I", os.path.dirname(_lib.HEADER_PATH), c, "-o", exe], check=True)
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()
    assert [int(x) for x in out] == [ctypes.sizeof(Plan), Plan.level_scale.offset,
                                     Plan.dom_off.offset, Plan.slot_capacity.offset,
                                     Plan.max_rays.offset, Plan.row_base.offset]


def _setup(K=2, N=3000, seed=0):
    from volsurfs_amd.neural_textures import NeuralTextureBank
    g = torch.Generator().manual_seed(seed)
    nr_tris = 500
    face_uvs = torch.rand(nr_tris, 6, generator=g)
    hit_slot = torch.randint(0, nr_tris, (K, N), generator=g, dtype=torch.int32)
    hit_slot[torch.rand(K, N, generator=g) < 0.25] = -1
    bu = torch.rand(K, N, generator=g)
    bv = torch.rand(K, N, generator=g) * (1 - bu)
    hit_uv = torch.stack([bu, bv], -1)
    # a few exact-edge uvs
    face_uvs[0] = torch.tensor([0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    face_uvs[1] = torch.tensor([1.0, 1.0, 1.0, 1.0, 1.0, 1.0])
    hit_slot[0, :4] = torch.tensor([0, 1, 0, 1], dtype=torch.int32)
    bank = NeuralTextureBank(K, N, device="cuda", seed=3)
    with torch.no_grad():
        bank.tables.copy_((torch.rand(bank.tables.shape, generator=g) * 2 - 1))
    bank.refresh_half_params()
    return bank, face_uvs, hit_slot, hit_uv


@pytest.mark.gpu
def test_mark_compact_encode_bit_exact():
    bank, face_uvs, hit_slot, hit_uv = _setup()
    K, N = hit_slot.shape
    tex_uv = bank.mark_and_compact(hit_slot.cuda(), hit_uv.cuda(), face_uvs.cuda(), want_texel_of_slot=True)
    assert not bank.marks.any()          # the frame compaction clears the marks it consumed
    bank.encode()
    feats = bank.features_level_major()
    torch.cuda.synchronize()
    # --- uv (volsurfs.py:511-514)
    hit = hit_slot >= 0
    bary = torch.stack([(1 - hit_uv[..., 0]) - hit_uv[..., 1], hit_uv[..., 0], hit_uv[..., 1]], -1)
    fu = face_uvs.view(-1, 3, 2)
    ref_uv = torch.zeros(K, N, 2)
    for s in range(K):
        ref_uv[s][hit[s]] = ONT.interp_uv(bary[s][hit[s]], fu, hit_slot[s][hit[s]].long())
    assert torch.equal(tex_uv.cpu(), ref_uv)
    # --- slots: exactly the set of corner texels of every hit, per (shell, degree)
    seg = bank.seg_start.cpu().numpy()
    tos = bank.texel_of_slot.cpu().numpy()
    slot_of = bank.slot_of.cpu().numpy()
    geom = tcnn_like.GridGeometry()
    for s in range(K):
        for d in range(4):
            R = bank.tex_res[d]
            W = R + 2
            _, _, corners = ONT.texel_corners(ref_uv[s][hit[s]].clone(), R)
            ij = torch.floor(corners).long() + 1                  # extended-grid coords [M,4,2]
            want = np.unique((ij[..., 1] * W + ij[..., 0]).numpy().ravel())
            sd = s * 4 + d
            dom = int(bank.plan.dom_off[sd])
            got = tos[seg[sd]:seg[sd + 1]] - dom
            assert np.array_equal(got, want), (s, d)
            assert np.array_equal(slot_of[dom + want], np.arange(seg[sd], seg[sd + 1]))
            # --- features of every slot, both types, all 16 levels
            ix, iy = torch.from_numpy(want % W), torch.from_numpy(want // W)
            xy = torch.stack([((ix - 1).float() + 0.5) / R, ((iy - 1).float() + 0.5) / R], -1)
            for typ in range(2):
                x = bank.tex_index(s, typ, d)
                ref = tcnn_like.hashgrid_forward(geom, bank.tables[x].detach().cpu(), xy)  # [P,32] f16
                got_f = feats[typ, :, seg[sd]:seg[sd + 1]].cpu()                        # [16,P,2]
                got_f = got_f.permute(1, 0, 2).reshape(-1, 32)
                assert torch.equal(got_f, ref), (s, typ, d)
    assert seg[-1] == seg[K * 4] and seg[K * 4] <= bank.slot_capacity


@pytest.mark.gpu
def test_compact_empty_and_full():
    from volsurfs_amd.neural_textures import NeuralTextureBank
    bank = NeuralTextureBank(1, 64, device="cuda", textures_res=(64, 32, 16, 8))
    hs = torch.full((1, 64), -1, dtype=torch.int32).cuda()
    bank.mark_and_compact(hs, torch.zeros(1, 64, 2).cuda(), torch.zeros(4, 6).cuda())
    assert bank.seg_start.cpu().tolist()[:5] == [0, 0, 0, 0, 0]
    assert (bank.slot_of == -1).all()


@pytest.mark.gpu
def test_dense_level_larger_than_the_pair_offset_is_not_paired():
    """the paired dense-level forward stages the alpha table at a fixed LDS offset of
    15 360 entries; a grid whose dense level is larger (per_level_scale 2: 128^2 = 16 384 entries) used to
    overwrite the colour table's tail.  Such a level is now walked once per texture: features bit-exact
    against the oracle on that geometry, forward and (same planes) backward finite."""
    from volsurfs_amd.neural_textures import NeuralTextureBank
    grid = dict(per_level_scale=2.0)
    geom = tcnn_like.GridGeometry(per_level_scale=2.0)
    assert max(s for s, r in zip(geom.size, geom.res) if r * r <= s + 7) == 16384
    K, N = 1, 2000
    g = torch.Generator().manual_seed(5)
    face_uvs = torch.rand(200, 6, generator=g)
    hit_slot = torch.randint(0, 200, (K, N), generator=g, dtype=torch.int32)
    bu = torch.rand(K, N, generator=g)
    hit_uv = torch.stack([bu, torch.rand(K, N, generator=g) * (1 - bu)], -1)
    bank = NeuralTextureBank(K, N, device="cuda", seed=3, grid=grid)
    assert bank.n_entries == geom.offset[-1]
    with torch.no_grad():
        bank.tables.copy_((torch.rand(bank.tables.shape, generator=g) * 2 - 1))
    bank.refresh_half_params()
    bank.mark_and_compact(hit_slot.cuda(), hit_uv.cuda(), face_uvs.cuda())
    bank.encode()
    feats = bank.features_level_major()
    torch.cuda.synchronize()
    seg = bank.seg_start.cpu().numpy()
    xy_all = bank.slot_xy.cpu()
    for d in range(4):
        xy = xy_all[seg[d]:seg[d + 1]]
        for typ in range(2):
            x = bank.tex_index(0, typ, d)
            ref = tcnn_like.hashgrid_forward(geom, bank.tables[x].detach().cpu(), xy)
            got = feats[typ, :, seg[d]:seg[d + 1]].cpu().permute(1, 0, 2).reshape(-1, 32)
            assert torch.equal(got, ref), (typ, d)
