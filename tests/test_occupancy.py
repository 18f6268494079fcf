"""OccupancyGrid + the grid-occupied foreground sampler: HIP kernels vs the
CPU restatement of the reference kernels (oracle/occupancy.py), bit for bit, plus properties."""
import numpy as np
import pytest
import torch

from oracle import occupancy as OO
from oracle.packed import Pcg32

EXT = [2.0, 1.5, 1.0]


def _rays(g, N, ext, inside=0.95):
    half = np.array(ext, np.float32) * 0.5 * inside
    p1 = g.uniform(-1, 1, (N, 3)).astype(np.float32) * half
    p2 = g.uniform(-1, 1, (N, 3)).astype(np.float32) * half
    d = p2 - p1
    L = np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    d = (d / L).astype(np.float32)
    return p1, d, np.zeros((N, 1), np.float32), L.astype(np.float32)


def _blob_occupancy(n, ext, g):
    """occupied = a few random balls; roi = everything but a slab."""
    idx = np.arange(n ** 3)
    c = OO.grid_points(idx, n, ext, True)
    occ = np.zeros(n ** 3, bool)
    for _ in range(4):
        centre = g.uniform(-0.3, 0.3, 3) * np.array(ext)
        occ |= np.linalg.norm(c - centre, axis=1) < 0.22
    roi = c[:, 0] < 0.6
    return occ, roi


def test_oracle_morton_and_voxel_geometry():
    n = 16
    for v in [0, 1, 5, 77, n ** 3 - 1]:
        x, y, z = (OO.morton3d_invert(v >> k) for k in range(3))
        assert OO.morton3d(x, y, z) == v and max(x, y, z) < n
        centre = OO.lin_idx_to_3d(v, n, EXT, True, True)
        assert OO.pos_to_lin_idx(centre, n, EXT) == v               # a voxel's centre maps back to it
        ll = OO.lin_idx_to_3d(v, n, EXT, True, False)
        assert all(abs((centre[k] - ll[k]) - EXT[k] / n / 2) < 1e-6 for k in range(3))
    assert not OO.voxel_ok(OO.pos_to_lin_idx([1.5, 0, 0], n, EXT), n)     # outside in +x
    assert OO.voxel_ok(OO.pos_to_lin_idx([-5.0, 0, 0], n, EXT), n)        # negative side saturates to voxel 0 (reference quirk)
    # DDA: from a voxel centre along +x the next boundary is half a voxel away
    t = OO.distance_to_next_voxel(OO.lin_idx_to_3d(9, n, EXT, True, True), [1, 0, 0], n, EXT)
    assert abs(t - (EXT[0] / n / 2 + 1e-6)) < 1e-6


def test_oracle_marchers_on_a_slab():
    """Occupied slab |x| < 0.25 of a unit cube: rays along +x enter / leave it where expected and
    the sampler puts equidistant samples inside it only."""
    n, ext = 16, [1.0, 1.0, 1.0]
    c = OO.grid_points(np.arange(n ** 3), n, ext, True)
    occ, roi = np.abs(c[:, 0]) < 0.25, np.ones(n ** 3, bool)
    o = np.array([[-0.45, 0.01, 0.02], [-0.45, 0.3, -0.2]], np.float32)
    d = np.array([[1, 0, 0], [1, 0, 0]], np.float32)
    t0, t1 = np.zeros((2, 1), np.float32), np.full((2, 1), 0.9, np.float32)
    near, far = OO.rays_t_near_t_far(o, d, t0[:, 0], t1[:, 0], n, ext, occ, roi)
    assert np.all(np.abs(near - 0.2) < 1.5 / n) and np.all(np.abs(far - 0.7) < 1.5 / n)
    se, z, s3d, dt = OO.sample_fg_occupied(o, d, t0[:, 0], t1[:, 0], 0.05, 1, 32, False, Pcg32(), n, ext, occ, roi)
    for i in range(2):
        k = se[i, 1] - se[i, 0]
        assert 8 <= k <= 11
        zs = z[i, :k]
        assert np.all(np.abs(s3d[i, :k, 0]) < 0.25 + 1e-5)
        assert np.allclose(np.diff(zs), dt[i], atol=2e-5)


def _grid(n, ext, seed=0):
    from volsurfs_amd.volsurfs import OccupancyGrid, _Pcg32State
    g = np.random.default_rng(seed)
    occ, roi = _blob_occupancy(n, ext, g)
    grid = OccupancyGrid(n, ext)
    OccupancyGrid.m_rng = _Pcg32State()
    grid.set_grid_occupancy(torch.from_numpy(occ).cuda())
    grid.m_grid_roi = torch.from_numpy(roi).cuda()
    vals = g.uniform(-1, 1, n ** 3).astype(np.float32)
    grid.set_grid_values(torch.from_numpy(vals).cuda())
    return grid, occ, roi, vals, g


@pytest.mark.gpu
def test_hip_grid_points_and_updates_vs_oracle():
    from volsurfs_amd.volsurfs import OccupancyGrid
    n = 16
    grid, occ, roi, vals, g = _grid(n, EXT)
    ll, idx = grid.get_grid_lower_left_voxels_vertices()
    assert np.array_equal(ll.cpu().numpy(), OO.grid_points(np.arange(n ** 3), n, EXT, False))
    pts, _ = grid.get_grid_samples(False)
    assert np.array_equal(pts.cpu().numpy(), OO.grid_points(np.arange(n ** 3), n, EXT, True))
    jit, _ = grid.get_grid_samples(True)
    assert np.array_equal(jit.cpu().numpy(), OO.grid_points(np.arange(n ** 3), n, EXT, True, True, Pcg32()))
    assert (jit - pts).abs().max().item() <= max(EXT) / n / 2 + 1e-6
    sel, sel_idx = grid.get_random_grid_samples_in_roi(500, False)
    assert roi[sel_idx.cpu().numpy()].all()
    assert np.array_equal(sel.cpu().numpy(), OO.grid_points(sel_idx.cpu().numpy(), n, EXT, True))
    # values: max(new, old * decay)
    pick = torch.from_numpy(g.permutation(n ** 3)[:1000].astype(np.int32)).cuda()
    new = g.uniform(-1, 1, (1000, 1)).astype(np.float32)
    ref = vals.copy()
    OO.update_values(ref, pick.cpu().numpy(), new[:, 0], 0.95)
    grid.update_grid_values(pick, torch.from_numpy(new).cuda(), 0.95)
    assert np.array_equal(grid.get_grid_values().cpu().numpy(), ref)
    # occupancy from densities, with and without the 27-neighbourhood
    for nb in (False, True):
        ref_occ = occ.copy()
        OO.update_occupancy_density(ref_occ, ref, pick.cpu().numpy(), n, EXT, 0.3, nb)
        grid.set_grid_occupancy(torch.from_numpy(occ).cuda())
        grid.update_grid_occupancy_with_density_values(pick, 0.3, nb)
        assert np.array_equal(grid.get_grid_occupancy().cpu().numpy(), ref_occ)
    # occupancy from sdf values (expf may differ by an ulp: skip voxels at the threshold)
    beta = g.uniform(5, 60, (1000, 1)).astype(np.float32)
    grid.set_grid_occupancy(torch.from_numpy(occ).cuda())
    grid.update_grid_occupancy_with_sdf_values(pick, torch.from_numpy(beta).cuda(), 1e-2, False)
    got = grid.get_grid_occupancy().cpu().numpy()
    pk = pick.cpu().numpy()
    w = np.array([OO.sdf_weight(ref[v], beta[i, 0], n, EXT) for i, v in enumerate(pk)])
    sure = np.abs(w - 1e-2) > 1e-5
    assert sure.mean() > 0.99 and np.array_equal(got[pk][sure], (w > 1e-2)[sure])
    assert 0 < got[pk].mean() < 1
    untouched = np.setdiff1d(np.arange(n ** 3), pk)
    assert np.array_equal(got[untouched], occ[untouched])
    assert grid.get_nr_occupied_voxels() == int(got.sum())
    assert grid.get_nr_voxels_in_roi() == int(roi.sum())
    with pytest.raises(Exception):
        OccupancyGrid(24, EXT)                                       # not a power of two


@pytest.mark.gpu
def test_hip_sphere_roi_and_check_occupancy():
    n = 32
    grid, occ, roi, vals, g = _grid(n, [1.0, 1.0, 1.0], seed=3)
    grid.init_sphere_roi(0.45, 0.02)
    r = grid.get_grid_roi().cpu().numpy()
    c = OO.grid_points(np.arange(n ** 3), n, [1.0] * 3, True)
    dist = np.linalg.norm(c, axis=1)
    assert r[dist < 0.43 - 0.87 / n].all() and not r[dist > 0.43].any()
    assert abs(r.sum() / n ** 3 - 4 / 3 * np.pi * 0.43 ** 3) < 0.05
    pts = g.uniform(-0.7, 0.7, (3000, 3)).astype(np.float32)
    o, v = grid.check_occupancy(torch.from_numpy(pts).cuda())
    ro, rv = OO.check_occupancy(pts, n, [1.0] * 3, vals, occ, r)
    assert np.array_equal(o.cpu().numpy()[:, 0], ro) and np.array_equal(v.cpu().numpy()[:, 0], rv)
    assert 0 < ro.mean() < 0.5


@pytest.mark.gpu
def test_hip_ray_marchers_vs_oracle():
    n = 16
    grid, occ, roi, vals, g = _grid(n, EXT, seed=1)
    o, d, t0, t1 = _rays(g, 300, EXT)
    cu = lambda x: torch.from_numpy(x).cuda()
    near, far = grid.get_rays_t_near_t_far(cu(o), cu(d), cu(t0), cu(t1))
    rn, rf = OO.rays_t_near_t_far(o, d, t0[:, 0], t1[:, 0], n, EXT, occ, roi)
    assert np.array_equal(near.cpu().numpy()[:, 0], rn) and np.array_equal(far.cpu().numpy()[:, 0], rf)
    assert (rf > rn).mean() > 0.3                                     # many rays cross a blob
    pack = grid.get_first_rays_sample_start_of_grid_occupied_regions(cu(o), cu(d), cu(t0), cu(t1))
    se, s3d, z = OO.first_sample(o, d, t0[:, 0], t1[:, 0], n, EXT, occ, roi)
    assert np.array_equal(pack.ray_start_end_idx.cpu().numpy(), se)
    has = se[:, 1] > se[:, 0]
    assert np.array_equal(pack.samples_3d.cpu().numpy()[has], s3d[has])
    assert np.array_equal(pack.samples_z.cpu().numpy()[has, 0], z[has])
    assert np.array_equal(pack.samples_dirs.cpu().numpy()[has], d[has])
    pts = cu(o.copy())
    new, within = grid.advance_ray_sample_to_next_occupied_voxel(cu(d), pts)
    rnew, rwithin = OO.advance_samples(d, o, n, EXT, occ, roi)
    assert new.data_ptr() == pts.data_ptr()                           # in place, as the reference
    assert np.array_equal(within.cpu().numpy()[:, 0], rwithin) and np.array_equal(new.cpu().numpy(), rnew)
    assert 0 < rwithin.mean() < 1


@pytest.mark.gpu
@pytest.mark.parametrize("jitter", [False, True])
def test_hip_samples_fg_in_grid_occupied_regions_vs_oracle(jitter):
    from volsurfs_amd.volsurfs import RaySampler, _Pcg32State
    n = 16
    grid, occ, roi, vals, g = _grid(n, EXT, seed=2)
    N, max_nr = 200, 24
    o, d, t0, t1 = _rays(g, N, EXT)
    cu = lambda x: torch.from_numpy(x).cuda()
    RaySampler.m_rng = _Pcg32State()
    pack = RaySampler.compute_samples_fg_in_grid_occupied_regions(
        cu(o), cu(d), cu(t0), cu(t1), 0.03, 2, max_nr, jitter, grid.get_nr_voxels_per_dim(),
        grid.get_grid_extent(), grid.get_grid_occupancy(), grid.get_grid_roi(), 0)
    se, z, s3d, dt = OO.sample_fg_occupied(o, d, t0[:, 0], t1[:, 0], 0.03, 2, max_nr, jitter, Pcg32(), n, EXT, occ, roi)
    counts = np.where(se[:, 0] >= 0, se[:, 1] - se[:, 0], 0)
    assert pack.is_compacted and pack.get_total_nr_samples() == counts.sum() > 0
    got_se = pack.ray_start_end_idx.cpu().numpy()
    assert np.array_equal(got_se[:, 1] - got_se[:, 0], counts)
    zs = np.concatenate([z[i, :counts[i]] for i in range(N)])
    ps = np.concatenate([s3d[i, :counts[i]] for i in range(N)])
    assert np.array_equal(pack.samples_z.cpu().numpy()[:, 0], zs)
    assert np.array_equal(pack.samples_3d.cpu().numpy(), ps)
    assert np.array_equal(pack.ray_max_dt.cpu().numpy()[counts > 0, 0], dt[counts > 0])
    # every sample sits in an occupied voxel of the region of interest
    inside, _ = grid.check_occupancy(pack.samples_3d)
    assert inside.all()


@pytest.mark.gpu
def test_hip_occupancy_edge_cases():
    """Empty inputs, an empty grid, a full grid, rays that start outside the grid."""
    from volsurfs_amd.volsurfs import OccupancyGrid, RaySampler
    n, ext = 8, [1.0, 1.0, 1.0]
    grid = OccupancyGrid(n, ext)
    z3 = torch.zeros(0, 3, device="cuda")
    z1 = torch.zeros(0, 1, device="cuda")
    o, v = grid.check_occupancy(z3)
    assert o.shape == (0, 1) and v.shape == (0, 1)
    near, far = grid.get_rays_t_near_t_far(z3, z3, z1, z1)
    assert near.shape == (0, 1)
    grid.update_grid_values(torch.zeros(0, dtype=torch.int32, device="cuda"), z1, 0.9)
    # a fresh grid is fully occupied with value 1 (src/OccupancyGrid.cu:131-147)
    assert grid.get_nr_occupied_voxels() == n ** 3 == grid.get_nr_voxels_in_roi()
    assert grid.get_grid_min_value() == grid.get_grid_max_value() == 1.0
    ro = torch.tensor([[-0.45, 0.0, 0.0], [-3.0, 0.0, 0.0]], device="cuda")
    rd = torch.tensor([[1.0, 0.0, 0.0], [1.0, 0.0, 0.0]], device="cuda")
    t0 = torch.zeros(2, 1, device="cuda")
    t1 = torch.tensor([[0.9], [0.5]], device="cuda")
    near, far = grid.get_rays_t_near_t_far(ro, rd, t0, t1)
    assert near[0].item() == 0.0 and abs(far[0].item() - 0.9) < 1e-6     # occupied all the way
    # (the second ray never reaches the grid: negative coordinates saturate to voxel 0, which is
    # occupied here — the reference's quirk, reproduced)
    pack = RaySampler.compute_samples_fg_in_grid_occupied_regions(
        ro[:1], rd[:1], t0[:1], t1[:1], 0.1, 1, 16, False, n, ext, grid.get_grid_occupancy(), grid.get_grid_roi(), 0)
    k = pack.get_total_nr_samples()
    assert 8 <= k <= 9
    assert torch.allclose(pack.samples_z[1:] - pack.samples_z[:-1], pack.ray_max_dt[0].expand(k - 1, 1), atol=2e-5)
    grid.set_grid_occupancy_empty()
    assert grid.get_nr_occupied_voxels() == 0
    near, far = grid.get_rays_t_near_t_far(ro, rd, t0, t1)
    assert torch.equal(near, t0) and torch.equal(far, t0)               # nothing occupied: both stay at t_entry
    pack = RaySampler.compute_samples_fg_in_grid_occupied_regions(
        ro, rd, t0, t1, 0.1, 1, 16, False, n, ext, grid.get_grid_occupancy(), grid.get_grid_roi(), 0)
    assert pack.is_empty() and pack.get_total_nr_samples() == 0
    first = grid.get_first_rays_sample_start_of_grid_occupied_regions(ro, rd, t0, t1)
    assert (first.ray_start_end_idx == 0).all()
    grid.set_grid_occupancy_full()
    assert grid.get_nr_occupied_voxels_in_roi() == n ** 3
