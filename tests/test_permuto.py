"""PermutoHashEncoder (/root/reference/volsurfs_py/encodings/permutohash.py):
oracle properties on the CPU, HIP kernels vs the oracle on the GPU.  The wrapped package is an
un-vendored fork: the oracle restates the published algorithm, PARITY UNPINNED."""
import numpy as np
import pytest
import torch

from oracle import permuto as P


def _setup(L=6, cap=1 << 12, D=3, N=3000, seed=0):
    g = np.random.default_rng(seed)
    vals = g.standard_normal((L, cap, 2)).astype(np.float32)
    x = g.random((N, D)).astype(np.float32)
    scales = np.geomspace(1.0, 1e-3, L)
    shift = (g.standard_normal((L, D)) * 10).astype(np.float32)
    return vals, x, scales, shift


@pytest.mark.parametrize("D", [2, 3, 4])
def test_oracle_simplex_properties(D):
    vals, x, scales, shift = _setup(D=D)
    sf = P.scale_factors(scales, D)
    for l in (0, 3, 5):
        rem0, rank, bary = P.simplex(x, shift[l], sf[l])
        assert (rem0.sum(1) == 0).all()                                  # on the hyperplane sum = 0
        assert (np.sort(rank, 1) == np.arange(D + 1)).all()              # a permutation
        w = bary[:, :D + 1]
        assert w.min() > -1e-4 and np.abs(w.sum(1) - 1).max() < 1e-5     # barycentric coordinates
        # the D+1 vertices of a simplex are distinct lattice points
        keys = np.stack([np.stack([rem0[:, i] + k - np.where(rank[:, i] > D - k, D + 1, 0)
                                   for i in range(D)], 1) for k in range(D + 1)], 1)
        for a in range(D + 1):
            for b in range(a + 1, D + 1):
                assert (keys[:, a] != keys[:, b]).any(1).all()


def test_oracle_is_continuous_and_interpolates():
    """Across simplex boundaries the encoding is continuous (the property the lattice is used
    for); moving a point by 1e-4 of the finest cell changes the features by a matching amount."""
    vals, x, scales, shift = _setup(L=4, N=2000)
    a = P.encode(vals, x, scales, shift)
    b = P.encode(vals, x + np.float32(1e-6), scales, shift)
    assert np.abs(a - b).max() < 0.05
    # window scales a level linearly; zero window removes it
    w = np.array([1.0, 0.5, 0.0, 1.0], np.float32)
    c = P.encode(vals, x, scales, shift, window=w)
    np.testing.assert_allclose(c[:, 2:4], 0.5 * a[:, 2:4], rtol=1e-6, atol=1e-7)
    assert (c[:, 4:6] == 0).all() and np.array_equal(c[:, :2], a[:, :2])


def test_oracle_wrapper_shapes_and_window():
    vals, x, scales, shift = _setup(L=24, cap=1 << 10, N=50)
    pts = (x - 0.5) * 1.5
    enc, oob = P.permuto_hash_encoder(vals, pts, shift, bb_sides=2.0)
    assert enc.shape == (50, 24 * 2 + 3 - 1) and not oob.any()                 # permutohash.py:38-41, 91-92
    np.testing.assert_allclose(enc[:, 48:], ((pts / 1.0 + 1) / 2)[:, :2], rtol=1e-6)
    _, oob = P.permuto_hash_encoder(vals, pts * 3, shift, bb_sides=2.0)
    assert oob.any()
    w = P.coarse2fine_window(1.0, 24)
    assert (w == 1).all()
    w = P.coarse2fine_window(0.3, 24)
    assert (w[:7] == 1).all() and 0 < w[7] < 1 and (w[8:] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("D", [2, 3, 4])
def test_hip_permuto_encoding_matches_oracle_values_and_gradients(D):
    from volsurfs_amd.encodings import PermutoEncoding
    vals, x, scales, shift = _setup(D=D, N=5000, seed=D)
    enc = PermutoEncoding(D, vals.shape[1], vals.shape[0], 2, scales)
    with torch.no_grad():
        enc.lattice_values.copy_(torch.from_numpy(vals))
        enc.random_shift_per_level.copy_(torch.from_numpy(shift))
    xs = torch.from_numpy(x).cuda()
    win = torch.tensor([1.0, 1.0, 0.75, 0.5, 0.25, 0.0])
    for window, wnp in ((None, None), (win, win.numpy())):
        out = enc(xs, window)
        ref = P.encode(vals, x, scales, shift, wnp)
        assert out.shape == ref.shape
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=0, atol=1e-6)
        g = np.random.default_rng(1).standard_normal(ref.shape).astype(np.float32)
        enc.lattice_values.grad = None
        out.backward(torch.from_numpy(g).cuda())
        gref = P.encode_backward(g, x, scales, shift, vals.shape[1], wnp)
        got = enc.lattice_values.grad.cpu().numpy()
        np.testing.assert_allclose(got, gref, rtol=1e-4, atol=1e-5)
        assert (got != 0).any()
    assert np.array_equal(enc(xs, None).detach().cpu().numpy(), enc(xs, None).detach().cpu().numpy())


@pytest.mark.gpu
def test_hip_permuto_backward_coarse_levels_through_lds_and_fine_levels_direct():
    """24 levels 1 .. 1e-3 at capacity 2^14, 20 k points: the first ten levels take the LDS-table
    kernel (several chunks per level; at this capacity the finer of them overflow the 4096 slots and
    fall back to memory atomics), the rest the plain kernel — all against the oracle."""
    from volsurfs_amd.encodings import PermutoEncoding
    g = np.random.default_rng(7)
    L, C, N = 24, 1 << 14, 20000
    scales = np.geomspace(1.0, 1e-3, L)
    vals = g.standard_normal((L, C, 2)).astype(np.float32)
    shift = (g.standard_normal((L, 3)) * 10).astype(np.float32)
    x = (g.random((N, 3), dtype=np.float32) - 0.5) * 0.8
    enc = PermutoEncoding(3, C, L, 2, scales)
    with torch.no_grad():
        enc.lattice_values.copy_(torch.from_numpy(vals))
        enc.random_shift_per_level.copy_(torch.from_numpy(shift))
    out = enc(torch.from_numpy(x).cuda(), None)
    go = g.standard_normal(out.shape).astype(np.float32)
    go[::7] = 0.0                                     # rows without gradient are skipped
    out.backward(torch.from_numpy(go).cuda())
    gref = P.encode_backward(go, x, scales, shift, C, None)
    got = enc.lattice_values.grad.cpu().numpy()
    np.testing.assert_allclose(got, gref, rtol=2e-4, atol=2e-4 * np.abs(gref).max())
    for l in (0, 5, 9, 10, 23):
        assert np.abs(got[l]).max() > 0


@pytest.mark.gpu
def test_hip_permuto_hash_encoder_reference_configuration():
    """The configuration the reference instantiates (permutohash.py:12-37 via get_encoder):
    24 levels, capacity 2^18, sigma 1 .. 1e-4, random shift, points concatenated, last channel
    dropped; coarse-to-fine window from iter_nr."""
    from volsurfs_amd.encodings import get_encoder
    e = get_encoder("permutohash", input_dim=3, nr_levels=24, nr_iters_for_c2f=1000, bb_sides=2.0)
    assert e.output_dim == 50 and e.encoder.output_dims() == 51
    assert tuple(e.encoder.lattice_values.shape) == (24, 1 << 18, 2)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        e.encoder.lattice_values.copy_(torch.randn(24, 1 << 18, 2, generator=g))
    pts = (torch.rand(4000, 3, generator=g) - 0.5) * 1.9
    pts[0] = torch.tensor([1.5, 0.0, 0.0])
    vals = e.encoder.lattice_values.detach().cpu().numpy()
    shift = e.encoder.random_shift_per_level.cpu().numpy()
    for it, t in ((None, 1.0), (300, 0.3 + 0.7 * 0.3), (5000, 1.0)):
        enc, oob = e(pts.cuda(), iter_nr=it)
        ref, oob_ref = P.permuto_hash_encoder(vals, pts.numpy(), shift, 2.0,
                                              window=P.coarse2fine_window(t, 24))
        assert enc.shape == (4000, 50)
        assert np.array_equal(oob.cpu().numpy(), oob_ref) and oob_ref[0]
        # sigma = 1e-4 puts coordinates at ~1e4: fp32 products differ by ulps of that scale only
        np.testing.assert_allclose(enc.detach().cpu().numpy(), ref, rtol=0, atol=2e-5)
    # gradient reaches the lattice values; state_dict round trip keeps the shift (and the plan)
    enc, _ = e(pts.cuda())
    enc.sum().backward()
    assert e.encoder.lattice_values.grad.abs().sum() > 0
    e2 = get_encoder("permutohash", input_dim=3, nr_levels=24, nr_iters_for_c2f=1000, bb_sides=2.0)
    e2.load_state_dict(e.state_dict())
    assert torch.equal(e2(pts.cuda())[0], e(pts.cuda())[0])


@pytest.mark.gpu
@pytest.mark.parametrize("sizes", [[3000, 0, 4100, 700, 2500], [200] * 9 + [0, 33]])
def test_hip_permuto_grouped_launch_equals_the_encoders_one_by_one(sizes):
    """encodings.permuto_hash_encode_grouped (vsa_permuto_encode_fwd_grouped / _bwd_grouped: group =
    blockIdx.z, plans from a device array; 11 groups = two launches) against PermutoHashEncoder.forward
    per segment: features bit for bit, lattice gradients to 1e-5 of their scale (float atomics, and the
    coarse levels' fixed-point tables, sum in a different order).  Every encoder has its own shift."""
    from volsurfs_amd.encodings import get_encoder, permuto_hash_encode_grouped, permuto_hash_encoders_groupable
    g = torch.Generator().manual_seed(4)
    encs = []
    for i in range(len(sizes)):
        e = get_encoder("permutohash", input_dim=3, nr_levels=24, nr_iters_for_c2f=1000, bb_sides=2.0)
        with torch.no_grad():
            e.encoder.lattice_values.copy_(torch.randn(24, 1 << 18, 2, generator=g))
            e.encoder.random_shift_per_level.copy_(torch.randn(24, 3, generator=g) * 10.0)
        encs.append(e)
    pts = ((torch.rand(sum(sizes), 3, generator=g) - 0.5) * 1.2).cuda()
    gy = torch.randn(sum(sizes), 50, generator=g).cuda()
    assert permuto_hash_encoders_groupable(encs, pts)
    for it in (None, 300):
        out = permuto_hash_encode_grouped(encs, pts, sizes, iter_nr=it)
        out.backward(gy)
        got = [e.encoder.lattice_values.grad.clone() for e in encs]
        for e in encs:
            e.encoder.lattice_values.grad = None
        refs, a = [], 0
        for e, n in zip(encs, sizes):
            if n:
                f, _ = e(pts[a:a + n], iter_nr=it)
                f.backward(gy[a:a + n])
                refs.append(f.detach())
            a += n
        assert torch.equal(out.detach(), torch.cat(refs, 0))
        for e, n, ga in zip(encs, sizes, got):
            gb = e.encoder.lattice_values.grad
            if n == 0:
                assert float(ga.abs().max()) == 0.0 and gb is None
                continue
            scale = float(gb.abs().max())
            np.testing.assert_allclose(ga.cpu().numpy(), gb.cpu().numpy(), rtol=1e-4, atol=1e-5 * scale)
            e.encoder.lattice_values.grad = None
