"""Mirror of the reference's pybind11 module `volsurfs` (src/PyBridge.cxx:19-139)
for the classes the K-shell path reaches: RaySamplesPacked, VolumeRendering,
RaySampler — same class / method names, argument order and return shapes, on the
HIP kernels of libvolsurfs_hip.so — plus OccupancyGrid and the samplers that only
the sibling methods (nerf / surf / offsets_surfs) use (not executed
by VolSurfs.render_rays, A12).

Differences a caller can observe: contract violations raise (the reference CHECKs
abort the process), kernels are asynchronous on the current stream.
"""
import torch

from . import _lib


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


class RaySamplesPacked:
    """include/volsurfs/RaySamplesPacked.cuh:7-80; ctor src/RaySamplesPacked.cu:13-48
    (everything initialised to -1)."""

    def __init__(self, nr_rays, max_nr_samples, first_sample_idx=0, values_dim=0):
        d = _dev()
        f = lambda *s: torch.full(s, -1.0, device=d)
        self.samples_idx = torch.arange(first_sample_idx, max_nr_samples + first_sample_idx,
                                        dtype=torch.int32, device=d).unsqueeze(1)
        self.samples_3d, self.samples_dirs = f(max_nr_samples, 3), f(max_nr_samples, 3)
        self.samples_z, self.samples_dt = f(max_nr_samples, 1), f(max_nr_samples, 1)
        self.ray_start_end_idx = torch.full((nr_rays, 2), -1, dtype=torch.int32, device=d)
        self.ray_o, self.ray_d = f(nr_rays, 3), f(nr_rays, 3)
        self.ray_enter, self.ray_exit, self.ray_max_dt = f(nr_rays, 1), f(nr_rays, 1), f(nr_rays, 1)
        self.samples_values = f(max_nr_samples, values_dim)
        self.has_samples_values, self.has_dt, self.is_compacted = False, False, True
        self._nr_rays, self._max_nr_samples, self._values_dim = nr_rays, max_nr_samples, values_dim

    def get_nr_rays(self):
        return self._nr_rays

    def get_max_nr_samples(self):
        return self._max_nr_samples

    def get_values_dim(self):
        return self._values_dim

    def get_ray_start_end_idx(self):
        return self.ray_start_end_idx

    def get_ray_o(self):
        return self.ray_o

    def get_ray_d(self):
        return self.ray_d

    def get_ray_enter(self):
        return self.ray_enter

    def get_ray_exit(self):
        return self.ray_exit

    def get_ray_max_dt(self):
        if not self.has_dt:
            raise ValueError("ray_max_dt is not set")          # RaySamplesPacked.cu:58-61
        return self.ray_max_dt

    def get_nr_samples_per_ray(self):
        return (self.ray_start_end_idx[:, 1] - self.ray_start_end_idx[:, 0]).unsqueeze(1)

    def get_total_nr_samples(self):
        if self._nr_rays == 0:
            return 0
        return int((self.ray_start_end_idx[:, 1] - self.ray_start_end_idx[:, 0]).sum().item())

    def is_empty(self):
        return self.get_total_nr_samples() == 0

    def get_samples_values(self):
        return self.samples_values

    def are_samples_values_set(self):
        return self.has_samples_values

    def set_samples_values(self, v):
        self.samples_values, self.has_samples_values, self._values_dim = v, True, v.shape[1]

    def remove_samples_values(self):
        self.samples_values = torch.full((self._max_nr_samples, 0), -1.0, device=_dev())
        self.has_samples_values, self._values_dim = False, 0

    def _ray_slice(self, t, ray_idx):
        a, b = self.ray_start_end_idx[ray_idx].tolist()
        return t[a:b]

    def get_ray_samples_idx(self, i):
        return self._ray_slice(self.samples_idx, i)

    def get_ray_samples_3d(self, i):
        return self._ray_slice(self.samples_3d, i)

    def get_ray_samples_dirs(self, i):
        return self._ray_slice(self.samples_dirs, i)

    def get_ray_samples_z(self, i):
        return self._ray_slice(self.samples_z, i)

    def get_ray_samples_dt(self, i):
        return self._ray_slice(self.samples_dt, i)

    def get_ray_samples_values(self, i):
        return self._ray_slice(self.samples_values, i)

    def copy(self):
        """RaySamplesPacked.cu:344-369 (deep clone)."""
        c = RaySamplesPacked.__new__(RaySamplesPacked)
        for k, v in self.__dict__.items():
            setattr(c, k, v.clone() if isinstance(v, torch.Tensor) else v)
        return c

    def update_dt(self, is_background):
        """RaySamplesPacked.cu:396-461."""
        if not self.is_compacted:
            raise _lib.VolsurfsHipError("RaySamplesPacked must be compacted before update_dt")
        _lib.call("vsa_packed_update_dt", self.ray_start_end_idx, self.ray_max_dt, self.ray_exit,
                  self.samples_z, bool(is_background), self.samples_dt, self._nr_rays,
                  _lib.stream_ptr())
        self.has_dt = True

    def compact_to_valid_samples(self):
        """src/RaySamplesPacked.cu:188-273: drop the unused slots of an uncompacted pack (one
        host read of the sample count, as in the reference)."""
        if self.is_compacted:
            raise _lib.VolsurfsHipError("RaySamplesPacked must not be compacted before compact_to_valid_samples")
        N, V = self._nr_rays, self._values_dim
        counts = (self.ray_start_end_idx[:, 1] - self.ray_start_end_idx[:, 0])
        total = int(counts.sum().item()) if N else 0
        c = RaySamplesPacked(N, total, 0, V)
        c.ray_o, c.ray_d = self.ray_o.clone(), self.ray_d.clone()
        c.ray_enter, c.ray_exit = self.ray_enter.clone(), self.ray_exit.clone()
        c.has_samples_values, c.has_dt = self.has_samples_values, self.has_dt
        c.ray_max_dt = self.ray_max_dt.clone()
        c.is_compacted = True
        if total == 0:
            return c
        out_start = (torch.cumsum(counts, 0) - counts).to(torch.int32).contiguous()
        _lib.call("vsa_pack_compact", self.ray_start_end_idx, out_start, self.samples_idx,
                  self.samples_3d, self.samples_dirs, self.samples_z, self.samples_dt,
                  self.samples_values if V else None, V, c.samples_idx, c.samples_3d, c.samples_dirs,
                  c.samples_z, c.samples_dt, c.samples_values if V else None, c.ray_start_end_idx, N,
                  _lib.stream_ptr())
        return c


def _check_pack(p, *tensors):
    if not p.is_compacted:
        raise _lib.VolsurfsHipError("RaySamplesPacked must be compacted")
    for t, cols in tensors:
        if t.dim() != 2 or t.shape[1] != cols or t.dtype != torch.float32 or not t.is_cuda:
            raise _lib.VolsurfsHipError(f"expected a CUDA float32 [S,{cols}] tensor, got {tuple(t.shape)} {t.dtype}")


class VolumeRendering:
    """include/volsurfs/VolumeRendering.cuh:69-97: static methods, same order of arguments."""
    # True (default) = compute what the reference computes, including its two slips:
    # integrate_with_weights_3d_backward reads the y lane of `values` for the z lane of grad_weights
    # (kernels/volsurfs/VolumeRenderingGPU.cuh:1021) and median_depth_over_rays falls back to the
    # reference's value when no sample crosses the threshold (:407).  Opt out with
    # `VolumeRendering.bug_compat = False` for the mathematically intended result.
    bug_compat = True

    @staticmethod
    def cumprod_one_minus_alpha_to_transmittance(pack, one_minus_alpha):
        _check_pack(pack, (one_minus_alpha, 1))
        N, S = pack.get_nr_rays(), one_minus_alpha.shape[0]
        T = torch.zeros(S, 1, device=one_minus_alpha.device)
        bgT = torch.ones(N, 1, device=one_minus_alpha.device)
        _lib.call("vsa_packed_cumprod_fwd", pack.ray_start_end_idx, one_minus_alpha.contiguous(), T,
                  bgT, N, _lib.stream_ptr())
        return T, bgT

    @staticmethod
    def cumprod_one_minus_alpha_to_transmittance_backward(grad_transmittance, grad_bg_transmittance,
                                                          pack, alpha, transmittance,
                                                          bg_transmittance, cumsumLV):
        S = alpha.shape[0]
        if grad_transmittance.shape[0] != S:
            raise _lib.VolsurfsHipError("grad_transmittance should have size nr_samples_total x 1")
        g = torch.zeros(S, 1, device=alpha.device)
        _lib.call("vsa_packed_cumprod_bwd", pack.ray_start_end_idx, grad_bg_transmittance.contiguous(),
                  alpha.contiguous(), bg_transmittance.contiguous(), cumsumLV.contiguous(), g,
                  pack.get_nr_rays(), _lib.stream_ptr())
        return g

    @staticmethod
    def cumsum_over_rays(pack, values, inverse):
        _check_pack(pack, (values, 1))
        out = torch.zeros_like(values)
        _lib.call("vsa_packed_cumsum", pack.ray_start_end_idx, values.contiguous(), bool(inverse),
                  out, pack.get_nr_rays(), _lib.stream_ptr())
        return out

    @staticmethod
    def _integrate(pack, values, weights, dim):
        _check_pack(pack, (values, dim), (weights, 1))
        out = torch.zeros(pack.get_nr_rays(), dim, device=values.device)
        _lib.call("vsa_packed_integrate_fwd", pack.ray_start_end_idx, values.contiguous(),
                  weights.contiguous(), out, pack.get_nr_rays(), dim, _lib.stream_ptr())
        return out

    @staticmethod
    def integrate_with_weights_1d(pack, values, weights):
        return VolumeRendering._integrate(pack, values, weights, 1)

    @staticmethod
    def integrate_with_weights_3d(pack, values, weights):
        return VolumeRendering._integrate(pack, values, weights, 3)

    @staticmethod
    def _integrate_backward(grad_result, pack, values, weights, dim):
        if grad_result.shape != (pack.get_nr_rays(), dim):
            raise _lib.VolsurfsHipError(f"grad_result should have size nr_rays x {dim}")
        gv, gw = torch.zeros_like(values), torch.zeros_like(weights)
        _lib.call("vsa_packed_integrate_bwd", pack.ray_start_end_idx, grad_result.contiguous(),
                  values.contiguous(), weights.contiguous(), gv, gw, pack.get_nr_rays(), dim,
                  bool(VolumeRendering.bug_compat), _lib.stream_ptr())
        return gv, gw

    @staticmethod
    def integrate_with_weights_1d_backward(grad_result, pack, values, weights, result):
        return VolumeRendering._integrate_backward(grad_result, pack, values, weights, 1)

    @staticmethod
    def integrate_with_weights_3d_backward(grad_result, pack, values, weights, result):
        return VolumeRendering._integrate_backward(grad_result, pack, values, weights, 3)

    @staticmethod
    def median_depth_over_rays(pack, weights, threshold):
        _check_pack(pack, (weights, 1))
        out = torch.zeros(pack.get_nr_rays(), 1, device=weights.device)
        _lib.call("vsa_packed_median_depth", pack.ray_start_end_idx, pack.samples_z,
                  weights.contiguous(), float(threshold), out, pack.get_nr_rays(),
                  bool(VolumeRendering.bug_compat), _lib.stream_ptr())
        return out

    @staticmethod
    def sdf2alpha(pack, samples_sdf, logistic_beta):
        """src/VolumeRendering.cu:178-229 (NeuS alpha; needs pack.samples_dt)."""
        _check_pack(pack, (samples_sdf, 1), (logistic_beta, 1))
        if not pack.has_dt:
            raise _lib.VolsurfsHipError("ray_samples_packed should have dt")
        alpha = torch.zeros_like(samples_sdf)
        _lib.call("vsa_packed_sdf2alpha", pack.ray_start_end_idx, pack.samples_dt,
                  samples_sdf.contiguous(), logistic_beta.contiguous(), alpha, pack.get_nr_rays(),
                  _lib.stream_ptr())
        return alpha

    @staticmethod
    def sum_over_rays(pack, samples_values):
        """src/VolumeRendering.cu:231-324 -> (sum_per_ray [N,D], sum_per_sample [S,D]), D in 1,2,3,32."""
        D = samples_values.shape[1]
        if D not in (1, 2, 3, 32):
            raise _lib.VolsurfsHipError(f"sum_over_rays: value dim {D} not supported (1, 2, 3, 32)")
        _check_pack(pack, (samples_values, D))
        per_ray = torch.zeros(pack.get_nr_rays(), D, device=samples_values.device)
        per_sample = torch.zeros_like(samples_values)
        _lib.call("vsa_packed_sum_over_rays", pack.ray_start_end_idx, samples_values.contiguous(),
                  per_ray, per_sample, pack.get_nr_rays(), D, _lib.stream_ptr())
        return per_ray, per_sample

    @staticmethod
    def sum_over_rays_backward(grad_per_ray, grad_per_sample, pack, samples_values):
        D = samples_values.shape[1]
        g = torch.zeros_like(samples_values)
        _lib.call("vsa_packed_sum_over_rays_bwd", pack.ray_start_end_idx, grad_per_ray.contiguous(),
                  grad_per_sample.contiguous(), g, pack.get_nr_rays(), D, _lib.stream_ptr())
        return g

    @staticmethod
    def compute_cdf(pack, samples_weights):
        """src/VolumeRendering.cu:418-465."""
        _check_pack(pack, (samples_weights, 1))
        cdf = torch.zeros_like(samples_weights)
        _lib.call("vsa_packed_compute_cdf", pack.ray_start_end_idx, samples_weights.contiguous(), cdf,
                  pack.get_nr_rays(), _lib.stream_ptr())
        return cdf

    m_rng = None   # static pcg32 m_rng of src/VolumeRendering.cu:19 (set below)

    @staticmethod
    def importance_sample(pack, samples_cdf, nr_importance_samples, jitter_samples):
        """src/VolumeRendering.cu:467-560: a new pack with nr_importance_samples per ray."""
        import ctypes
        if pack.is_empty():
            raise _lib.VolsurfsHipError("RaySamplesPacked should not be empty")
        _check_pack(pack, (samples_cdf, 1))
        N, n = pack.get_nr_rays(), int(nr_importance_samples)
        imp = RaySamplesPacked(N, N * n, pack.get_max_nr_samples(), pack.get_values_dim())
        imp.has_samples_values = imp.has_dt = imp.is_compacted = False
        rng = VolumeRendering.m_rng
        _lib.call("vsa_importance_sample", pack.ray_o, pack.ray_d, pack.ray_start_end_idx,
                  pack.samples_z, samples_cdf.contiguous(), n, bool(jitter_samples),
                  ctypes.c_uint64(rng.state), ctypes.c_uint64(rng.inc), imp.samples_3d,
                  imp.samples_dirs, imp.samples_z, imp.ray_start_end_idx, N, _lib.stream_ptr())
        if jitter_samples:
            rng.advance()
        imp = imp.compact_to_valid_samples()
        if imp.get_total_nr_samples() <= 0:
            raise _lib.VolsurfsHipError("nr_samples_imp should be > 0")
        return imp

    @staticmethod
    def combine_ray_samples_packets(pack_1, pack_2, min_dist_between_samples):
        """src/VolumeRendering.cu:562-670."""
        for p_ in (pack_1, pack_2):
            if not p_.is_compacted:
                raise _lib.VolsurfsHipError("RaySamplesPacked must be compacted before combine_ray_samples_packets")
        if pack_1.has_samples_values != pack_2.has_samples_values or \
                pack_1.get_values_dim() != pack_2.get_values_dim() or \
                pack_1.get_nr_rays() != pack_2.get_nr_rays():
            raise _lib.VolsurfsHipError("the two packs must agree in rays, values and values_dim")
        e1, e2 = pack_1.is_empty(), pack_2.is_empty()
        if e1 and e2:
            raise _lib.VolsurfsHipError("Both ray_samples_packed are empty")
        if e1:
            return pack_2
        if e2:
            return pack_1
        N, V = pack_1.get_nr_rays(), pack_1.get_values_dim()
        total = pack_1.get_total_nr_samples() + pack_2.get_total_nr_samples()
        c = RaySamplesPacked(N, total, 0, V)
        c.ray_o, c.ray_d = pack_1.ray_o.clone(), pack_1.ray_d.clone()
        c.ray_enter, c.ray_exit = pack_1.ray_enter.clone(), pack_1.ray_exit.clone()
        c.has_samples_values, c.has_dt = pack_1.has_samples_values, False
        c.ray_max_dt = pack_1.ray_max_dt.clone()
        c.is_compacted = False
        counts = (pack_1.get_nr_samples_per_ray() + pack_2.get_nr_samples_per_ray())[:, 0]
        out_start = (torch.cumsum(counts, 0) - counts).to(torch.int32).contiguous()
        v1 = pack_1.samples_values if V else None
        v2 = pack_2.samples_values if V else None
        _lib.call("vsa_combine_packs", pack_1.ray_start_end_idx, pack_1.samples_idx, pack_1.samples_3d,
                  pack_1.samples_dirs, pack_1.samples_z, v1, pack_2.ray_start_end_idx,
                  pack_2.samples_idx, pack_2.samples_3d, pack_2.samples_dirs, pack_2.samples_z, v2,
                  out_start, float(min_dist_between_samples), V, c.samples_idx, c.samples_3d,
                  c.samples_dirs, c.samples_z, c.samples_values if V else None, c.ray_start_end_idx,
                  N, _lib.stream_ptr())
        c = c.compact_to_valid_samples()
        if c.get_total_nr_samples() <= 0:
            raise _lib.VolsurfsHipError("total_nr_samples should be > 0")
        return c


class _Pcg32State:
    """Process-global RNG of the reference's RaySampler (static pcg32 m_rng,
    src/RaySampler.cu:19), advanced by 2^32 after every jittered call (:139-142)."""
    MULT, M64 = 0x5851f42d4c957f2d, (1 << 64) - 1

    def __init__(self):
        self.state, self.inc = 0x853c49e6748fea9b, 0xda3e39cb94b95bdb

    def advance(self, delta=1 << 32):
        cur_mult, cur_plus, acc_mult, acc_plus = self.MULT, self.inc, 1, 0
        while delta > 0:
            if delta & 1:
                acc_mult = (acc_mult * cur_mult) & self.M64
                acc_plus = (acc_plus * cur_mult + cur_plus) & self.M64
            cur_plus = ((cur_mult + 1) * cur_plus) & self.M64
            cur_mult = (cur_mult * cur_mult) & self.M64
            delta >>= 1
        self.state = (acc_mult * self.state + acc_plus) & self.M64


VolumeRendering.m_rng = _Pcg32State()


class RaySampler:
    """include/volsurfs/RaySampler.cuh:15-58."""
    m_rng = _Pcg32State()

    @staticmethod
    def compute_samples_fg(rays_o, rays_d, ray_t_entry, ray_t_exit, min_dist_between_samples,
                           min_nr_samples_per_ray, max_nr_samples_per_ray, jitter_samples, values_dim):
        """src/RaySampler.cu:158-240: uniform foreground samples, compacted."""
        import ctypes
        for t, n in ((rays_o, "rays_o"), (rays_d, "rays_d"), (ray_t_entry, "ray_t_entry"), (ray_t_exit, "ray_t_exit")):
            if t.dim() != 2:
                raise _lib.VolsurfsHipError(f"{n} should be 2-D, it has sizes {tuple(t.shape)}")
        N = rays_o.shape[0]
        p = RaySamplesPacked(N, N * int(max_nr_samples_per_ray), 0, int(values_dim))
        p.ray_o, p.ray_d = rays_o.clone().contiguous(), rays_d.clone().contiguous()
        p.ray_enter, p.ray_exit = ray_t_entry.clone().contiguous(), ray_t_exit.clone().contiguous()
        p.is_compacted = False
        rng = RaySampler.m_rng
        _lib.call("vsa_sample_fg", p.ray_o, p.ray_d, p.ray_enter, p.ray_exit,
                  float(min_dist_between_samples), int(min_nr_samples_per_ray),
                  int(max_nr_samples_per_ray), bool(jitter_samples), ctypes.c_uint64(rng.state),
                  ctypes.c_uint64(rng.inc), p.ray_max_dt, p.samples_idx, p.samples_3d, p.samples_dirs,
                  p.samples_z, p.ray_start_end_idx, N, _lib.stream_ptr())
        if jitter_samples:
            rng.advance()
        return p.compact_to_valid_samples()

    @staticmethod
    def compute_samples_bg(rays_o, rays_d, ray_t_start, ray_t_far, nr_samples_per_ray, jitter_samples):
        """src/RaySampler.cu:70-156."""
        import ctypes
        if rays_o.dim() != 2 or rays_d.dim() != 2 or ray_t_start.dim() != 2:
            raise _lib.VolsurfsHipError("rays_o/rays_d should be nr_rays x 3, ray_t_start nr_rays x 1")
        N = rays_o.shape[0]
        p = RaySamplesPacked(N, N * nr_samples_per_ray, 0, 0)
        p.ray_o, p.ray_d = rays_o.clone().contiguous(), rays_d.clone().contiguous()
        p.ray_enter = ray_t_start.clone().contiguous()
        p.ray_exit = torch.full((N, 1), float(ray_t_far), device=rays_o.device)
        p.is_compacted = True
        rng = RaySampler.m_rng
        _lib.call("vsa_sample_bg", p.ray_o, p.ray_d, p.ray_enter, float(ray_t_far),
                  int(nr_samples_per_ray), bool(jitter_samples), ctypes.c_uint64(rng.state),
                  ctypes.c_uint64(rng.inc), p.ray_max_dt, p.samples_3d, p.samples_dirs, p.samples_z,
                  p.ray_start_end_idx, N, _lib.stream_ptr())
        if jitter_samples:
            rng.advance()
        return p

    @staticmethod
    def contract_samples(pack):
        """src/RaySampler.cu:336-381: copy, contract (scale 2), update_dt(True)."""
        if not pack.is_compacted:
            raise _lib.VolsurfsHipError("RaySamplesPacked should be compacted before contract_samples")
        if pack.get_nr_rays() == 0:
            raise _lib.VolsurfsHipError("RaySamplesPacked must not be empty before contract_samples")
        c = pack.copy()
        _lib.call("vsa_contract_samples", pack.ray_o, pack.ray_start_end_idx, pack.samples_3d,
                  pack.samples_z, c.samples_3d, c.samples_z, pack.get_nr_rays(), _lib.stream_ptr())
        c.update_dt(True)
        return c

    @staticmethod
    def compute_samples_fg_in_grid_occupied_regions(rays_o, rays_d, ray_t_entry, ray_t_exit,
                                                    min_dist_between_samples, min_nr_samples_per_ray,
                                                    max_nr_samples_per_ray, jitter_samples,
                                                    nr_voxels_per_dim, grid_extent, grid_occupancy,
                                                    grid_roi, values_dim):
        """src/RaySampler.cu:243-334: foreground samples only where the occupancy grid is occupied
        (equidistant in occupied-space arc length), compacted."""
        import ctypes
        for t, n in ((rays_o, "rays_o"), (rays_d, "rays_d"), (ray_t_entry, "ray_t_entry"), (ray_t_exit, "ray_t_exit")):
            if t.dim() != 2:
                raise _lib.VolsurfsHipError(f"{n} should be 2-D, it has sizes {tuple(t.shape)}")
        N = rays_o.shape[0]
        p = RaySamplesPacked(N, N * int(max_nr_samples_per_ray), 0, int(values_dim))
        p.ray_o, p.ray_d = rays_o.clone().contiguous(), rays_d.clone().contiguous()
        p.ray_enter, p.ray_exit = ray_t_entry.clone().contiguous(), ray_t_exit.clone().contiguous()
        p.is_compacted = False
        ex = [float(v) for v in grid_extent]
        rng = RaySampler.m_rng
        _lib.call("vsa_sample_fg_occupied", p.ray_o, p.ray_d, p.ray_enter, p.ray_exit,
                  float(min_dist_between_samples), int(min_nr_samples_per_ray),
                  int(max_nr_samples_per_ray), bool(jitter_samples), ctypes.c_uint64(rng.state),
                  ctypes.c_uint64(rng.inc), int(nr_voxels_per_dim), ex[0], ex[1], ex[2],
                  _as_flags(grid_occupancy), _as_flags(grid_roi), p.ray_max_dt, p.samples_idx,
                  p.samples_3d, p.samples_dirs, p.samples_z, p.ray_start_end_idx, N, _lib.stream_ptr())
        if jitter_samples:
            rng.advance()
        return p.compact_to_valid_samples()

    @staticmethod
    def init_with_one_sample_per_ray(samples_3d, samples_dir):
        """src/RaySampler.cu:30-68: a pack with exactly one sample per ray (z = dt = 0)."""
        N = samples_3d.shape[0]
        p = RaySamplesPacked(N, N, 0, 1)
        p.samples_3d, p.samples_dirs = samples_3d.clone().contiguous(), samples_dir.clone().contiguous()
        p.samples_z, p.samples_dt = torch.zeros(N, 1, device=samples_3d.device), torch.zeros(N, 1, device=samples_3d.device)
        i = torch.arange(N, dtype=torch.int32, device=samples_3d.device)
        p.ray_start_end_idx = torch.stack([i, i + 1], 1).contiguous()
        return p

    @staticmethod
    def uncontract_samples(pack):
        """src/RaySampler.cu:383-428: copy, un-contract (scale 2), update_dt(True)."""
        if not pack.is_compacted:
            raise _lib.VolsurfsHipError("RaySamplesPacked should be compacted before uncontract_samples")
        if pack.is_empty():
            raise _lib.VolsurfsHipError("RaySamplesPacked must not be empty before uncontract_samples")
        c = pack.copy()
        _lib.call("vsa_uncontract_samples", pack.ray_o, pack.ray_start_end_idx, pack.samples_3d,
                  pack.samples_z, c.samples_3d, c.samples_z, pack.get_nr_rays(), _lib.stream_ptr())
        c.update_dt(True)
        return c


# ---- autograd glue, same classes as volume_rendering/volume_rendering_funcs.py:91-241
class CumprodOneMinusAlphaToTransmittanceFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pack, alpha):
        T, bgT = VolumeRendering.cumprod_one_minus_alpha_to_transmittance(pack, alpha)
        ctx.save_for_backward(alpha, T, bgT)
        ctx.pack = pack
        return T, bgT

    @staticmethod
    def backward(ctx, g_T, g_bgT):
        alpha, T, bgT = ctx.saved_tensors
        lv = g_T * T
        cumsum_lv = VolumeRendering.cumsum_over_rays(ctx.pack, lv, True)
        g = VolumeRendering.cumprod_one_minus_alpha_to_transmittance_backward(
            g_T, g_bgT, ctx.pack, alpha, T, bgT, cumsum_lv)
        ctx.pack = None
        return None, g


class IntegrateWithWeights3DFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pack, values, weights):
        out = VolumeRendering.integrate_with_weights_3d(pack, values, weights)
        ctx.save_for_backward(values, weights, out)
        ctx.pack = pack
        return out

    @staticmethod
    def backward(ctx, g):
        values, weights, out = ctx.saved_tensors
        gv, gw = VolumeRendering.integrate_with_weights_3d_backward(g.contiguous(), ctx.pack, values,
                                                                    weights, out)
        ctx.pack = None
        return None, gv, gw


class IntegrateWithWeights1DFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pack, values, weights):
        out = VolumeRendering.integrate_with_weights_1d(pack, values, weights)
        ctx.save_for_backward(values, weights, out)
        ctx.pack = pack
        return out

    @staticmethod
    def backward(ctx, g):
        values, weights, out = ctx.saved_tensors
        gv, gw = VolumeRendering.integrate_with_weights_1d_backward(g.contiguous(), ctx.pack, values,
                                                                    weights, out)
        ctx.pack = None
        return None, gv, gw


class SumOverRaysFunc(torch.autograd.Function):
    """volume_rendering_funcs.py:244-272."""

    @staticmethod
    def forward(ctx, ray_samples_packed, sample_values):
        per_ray, per_sample = VolumeRendering.sum_over_rays(ray_samples_packed, sample_values)
        ctx.save_for_backward(sample_values)
        ctx.ray_samples_packed = ray_samples_packed
        return per_ray, per_sample

    @staticmethod
    def backward(ctx, grad_per_ray, grad_per_sample):
        (sample_values,) = ctx.saved_tensors
        g = VolumeRendering.sum_over_rays_backward(grad_per_ray, grad_per_sample,
                                                   ctx.ray_samples_packed, sample_values)
        ctx.ray_samples_packed = None
        return None, g


def _as_flags(t):
    """torch.bool grid -> the same storage viewed as uint8 (what the kernels index)."""
    if t.dtype not in (torch.bool, torch.uint8) or not t.is_contiguous():
        raise _lib.VolsurfsHipError("occupancy / roi grids must be contiguous bool tensors")
    return t.view(torch.uint8)


class OccupancyGrid:
    """include/volsurfs/OccupancyGrid.cuh:9-68, src/OccupancyGrid.cu: nr_voxels_per_dim^3 voxels
    in Morton order, centred on the origin; values (density or sdf), occupancy flags and a
    region-of-interest mask, all resident on the device."""
    m_rng = _Pcg32State()

    def __init__(self, nr_voxels_per_dim, grid_extent):
        self.m_nr_voxels_per_dim = int(nr_voxels_per_dim)
        self.m_grid_extent = [float(v) for v in grid_extent]
        if len(self.m_grid_extent) != 3:
            raise _lib.VolsurfsHipError("grid_extent must have 3 components")
        self.m_grid_values = OccupancyGrid.make_grid_values(nr_voxels_per_dim)
        self.m_grid_occupancy = OccupancyGrid.make_grid_occupancy(nr_voxels_per_dim)
        self.m_grid_roi = OccupancyGrid.make_grid_occupancy(nr_voxels_per_dim)

    # ---- construction / accessors (src/OccupancyGrid.cu:18-165)
    @staticmethod
    def _check_n(n):
        n = int(n)
        if n < 2 or n > 1024 or n & (n - 1):
            raise _lib.VolsurfsHipError("nr_voxels_per_dim must be a power of two (Morton codes), 2..1024")
        return n

    @staticmethod
    def make_grid_values(nr_voxels_per_dim):
        n = OccupancyGrid._check_n(nr_voxels_per_dim)
        return torch.ones(n ** 3, dtype=torch.float32, device=_dev())

    @staticmethod
    def make_grid_occupancy(nr_voxels_per_dim):
        n = OccupancyGrid._check_n(nr_voxels_per_dim)
        return torch.ones(n ** 3, dtype=torch.bool, device=_dev())

    def get_grid_values(self):
        return self.m_grid_values

    def get_grid_occupancy(self):
        return self.m_grid_occupancy

    def get_grid_roi(self):
        return self.m_grid_roi

    def get_grid_occupancy_in_roi(self):
        return self.m_grid_occupancy.masked_select(self.m_grid_roi)

    def set_grid_values(self, grid_values):
        self.m_grid_values = grid_values

    def set_grid_occupancy(self, grid_occupancy):
        self.m_grid_occupancy = grid_occupancy

    def set_grid_occupancy_full(self):
        self.m_grid_occupancy.fill_(True)

    def set_grid_occupancy_empty(self):
        self.m_grid_occupancy.fill_(False)

    def get_nr_voxels(self):
        return self.m_nr_voxels_per_dim ** 3

    def get_nr_voxels_per_dim(self):
        return self.m_nr_voxels_per_dim

    def get_grid_extent(self):
        return list(self.m_grid_extent)

    def get_nr_voxels_in_roi(self):
        return int(self.m_grid_roi.sum().item())

    def get_nr_occupied_voxels(self):
        return int(self.m_grid_occupancy.sum().item())

    def get_nr_occupied_voxels_in_roi(self):
        return int(self.get_grid_occupancy_in_roi().sum().item())

    def get_grid_max_value(self):
        return float(self.m_grid_values.max().item())

    def get_grid_min_value(self):
        return float(self.m_grid_values.min().item())

    def get_grid_max_value_in_roi(self):
        return float(self.m_grid_values.masked_select(self.m_grid_roi).max().item())

    def get_grid_min_value_in_roi(self):
        return float(self.m_grid_values.masked_select(self.m_grid_roi).min().item())

    def init_sphere_roi(self, radius, padding):
        """:112-129: a voxel is in the region of interest when all its 8 vertices lie inside the
        sphere of `radius - padding`."""
        ll, _ = self.get_grid_lower_left_voxels_vertices()
        dist = self.get_grid_all_voxels_vertices(ll).reshape(-1, 3).norm(2, 1, True)
        self.m_grid_roi = (dist < (radius - padding)).reshape(-1, 8).all(-1)

    def get_grid_all_voxels_vertices(self, ll_vertices):
        """:167-186: [nr_voxels, 8, 3]."""
        n = self.m_nr_voxels_per_dim
        voxel = torch.tensor(self.m_grid_extent, dtype=torch.float32, device=ll_vertices.device) / n
        offs = torch.tensor([[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1], [1, 0, 0], [1, 0, 1], [1, 1, 0],
                             [1, 1, 1]], dtype=torch.float32, device=ll_vertices.device) * voxel
        return ll_vertices.view(-1, 1, 3) + offs.view(1, -1, 3)

    # ---- voxel positions (:188-310)
    def _points(self, indices, count, centre, jitter):
        import ctypes
        out = torch.empty(count, 3, dtype=torch.float32, device=_dev())
        rng = OccupancyGrid.m_rng
        e = self.m_grid_extent
        _lib.call("vsa_occ_grid_points", indices, int(count), self.m_nr_voxels_per_dim, e[0], e[1], e[2],
                  int(centre), bool(jitter), ctypes.c_uint64(rng.state), ctypes.c_uint64(rng.inc), out,
                  _lib.stream_ptr())
        if jitter:
            rng.advance()
        return out

    def get_grid_lower_left_voxels_vertices(self):
        n = self.get_nr_voxels()
        idx = torch.arange(n, dtype=torch.int32, device=_dev())
        return self._points(idx, n, 0, False), idx

    def get_grid_samples(self, jitter_samples):
        n = self.get_nr_voxels()
        idx = torch.arange(n, dtype=torch.int32, device=_dev())
        return self._points(idx, n, 1, jitter_samples), idx

    def get_random_grid_samples(self, nr_voxels_to_select, jitter_samples):
        idx = torch.randint(0, self.get_nr_voxels(), (int(nr_voxels_to_select),), dtype=torch.int32,
                            device=_dev())
        return self._points(idx, int(nr_voxels_to_select), 1, jitter_samples), idx

    def get_random_grid_samples_in_roi(self, nr_voxels_to_select, jitter_samples):
        roi_idx = torch.nonzero(self.m_grid_roi).to(torch.int32)
        pick = torch.randint(0, roi_idx.shape[0], (int(nr_voxels_to_select),), dtype=torch.int32,
                             device=_dev())
        idx = roi_idx.index_select(0, pick).squeeze(1).contiguous()
        return self._points(idx, int(nr_voxels_to_select), 1, jitter_samples), idx

    # ---- updates (:402-474)
    def update_grid_values(self, point_indices, values, decay):
        if values.dim() != 2 or point_indices.dim() != 1:
            raise _lib.VolsurfsHipError("values should be [nr_points,1] and point_indices [nr_points]")
        if decay > 1.0:
            raise _lib.VolsurfsHipError(f"decay should be <= 1.0 but it is {decay}")
        _lib.call("vsa_occ_update_values", point_indices.contiguous(), values.contiguous(),
                  point_indices.shape[0], float(decay), self.m_grid_values, _lib.stream_ptr())

    def update_grid_occupancy_with_density_values(self, point_indices, occupancy_tresh, check_neighbours):
        if point_indices.dim() != 1:
            raise _lib.VolsurfsHipError("point_indices should have dim 1")
        e = self.m_grid_extent
        _lib.call("vsa_occ_update_occupancy_density", point_indices.contiguous(), point_indices.shape[0],
                  self.m_nr_voxels_per_dim, e[0], e[1], e[2], float(occupancy_tresh),
                  bool(check_neighbours), self.m_grid_values, _as_flags(self.m_grid_occupancy),
                  _lib.stream_ptr())

    def update_grid_occupancy_with_sdf_values(self, point_indices, logistic_beta, occupancy_thresh,
                                              check_neighbours):
        if point_indices.dim() != 1:
            raise _lib.VolsurfsHipError("point_indices should have dim 1")
        e = self.m_grid_extent
        _lib.call("vsa_occ_update_occupancy_sdf", point_indices.contiguous(), logistic_beta.contiguous(),
                  point_indices.shape[0], self.m_nr_voxels_per_dim, e[0], e[1], e[2],
                  float(occupancy_thresh), self.m_grid_values, _as_flags(self.m_grid_occupancy),
                  _lib.stream_ptr())

    # ---- queries along points / rays (:312-400, 476-607)
    def check_occupancy(self, points):
        if points.dtype != torch.float32 or points.dim() != 2:
            raise _lib.VolsurfsHipError("positions should be float [nr_points,3]")
        P = points.shape[0]
        occ = torch.ones(P, 1, dtype=torch.bool, device=points.device)
        val = torch.ones(P, 1, dtype=torch.float32, device=points.device)
        e = self.m_grid_extent
        _lib.call("vsa_occ_check", points.contiguous(), P, self.m_nr_voxels_per_dim, e[0], e[1], e[2],
                  self.m_grid_values, _as_flags(self.m_grid_occupancy), _as_flags(self.m_grid_roi),
                  occ.view(torch.uint8), val, _lib.stream_ptr())
        return occ, val

    def get_rays_t_near_t_far(self, rays_o, rays_d, ray_t_entry, ray_t_exit):
        for t, n in ((rays_o, "rays_o"), (rays_d, "rays_d"), (ray_t_entry, "ray_t_entry"), (ray_t_exit, "ray_t_exit")):
            if t.dim() != 2:
                raise _lib.VolsurfsHipError(f"{n} should be 2-D, it has sizes {tuple(t.shape)}")
        N = rays_o.shape[0]
        near = torch.empty(N, 1, dtype=torch.float32, device=rays_o.device)
        far = torch.empty(N, 1, dtype=torch.float32, device=rays_o.device)
        e = self.m_grid_extent
        _lib.call("vsa_occ_rays_t_near_t_far", rays_o.contiguous(), rays_d.contiguous(),
                  ray_t_entry.contiguous(), ray_t_exit.contiguous(), N, self.m_nr_voxels_per_dim, e[0],
                  e[1], e[2], _as_flags(self.m_grid_occupancy), _as_flags(self.m_grid_roi), near, far,
                  _lib.stream_ptr())
        return near, far

    def get_first_rays_sample_start_of_grid_occupied_regions(self, rays_o, rays_d, ray_t_entry, ray_t_exit):
        N = rays_o.shape[0]
        p = RaySamplesPacked(N, N, 0, 1)
        e = self.m_grid_extent
        _lib.call("vsa_occ_first_sample", rays_o.contiguous(), rays_d.contiguous(),
                  ray_t_entry.contiguous(), ray_t_exit.contiguous(), N, self.m_nr_voxels_per_dim, e[0],
                  e[1], e[2], _as_flags(self.m_grid_occupancy), _as_flags(self.m_grid_roi), p.samples_3d,
                  p.samples_dirs, p.samples_z, p.samples_dt, p.ray_start_end_idx, _lib.stream_ptr())
        return p

    def advance_ray_sample_to_next_occupied_voxel(self, samples_dirs, samples_3d):
        """The reference updates samples_3d IN PLACE and returns it (:580-607)."""
        P = samples_3d.shape[0]
        if not samples_3d.is_contiguous():
            raise _lib.VolsurfsHipError("samples_3d must be contiguous (it is updated in place)")
        within = torch.ones(P, 1, dtype=torch.bool, device=samples_3d.device)
        e = self.m_grid_extent
        _lib.call("vsa_occ_advance_samples", samples_dirs.contiguous(), samples_3d, P,
                  self.m_nr_voxels_per_dim, e[0], e[1], e[2], _as_flags(self.m_grid_occupancy),
                  _as_flags(self.m_grid_roi), samples_3d, within.view(torch.uint8), _lib.stream_ptr())
        return samples_3d, within
