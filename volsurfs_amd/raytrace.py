"""RayTracer — raytracelib-shaped ray / K-shell intersector (SURVEY §8a A2, §8b
"Secondary boundaries").

Mirrors `raytracelib.RayTracer(list[TensorMesh])` / `.trace(rays_o, rays_d,
mesh_id=int) -> dict` as called at
/root/reference/volsurfs_py/methods/volsurfs.py:128 and :476-501, and adds
`trace_all` (all K shells in one launch, no host sync) used by the fused path.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib


class RayTracer:
    def __init__(self, tensor_meshes, leaf_size=None, node_format=None):
        """node_format: "q16" (default; binary 32-byte quantised nodes, vsa_trace_q / vsa_trace_q_fb) or
        "f32" (binary 64-byte fp32 nodes, vsa_trace); also selected by VSA_TRACE_NODES.  Both give
        identical hits; the quantised format assumes ray origins within ~60 mesh extents of the mesh
        (include/volsurfs_hip.h).  (The 4-wide nodes, the budgeted three-pass walk and the persistent-lane
        kernel of round 3 — bit-exact, measured slower: profiles/NOTEBOOK.md A9.4 — left the library in round 5.)"""
        self.node_format = node_format or os.environ.get("VSA_TRACE_NODES", "q16")
        if self.node_format not in ("q16", "f32"):
            raise _lib.VolsurfsHipError(f"unknown node_format {self.node_format}")
        # q16 only: launch order from the previous call's measured cost (vsa_trace_q_fb; identical hits)
        self.cost_feedback = os.environ.get("VSA_TRACE_FEEDBACK", "1") != "0"
        self._fb = None
        self.nr_meshes = len(tensor_meshes)
        if not 1 <= self.nr_meshes <= 16:
            raise _lib.VolsurfsHipError("RayTracer supports 1..16 meshes")
        L = _lib.lib()
        if leaf_size is None:
            leaf_size = int(os.environ.get("VSA_LEAF_SIZE", "4"))
        self._bvh, self._layout = [], []            # builder handles (kept for refit) and (node_base, nr_nodes, tri_base, nr_tris)
        self.mesh_tri_offset, self.mesh_nr_tris = [], []
        self.max_depth = 0
        node_base = tri_base = 0
        # The K shells' trees are independent host builds (binned SAH, csrc/bvh_build.cpp, one thread each): built side by
        # side on a thread pool — ctypes drops the GIL inside the call — 7 x 1.31 M triangles (configs[4]) take the time of one
        # shell (2.1 s) instead of 15 s.  Handles are collected in mesh order: the layout below is the sequential build's.
        arrays = [(np.ascontiguousarray(m.vertices.detach().cpu().numpy(), np.float32),
                   np.ascontiguousarray(m.faces.detach().cpu().numpy(), np.int32)) for m in tensor_meshes]

        def build_one(vf):
            v, f = vf
            h = ctypes.c_void_p()
            rc = L.vsa_bvh_build(v.ctypes.data_as(ctypes.c_void_p), f.ctypes.data_as(ctypes.c_void_p),
                                 ctypes.c_int(v.shape[0]), ctypes.c_int(f.shape[0]),
                                 ctypes.c_int(leaf_size), ctypes.byref(h))
            return rc, h
        if self.nr_meshes > 1 and sum(f.shape[0] for _, f in arrays) >= 100000 and os.environ.get("VSA_BVH_THREADS", "1") != "0":
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(self.nr_meshes, os.cpu_count() or 1)) as pool:
                built = list(pool.map(build_one, arrays))
        else:
            built = [build_one(vf) for vf in arrays]
        for rc, h in built:
            if rc != 0:
                raise _lib.VolsurfsHipError(f"vsa_bvh_build failed with status {rc}")
        for _, h in built:
            self._bvh.append(h)
            nn, nt, md = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            L.vsa_bvh_sizes(h, ctypes.byref(nn), ctypes.byref(nt), ctypes.byref(md))
            self._layout.append((node_base, nn.value, tri_base, nt.value))
            self.mesh_tri_offset.append(tri_base)
            self.mesh_nr_tris.append(nt.value)
            self.max_depth = max(self.max_depth, md.value)
            node_base += nn.value
            tri_base += nt.value
        dev = tensor_meshes[0].vertices.device
        self.device = dev
        nodes, qnodes, tris, frames = self._export()
        self.nodes = torch.from_numpy(nodes).to(dev)
        self.qnodes = torch.from_numpy(qnodes.view(np.int32)).to(dev)
        self._frames = (ctypes.c_float * (6 * self.nr_meshes))(*frames.tolist())
        self.tris = torch.from_numpy(tris).to(dev)
        # original face id of every leaf-ordered triangle slot (for uv tables etc.)
        self.slot_face_id = torch.from_numpy(tris[:, 3].copy().view(np.int32)).to(dev)
        self.roots = [lay[0] for lay in self._layout]
        self._roots = (ctypes.c_int32 * self.nr_meshes)(*self.roots)

    def _export(self):
        """Concatenated fp32 nodes [*,16], quantised nodes [*,8] u32, triangles [*,12] and the K
        quantisation frames [K*6] of the builder handles."""
        L = _lib.lib()
        n_nodes = sum(lay[1] for lay in self._layout)
        n_tris = sum(lay[3] for lay in self._layout)
        nodes = np.empty((n_nodes, 16), np.float32)
        qnodes = np.empty((n_nodes, 8), np.uint32)
        tris = np.empty((n_tris, 12), np.float32)
        frames = np.empty(6 * self.nr_meshes, np.float32)
        for i, (h, (nb, nn, tb, nt)) in enumerate(zip(self._bvh, self._layout)):
            rc = L.vsa_bvh_export(h, nodes[nb:nb + nn].ctypes.data_as(ctypes.c_void_p),
                                  tris[tb:tb + nt].ctypes.data_as(ctypes.c_void_p), ctypes.c_int(nb),
                                  ctypes.c_int(tb))
            rc2 = L.vsa_bvh_export_q(h, qnodes[nb:nb + nn].ctypes.data_as(ctypes.c_void_p),
                                     tris[tb:tb + nt].ctypes.data_as(ctypes.c_void_p), ctypes.c_int(nb),
                                     ctypes.c_int(tb), frames[6 * i:6 * i + 6].ctypes.data_as(ctypes.c_void_p))
            if rc != 0 or rc2 != 0:
                raise _lib.VolsurfsHipError(f"vsa_bvh_export failed with status {rc} / {rc2}")
        return nodes, qnodes, tris, frames

    def refit(self, tensor_meshes):
        """The shells' vertices moved (same faces): recompute triangle records and boxes bottom-up in the
        existing trees (vsa_bvh_refit) and overwrite the device arrays in place — no SAH rebuild, triangle
        slots (and with them `slot_face_id` and any per-slot uv table) unchanged.  Hits through the
        refitted trees are bit-identical to a rebuild's (tests/test_raytrace.py::test_refit_*).
        SURVEY §8f row 1; replaces re-running RayTracer(tensor_meshes) (volsurfs.py:82-128)."""
        if len(tensor_meshes) != self.nr_meshes:
            raise _lib.VolsurfsHipError("refit needs the meshes the tracer was built on")
        L = _lib.lib()
        for h, m in zip(self._bvh, tensor_meshes):
            v = np.ascontiguousarray(m.vertices.detach().cpu().numpy(), np.float32)
            rc = L.vsa_bvh_refit(h, v.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(v.shape[0]))
            if rc != 0:
                raise _lib.VolsurfsHipError(f"vsa_bvh_refit failed with status {rc} (vertex count changed?)")
        nodes, qnodes, tris, frames = self._export()
        self.nodes.copy_(torch.from_numpy(nodes))
        self.qnodes.copy_(torch.from_numpy(qnodes.view(np.int32)))
        self.tris.copy_(torch.from_numpy(tris))
        self._frames = (ctypes.c_float * (6 * self.nr_meshes))(*frames.tolist())
        self._fb = None        # the measured launch order belonged to the old geometry
        return self

    def __del__(self):
        try:
            L = _lib.lib()
            for h in getattr(self, "_bvh", []):
                L.vsa_bvh_destroy(h)
        except Exception:
            pass
        self._bvh = []

    def trace_all(self, rays_o, rays_d, t_min=0.0, out=None):
        """All K shells, one launch.  Returns hit_t [K,N] f32, hit_slot [K,N]
        i32 (global index into self.tris, -1 = miss), hit_uv [K,N,2] f32 (written into `out` = that triple when given)."""
        N = rays_o.shape[0]
        rays_o = _lib.check_f32(rays_o.contiguous(), N, 3)
        rays_d = _lib.check_f32(rays_d.contiguous(), N, 3)
        K = self.nr_meshes
        if out is not None:
            hit_t, hit_slot, hit_uv = out
            _lib.check_f32(hit_t, K, N)
            _lib.check_f32(hit_uv, K, N, 2)
            if hit_slot.dtype != torch.int32 or tuple(hit_slot.shape) != (K, N) or not hit_slot.is_contiguous():
                raise _lib.VolsurfsHipError("trace_all: out[1] must be a contiguous int32 [K, N] tensor")
        else:
            hit_t = torch.empty(K, N, device=rays_o.device)
            hit_slot = torch.empty(K, N, dtype=torch.int32, device=rays_o.device)
            hit_uv = torch.empty(K, N, 2, device=rays_o.device)
        # small batches (a training batch's few ten thousand random rays: ~2 waves per SIMD, each the maximum of 64
        # unrelated walks): narrow waves — fewer rays per wave, more waves (vsa_trace_q_narrow; same hits).  The launch-order
        # feedback has nothing to learn from random rays (profiles/NOTEBOOK.md round 5)
        rpw = self.narrow_rays_per_wave(N, K) if self.node_format == "q16" else 64
        if rpw < 64:
            _lib.call("vsa_trace_q_narrow", self.qnodes, self.tris, self._roots, self._frames, K,
                      self.max_depth, rays_o, rays_d, N, float(t_min), hit_t, hit_slot, hit_uv, int(rpw),
                      _lib.stream_ptr())
        elif self.node_format == "q16" and self.cost_feedback and self.max_depth < 48:
            if self._fb is None or self._fb[1] < N or self._fb[0].device != rays_o.device:   # grows only
                fn = _lib.lib().vsa_trace_feedback_bytes
                fn.restype = ctypes.c_longlong
                nbytes = int(fn(ctypes.c_int(N), ctypes.c_int(K)))
                if nbytes < 0:
                    raise _lib.VolsurfsHipError("vsa_trace_feedback_bytes failed")
                self._fb = [torch.zeros(nbytes, dtype=torch.uint8, device=rays_o.device), N, nbytes]
            # phase 2: the read / written halves alternate through a word in the buffer, flipped on the
            # device in front of every launch, so a captured graph alternates them on every replay too
            _lib.call("vsa_trace_q_fb", self.qnodes, self.tris, self._roots, self._frames, K,
                      self.max_depth, rays_o, rays_d, N, float(t_min), hit_t, hit_slot, hit_uv,
                      self._fb[0], ctypes.c_longlong(self._fb[2]), 2, _lib.stream_ptr())
        elif self.node_format == "q16":
            _lib.call("vsa_trace_q", self.qnodes, self.tris, self._roots, self._frames, K,
                      self.max_depth, rays_o, rays_d, N, float(t_min), hit_t, hit_slot, hit_uv,
                      _lib.stream_ptr())
        else:
            _lib.call("vsa_trace", self.nodes, self.tris, self._roots, K, self.max_depth, rays_o,
                      rays_d, N, float(t_min), hit_t, hit_slot, hit_uv, _lib.stream_ptr())
        return hit_t, hit_slot, hit_uv

    # narrow waves (vsa_trace_q_narrow: NARROW_RPW rays per 64-lane wave) below this many (ray, shell) walks; 0 = never.
    # Measured on MI355X (tools/trace_narrow_ab.py, profiles/r06/trace_narrow_ab.txt): 34 000 random rays x 5 shells
    # 0.108 ms -> 0.132 (worse), 8 000 rays 0.080 -> 0.069 at 8 per wave: off by default, the hits do not depend on it
    NARROW_BELOW = int(os.environ.get("VSA_TRACE_NARROW_BELOW", "0"))
    NARROW_RPW = int(os.environ.get("VSA_TRACE_RPW", "16"))

    @staticmethod
    def coop_config(chunk=16, lanes=24, max_waves=4096):
        """Process-wide setting of the traversal's cooperative finish (vsa_trace_coop_config: in launches of at most
        `max_waves` waves a wave whose last `lanes` rays are still walking finishes them together; lanes = 0: never).
        The hits do not depend on it."""
        _lib.call("vsa_trace_coop_config", int(chunk), int(lanes), ctypes.c_longlong(int(max_waves)))

    def narrow_rays_per_wave(self, N, K):
        return self.NARROW_RPW if N * K < self.NARROW_BELOW else 64

    def walk_stats(self, rays_o, rays_d, t_min=0.0):
        """{lane_visits, tri_tests, wave_trips, waves, max_wave_trips} of one traversal of these rays
        (vsa_trace_q_stats: the same walk with counters; q16 nodes).  Synchronises; measurement only."""
        N = rays_o.shape[0]
        rays_o = _lib.check_f32(rays_o.contiguous(), N, 3)
        rays_d = _lib.check_f32(rays_d.contiguous(), N, 3)
        st = torch.zeros(5, dtype=torch.int64, device=rays_o.device)
        _lib.call("vsa_trace_q_stats", self.qnodes, self.tris, self._roots, self._frames, self.nr_meshes,
                  self.max_depth, rays_o, rays_d, N, float(t_min), st, _lib.stream_ptr())
        v = st.cpu().tolist()
        return dict(zip(("lane_visits", "tri_tests", "wave_trips", "waves", "max_wave_trips"), v))

    def feedback_header(self):
        """{tag, n0, n1, n2} of the half the LAST cost-feedback launch wrote (tests, diagnostics)."""
        half = ((self._fb[2] - 256) // 2) & ~255
        ph = int(self._fb[0][2 * half:2 * half + 4].view(torch.int32).item()) & 1
        return self._fb[0][(ph ^ 1) * half:][:16].view(torch.int32).cpu().tolist()

    def trace(self, rays_o, rays_d, mesh_id=0, t_min=0.0):
        """raytracelib-shaped single-mesh trace (volsurfs.py:480-501)."""
        N = rays_o.shape[0]
        rays_o = _lib.check_f32(rays_o.contiguous(), N, 3)
        rays_d = _lib.check_f32(rays_d.contiguous(), N, 3)
        dev = rays_o.device
        hit_t = torch.empty(1, N, device=dev)
        hit_slot = torch.empty(1, N, dtype=torch.int32, device=dev)
        hit_uv = torch.empty(1, N, 2, device=dev)
        root = (ctypes.c_int32 * 1)(self.roots[mesh_id])
        st = _lib.stream_ptr()
        _lib.call("vsa_trace", self.nodes, self.tris, root, 1, self.max_depth, rays_o, rays_d, N,
                  float(t_min), hit_t, hit_slot, hit_uv, st)
        is_hit = torch.empty(N, dtype=torch.uint8, device=dev)
        tri_id = torch.empty(N, dtype=torch.int32, device=dev)
        pos = torch.empty(N, 3, device=dev)
        nrm = torch.empty(N, 3, device=dev)
        bary = torch.empty(N, 3, device=dev)
        _lib.call("vsa_hit_attributes", self.tris, rays_o, rays_d, hit_t, hit_slot, hit_uv, N,
                  is_hit, tri_id, pos, nrm, bary, st)
        is_hit = is_hit.bool()
        return {
            "any_hit": bool(is_hit.any().item()),  # host sync, as in the reference (volsurfs.py:481)
            "is_hit": is_hit,
            "triangles_id": tri_id.long(),
            "depth": hit_t[0],
            "positions": pos,
            "normals": nrm,
            "barycentric": bary,
        }
