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
        """node_format: "q16" (default; binary 32-byte quantised nodes, vsa_trace_q), "q16x4" (the
        same tree collapsed to 4-wide 64-byte nodes, vsa_trace_q4: half the dependent node fetches,
        measured 8 % SLOWER — every visit tests four boxes where the binary walk prunes after two — kept
        as a tested option) or "f32" (binary 64-byte fp32 nodes, vsa_trace); also selected by
        VSA_TRACE_NODES.  All give identical hits; the quantised formats assume ray origins within
        ~60 mesh extents of the mesh (include/volsurfs_hip.h)."""
        self.node_format = node_format or os.environ.get("VSA_TRACE_NODES", "q16")
        if self.node_format not in ("q16x4", "q16", "f32"):
            raise _lib.VolsurfsHipError(f"unknown node_format {self.node_format}")
        # q16 only: trips of the walk loop after which a wave hands its unfinished rays' subtrees to a
        # second pass (vsa_trace_q_budgeted; identical results).  0 = the one-pass kernel, the default:
        # measured at 800x800, K = 5: one pass 0.258 ms; budget 96 / 64 / 48 / 24: 0.256 / 0.287 / 0.354 /
        # 0.818 ms (profiles/NOTEBOOK.md A9.4: pass A loses its tail, 0.185 ms at 48, but a ray that has no hit yet
        # hands over subtrees the one-pass walk would have pruned after its first hit)
        self.round_budget = int(os.environ.get("VSA_TRACE_BUDGET", "0"))
        self._ws = None
        self.workspace_bytes = None
        # q16 only: launch order from the previous call's measured cost (vsa_trace_q_fb; identical hits)
        self.cost_feedback = os.environ.get("VSA_TRACE_FEEDBACK", "1") != "0"
        self._fb = None
        self.nr_meshes = len(tensor_meshes)
        if not 1 <= self.nr_meshes <= 16:
            raise _lib.VolsurfsHipError("RayTracer supports 1..16 meshes")
        L = _lib.lib()
        nodes_all, tris_all, roots, qnodes_all, frames = [], [], [], [], []
        q4_all, roots4, node4_base, self.max_depth4 = [], [], 0, 0
        self.mesh_tri_offset, self.mesh_nr_tris = [], []
        self.max_depth = 0
        node_base = tri_base = 0
        for m in tensor_meshes:
            if leaf_size is None:
                leaf_size = int(os.environ.get("VSA_LEAF_SIZE", "4"))
            v = np.ascontiguousarray(m.vertices.detach().cpu().numpy(), np.float32)
            f = np.ascontiguousarray(m.faces.detach().cpu().numpy(), np.int32)
            h = ctypes.c_void_p()
            rc = L.vsa_bvh_build(v.ctypes.data_as(ctypes.c_void_p), f.ctypes.data_as(ctypes.c_void_p),
                                 ctypes.c_int(v.shape[0]), ctypes.c_int(f.shape[0]),
                                 ctypes.c_int(leaf_size), ctypes.byref(h))
            if rc != 0:
                raise _lib.VolsurfsHipError(f"vsa_bvh_build failed with status {rc}")
            nn, nt, md = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            L.vsa_bvh_sizes(h, ctypes.byref(nn), ctypes.byref(nt), ctypes.byref(md))
            nodes = np.empty((nn.value, 16), np.float32)
            tris = np.empty((nt.value, 12), np.float32)
            rc = L.vsa_bvh_export(h, nodes.ctypes.data_as(ctypes.c_void_p),
                                  tris.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(node_base),
                                  ctypes.c_int(tri_base))
            qnodes = np.empty((nn.value, 8), np.uint32)
            frame = np.empty(6, np.float32)
            rc2 = L.vsa_bvh_export_q(h, qnodes.ctypes.data_as(ctypes.c_void_p),
                                     tris.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(node_base),
                                     ctypes.c_int(tri_base), frame.ctypes.data_as(ctypes.c_void_p))
            if self.node_format == "q16x4":      # the 4-wide collapse (host time + a second node array) only when asked for
                q4 = np.empty((nn.value, 16), np.uint32)
                n4, d4 = ctypes.c_int(), ctypes.c_int()
                rc3 = L.vsa_bvh_export_q4(h, q4.ctypes.data_as(ctypes.c_void_p), tris.ctypes.data_as(ctypes.c_void_p),
                                          ctypes.c_int(node4_base), ctypes.c_int(tri_base),
                                          frame.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n4), ctypes.byref(d4))
                rc = rc or rc3
                q4_all.append(q4[:n4.value])
                roots4.append(node4_base)
                node4_base += n4.value
                self.max_depth4 = max(self.max_depth4, d4.value)
            L.vsa_bvh_destroy(h)
            if rc != 0 or rc2 != 0:
                raise _lib.VolsurfsHipError(f"vsa_bvh_export failed with status {rc} / {rc2}")
            qnodes_all.append(qnodes)
            frames.append(frame)
            roots.append(node_base)
            self.mesh_tri_offset.append(tri_base)
            self.mesh_nr_tris.append(nt.value)
            self.max_depth = max(self.max_depth, md.value)
            node_base += nn.value
            tri_base += nt.value
            nodes_all.append(nodes)
            tris_all.append(tris)
        dev = tensor_meshes[0].vertices.device
        self.device = dev
        self.nodes = torch.from_numpy(np.concatenate(nodes_all, 0)).to(dev)
        self.qnodes = torch.from_numpy(np.concatenate(qnodes_all, 0).view(np.int32)).to(dev)
        self.qnodes4 = torch.from_numpy(np.concatenate(q4_all, 0).view(np.int32)).to(dev) if q4_all else None
        self._roots4 = (ctypes.c_int32 * self.nr_meshes)(*roots4) if q4_all else None
        self._frames = (ctypes.c_float * (6 * self.nr_meshes))(*np.concatenate(frames).tolist())
        tris_np = np.concatenate(tris_all, 0)
        self.tris = torch.from_numpy(tris_np).to(dev)
        # original face id of every leaf-ordered triangle slot (for uv tables etc.)
        self.slot_face_id = torch.from_numpy(tris_np[:, 3].copy().view(np.int32)).to(dev)
        self._roots = (ctypes.c_int32 * self.nr_meshes)(*roots)
        self.roots = roots

    def _workspace(self, N, device):
        """Scratch of vsa_trace_q_budgeted for N rays (kept between calls; contents are not)."""
        if self._ws is None or self._ws[1] != N or self._ws[0].device != device:
            fn = _lib.lib().vsa_trace_q_workspace_bytes
            fn.restype = ctypes.c_longlong
            nbytes = int(fn(ctypes.c_int(N), ctypes.c_int(self.nr_meshes)))
            if nbytes < 0:
                raise _lib.VolsurfsHipError("vsa_trace_q_workspace_bytes failed")
            if self.workspace_bytes is not None:     # tests: a workspace too small for the hand-overs
                nbytes = int(self.workspace_bytes)
            self._ws = (torch.empty(nbytes, dtype=torch.uint8, device=device), N, nbytes)
        return self._ws[0], self._ws[2]

    def trace_all(self, rays_o, rays_d, t_min=0.0):
        """All K shells, one launch.  Returns hit_t [K,N] f32, hit_slot [K,N]
        i32 (global index into self.tris, -1 = miss), hit_uv [K,N,2] f32."""
        N = rays_o.shape[0]
        rays_o = _lib.check_f32(rays_o.contiguous(), N, 3)
        rays_d = _lib.check_f32(rays_d.contiguous(), N, 3)
        K = self.nr_meshes
        hit_t = torch.empty(K, N, device=rays_o.device)
        hit_slot = torch.empty(K, N, dtype=torch.int32, device=rays_o.device)
        hit_uv = torch.empty(K, N, 2, device=rays_o.device)
        if self.node_format == "q16x4":
            _lib.call("vsa_trace_q4", self.qnodes4, self.tris, self._roots4, self._frames, K,
                      self.max_depth4, rays_o, rays_d, N, float(t_min), hit_t, hit_slot, hit_uv,
                      _lib.stream_ptr())
        elif self.node_format == "q16" and self.round_budget > 0 and self.max_depth < 48:
            ws, ws_bytes = self._workspace(N, rays_o.device)
            _lib.call("vsa_trace_q_budgeted", self.qnodes, self.tris, self._roots, self._frames, K,
                      self.max_depth, rays_o, rays_d, N, float(t_min), hit_t, hit_slot, hit_uv,
                      self.round_budget, ws, ctypes.c_longlong(ws_bytes), _lib.stream_ptr())
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
