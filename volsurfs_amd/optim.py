"""FusedAdam — the optimiser of the reference's training loop on one HIP kernel.

The reference builds `apex.optimizers.FusedAdam(param_groups, amsgrad=False, betas=(0.9, 0.99),
eps=1e-15, weight_decay=0.0, lr=...)` (/root/reference/volsurfs_py/methods/base_method.py:87-94)
and calls `.step()` once per iteration (trainer.py:278) — the same update as
`torch.optim.Adam`, against which tests/test_optim.py pins this class at 1e-6.

`vsa_adam_step` (csrc/adam.hip) updates ALL tensors of a parameter group in one launch and
fuses two passes the surrounding loop otherwise makes over the same memory: the refresh of a
parameter's f16 compute copy (`half_copies`) and the zeroing of its gradient for the next
iteration.  Gradients therefore live in persistent buffers (`p.grad` is allocated once and never
set to None): `zero_grad()` is free after a `step()`.
"""
import ctypes
import os
import weakref

import torch

from . import _lib


class AdamTensor(ctypes.Structure):
    """Mirror of `vsa_adam_tensor` (include/volsurfs_hip.h)."""
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("param_f16", ctypes.c_void_p), ("n", ctypes.c_int64)]


_NO_CACHE = __import__("os").environ.get("VSA_NO_DESC_CACHE", "0") == "1"     # A/B switch


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.0,
                 amsgrad=False, half_copies=None):
        if weight_decay != 0.0 or amsgrad:
            raise _lib.VolsurfsHipError("FusedAdam: weight_decay / amsgrad are not used by the "
                                        "reference (base_method.py:87-94) and not built")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0.0, amsgrad=False))
        self._half = {id(p): h for p, h in (half_copies or {}).items()}
        self._plans = {}          # group index -> (key, descriptor tensor, chunk tensor, nr_chunks, tensor objects)
        self._plan_age = {}
        self._grads_clean = False
        for g in self.param_groups:
            g.setdefault("step", 0)
            for p in g["params"]:
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise _lib.VolsurfsHipError("FusedAdam: contiguous CUDA float32 parameters only")
                # any gradient autograd accumulates makes the buffers dirty again (kernels that
                # accumulate into .grad directly call mark_grads_dirty themselves)
                p.register_post_accumulate_grad_hook(self._dirty_hook)
                p._vsa_optimizer = weakref.ref(self)     # kernels that add into .grad themselves (accumulate_into_grad)

    def _ensure(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["exp_avg"] = torch.zeros_like(p)
            st["exp_avg_sq"] = torch.zeros_like(p)
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        return st

    def _plan(self, gi, group):
        # Fast path: the tensors behind the cached descriptors are still the same OBJECTS (a
        # parameter's .grad or a state tensor that gets replaced is a new object) and the parameters
        # sit where they sat; the full pointer-level check below runs every 64th step and whenever one of these differs.  (Four data_ptr() calls
        # per parameter and step were 0.2 ms of host time with the 90 tensors of BASELINE configs[2].)
        cur = self._plans.get(gi)
        if cur is not None and not _NO_CACHE:
            self._plan_age[gi] = self._plan_age.get(gi, 0) + 1
            if self._plan_age[gi] % 64:
                state = self.state
                for p, g, ea, eas, ptr in cur[4]:
                    st = state.get(p)
                    if p.grad is not g or st is None or st.get("exp_avg") is not ea or st.get("exp_avg_sq") is not eas \
                            or not p.requires_grad or p.data_ptr() != ptr:
                        break
                else:
                    if len(cur[4]) == sum(1 for p in group["params"] if p.requires_grad):
                        return cur
        ps = [p for p in group["params"] if p.requires_grad]
        sts = [self._ensure(p) for p in ps]
        key = tuple((p.data_ptr(), p.grad.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr())
                    for p, s in zip(ps, sts))
        if cur is not None and cur[0] == key:
            return cur
        chunk = _lib.lib().vsa_adam_chunk_elems()
        arr = (AdamTensor * max(len(ps), 1))()
        chunks = []
        for i, (p, s) in enumerate(zip(ps, sts)):
            if not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
                raise _lib.VolsurfsHipError("FusedAdam: gradients must be contiguous float32")
            h = self._half.get(id(p))
            if h is not None and (h.shape != p.shape or h.dtype != torch.float16 or not h.is_contiguous()):
                raise _lib.VolsurfsHipError("FusedAdam: an f16 copy must match its parameter's shape")
            arr[i] = AdamTensor(p.data_ptr(), p.grad.data_ptr(), s["exp_avg"].data_ptr(),
                                s["exp_avg_sq"].data_ptr(), h.data_ptr() if h is not None else None,
                                p.numel())
            chunks += [(i, c) for c in range((p.numel() + chunk - 1) // chunk)]
        dev = ps[0].device if ps else "cuda"
        desc = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        ck = torch.tensor(chunks, dtype=torch.int32).reshape(-1, 2).to(dev)
        self._plans[gi] = (key, desc, ck, len(chunks),
                           [(p, p.grad, s["exp_avg"], s["exp_avg_sq"], p.data_ptr()) for p, s in zip(ps, sts)])
        return self._plans[gi]

    def load_state_dict(self, state_dict):
        """Accepts its own state dicts and the world-size independent ones ShardedFusedAdam writes
        (full-size moments + a "layout" tag).  The descriptor cache notices the replaced state
        tensors by identity (`_plan`)."""
        sd = dict(state_dict)
        sd.pop("layout", None)
        super().load_state_dict(sd)

    def zero_grad(self, set_to_none=False):
        """Gradients stay allocated (the step kernel holds their addresses and has already
        cleared them); only a backward without a following step leaves something to clear."""
        if self._grads_clean:
            return
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    p.grad.zero_()
        self._note_clean()

    def mark_grads_dirty(self):
        self._grads_clean = False

    def _note_clean(self):
        """The gradient buffers are all zero NOW: remember which tensor objects, at which version (a `p.grad = ...`
        assignment or an in-place torch op on a gradient — a regulariser, a manual accumulate — bumps neither flag)."""
        self._grads_clean = True
        self._clean_sig = [(id(p.grad), p.grad._version) for g in self.param_groups for p in g["params"] if p.grad is not None]

    def grads_are_clean(self):
        """True iff every gradient buffer is still the all-zero buffer the last step() / zero_grad() left: what a kernel
        that STORES instead of accumulating (vsa_nt_plan.grads_zeroed) may rely on."""
        if not self._grads_clean:
            return False
        sig = [(id(p.grad), p.grad._version) for g in self.param_groups for p in g["params"] if p.grad is not None]
        if sig != getattr(self, "_clean_sig", None):
            self._grads_clean = False
        return self._grads_clean

    def _dirty_hook(self, _param):
        self._grads_clean = False

    shared_workgroups = int(os.environ.get("VSA_ADAM_SHARED_WGS", "512"))   # step(stream=...): grid of the side-stream launch

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, stream=None):
        """stream: a side `torch.cuda.Stream` to run the update on (ordered after everything
        queued on the current stream); the returned event marks its completion and whoever reads
        the parameters next must wait for it.  The update is HBM-bound and the start of the next
        iteration (ray batch, traversal, texel compaction) neither reads nor writes parameters,
        so the two overlap."""
        loss = closure() if closure is not None else None
        if stream is not None:
            stream.wait_stream(torch.cuda.current_stream())
        sp = ctypes.c_void_p(stream.cuda_stream) if stream is not None else _lib.stream_ptr()
        # beside another stream's kernels: a bounded grid (vsa_adam_step_shared), else one workgroup per chunk
        bound = int(self.shared_workgroups) if stream is not None else 0
        for gi, group in enumerate(self.param_groups):
            _, desc, ck, n, _ = self._plan(gi, group)
            group["step"] += 1
            b1, b2 = group["betas"]
            _lib.call("vsa_adam_step_shared", desc, ck, n, float(group["lr"]), float(b1), float(b2),
                      float(group["eps"]), int(group["step"]), float(grad_scale), 1, bound, sp)
        self._note_clean()
        if stream is not None:
            ev = torch.cuda.Event()
            ev.record(stream)
            return ev
        return loss


class ShardedFusedAdam(FusedAdam):
    """Data-parallel Adam with the optimiser state and the update SHARDED over the ranks (ZeRO
    stage 1) — DESIGN.md §8: instead of all-reducing 114 MB of fp32 texture gradients and running the
    same Adam over all 28.7 M parameters on every rank,

      1. reduce-scatter the gradients: rank r receives the SUM of slice r of every tensor
         (RCCL: (w-1)/w x 114 MB on the ring instead of 2 (w-1)/w x 114 MB),
      2. Adam on that slice only (moments exist for the slice only: 1/w of the state, 1/w of the
         HBM traffic of the update),
      3. all-gather what the kernels read — the f16 compute copies where a parameter has one (half
         the bytes), the fp32 values otherwise.

    The fp32 masters are current on their owner's slice only; `gather_masters()` (called by
    VolSurfs.sync_params before save / bake) completes them.  The update is elementwise, so the
    result equals FusedAdam after an all-reduce bit for bit wherever the summed gradients do
    (tests/test_parallel.py::test_sharded_adam_equals_allreduce_adam).  A tensor whose size is not a
    multiple of world x 4 elements is all-reduced and updated in full on every rank.
    Backends without reduce_scatter (gloo) get all-reduce + slice: same numbers."""

    def __init__(self, params, world, rank, group=None, force_collectives=False, **kw):
        """force_collectives: shard (into ONE slice) and run reduce-scatter / all-gather even when
        world == 1 — identities that put RCCL's real entry points, its stream and this class's
        ordering against the Adam kernel under test on a single GPU."""
        super().__init__(params, **kw)
        self.world, self.rank, self.group = int(world), int(rank), group
        self.force_collectives = bool(force_collectives)
        self._shards = {}            # id(param) -> (a, b, grad shard, param slice, f16 slice | None)

    def _shard(self, p):
        s = self._shards.get(id(p))
        if s is None:
            n = p.numel()
            if (self.world == 1 and not self.force_collectives) or n % (self.world * 4):
                # not divisible: all-reduced and updated in full on every rank, through a private
                # gradient buffer like the slices (the descriptors never hold p.grad's address)
                h = self._half.get(id(p))
                s = (0, n, torch.zeros(n, device=p.device), p.detach().view(-1), h.view(-1) if h is not None else None)
                self._full = getattr(self, "_full", set()) | {id(p)}
            else:
                per = n // self.world
                a = self.rank * per
                h = self._half.get(id(p))
                s = (a, a + per, torch.zeros(per, device=p.device), p.detach().view(-1)[a:a + per],
                     h.view(-1)[a:a + per] if h is not None else None)
            self._shards[id(p)] = s
        return s

    def _plan(self, gi, group):
        # The cached descriptors hold raw addresses of the slice moments: they are only valid while
        # the state tensors are the same OBJECTS (load_state_dict replaces them) and the set of
        # trainable parameters is unchanged — the same identity check as the base class.
        cur = self._plans.get(gi)
        if cur is not None:
            state = self.state
            for p, ea, eas, ptr in cur[4]:
                st = state.get(p)
                if st is None or st.get("exp_avg") is not ea or st.get("exp_avg_sq") is not eas \
                        or not p.requires_grad or p.data_ptr() != ptr:
                    break
            else:
                if len(cur[4]) == sum(1 for p in group["params"] if p.requires_grad):
                    return cur
        ps = [p for p in group["params"] if p.requires_grad]
        chunk = _lib.lib().vsa_adam_chunk_elems()
        arr = (AdamTensor * max(len(ps), 1))()
        chunks = []
        for i, p in enumerate(ps):
            st = self.state[p]
            sh = self._shard(p)
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            a, b, g, ps_, hs = sh
            for k in ("exp_avg", "exp_avg_sq"):                  # the slice's moments only
                m = st.get(k)
                if m is None:
                    st[k] = torch.zeros(b - a, device=p.device)
                elif m.numel() != b - a or m.dtype != torch.float32 or not m.is_contiguous():
                    raise _lib.VolsurfsHipError(
                        "ShardedFusedAdam: state '%s' holds %d elements, this rank's slice has %d "
                        "(load checkpoints through load_state_dict)" % (k, m.numel(), b - a))
            arr[i] = AdamTensor(ps_.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                st["exp_avg_sq"].data_ptr(), hs.data_ptr() if hs is not None else None, b - a)
            n = b - a
            chunks += [(i, c) for c in range((n + chunk - 1) // chunk)]
        dev = ps[0].device if ps else "cuda"
        desc = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        ck = torch.tensor(chunks, dtype=torch.int32).reshape(-1, 2).to(dev)
        self._plans[gi] = (None, desc, ck, len(chunks),
                           [(p, self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"], p.data_ptr()) for p in ps])
        return self._plans[gi]

    # ---- checkpoints: world-size independent.  state_dict() is a COLLECTIVE (every rank calls it):
    # the moment slices are all-gathered into full tensors, so the result has the layout of a plain
    # FusedAdam / torch.optim.Adam state dict and loads into either class under any world size;
    # only rank 0 needs to write it (VolSurfs.save).  load_state_dict() takes that full layout (or
    # this rank's own slices) and keeps the slice this rank owns.
    @torch.no_grad()
    def state_dict(self):
        import torch.distributed as dist
        sd = super().state_dict()
        ps = [p for g in self.param_groups for p in g["params"]]
        full = getattr(self, "_full", set())
        state = {}
        for idx, p in enumerate(ps):
            st = sd["state"].get(idx)
            if st is None:
                continue
            st = dict(st)
            if id(p) not in full and (self.world > 1 or self.force_collectives):
                for k in ("exp_avg", "exp_avg_sq"):
                    mine = st[k].contiguous()
                    whole = torch.empty(p.numel(), device=mine.device, dtype=mine.dtype)
                    self._all_gather(whole, mine)
                    st[k] = whole.view(p.shape)
            else:
                for k in ("exp_avg", "exp_avg_sq"):
                    st[k] = st[k].view(p.shape)
            state[idx] = st
        sd["state"] = state
        sd["layout"] = "full"
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        sd = dict(state_dict)
        sd.pop("layout", None)
        ps = [p for g in self.param_groups for p in g["params"]]
        state = {}
        for idx, st in sd["state"].items():
            p = ps[int(idx)]
            a, b = self._shard(p)[0], self._shard(p)[1]
            st = dict(st)
            for k in ("exp_avg", "exp_avg_sq"):
                m = st[k]
                if m.numel() == p.numel():
                    st[k] = m.reshape(-1)[a:b].clone()
                elif m.numel() != b - a:
                    raise _lib.VolsurfsHipError(
                        "ShardedFusedAdam.load_state_dict: '%s' of parameter %d holds %d elements; expected the "
                        "full tensor (%d) or this rank's slice (%d)" % (k, int(idx), m.numel(), p.numel(), b - a))
            state[idx] = st
        sd["state"] = state
        super().load_state_dict(sd)
        self._plans.clear()           # the descriptors point at the replaced state tensors

    def __setstate__(self, state):
        super().__setstate__(state)
        self._plans = {}

    def _reduce_scatter(self, p, sh):
        import torch.distributed as dist
        a, b, g = sh[0], sh[1], sh[2]
        flat = p.grad.view(-1)
        if dist.get_backend(self.group) == "gloo":      # gloo has no reduce_scatter: all-reduce + slice, same numbers
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            g.copy_(flat[a:b])
        else:                                            # RCCL: an error here is an error (no silent fallback)
            dist.reduce_scatter_tensor(g, flat, op=dist.ReduceOp.SUM, group=self.group)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, stream=None):
        """Gradients of THIS rank's backward in p.grad -> parameters (and f16 copies) updated on
        every rank.  Replaces `allreduce_gradients` + `FusedAdam.step`."""
        import torch.distributed as dist
        if stream is not None:
            raise _lib.VolsurfsHipError("ShardedFusedAdam: the collectives order the step; no side stream")
        ps = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        for g in self.param_groups:
            self._plan(self.param_groups.index(g), g)        # state and slices exist
        full = getattr(self, "_full", set())
        for p in ps:
            sh = self._shard(p)
            if id(p) in full:
                if self.world > 1:
                    dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group)
                sh[2].copy_(p.grad.view(-1))
            else:
                self._reduce_scatter(p, sh)
        sp = _lib.stream_ptr()
        for gi, group in enumerate(self.param_groups):
            _, desc, ck, n, _ = self._plan(gi, group)
            group["step"] += 1
            b1, b2 = group["betas"]
            _lib.call("vsa_adam_step", desc, ck, n, float(group["lr"]), float(b1), float(b2),
                      float(group["eps"]), int(group["step"]), float(grad_scale), 1, sp)
        for p in ps:                       # what the kernels read, complete on every rank again
            sh = self._shard(p)
            p.grad.zero_()                 # the kernel cleared the slice buffer; the full-size buffer is this rank's backward target
            if id(p) in full:
                continue
            a, b, _, p_slice, h_slice = sh
            h = self._half.get(id(p))
            whole = (h if h is not None else p.detach()).view(-1)
            mine = (h_slice if h is not None else p_slice).clone()
            self._all_gather(whole, mine)
        self._note_clean()
        return None

    def _all_gather(self, whole, mine):
        """whole[r * per:(r + 1) * per] <- rank r's `mine`: the slices lie in rank order, so RCCL gathers
        straight into the destination (all_gather_into_tensor); gloo takes the list form."""
        import torch.distributed as dist
        per = mine.numel()
        if dist.get_backend(self.group) == "gloo":
            dist.all_gather([whole[r * per:(r + 1) * per] for r in range(self.world)], mine, group=self.group)
        else:
            dist.all_gather_into_tensor(whole, mine, group=self.group)

    @torch.no_grad()
    def gather_masters(self):
        """Complete the fp32 masters on every rank (before a checkpoint or a bake)."""
        import torch.distributed as dist
        for g in self.param_groups:
            for p in g["params"]:
                sh = self._shard(p)
                if id(p) in getattr(self, "_full", set()) or self._half.get(id(p)) is None:
                    continue
                a, b, _, p_slice, _ = sh
                self._all_gather(p.detach().view(-1), p_slice.clone())


def accumulate_into_grad(param):
    """For autograd Functions whose backward kernel ACCUMULATES (+=) into a table-sized gradient:
    if `param` already owns a contiguous fp32 `.grad` (the persistent buffers FusedAdam keeps),
    return it — the kernel then adds straight into it and the Function returns None for that input,
    saving a zero-filled temporary and autograd's separate accumulation pass over it.  Returns None
    when there is no such buffer (the Function allocates and returns a gradient as usual)."""
    g = param.grad
    if g is None or not g.is_contiguous() or g.dtype != torch.float32 or g.shape != param.shape:
        return None
    opt = getattr(param, "_vsa_optimizer", None)
    opt = opt() if opt is not None else None
    if opt is not None:
        opt.mark_grads_dirty()
    return g
