"""One training iteration of the K-shell method in the reference's order
(/root/reference/volsurfs_py/trainer.py:118-308): zero_grad -> ray batch -> forward ->
backward -> optimiser step -> dynamic ray count -> lr scheduler, and the loop around it with
the reference's callback hooks (`train`).  Logging, checkpoints and evaluation (the rest of
trainer.py) are outside SURVEY §8 and live in the callbacks.
"""
import os

import torch


def loss_l1(gt, pred, mask=None):
    """utils/losses.py:14-19."""
    d = (gt - pred).abs()
    return (d * mask).mean() if mask is not None else d.mean()


def loss_l2(gt, pred, mask=None):
    """utils/losses.py:6-11."""
    d = (gt - pred) ** 2
    return (d * mask).mean() if mask is not None else d.mean()


def dynamic_nr_rays(nr_rays, nr_samples, target_nr_samples):
    """trainer.py:288-304: scale the next batch so that it yields ~target_nr_samples hits."""
    if nr_samples is None or nr_samples <= 0:
        return nr_rays
    return int(nr_rays * (float(target_nr_samples) / nr_samples))


class _HostCount:
    """A device counter copied to pinned host memory without stalling the stream: the copy is
    queued right where the value is produced and only waited for when it is read."""

    def __init__(self):
        self.buf = torch.zeros(1, dtype=torch.int64).pin_memory()
        self.event = torch.cuda.Event()

    def post(self, t):
        self.buf.copy_(t.reshape(1), non_blocking=True)
        self.event.record()

    def get(self):
        self.event.synchronize()
        return int(self.buf[0])


def train_step(method, rays_o, rays_d, gt_rgb, gt_mask=None, iter_nr=0, is_first_iter=False,
               nr_rays=None, target_nr_of_training_samples=None, world=1, is_training_masked=False,
               group=None, sync_losses=True, fused=True, overlap_optimizer=False, ahead=None,
               prefetch=None):
    """Returns (losses dict with a float "loss", next nr_rays).  `method` is a
    volsurfs_amd.methods.VolSurfs with init_optim() called.

    A batch larger than `method.max_rays` (the dynamic ray count of trainer.py:288-304 grows
    without bound) is run as chunks of at most max_rays rays whose losses are weighted by their
    share of the batch, gradients accumulating: the step equals the one-shot step.

    world > 1: the caller feeds this rank's shard.  The reference's loss is a MEAN over the
    batch (utils/losses.py:14-19), so each rank's loss is weighted by local_rays / global_rays
    (one all-reduce of the ray counts) before backward and the gradients are summed over the
    ranks: the result is the gradient of the global mean for even and uneven shards alike
    (SURVEY §8e).

    fused=True uses `method.fused_forward_backward` when the method offers it for this
    configuration (neural textures, constant background, unmasked L1): the same arithmetic as
    forward + backward, as one launch sequence without an autograd graph.  The only host read of
    an iteration is then the hit count the dynamic ray count needs (trainer.py:293-304), copied
    asynchronously right after the traversal, so the host queues the next iteration while this
    one still runs.  sync_losses=False leaves the losses as device tensors (no .item()).
    overlap_optimizer=True (fused path only) runs the Adam launch on a side stream; the next
    reader of the texture parameters waits for it (methods.VolSurfs.optim_step).

    ahead / prefetch (the autograd path of the legacy models, one chunk): `ahead` is this batch's
    `method.trace_ahead` context; `prefetch(next_nr_rays)` is called between the forward and the
    backward pass so that the caller can queue the NEXT batch's traversal there
    (train_step_from_reel does): the host then never waits for the backward pass and the
    optimiser step before it can size the next forward pass."""
    method.is_training = True
    method.optimizer.zero_grad()                                            # trainer.py:118
    n_local = rays_o.shape[0]
    share = 1.0
    if world > 1:
        import torch.distributed as dist
        cnt = torch.tensor([float(n_local)], device=rays_o.device, dtype=torch.float64)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
        share = n_local / cnt.item()
    cap = int(getattr(method, "max_rays", n_local) or n_local)
    bounds = list(range(0, n_local, cap)) if n_local > cap else [0]
    use_fused = fused and hasattr(method, "supports_fused_step") and \
        method.supports_fused_step(gt_mask, is_training_masked)
    losses, nr_samples, counts = {}, 0, []
    for ci, a in enumerate(bounds):
        b = min(n_local, a + cap) if len(bounds) > 1 else n_local
        sl = slice(a, b)
        w = (b - a) / max(n_local, 1)
        if use_fused:
            loss, nr_hits, _ = method.fused_forward_backward(rays_o[sl], rays_d[sl], gt_rgb[sl],
                                                            loss_weight=w * share,
                                                            is_first_iter=is_first_iter and ci == 0)
            pool = method.__dict__.setdefault("_host_counts", [])
            while len(pool) <= ci:
                pool.append(_HostCount())
            pool[ci].post(nr_hits)
            counts.append(pool[ci])
            l = {"loss": loss, "rgb": loss}
        elif (fused and len(bounds) == 1 and hasattr(method, "supports_fused_legacy_step")
              and (ahead is not None or not target_nr_of_training_samples)
              and method.supports_fused_legacy_step(rays_o, gt_mask, is_training_masked)):
            # the legacy branch without autograd (methods.VolSurfs.fused_legacy_forward / _backward): forward and
            # loss, then — where the autograd path does it too — the NEXT batch's traversal, then backward
            loss, state = method.fused_legacy_forward(rays_o, rays_d, gt_rgb, iter_nr=iter_nr, ahead=ahead,
                                                      loss_weight=w * share, is_first_iter=is_first_iter)
            if ahead is not None:
                nr_samples += int(getattr(method, "last_nr_hits", 0))
            if prefetch is not None:
                nxt = nr_rays
                if nr_rays is not None and target_nr_of_training_samples and nr_samples:
                    nxt = dynamic_nr_rays(nr_rays, nr_samples, target_nr_of_training_samples)
                prefetch(nxt)
            opt = getattr(method, "optimizer", None)
            if hasattr(opt, "mark_grads_dirty"):
                opt.mark_grads_dirty()
            method.fused_legacy_backward(state)
            l = {"loss": loss, "rgb": loss}
        else:
            single = len(bounds) == 1
            extra = {"ahead": ahead} if (ahead is not None and single) else {}
            if not target_nr_of_training_samples and hasattr(method, "trace_ahead"):
                extra["return_samples"] = False      # the hit points are only counted, and only for the dynamic ray count
            l, _, samples_3d = method(rays_o[sl], rays_d[sl], gt_rgb[sl],
                                      None if gt_mask is None else gt_mask[sl], iter_nr,
                                      is_first_iter=is_first_iter and ci == 0,
                                      is_training_masked=is_training_masked, **extra)   # :229
            if samples_3d is not None:
                nr_samples += samples_3d.shape[0]
            elif "ahead" in extra:
                nr_samples += int(getattr(method, "last_nr_hits", 0))
            if prefetch is not None and single:      # the next batch's traversal goes in HERE
                nxt = nr_rays
                if nr_rays is not None and target_nr_of_training_samples and nr_samples:
                    nxt = dynamic_nr_rays(nr_rays, nr_samples, target_nr_of_training_samples)
                prefetch(nxt)
            (l["loss"] if w * share == 1.0 else l["loss"] * (w * share)).backward()      # :264
        if len(bounds) == 1:          # w = 1: the values as they are (two elementwise launches per key otherwise)
            losses = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in l.items()}
        else:
            for k, v in l.items():
                losses[k] = losses.get(k, 0.0) + (v.detach() if isinstance(v, torch.Tensor) else v) * w
    if world > 1 and not hasattr(method.optimizer, "gather_masters"):     # (a sharded optimiser reduces inside step())
        from .parallel import allreduce_gradients
        allreduce_gradients([p for g in method.optimizer.param_groups for p in g["params"]], world, group)
    if overlap_optimizer and use_fused:                                      # :278
        method.optim_step(overlap=True)          # Adam beside the next iteration's traversal
    else:
        method.optim_step()
    nr_samples += sum(c.get() for c in counts)
    method.last_nr_samples = nr_samples
    if sync_losses:
        losses = {k: (v.item() if isinstance(v, torch.Tensor) else v) for k, v in losses.items()}
    if nr_rays is not None and target_nr_of_training_samples and nr_samples:
        nr_rays = dynamic_nr_rays(nr_rays, nr_samples, target_nr_of_training_samples)
    if method.lr_scheduler is not None:                                     # :306-308
        method.lr_scheduler.step()
    return losses, nr_rays


def train_step_from_reel(method, reel, nr_rays, jitter_pixels=True, nr_rays_per_pixel=1, iter_nr=0,
                         is_training_masked=False, **kw):
    """trainer.py:176-235: draw the batch on the device (TensorReel.get_next_rays_batch), apply
    the mask to the ground truth as :203-207 does, and step.  With nr_rays_per_pixel > 1 each
    ray is compared with its pixel's value."""
    # look-ahead (methods.VolSurfs.trace_ahead): the batch may have been drawn, and its traversal
    # queued, during the previous step; and this step queues the next one's
    can_ahead = hasattr(method, "trace_ahead") and not getattr(method, "using_neural_textures", True) \
        and getattr(method, "look_ahead", True)
    key = (id(reel), int(nr_rays), bool(jitter_pixels), int(nr_rays_per_pixel))
    la = method.__dict__.pop("_lookahead", None) if can_ahead else None
    if la is not None and la["key"] == key:
        rays_o, rays_d, vals, ahead = la["rays_o"], la["rays_d"], la["vals"], la["ahead"]
    else:
        _, rays_o, rays_d, vals, _ = reel.get_next_rays_batch(nr_rays, jitter_pixels, nr_rays_per_pixel)
        ahead = method.trace_ahead(rays_o, rays_d) if can_ahead and rays_o.shape[0] <= method.max_rays else None

    def prefetch(next_nr_rays):
        n = max(64, int(next_nr_rays))
        _, ro, rd, v, _ = reel.get_next_rays_batch(n, jitter_pixels, nr_rays_per_pixel)
        if ro.shape[0] > method.max_rays:
            return
        method._lookahead = {"key": (id(reel), n, bool(jitter_pixels), int(nr_rays_per_pixel)), "rays_o": ro,
                             "rays_d": rd, "vals": v, "ahead": method.trace_ahead(ro, rd)}
    if can_ahead and ahead is not None:
        kw = dict(kw, ahead=ahead, prefetch=prefetch)
        rays_o, rays_d = ahead.rays_o, ahead.rays_d     # the contiguous tensors the context was traced with
    gt_rgb = vals["rgb"]
    gt_mask = vals["mask"] if "mask" in vals else (torch.ones_like(gt_rgb[:, :1]) if is_training_masked else None)
    if is_training_masked:
        gt_rgb = gt_rgb * gt_mask
        if method.bg_color is not None:
            gt_rgb = gt_rgb + (1 - gt_mask) * method.bg_color.reshape(1, 3).to(gt_rgb)
    if nr_rays_per_pixel > 1:
        gt_rgb = gt_rgb.repeat_interleave(nr_rays_per_pixel, 0)
        gt_mask = None if gt_mask is None else gt_mask.repeat_interleave(nr_rays_per_pixel, 0)
    return train_step(method, rays_o, rays_d, gt_rgb, gt_mask, iter_nr, nr_rays=nr_rays,
                      is_training_masked=is_training_masked, **kw)


class TrainingState:
    """utils/training.py's state carried through the callbacks (trainer.py:84-86)."""

    def __init__(self):
        self.iter_nr = 0
        self.is_first_iter = True
        self.loss_rgb = 0.0


def train(train_data_tensor_reel, method, start_iter_nr=0, iter_finish_nr=0, callbacks=None,
          nr_training_rays=None, jitter_training_rays=True, nr_training_rays_per_pixel=1,
          is_training_masked=False, target_nr_of_training_samples=None, world=1):
    """The loop of trainer.py:57-440 around `train_step_from_reel`, with the reference's callback
    hooks (callbacks/callback.py:25-47: training_started / iter_started / iter_ended /
    training_ended, each called with phase=TrainingState).  Logging, checkpoints, test-loss
    estimates and evaluation renders of the reference loop are outside SURVEY §8 and belong to
    the callbacks.  Returns the number of completed iterations."""
    def hook(name, **kw):
        for cb in (callbacks or []):
            fn = getattr(cb, name, None)
            if fn is not None:
                fn(**kw)

    phase = TrainingState()
    phase.iter_nr = start_iter_nr
    nr_rays = nr_training_rays if nr_training_rays is not None else 512
    if getattr(method, "optimizer", None) is None:
        method.init_optim()
    hook("training_started")
    for _ in range(start_iter_nr, iter_finish_nr):
        hook("iter_started", phase=phase)
        losses, nr_rays = train_step_from_reel(
            method, train_data_tensor_reel, nr_rays, jitter_pixels=jitter_training_rays,
            nr_rays_per_pixel=nr_training_rays_per_pixel, iter_nr=phase.iter_nr,
            is_training_masked=is_training_masked, is_first_iter=phase.is_first_iter,
            target_nr_of_training_samples=target_nr_of_training_samples, world=world)
        phase.loss_rgb = losses["loss"]
        method.is_training = False                                            # set_eval_mode (:324)
        hook("iter_ended", phase=phase, losses=losses, nr_rays=nr_rays)
        phase.iter_nr += 1
        phase.is_first_iter = False
    hook("training_ended")
    return phase.iter_nr


def get_last_checkpoint_in_path(path):
    """utils/training.py: the lexicographically last <iter:07d> directory, or None."""
    import os
    if path is None or not os.path.isdir(path):
        return None
    names = sorted(n for n in os.listdir(path) if n.isdigit() and os.path.isdir(os.path.join(path, n)))
    return names[-1] if names else None


def save_checkpoints(method, iter_nr, remove_previous=True):
    """utils/training.py:59-78: optionally drop the newest previous checkpoint, then method.save."""
    import os
    import shutil
    # Under a sharded optimiser method.save() is a collective every rank calls: only rank 0 touches the
    # directory tree (the ranks would race on the same rmtree), and it does so BEFORE the collectives of
    # save(), which order the other ranks behind it.
    if remove_previous and getattr(method, "_dist_rank", 0) == 0 and method.save_checkpoints_path is not None:
        last = get_last_checkpoint_in_path(method.save_checkpoints_path)
        if last is not None:
            shutil.rmtree(os.path.join(method.save_checkpoints_path, last), ignore_errors=True)
    return method.save(iter_nr)


class Profiler:
    """mvdatasets.utils.profiler.Profiler-shaped section timer (trainer.py:555, 703; threaded
    into the method as `profiler`): start(name) / end(name) bracket a section, `get_avg_time`
    and `print_avg_times` report.  Sections are timed with events on the current stream and
    resolved at report time, so the timed code keeps running asynchronously (the reference's
    wall-clock timer relies on CUDA_LAUNCH_BLOCKING=1, trainer.py:53)."""

    def __init__(self, verbose=False):
        self.verbose = verbose
        self._open, self._pairs = {}, {}

    def start(self, name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self._open[name] = e

    def end(self, name):
        if name not in self._open:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self._pairs.setdefault(name, []).append((self._open.pop(name), e))

    def get_avg_time(self, name):
        """seconds"""
        pairs = self._pairs.get(name)
        if not pairs:
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in pairs) / len(pairs) * 1e-3

    def get_avg_times(self):
        return {k: self.get_avg_time(k) for k in self._pairs}

    def print_avg_times(self):
        for k, v in self.get_avg_times().items():
            print(f"{k}: {v * 1e3:.3f} ms (n={len(self._pairs[k])})")

    def reset(self):
        self._open, self._pairs = {}, {}


# ----------------------------------------------------------------------------------------------------------------
# The training iteration as ONE replayed HIP graph (round 6; include/volsurfs_hip.h: vsa_train_ctl)
import ctypes as _ct


class TrainCtl(_ct.Structure):
    """Mirror of `vsa_train_ctl` (include/volsurfs_hip.h)."""
    _fields_ = [("iter", _ct.c_int32), ("nr_rays", _ct.c_int32), ("capacity", _ct.c_int32), ("adam_step", _ct.c_int32),
                ("loss_scale", _ct.c_float), ("adam_lr", _ct.c_float), ("loss", _ct.c_float), ("clamped", _ct.c_int32),
                ("rng_state", _ct.c_uint64), ("rng_inc", _ct.c_uint64), ("nr_hits", _ct.c_int64),
                ("target_hits", _ct.c_int32), ("nr_warmup", _ct.c_int32), ("nr_milestones", _ct.c_int32),
                ("milestone", _ct.c_int32 * 8), ("lr_stage", _ct.c_float * 9), ("adam_pending", _ct.c_int32),
                ("lr_base", _ct.c_double), ("loss_weight", _ct.c_double), ("sum_rays", _ct.c_int64), ("sum_hits", _ct.c_int64)]


class GraphTrainLoop:
    """trainer.py:118-308 for the neural-texture method with a constant background, as ONE HIP graph per iteration.

    The eager loop (`train_step_from_reel`) issues ~20 dependent launches per iteration from Python and reads the hit
    count back to size the next batch; at the reference's batch (49 152 hits) its host side and the launch gaps between
    its kernels are as long as the kernels themselves (profiles/r06/train_host_floor.txt).  Here every launch runs at a
    fixed `capacity` of rays — the sampler fills the rays beyond the iteration's count with rays that miss the scene —
    and what changes from one iteration to the next lives in a device control block that a one-lane kernel advances by
    the reference's own rules (vsa_train_ctl_tick: dynamic ray count, warm-up + MultiStepLR, Adam's step count, the
    sampler's stream).  The Adam update of iteration i runs at the head of iteration i + 1's graph on a side stream,
    beside the ray batch, traversal and texel compaction that read no parameter.  `step()` is one graph replay: no host
    read, no host write.

    Same batches, same rays, same learning rates as the eager loop from the same state (tests/test_train_graph.py);
    parameters agree up to the order of the gradient atomics, as two eager runs do."""

    def __init__(self, method, reel, nr_rays, target_nr_of_training_samples, iter_nr=0, capacity=None,
                 jitter_pixels=True, loss_weight=1.0):
        from . import _lib
        from .optim import FusedAdam
        from .schedulers import lr_at
        if not method.supports_fused_step(None, False) or not getattr(method, "using_neural_textures", False):
            raise _lib.VolsurfsHipError("GraphTrainLoop: neural-texture method with a constant background only")
        opt = method.optimizer
        if type(opt) is not FusedAdam or len(opt.param_groups) != 1:
            raise _lib.VolsurfsHipError("GraphTrainLoop: one FusedAdam parameter group (not the sharded optimiser)")
        if reel.masks is not None:
            raise _lib.VolsurfsHipError("GraphTrainLoop: unmasked training only")
        self.method, self.reel, self.jitter = method, reel, bool(jitter_pixels)
        dev = method.bank.tables.device
        n = int(nr_rays)
        if capacity is None:      # headroom for the dynamic count's fluctuation; a multiple of 4 096
            capacity = min(int(method.max_rays), (int(n * 1.25) + 4096 + 4095) // 4096 * 4096)
        self.capacity = int(capacity)
        if not 1 <= n <= self.capacity <= method.max_rays:
            raise _lib.VolsurfsHipError(f"GraphTrainLoop: 1 <= nr_rays {n} <= capacity {self.capacity} <= max_rays")
        g = opt.param_groups[0]
        base_lr = float(g.get("initial_lr", method.lr))
        ms = sorted(int(m) for m in method.lr_milestones)[:8]
        gamma = float(getattr(method.scheduler_lr_decay, "gamma", 0.3))
        c = TrainCtl()
        c.iter, c.nr_rays, c.capacity, c.adam_step = int(iter_nr), n, self.capacity, int(g.get("step", 0))
        c.loss_scale = float(loss_weight) / (3.0 * n)
        c.rng_state, c.rng_inc = reel.rng.state, reel.rng.inc
        c.target_hits = int(target_nr_of_training_samples or 0)
        c.nr_warmup, c.nr_milestones = int(method.nr_warmup_iters), len(ms)
        for i, m_ in enumerate(ms):
            c.milestone[i] = m_
        for k in range(len(ms) + 1):
            c.lr_stage[k] = base_lr * gamma ** k           # (double -> float, as the eager loop's c_float(group["lr"]))
        c.lr_base, c.loss_weight = base_lr, float(loss_weight)
        self._host = c
        self.ctl = torch.frombuffer(bytearray(bytes(c)), dtype=torch.uint8).to(dev)
        off = TrainCtl.nr_hits.offset
        self._hits_out = self.ctl[off:off + 8].view(torch.int64)
        fn = _lib.lib().vsa_reduce_scratch_bytes
        fn.restype = _ct.c_longlong
        self._scr_hits = torch.zeros(int(fn()), dtype=torch.uint8, device=dev)
        self._scr_loss = torch.zeros(int(fn()), dtype=torch.uint8, device=dev)
        # a ray that misses every shell: from the first camera's centre, straight away from the scene's centre
        o = reel.c2w[0, :, 3].detach().cpu().double()
        d = o / o.norm() if float(o.norm()) > 0 else torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
        self._dummy = [(_ct.c_float * 3)(*[float(v) for v in x]) for x in
                       (o, d, method.bg_color.reshape(3).detach().cpu())]
        self._side = torch.cuda.Stream(device=dev)
        self.adam_beside_head = os.environ.get("VSA_TRAIN_GRAPH_ADAM_SIDE", "1") != "0"
        self.graph = None
        self._lr_at = lambda it: lr_at(it, base_lr, int(method.nr_warmup_iters), ms, gamma)
        opt._plan(0, g)                    # descriptors exist before anything is captured
        if not opt.grads_are_clean():
            opt.zero_grad()
        method.bank._ensure_grads()

    # -- one iteration: what a replay runs
    def _iteration(self):
        from . import _lib
        m, bank, opt, cap = self.method, self.method.bank, self.method.optimizer, self.capacity
        reel, ctl = self.reel, self.ctl
        dev = ctl.device
        main = torch.cuda.current_stream()
        g = opt.param_groups[0]
        _, desc, ck, nck, _ = opt._plan(0, g)
        b1, b2 = g["betas"]
        # the update of the PREVIOUS iteration beside this iteration's parameter-free head
        if self.adam_beside_head:
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                _lib.call("vsa_adam_step_ctl", desc, ck, nck, float(b1), float(b2), float(g["eps"]), 1.0, 1,
                          int(opt.shared_workgroups), ctl, _ct.c_void_p(self._side.cuda_stream))
        else:
            _lib.call("vsa_adam_step_ctl", desc, ck, nck, float(b1), float(b2), float(g["eps"]), 1.0, 1, 0, ctl,
                      _lib.stream_ptr())
        cam = torch.empty(cap, dtype=torch.int32, device=dev)
        o, d, gt = (torch.empty(cap, 3, device=dev) for _ in range(3))
        _lib.call("vsa_reel_next_rays_batch_ctl", reel.c2w, reel.intrinsics_inv, reel.rgbs, None, reel.nr_cameras,
                  reel.height, reel.width, cap, 1, self.jitter, ctl, self._dummy[0], self._dummy[1], self._dummy[2],
                  cam, o, d, gt, None, None, _lib.stream_ptr())
        hit_t, hit_slot, hit_uv = m._trace_now(o, d)
        _lib.call("vsa_count_hits", hit_slot, _ct.c_longlong(hit_slot.numel()), self._scr_hits, self._hits_out,
                  _lib.stream_ptr())
        tex_uv = bank.mark_and_compact(hit_slot, hit_uv, m.face_uvs)
        if self.adam_beside_head:
            main.wait_stream(self._side)        # the parameters (and their f16 copies) are final; the gradients are zero
        bank.evaluate()
        act = torch.empty(m.nr_meshes, cap, 4, device=dev)
        tris = m.raytracer.tris
        rgb_k, alpha_k, _, _ = bank.shade(hit_slot, tex_uv, d, tris, act_out=act)
        rgb = torch.empty(cap, 3, device=dev)
        g_c, g_a = torch.empty_like(rgb_k), torch.empty_like(alpha_k)
        _lib.call("vsa_composite_dense_fwd_bwd_l1_ctl", rgb_k, alpha_k, m.bg_color, True, gt, ctl, rgb, g_c, g_a, cap,
                  m.nr_meshes, 0, _lib.stream_ptr())
        # (the f16 gradient chain's scale is any constant the kernels divide out again: the capacity's)
        from .pipeline import GRAD_CHAIN_GAIN
        scale = m.grad_scale if m.grad_scale is not None else GRAD_CHAIN_GAIN * float(cap)
        bank.backward(hit_slot, tex_uv, d, tris, g_c, g_a, scale, act, grads_zeroed=True)
        _lib.call("vsa_l1_mean_ctl", rgb, gt, cap, self._scr_loss, ctl, _lib.stream_ptr())
        _lib.call("vsa_train_ctl_tick", ctl, _lib.stream_ptr())

    def capture(self, warm_iterations=2):
        """`warm_iterations` real iterations run eagerly first (every lazily allocated buffer exists afterwards), then one
        iteration is captured — a capture pass only records."""
        from .pipeline import _CAPTURE_MODE, _no_gc
        m = self.method
        m.is_training = True
        opt = m.optimizer
        opt.mark_grads_dirty()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(int(warm_iterations)):
                self._iteration()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with _no_gc(), torch.cuda.graph(self.graph, capture_error_mode=_CAPTURE_MODE):
            self._iteration()
        return self

    def step(self):
        self.graph.replay()

    def read(self):
        """The control block as it stands on the device (synchronises): iter, nr_rays, nr_hits, loss, lr, clamped."""
        c = TrainCtl.from_buffer_copy(bytes(self.ctl.cpu().numpy().tobytes()))
        return {"iter": c.iter, "nr_rays": c.nr_rays, "nr_hits": c.nr_hits, "loss": c.loss, "adam_step": c.adam_step,
                "adam_lr": c.adam_lr, "clamped": c.clamped, "capacity": c.capacity, "rng_state": c.rng_state,
                "sum_rays": c.sum_rays, "sum_hits": c.sum_hits}

    def finish(self):
        """Apply the last iteration's pending update and hand the loop's state back to the host objects (optimiser step
        count and lr, scheduler position, the reel's stream), so that eager training, checkpoints or another loop can go on."""
        from . import _lib
        m, opt = self.method, self.method.optimizer
        g = opt.param_groups[0]
        _, desc, ck, nck, _ = opt._plan(0, g)
        b1, b2 = g["betas"]
        _lib.call("vsa_adam_step_ctl", desc, ck, nck, float(b1), float(b2), float(g["eps"]), 1.0, 1, 0, self.ctl,
                  _lib.stream_ptr())
        off = TrainCtl.adam_pending.offset
        self.ctl[off:off + 4].zero_()
        st = self.read()
        g["step"] = int(st["adam_step"])
        sched = getattr(m, "lr_scheduler", None)
        if sched is not None and hasattr(sched, "it"):
            sched.it = int(st["iter"])
            sched._apply()
        else:
            g["lr"] = self._lr_at(int(st["iter"]))
        self.reel.rng.state = int(st["rng_state"])
        opt._note_clean()
        m.last_nr_samples = int(st["nr_hits"])
        return st
