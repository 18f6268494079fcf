"""Tile-parallel data parallelism for the K-shell path (SURVEY §5 "Distributed
communication backend", §8e).  The reference is single-GPU (G1); rays are
independent end to end, so a frame / batch is sharded by contiguous ray chunks
(16 384 = the reference's eval chunk, base_method.py:407-418) dealt round-robin
to the ranks, meshes + BVH + textures are replicated, and the only collective is
one all-reduce(sum) of the parameter gradients per training step (RCCL over xGMI
on MI355X; gloo in the CPU tests)."""
import torch


def shard_chunks(nr_rays, rank, world, chunk=16384):
    """[(start, end)] ray ranges owned by `rank`: chunk c goes to rank c % world.
    Round-robin (not one contiguous block per rank) balances the load, because rays
    that miss every shell are almost free."""
    out = []
    nchunks = (nr_rays + chunk - 1) // chunk
    for c in range(rank, nchunks, world):
        out.append((c * chunk, min(nr_rays, (c + 1) * chunk)))
    return out


def shard_indices(nr_rays, rank, world, chunk=16384, device="cpu"):
    parts = [torch.arange(a, b, device=device) for a, b in shard_chunks(nr_rays, rank, world, chunk)]
    return torch.cat(parts) if parts else torch.zeros(0, dtype=torch.long, device=device)


def shard_bands(height, rank, world, band=8):
    """Rows of an image owned by `rank` when the frame is dealt round-robin in horizontal bands
    of `band` rows (8 = the pixel-tile edge of KShellPipeline's ray order, so a rank's rows form
    an image of its own whose 8x8 tiles are tiles of the full frame).  Strong scaling of ONE
    frame: rays that miss every shell are almost free, so contiguous blocks would leave the
    ranks holding the frame's border idle."""
    if height % band:
        raise ValueError(f"image height {height} is not a multiple of the band height {band}")
    rows = [r for b in range(rank, height // band, world) for r in range(b * band, (b + 1) * band)]
    return torch.tensor(rows, dtype=torch.long)


def allreduce_gradients(params, world, group=None):
    """Sum the gradients over ranks (each rank back-propagated its shard of a loss
    normalised by the GLOBAL ray count, so the sum is the gradient of the global
    mean loss; uneven shards stay exact — SURVEY §8e)."""
    if world == 1:
        return
    import torch.distributed as dist
    works = []
    for p in params:
        if p.grad is not None:
            works.append(dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()


class GradientOverlap:
    """All-reduce gradient tensors as they become final, overlapped with the rest of
    backward: `reduce_async(t)` enqueues an asynchronous all-reduce(sum) of `t` (ordered
    after the work already queued on the current stream), `wait()` makes the current
    stream wait for all of them.  One instance per training loop.

    wire_dtype (opt-in, e.g. torch.bfloat16): the tensor is converted, reduced at that width and
    converted back in `wait()` — half the bytes on the xGMI ring (the 114 MB of fp32 texture
    gradients are what bounds strong scaling) at the price of a bf16-rounded SUM
    (~2^-9 relative per hop); the default (None) reduces the fp32 tensors in place, exactly."""

    def __init__(self, world, group=None, wire_dtype=None, force=False):
        """force: issue the collectives even in a one-rank group (they are identities there) — how a
        single MI355X exercises the real RCCL path: communicator init, RCCL's stream against the
        kernels' stream (bench.py --force-dist, tests/test_parallel.py::test_rccl_one_rank_*)."""
        self.world, self.group, self.works, self.wire_dtype, self.force = world, group, [], wire_dtype, force

    def reduce_async(self, t):
        if self.world == 1 and not self.force:
            return
        import torch.distributed as dist
        if self.wire_dtype is None:
            self.works.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True),
                               None, None))
        else:
            buf = t.to(self.wire_dtype)
            self.works.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True),
                               t, buf))

    def wait(self):
        for w, t, buf in self.works:
            w.wait()
            if buf is not None:
                t.copy_(buf)
        self.works = []


def default_phases(nr_shells, max_phases=3):
    """Shell-range ends of the hash-grid backward's phases.  Every phase costs the one launch ~26 us on
    one MI355X (each re-splits its shells' planes over all 256 workgroups: more, smaller LDS pieces) and
    removes 1/P - 1/(P+1) of the all-reduce from behind the last kernel: with the 114 MB of table
    gradients at 0.35-0.6 ms on the wire (8 GPUs, bf16 / fp32) the modelled exposed time is flat from 3
    phases on, so 3 it is; K < 3 shells: one phase per shell."""
    n = max(1, min(int(max_phases), int(nr_shells)))
    # (rounded up: the LAST phase, whose all-reduce nothing hides, is the smallest)
    return sorted({-(-nr_shells * (i + 1) // n) for i in range(n)})


def choose_phases(nr_shells, t_comm_ms, t_bwd_ms, head_ms, boundary_ms=0.045):
    """Phase count of the hash-grid backward from MEASURED times: t_comm_ms = the all-reduce of all
    table gradients on this group, t_bwd_ms = the un-phased hash-grid backward, head_ms = the next step's parameter-free head
    that runs before the wait (OverlappedStep.run_split), boundary_ms = what one more phase costs the launch (0.028 ms of
    piece visits + 0.015 ms for the wait -> collective -> event chain, profiles/r05/dp_schedule_one_gpu.txt).
    Phase p's slice is final at (p + 1) / n of the launch and its reduction takes its share of t_comm on the wire behind the
    previous slice's; what is left behind the launch, minus the head, is exposed.  One phase when the whole reduction hides
    behind the head (one GPU, a one-rank group: the step then pays nothing for being data-parallel).  Returns the ends."""
    best = None
    for n in range(1, max(1, int(nr_shells)) + 1):
        ends = default_phases(nr_shells, n)
        m = len(ends)
        t_b = t_bwd_ms + (m - 1) * boundary_ms
        done, prev = 0.0, 0
        for p, e in enumerate(ends):
            ready = t_b * (p + 1) / m
            done = max(done, ready) + t_comm_ms * (e - prev) / nr_shells
            prev = e
        cost = (m - 1) * boundary_ms + max(0.0, done - t_b - head_ms)
        if best is None or cost < best[0] - 1e-9:
            best = (cost, ends)
    return best[1]


class _EventWork:
    """What GradientOverlap.wait() needs of a Work: wait() makes the current stream wait."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class StepSignals:
    """Device-side completion flags of one data-parallel step (include/volsurfs_hip.h:
    vsa_dp_signal, vsa_nt_encode_bwd_phased, vsa_dp_stream_wait).  The step itself stays ONE
    stream of launches (so it captures into a HIP graph and replays): behind the MLP backward a
    one-lane kernel advances the epoch and publishes it in `flag_w` (weights.grad is final); the
    hash-grid backward is ONE launch whose workgroups walk the shells phase by phase and publish
    the epoch in flags[p] when the last of them leaves phase p.  A communication stream waits for
    `flag >= epoch` in front of each all-reduce.

    phases: list of shell-range ends, strictly increasing, last = K (default: default_phases(K))."""

    def __init__(self, nr_shells, device, phases=None, wait_mode=0, reserve_cus=0):
        self.reserve_cus = int(reserve_cus)      # compute units the hash-grid backward leaves to the collectives' kernels
        self.phase_end = [int(x) for x in (phases or default_phases(nr_shells))]
        if self.phase_end[-1] != nr_shells or any(b <= a for a, b in zip([0] + self.phase_end, self.phase_end)):
            raise ValueError(f"phases {self.phase_end} do not cut shells 0..{nr_shells}")
        n = len(self.phase_end)
        self.n = n
        import ctypes
        from . import _lib
        # flags: n phase words + the weights word, each its own signal allocation (include/volsurfs_hip.h:
        # vsa_dp_flags — the command processor waits on those, no wave is parked on a CU); epoch and the
        # per-phase workgroup counters: plain device words
        self._flags = ctypes.c_void_p()
        rc = _lib.lib().vsa_dp_flags_create(ctypes.c_int(n + 1), ctypes.byref(self._flags))
        if rc != 0:
            raise _lib.VolsurfsHipError(f"vsa_dp_flags_create failed with status {rc}")
        self.words = torch.zeros(n + 1, dtype=torch.int32, device=device)
        self.epoch, self.counters = self.words[:1], self.words[1:]
        # the device epoch after the steps launched so far.  Kept in ONE place per kind of launch: signal_weights()
        # counts an eager step, KShellPipeline.replay* a graph replay (capture passes only record) — callers never
        # patch it (a missed bump released every all-reduce on the previous step's flag values: ADVICE r5)
        self.epoch_host = 0
        self.on_weights_final = None   # eager steps call it behind the MLP backward (OverlappedStep: records an event)
        self.wait_mode = int(wait_mode)
        self.phase_end_c = (ctypes.c_int32 * n)(*self.phase_end)

    W = property(lambda self: self.n)     # index of the weights word in the flags

    def shell_range(self, p):
        return (self.phase_end[p - 1] if p else 0), self.phase_end[p]

    def signal_weights(self):
        """Stream-ordered: epoch += 1, weights word = epoch (a one-lane kernel: graph-capturable)."""
        from . import _lib
        _lib.call("vsa_dp_signal", self._flags, self.n, self.epoch, 1, _lib.stream_ptr())
        if not torch.cuda.is_current_stream_capturing():
            self.epoch_host += 1

    def stream_wait(self, index, value):
        """The CURRENT stream waits until flag word `index` (0..n-1: phases, n: weights) holds `value`."""
        from . import _lib
        _lib.call("vsa_dp_stream_wait", self._flags, int(index), int(value), self.wait_mode, _lib.stream_ptr())

    def read(self):
        """(flag words [n + 1], device epoch, counters [n]) as Python ints; synchronises (tests)."""
        import ctypes
        from . import _lib
        out = (ctypes.c_uint32 * (self.n + 1))()
        rc = _lib.lib().vsa_dp_flags_read(self._flags, out)
        if rc != 0:
            raise _lib.VolsurfsHipError(f"vsa_dp_flags_read failed with status {rc}")
        w = self.words.cpu().tolist()
        return list(out), w[0], w[1:]

    def __del__(self):
        try:
            from . import _lib
            if getattr(self, "_flags", None):
                _lib.lib().vsa_dp_flags_destroy(self._flags)
        except Exception:
            pass
        self._flags = None


class OverlappedStep:
    """The data-parallel training step: `launch()` enqueues one whole step built with
    `pipe.step(dp=signals)` — eagerly or as a graph replay — and the gradients go out on a side
    stream as the device publishes them: weights.grad behind the MLP backward, then each phase's
    slice of tables.grad while the later phases still accumulate.  Nothing in the step is split,
    re-ordered or synchronised with the host for this; on one GPU it costs what the phased walk of
    the hash-grid backward costs (profiles/r05/dp_schedule_one_gpu.txt).

    run() returns with the current stream waiting for the reduced gradients (as GradientOverlap.wait)."""

    def __init__(self, pipe, world, group=None, wire_dtype=None, force=False, phases=None, wait_mode=0,
                 reserve_cus=0, direct_rccl=None, rank=0):
        """direct_rccl: enqueue the all-reduces with RCCL itself on the side stream (volsurfs_amd.rccl: no
        ProcessGroupNCCL stream, no hand-off per collective) — True / False, default: when the group's backend is
        "nccl" and no wire dtype is asked for.  rank: this process's rank in `group` (forming the communicator)."""
        dev = pipe.bank.tables.device
        self.pipe, self.world, self.force = pipe, world, force
        self.rccl = None
        if world > 1 or force:
            import torch.distributed as dist
            if direct_rccl is None:
                direct_rccl = dist.is_initialized() and dist.get_backend(group) == "nccl" and wire_dtype is None
            if direct_rccl:
                # all ranks use the direct communicator or none does: a rank whose communicator did not come up
                # makes everybody fall back to torch.distributed (agreed through the group itself)
                try:
                    from .rccl import RcclComm
                    self.rccl = RcclComm(rank, world, group)
                except Exception as exc:
                    import warnings
                    warnings.warn(f"direct RCCL communicator not formed ({exc}); all-reduces go through torch.distributed")
                    self.rccl = None
                if world > 1:
                    ok = torch.tensor([1 if self.rccl is not None else 0], device=dev, dtype=torch.int32)
                    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
                    if int(ok.item()) == 0 and self.rccl is not None:
                        self.rccl.destroy()
                        self.rccl = None
        self.signals = StepSignals(pipe.K, dev, phases, wait_mode, reserve_cus)
        self.overlap = GradientOverlap(world, group, wire_dtype, force)
        self.side = torch.cuda.Stream(device=dev)
        self.active = world > 1 or force
        self.signals.on_weights_final = self._weights_final

    def eager(self, record=False):
        return self.pipe.step(record=record, dp=self.signals)

    def autotune_phases(self, group=None, reps=3):
        """Replace the default phase split by choose_phases() of times measured HERE: the all-reduce of the gradient
        buffers on this group's wire (side stream, events), the hash-grid backward and the step's head (one recorded eager
        step).  Collective: every rank calls it at the same point, before any graph is captured; the ranks take the
        slowest rank's times, so they all build the same launch.  Returns (phase ends, measured times)."""
        pipe, bank = self.pipe, self.pipe.bank
        pipe.step()                                    # gradient buffers exist
        torch.cuda.synchronize()
        t_comm = 0.0
        if self.active:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            with torch.cuda.stream(self.side):
                for i in range(reps + 1):              # (the first run pays the channel set-up)
                    if i == 1:
                        ev[0].record()
                    if self.rccl is not None:
                        self.rccl.all_reduce_sum_(bank.tables.grad, self.side)
                    else:
                        self.overlap.reduce_async(bank.tables.grad)
                        self.overlap.wait()
                ev[1].record()
            torch.cuda.synchronize()
            t_comm = ev[0].elapsed_time(ev[1]) / reps
        pipe.reset_stage_timers()
        pipe.step(record=True)
        st = pipe.stage_report()
        pipe.reset_stage_timers()
        t_bwd = float(st.get("nt_encode_bwd", {}).get("ms", 0.6))
        head = sum(float(st.get(k, {}).get("ms", 0.0)) for k in ("ray_tile_order", "trace", "nt_mark_compact"))
        times = torch.tensor([t_comm, t_bwd, -head], device=bank.tables.device, dtype=torch.float64)
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(times, op=dist.ReduceOp.MAX, group=group)       # slowest wire, slowest launch, shortest head
        t_comm, t_bwd, head = float(times[0]), float(times[1]), -float(times[2])
        ends = choose_phases(pipe.K, t_comm, t_bwd, head)
        if ends != self.signals.phase_end:
            sg = self.signals
            self.signals = StepSignals(pipe.K, bank.tables.device, ends, sg.wait_mode, sg.reserve_cus)
            self.signals.on_weights_final = self._weights_final
        return ends, {"t_comm_ms": t_comm, "t_encode_bwd_ms": t_bwd, "head_ms": head}

    def run_split(self, prefix, mid, tail):
        """The pipelined form (pipe.capture_graph_split; replay_prefix / replay_mid / replay_tail):
        `prefix()` — the step's parameter- and gradient-free head: ray order, traversal, mark / compact,
        0.33 ms of the 2.5 — is enqueued BEFORE the current stream waits for the previous step's reduction, so
        the tail of that reduction (its last phase, the hand-offs between the streams; in training also the
        optimiser on its side stream) runs beside it; then the wait, then `mid()` (zero_grad .. MLP backward),
        an event (weights.grad is final), `tail()` (the hash-grid backward, ONE launch) and this step's
        collectives.  Every step's gradients are still fully reduced before anything reads them; call
        finish() behind the last step."""
        prefix()
        self.overlap.wait()
        mid()
        self._weights_final()
        out = tail()
        self._enqueue_reductions()
        return out

    def finish(self):
        self.overlap.wait()

    def _weights_final(self):
        if self.active:
            self._mid_event = torch.cuda.Event()
            self._mid_event.record()

    def _enqueue_reductions(self):
        """The side stream's program for the step just launched.  It first waits for the EVENT behind the MLP
        backward and only then for the phase flags: a flag wait that sits at the head of another hardware
        queue while the step's twenty-odd launches go by costs the step 0.6-0.9 ms on this stack (whether the
        command processor polls a signal word or a one-lane kernel does: profiles/r05/dp_schedule_one_gpu.txt);
        a pending event wait costs nothing, and behind it the flag waits are only ever pending beside the
        hash-grid backward itself."""
        sg, bank = self.signals, self.pipe.bank
        if not self.active:
            return
        e = sg.epoch_host          # (the launch itself counted: StepSignals.signal_weights / KShellPipeline.replay*)
        ev = getattr(self, "_mid_event", None)
        self._mid_event = None
        if self.rccl is not None:
            return self._enqueue_reductions_rccl(e, ev)
        # (after the producer: a wait queued ahead of the kernel that satisfies it could share its hardware queue)
        with torch.cuda.stream(self.side):
            if ev is not None:
                self.side.wait_event(ev)
            else:
                sg.stream_wait(sg.W, e)         # a one-graph / un-instrumented launch: the weights word instead
            self.overlap.reduce_async(bank.weights.grad)
            for p in range(sg.n - 1):
                a, b = sg.shell_range(p)
                sg.stream_wait(p, e)
                self.overlap.reduce_async(bank.tables.grad[a * 8:b * 8])
        # the last phase is final when the launch ends: its all-reduce is simply stream-ordered behind the
        # step (one cross-stream hand-off less on the only part of the reduction that nothing hides)
        a, b = sg.shell_range(sg.n - 1)
        self.overlap.reduce_async(bank.tables.grad[a * 8:b * 8])

    def _enqueue_reductions_rccl(self, e, ev):
        """The same program with RCCL called directly: every all-reduce is a launch on the side stream itself, right
        behind the wait that releases it; the current stream waits for ONE event behind the last of them."""
        sg, bank = self.signals, self.pipe.bank
        tail = torch.cuda.Event()
        tail.record()                                   # the step's last launch (the hash-grid backward) is behind this
        with torch.cuda.stream(self.side):
            if ev is not None:
                self.side.wait_event(ev)
            else:
                sg.stream_wait(sg.W, e)
            self.rccl.all_reduce_sum_(bank.weights.grad, self.side)
            for p in range(sg.n - 1):
                a, b = sg.shell_range(p)
                sg.stream_wait(p, e)
                self.rccl.all_reduce_sum_(bank.tables.grad[a * 8:b * 8], self.side)
            a, b = sg.shell_range(sg.n - 1)
            self.side.wait_event(tail)                  # the last phase is final when the launch ends
            self.rccl.all_reduce_sum_(bank.tables.grad[a * 8:b * 8], self.side)
            done = torch.cuda.Event()
            done.record()
        self.overlap.works.append((_EventWork(done), None, None))

    def run(self, launch=None, record=False):
        """launch: a callable that enqueues one step built with dp=self.signals (default: the eager
        step; a graph's replay otherwise).  record: time the eager step's stages and the exposed wait."""
        if launch is not None:
            # a graph replay never calls _weights_final: an event a warm-up or an earlier eager step left behind
            # would release weights.grad's all-reduce before this step's MLP backward -> the flag word
            self._mid_event = None
        out = launch() if launch is not None else self.eager(record=record)
        self._enqueue_reductions()
        bank = self.pipe.bank
        self.pipe.timer.run("grad_allreduce", self.overlap.wait, record,
                            bytes=(bank.tables.numel() + bank.weights.numel()) * 4)
        return out


def gather_frame(local_rgb, nr_rays, rank, world, chunk=16384, group=None, force=False):
    """Rank 0 receives the full [nr_rays,3] frame (rendering needs no other
    collective: every rank writes its own tiles).  force: run the collective in a one-rank group too."""
    import torch.distributed as dist
    if world == 1 and not force:
        return local_rgb
    sizes = [sum(b - a for a, b in shard_chunks(nr_rays, r, world, chunk)) for r in range(world)]
    bufs = [torch.empty(s, 3, dtype=local_rgb.dtype, device=local_rgb.device) for s in sizes]
    dist.all_gather(bufs, local_rgb.contiguous(), group=group) if len(set(sizes)) == 1 else \
        _all_gather_uneven(bufs, local_rgb, rank, world, group)
    if rank != 0:
        return None
    frame = torch.empty(nr_rays, 3, dtype=local_rgb.dtype, device=local_rgb.device)
    for r in range(world):
        frame[shard_indices(nr_rays, r, world, chunk, local_rgb.device)] = bufs[r]
    return frame


def _all_gather_uneven(bufs, local, rank, world, group):
    import torch.distributed as dist
    for r in range(world):
        if r == rank:
            bufs[r].copy_(local)
        dist.broadcast(bufs[r], src=r, group=group)
