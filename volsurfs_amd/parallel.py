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
