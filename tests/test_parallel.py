"""N>1 path on CPU: ray sharding + gradient all-reduce with gloo, world_size 2."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from volsurfs_amd.parallel import (GradientOverlap, allreduce_gradients, gather_frame, shard_chunks,
                                   shard_indices)


def test_shards_partition_the_frame():
    for n in (0, 1, 16383, 16384, 640000, 1920000):
        for world in (1, 2, 3, 8):
            seen = torch.zeros(n, dtype=torch.int32)
            for r in range(world):
                for a, b in shard_chunks(n, r, world):
                    assert 0 <= a < b <= n and (a % 16384 == 0)
                    seen[a:b] += 1
            assert (seen == 1).all()
    # round-robin balance: ranks differ by at most one chunk
    sizes = [shard_indices(640000, r, 8).numel() for r in range(8)]
    assert max(sizes) - min(sizes) <= 16384


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rays, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    # a stand-in differentiable "renderer": per-ray colour from shared parameters
    w = torch.nn.Parameter(torch.linspace(-1, 1, 12).view(4, 3))
    feats = torch.rand(n_rays, 4, generator=torch.Generator().manual_seed(1))
    gt = torch.rand(n_rays, 3, generator=torch.Generator().manual_seed(2))
    idx = shard_indices(n_rays, rank, world, chunk=1000)
    pred = feats[idx] @ w
    # loss normalised by the GLOBAL ray count (volsurfs utils/losses.py:14-19 is a mean)
    loss = (gt[idx] - pred).abs().sum() / (n_rays * 3)
    loss.backward()
    allreduce_gradients([w], world)
    frame = gather_frame(pred.detach(), n_rays, rank, world, chunk=1000)
    if rank == 0:
        q.put((w.grad.clone(), frame))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank():
    n_rays = 7300            # uneven shards: 4 chunks of 1000 vs 3 + a remainder
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_rays, q)) for r in range(2)]
    for p in procs:
        p.start()
    grad, frame = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process reference
    w = torch.nn.Parameter(torch.linspace(-1, 1, 12).view(4, 3))
    feats = torch.rand(n_rays, 4, generator=torch.Generator().manual_seed(1))
    gt = torch.rand(n_rays, 3, generator=torch.Generator().manual_seed(2))
    pred = feats @ w
    ((gt - pred).abs().mean()).backward()
    torch.testing.assert_close(grad, w.grad, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(frame, pred.detach())


def _overlap_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the bench's pattern: slices of one gradient tensor are reduced as they become final
    g = torch.arange(40 * 6, dtype=torch.float32).view(40, 6) * (rank + 1)
    w = torch.full((5,), float(rank + 1))
    ov = GradientOverlap(world)
    ov.reduce_async(w)
    for s in range(5):
        ov.reduce_async(g[s * 8:(s + 1) * 8])      # contiguous slice, reduced in place
    ov.wait()
    assert ov.works == []
    if rank == 0:
        q.put((g.clone(), w.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_overlap_slices_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    g, w = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    torch.testing.assert_close(g, torch.arange(40 * 6, dtype=torch.float32).view(40, 6) * 3)
    torch.testing.assert_close(w, torch.full((5,), 3.0))
