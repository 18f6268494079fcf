"""Does a training batch's traversal run BESIDE the persistent kernels of another stream?  Stream A replays the graph of a
whole step at the training batch (34 000 random rays: trace, compaction, encode / MLP / shade forward and backward); stream B
launches traversals of another 34 000 rays back to back.  Times: A alone, B alone, both together (A's replays and as many
traversals as fit one per replay).  usage: python tools/overlap_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volsurfs_amd.camera import pinhole_rays                   # noqa: E402
from volsurfs_amd.mesh import nested_shells                    # noqa: E402
from volsurfs_amd.pipeline import KShellPipeline               # noqa: E402
from volsurfs_amd.raytrace import RayTracer                    # noqa: E402

n = 34000
meshes = nested_shells(K=5, subdiv=6)
o, d = pinhole_rays(800, 800, focal=1111.1, cam_pos=(0.0, 0.0, -1.5))
g = torch.Generator(device="cuda").manual_seed(0)
idx = torch.randperm(o.shape[0], device="cuda", generator=g)
p = KShellPipeline(meshes, o[idx[:n]].contiguous(), d[idx[:n]].contiguous(), torch.rand(n, 3, device="cuda", generator=g))
for _ in range(5):
    p.step()
p.capture_graph()
tr = RayTracer(meshes)
o2, d2 = o[idx[n:2 * n]].contiguous(), d[idx[n:2 * n]].contiguous()
out = tuple(torch.empty_like(x) for x in tr.trace_all(o2, d2))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
reps = 200


def run(a, b):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if a:
            with torch.cuda.stream(sa):
                p.replay()
        if b:
            with torch.cuda.stream(sb):
                tr.trace_all(o2, d2, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for _ in range(2):
    print(f"step graph alone {run(True, False):.4f} ms | traversal alone {run(False, True):.4f} ms | together {run(True, True):.4f} ms per pair")
