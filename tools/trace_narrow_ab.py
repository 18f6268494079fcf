"""Traversal of a TRAINING batch (34 000 random rays of 100 views, K = 5 subdiv-6 shells) with narrow waves: ms per launch
for rays-per-wave 64 (plain vsa_trace_q_fb) / 32 / 16 / 8 / 4 (vsa_trace_q_narrow), hits compared bit for bit.
usage: python tools/trace_narrow_ab.py [rays]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402
from volsurfs_amd.mesh import nested_shells                    # noqa: E402
from volsurfs_amd.raytrace import RayTracer                    # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 34000
dev = torch.device("cuda:0")
tr = RayTracer(nested_shells(K=5, subdiv=6, device=dev))
reel = bench.synthetic_reel(100, 800, dev, seed=42)
_, o, d, _, _ = reel.get_next_rays_batch(n, True, 1)
ref = None
for rpw in (64, 32, 16, 8, 4):
    tr.NARROW_BELOW, tr.NARROW_RPW = (0, 64) if rpw == 64 else (1 << 30, rpw)
    for _ in range(5):
        out = tr.trace_all(o, d)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(50):
        out = tr.trace_all(o, d)
    ev[1].record()
    torch.cuda.synchronize()
    same = True if ref is None else all(torch.equal(a, b) for a, b in zip(out, ref))
    ref = ref or out
    print(f"rays_per_wave {rpw:3d}: {ev[0].elapsed_time(ev[1]) / 50:.4f} ms per launch (50 back to back), hits identical: {same}")
