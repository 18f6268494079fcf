"""Secondary measurement (config C4's background branch):
fwd+bwd of VolSurfs.forward with bg_color=None, i.e. the K-shell path plus the contracted
background (32 inverse-depth samples per ray through the packed ops, NerfHash field: 3-D hash
grid on HIP, fp32 MLPs on rocBLAS) on a batch of rays.  Prints one JSON line.
usage: python tools/bench_bg.py [--rays 65536] [--steps 20]"""
import argparse
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from volsurfs_amd.background import BoundingSphere          # noqa: E402
from volsurfs_amd.camera import pinhole_rays                # noqa: E402
from volsurfs_amd.mesh import nested_shells                 # noqa: E402
from volsurfs_amd.methods import VolSurfs                   # noqa: E402
from volsurfs_amd.models import NerfHash                    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=65536)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--shells", type=int, default=5)
args = ap.parse_args()

meshes = nested_shells(K=args.shells, subdiv=6)
bg = NerfHash(3, "gridhash", "spherical_harmonics")
m = VolSurfs(meshes, max_rays=args.rays, bg_color=None, bg_model=bg,
             bounding_primitive=BoundingSphere(0.5), nr_samples_bg=32)
m.grad_scale = float(args.rays)
opt = m.init_optim()
o, d = pinhole_rays(800, 800, focal=1111.1, cam_pos=(0.0, 0.0, -1.5))
g = torch.Generator(device="cuda").manual_seed(0)
idx = torch.randperm(o.shape[0], device="cuda", generator=g)[:args.rays]
o, d = o[idx].contiguous(), d[idx].contiguous()
gt = torch.rand(args.rays, 3, device="cuda", generator=g)


def step(it):
    opt.zero_grad()
    losses, _, _ = m(o, d, gt, None, it)
    losses["loss"].backward()
    m.optim_step()


for it in range(5):
    step(it)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(args.steps):
    step(it)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
print(json.dumps({"workload": f"K={args.shells} shells + NerfHash background, {args.rays} random rays of an "
                              "800x800 view, 32 bg samples/ray, fwd+bwd+Adam",
                  "ms_per_step": dt * 1e3, "Mrays/s": args.rays / dt / 1e6,
                  "bg_samples_per_step": args.rays * 32}))
