"""Host-side profile of the config-3 training iteration (torch.profiler): which ops the Python
loop issues per iteration, how often, and what they cost on the CPU.  Run via gpurun:
    python tools/prof_host.py [train|train-permuto]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from volsurfs_amd.mesh import nested_shells
from volsurfs_amd.methods import VolSurfs
from volsurfs_amd.trainer import train_step_from_reel
workload = sys.argv[1] if len(sys.argv) > 1 else "train-permuto"
dev = torch.device("cuda:0")
meshes = nested_shells(K=5, subdiv=6, device=dev)
if workload == "dtu":          # BASELINE configs[3]: learned background, batches of 65 536 rays of a 1600x1200 frame
    from volsurfs_amd.background import BoundingSphere
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.models import NerfHash
    from volsurfs_amd.trainer import train_step
    H, W, batch = 1200, 1600, 65536
    m = VolSurfs(meshes, max_rays=batch, bg_color=None, bg_model=NerfHash(3, "gridhash", "spherical_harmonics", device=dev),
                 bounding_primitive=BoundingSphere(0.5), nr_samples_bg=32, nr_warmup_iters=0)
    m.init_optim()
    o, d = pinhole_rays(H, W, focal=1111.1 * H / 800.0, cam_pos=(0.0, 0.0, -1.5), device=dev)
    g = torch.Generator(device=dev).manual_seed(42)
    gt = torch.rand(H * W, 3, device=dev, generator=g)
    perm = torch.randperm(H * W, device=dev, generator=g)
    o, d, gt = o[perm].contiguous(), d[perm].contiguous(), gt[perm].contiguous()
    st = {"it": 0}

    def one_dtu():
        a = (st["it"] % 29) * batch
        m.grad_scale = float(batch)
        train_step(m, o[a:a + batch], d[a:a + batch], gt[a:a + batch], iter_nr=st["it"],
                   is_first_iter=st["it"] == 0, world=1, sync_losses=False)
        st["it"] += 1
    for _ in range(8):
        one_dtu()
    torch.cuda.synchronize()
    N = 5
    import time
    t0 = time.perf_counter()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        for _ in range(N):
            one_dtu()
        torch.cuda.synchronize()
    print("ms per batch (profiled):", (time.perf_counter() - t0) / N * 1e3)
    ev = prof.key_averages()
    print("%-58s %8s %10s %10s" % ("op", "calls/it", "self us/it", "total us/it"))
    for e in sorted(ev, key=lambda e: -e.self_cpu_time_total)[:25]:
        print("%-58s %8.1f %10.1f %10.1f" % (e.key[:58], e.count / N, e.self_cpu_time_total / N, e.cpu_time_total / N))
    print("sum of self CPU time per batch: %.0f us" % (sum(e.self_cpu_time_total for e in ev) / N))
    sys.exit(0)
kw = dict(using_neural_textures=False, rgb_pos_encoder_type="permutohash",
          rgb_mlp_layers_dims=(128, 128, 64)) if workload == "train-permuto" else {}
method = VolSurfs(meshes, max_rays=1 << 17, nr_warmup_iters=500, seed=42, **kw)
method.init_optim()
reel = bench.synthetic_reel(50, 800, dev, seed=42)
state = {"n": 512, "it": 0}


def one():
    method.grad_scale = float(state["n"])
    _, nxt = train_step_from_reel(method, reel, state["n"], jitter_pixels=True, iter_nr=state["it"],
                                  is_first_iter=state["it"] == 0, target_nr_of_training_samples=49152,
                                  world=1, sync_losses=False)
    state["n"] = max(64, min(int(nxt), 4 << 17))
    state["it"] += 1


for _ in range(60):
    one()
torch.cuda.synchronize()
N = 10
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
    for _ in range(N):
        one()
    torch.cuda.synchronize()
ev = prof.key_averages()
rows = sorted(ev, key=lambda e: -e.self_cpu_time_total)[:40]
print("%-58s %8s %10s %10s" % ("op", "calls/it", "self us/it", "total us/it"))
for e in rows:
    print("%-58s %8.1f %10.1f %10.1f" % (e.key[:58], e.count / N, e.self_cpu_time_total / N, e.cpu_time_total / N))
print("sum of self CPU time per iteration: %.0f us" % (sum(e.self_cpu_time_total for e in ev) / N))
