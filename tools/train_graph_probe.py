"""The graph-replayed training iteration alone (for rocprofv3 --kernel-trace + tools/kernel_timeline.py): eager warm-up until
the dynamic ray count has settled, then `n` replays of trainer.GraphTrainLoop.  usage: python tools/train_graph_probe.py [replays]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402
from volsurfs_amd.mesh import nested_shells                    # noqa: E402
from volsurfs_amd.methods import VolSurfs                      # noqa: E402
from volsurfs_amd.trainer import GraphTrainLoop, train_step_from_reel   # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = torch.device("cuda:0")
torch.manual_seed(42)
m = VolSurfs(nested_shells(K=5, subdiv=6, device=dev), max_rays=1 << 17, nr_warmup_iters=500, seed=42)
m.init_optim()
reel = bench.synthetic_reel(100, 800, dev, seed=42)
n = 512
for it in range(60):
    m.grad_scale = 16.0 * n
    _, n = train_step_from_reel(m, reel, n, jitter_pixels=True, iter_nr=it, is_first_iter=it == 0,
                                target_nr_of_training_samples=49152, sync_losses=False, overlap_optimizer=True)
m.grad_scale = None
loop = GraphTrainLoop(m, reel, n, 49152, iter_nr=60).capture()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(reps):
    loop.step()
torch.cuda.synchronize()
print("it/s", reps / (time.perf_counter() - t0), loop.read())
