"""Diagnostic (NT_STAMP build only: make -C volsurfs_amd/csrc EXTRA=-DNT_STAMP): per-workgroup timeline of nt_mlp_bwd_pc_kernel
at the TRAINING batch (34 000 random rays of an 800 x 800 view, ~49 k hits, ~0.7 M unique texels).  Only each workgroup's LAST
run is stamped.  usage: python tools/wg_timeline_train.py [rays]"""
import collections
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from volsurfs_amd import _lib                                  # noqa: E402
from volsurfs_amd.camera import pinhole_rays                   # noqa: E402
from volsurfs_amd.mesh import nested_shells                    # noqa: E402
from volsurfs_amd.pipeline import KShellPipeline               # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 34000
meshes = nested_shells(K=5, subdiv=6)
o, d = pinhole_rays(800, 800, focal=1111.1, cam_pos=(0.0, 0.0, -1.5))
g = torch.Generator(device="cuda").manual_seed(0)
idx = torch.randperm(o.shape[0], device="cuda", generator=g)[:n]
gt = torch.rand(n, 3, device="cuda", generator=g)
p = KShellPipeline(meshes, o[idx].contiguous(), d[idx].contiguous(), gt)
for _ in range(5):
    p.step()
torch.cuda.synchronize()
hits, slots = p.stats()
buf = np.zeros(16384 * 20, dtype=np.uint64)
_lib.lib().vsa_debug_read(buf.ctypes.data_as(ctypes.c_void_p))
r = buf[:16384 * 8].reshape(-1, 8)
role = buf[16384 * 8:16384 * 12].reshape(-1, 4)[r[:, 7] > 0]
r = r[r[:, 7] > 0]
t0, t1 = int(r[:, 0].min()), int(r[:, 1].max())
print("rays", n, "hits", hits, "unique texels", slots)
print("active WGs (last runs)", len(r), "span of the last runs us", (t1 - t0) / 100.0)
dur = (r[:, 1] - r[:, 0]).astype(np.float64) / 100
print("last-run duration us: mean %.1f min %.1f max %.1f" % (dur.mean(), dur.min(), dur.max()))
print("trips of the last run:", sorted(collections.Counter(r[:, 3].tolist()).items()))
cyc = (r[:, 4] + r[:, 5] + r[:, 6]).astype(np.float64)
print("cycles of the last run: staging %.0f  loop %.0f  epilogue %.0f  | cycles per us %.0f" %
      (r[:, 4].mean(), r[:, 5].mean(), r[:, 6].mean(), (cyc / np.maximum(dur, 1e-3)).mean()))
it_ = np.maximum(r[:, 3], 1).astype(np.float64)
print("loop cycles per trip: mean %.0f" % (r[:, 5] / it_).mean(), " producer work / wait per trip %.0f / %.0f" %
      ((role[:, 0] / it_).mean(), (role[:, 1] / it_).mean()))
st = np.sort(r[:, 0] - t0) / 100
en = np.sort(r[:, 1] - t0) / 100
print("last-run start us: median %.1f max %.1f; end us: median %.1f max %.1f" % (np.median(st), st[-1], np.median(en), en[-1]))
# events around the launch itself
_lib.kernel_events = {}
for _ in range(5):
    p.step()
print({k: round(v, 4) for k, v in _lib.kernel_ms().items() if "mlp_bwd" in k or "encode_bwd" in k or "trace" in k or "compact" in k})
