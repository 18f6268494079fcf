"""Diagnostic (make -C volsurfs_amd/csrc EXTRA=-DNT_SPAN): begin / end of every workgroup of the
persistent kernels in the last bench step -> how evenly the cost axis split the work."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volsurfs_amd import _lib
from volsurfs_amd.pipeline import KShellPipeline
import os
if os.environ.get("SPAN_RAYS"):        # a training batch: that many random rays of the 800 x 800 view (tools/wg_timeline_train.py)
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import nested_shells
    n = int(os.environ["SPAN_RAYS"])
    o, d = pinhole_rays(800, 800, focal=1111.1, cam_pos=(0.0, 0.0, -1.5))
    g = torch.Generator(device="cuda").manual_seed(0)
    idx = torch.randperm(o.shape[0], device="cuda", generator=g)[:n]
    p = KShellPipeline(nested_shells(K=5, subdiv=6), o[idx].contiguous(), d[idx].contiguous(),
                       torch.rand(n, 3, device="cuda", generator=g))
else:
    p = KShellPipeline.synthetic(res=int(os.environ.get("SPAN_RES", "800")))
for _ in range(3):
    p.step()
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
names = {"mlp": ["nt_mlp_fwd", "nt_mlp_bwd", None, None],
         "encode": ["encode_fwd dense", "encode_fwd hashed", "encode_bwd dense", "encode_bwd hashed"]}
for tu, ks in names.items():
    buf = np.zeros(4 * 2048 * 3, dtype=np.uint64)
    getattr(L, "vsa_span_read_" + tu)(buf.ctypes.data_as(ctypes.c_void_p))
    r = buf.reshape(4, 2048, 3)[:, :, :2]
    for i, k in enumerate(ks):
        if k is None:
            continue
        a = r[i][(r[i][:, 0] > 0) & (r[i][:, 1] > 0)].astype(np.int64)
        if not len(a):
            continue
        t0, t1 = a[:, 0].min(), a[:, 1].max()
        dur = (a[:, 1] - a[:, 0]) / 100.0
        end = (a[:, 1] - t0) / 100.0
        print(f"{k:20s} WGs {len(a):5d} span {(t1 - t0) / 100.0:7.1f} us | WG busy mean {dur.mean():7.1f} "
              f"min {dur.min():7.1f} max {dur.max():7.1f} | end p10 {np.percentile(end, 10):7.1f} "
              f"p50 {np.percentile(end, 50):7.1f} p90 {np.percentile(end, 90):7.1f} | efficiency {dur.mean() / ((t1 - t0) / 100.0):.2f}")
