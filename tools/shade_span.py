"""Diagnostic (tools/build_variant.sh sspan "-DSHADE_SPAN", library swapped in): begin / end of every
workgroup of nt_shade_bwd in the last frame -> resident workgroups over time, busy-time percentiles by shell.
usage: python tools/shade_span.py [--res 800]"""
import argparse, ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volsurfs_amd import _lib
from volsurfs_amd.pipeline import KShellPipeline

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=800)
args = ap.parse_args()
p = KShellPipeline.synthetic(res=args.res)
for _ in range(3):
    p.step()
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
W = 1 << 16
buf = np.zeros(W * 2, dtype=np.uint64)
L.vsa_span_read_shade(buf.ctypes.data_as(ctypes.c_void_p))
r = buf.reshape(W, 2)
ok = (r[:, 0] > 0) & (r[:, 1] > 0)
a = r[ok]
t0 = a[:, 0].min()
beg = (a[:, 0] - t0).astype(np.int64) / 100.0
end = (a[:, 1] - t0).astype(np.int64) / 100.0
dur = end - beg
span = end.max()
print(f"workgroups {len(a)}  kernel span {span:.1f} us  sum of busy time {dur.sum() / 1e3:.2f} ms -> mean resident "
      f"{dur.sum() / span:.0f} ({dur.sum() / span / 256:.1f} per CU)")
print("busy us: " + " ".join(f"p{q} {np.percentile(dur, q):.1f}" for q in (10, 50, 75, 90, 99, 100)))
edges = np.linspace(0, span, 31)
for lo, hi in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (lo + hi)
    n = int(((beg <= mid) & (end > mid)).sum())
    print(f"  t {mid:6.1f} us  resident {n:5d} ({n / 256:5.1f}/CU)  started in bin {int(((beg >= lo) & (beg < hi)).sum()):5d}")
