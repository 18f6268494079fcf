"""Diagnostic (NT_STAMP build only): per-workgroup timeline of nt_mlp_bwd_pc_kernel."""
import ctypes, collections, sys
import numpy as np, torch
sys.path.insert(0, ".")
from volsurfs_amd import _lib
from volsurfs_amd.pipeline import KShellPipeline
p = KShellPipeline.synthetic()
for _ in range(3):
    p.step()
torch.cuda.synchronize()
buf = np.zeros(16384 * 20, dtype=np.uint64)
L = _lib.lib()
rc = L.vsa_debug_read(buf.ctypes.data_as(ctypes.c_void_p))
stage = buf[16384 * 12:].reshape(-1, 8)
role = buf[16384 * 8:16384 * 12].reshape(-1, 4)
r = buf[:16384 * 8].reshape(-1, 8)
role = role[r[:, 7] > 0]
stage = stage[r[:, 7] > 0]
r = r[r[:, 7] > 0]
t0 = int(r[:, 0].min()); t1 = int(r[:, 1].max())
print("active WGs", len(r), "kernel span us", (t1 - t0) / 100.0)
dur = (r[:, 1] - r[:, 0]).astype(np.float64) / 100
print("WG dur us mean %.1f min %.1f max %.1f" % (dur.mean(), dur.min(), dur.max()))
print("iters hist", sorted(collections.Counter(r[:, 3].tolist()).items()))
print("cycles stage/loop/epi mean", r[:, 4].mean(), r[:, 5].mean(), r[:, 6].mean(), " loop/iter", (r[:, 5] / np.maximum(r[:, 3], 1)).mean())
it_ = np.maximum(r[:, 3], 1).astype(np.float64)
big = r[:, 3] > 100
print("per tile (runs > 100 trips): producer work %.0f wait %.0f | consumer work %.0f wait %.0f cycles" % tuple((role[big, k] / it_[big]).mean() for k in range(4)))
print("consumer stages per tile (dH2+mask | dH1 chain | dW2 (+dW3) | dX chain | dW1; the rest = epilogue):", " ".join("%.0f" % (stage[big, k] / it_[big]).mean() for k in range(5)))
# loop cycles per trip by (type, degree) of the LAST run of each workgroup (runs of > 100 trips)
by = collections.defaultdict(list)
for row in r:
    if row[3] > 100:
        tex = int(row[7]) - 1
        by[((tex // 4) & 1, tex % 4)].append(float(row[5]) / float(row[3]))
print("loop cycles per trip by (type, degree):", {k: round(float(np.mean(v))) for k, v in sorted(by.items())})
cu = collections.defaultdict(list)
for row in r:
    hw = int(row[2]) & 0xffffffff; xcc = int(row[2]) >> 32
    cu[(xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)].append(row)
busy = np.array([sum(int(x[1] - x[0]) for x in v) / 100 for v in cu.values()])
print("distinct CUs", len(cu), "busy us mean %.1f min %.1f max %.1f" % (busy.mean(), busy.min(), busy.max()))
xw = collections.Counter()
for row in r:
    xw[int(row[2]) >> 32] += int(row[3])
print("iters per xcc", sorted(xw.items()))
# concurrency over time
ev = sorted([(int(x[0]), 1) for x in r] + [(int(x[1]), -1) for x in r])
c = 0; last = t0; area = 0
for t, d in ev:
    area += c * (t - last); last = t; c += d
print("mean concurrent WGs %.1f" % (area / (t1 - t0)))
# start time of last WG, end of first round
st = np.sort(r[:, 0] - t0) / 100
print("WG start times us: 256th %.1f, median %.1f, last %.1f" % (st[min(255, len(st) - 1)], np.median(st), st[-1]))
