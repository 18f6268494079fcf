"""Diagnostic (tools/build_variant.sh span "-DTRACE_SPAN", library swapped in by tools/span_ab.sh-style
copy): begin / end / node visits of every wave of trace_q_kernel in the last frame -> is the kernel a
latency-bound steady state or a tail of a few long waves?
usage: python tools/trace_span.py [--res 800] [--noise 0.0]"""
import argparse, ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volsurfs_amd import _lib
from volsurfs_amd.pipeline import KShellPipeline

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=800)
ap.add_argument("--noise", type=float, default=0.0)
ap.add_argument("--order", default="identity", help="identity | ljf (longest first by the previous frame's trips: "
                "the upper bound of any cost predictor) | ljf8 | shell-fast | reverse")
args = ap.parse_args()
p = KShellPipeline.synthetic(res=args.res, noise=args.noise)
p.tracer.cost_feedback = False          # the stamps are in the stateless kernel
for _ in range(3):
    p.step()
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
W = 1 << 17
buf = np.zeros(W * 3, dtype=np.uint64)
L.vsa_span_read_trace(buf.ctypes.data_as(ctypes.c_void_p))
if args.order != "identity":
    r0 = buf.reshape(W, 3)
    nw = int(((r0[:, 0] > 0) & (r0[:, 1] > 0)).sum())
    rounds0 = (r0[:nw, 2] >> np.uint64(48)).astype(np.int64)
    if args.order == "ljf":
        order = np.argsort(-rounds0, kind="stable")
    elif args.order == "ljf8":          # 8 classes only
        order = np.argsort(-np.minimum(rounds0 // 8, 7), kind="stable")
    elif args.order == "reverse":
        order = np.arange(nw)[::-1]
    elif args.order == "shell-fast":
        K = 5
        order = np.arange(nw).reshape(K, nw // K).T.reshape(-1)
    order = np.ascontiguousarray(order.astype(np.int32))
    L.vsa_span_set_order(order.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(nw))
    for _ in range(2):
        p.step()
    torch.cuda.synchronize()
    L.vsa_span_read_trace(buf.ctypes.data_as(ctypes.c_void_p))
r = buf.reshape(W, 3)
ok = (r[:, 0] > 0) & (r[:, 1] > 0)
a = r[ok]
t0 = a[:, 0].min()
beg = (a[:, 0] - t0).astype(np.int64) / 100.0       # us (100 MHz)
end = (a[:, 1] - t0).astype(np.int64) / 100.0
dur = end - beg
rounds = (a[:, 2] >> np.uint64(48)).astype(np.int64)
outer = ((a[:, 2] >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64)
visits = (a[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
span = end.max()
print(f"waves {len(a)}  kernel span {span:.1f} us  sum of wave time {dur.sum() / 1e3:.2f} ms "
      f"-> mean resident waves {dur.sum() / span:.0f} ({dur.sum() / span / 256:.1f} per CU)")
print("wave life us: " + " ".join(f"p{q} {np.percentile(dur, q):.1f}" for q in (10, 50, 75, 90, 99, 100)))
live = rounds > 1
print(f"waves with more than one round: {live.sum()} ({100 * live.mean():.0f} %), their share of the wave time "
      f"{100 * dur[live].sum() / dur.sum():.0f} %; rounds p50 {np.percentile(rounds[live], 50):.0f} p99 "
      f"{np.percentile(rounds[live], 99):.0f} max {rounds.max()}; lane utilisation of the walk "
      f"{visits.sum() / (64.0 * rounds.sum()):.2f}; us per round {dur[live].sum() / rounds[live].sum():.3f}")
A = np.stack([rounds, outer, np.ones_like(rounds)], 1).astype(np.float64)
for name, sel in (("all waves", np.ones(len(a), bool)), ("waves ending in the last third of the kernel", end > 2 * span / 3),
                  ("waves that began in the first 20 us", beg < 20)):
    if sel.sum() > 10:
        coef = np.linalg.lstsq(A[sel], dur[sel], rcond=None)[0]
        print(f"fit over {name} ({int(sel.sum())}): life = {coef[0]:.3f} us x inner rounds + {coef[1]:.3f} us x leaf phases + {coef[2]:.2f} us"
              f"   (leaf phases per wave p50 {np.percentile(outer[sel], 50):.0f} max {outer[sel].max()})")
# resident waves over time
edges = np.linspace(0, span, 21)
for lo, hi in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (lo + hi)
    n = int(((beg <= mid) & (end > mid)).sum())
    started = int(((beg >= lo) & (beg < hi)).sum())
    print(f"  t {mid:6.1f} us  resident {n:5d} ({n / 256:5.1f}/CU)  started in bin {started:5d}")
