"""How much of the traversal's time is cold misses?  The same trace launch (a) back to back (tree warm in
L2 / the memory-side cache), (b) after a pass that streams 1 GB (tree evicted: what a pipeline step sees).
usage: python tools/trace_hot_cold.py [--res 800] [--noise 0.0]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volsurfs_amd.pipeline import KShellPipeline

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=800)
ap.add_argument("--noise", type=float, default=0.0)
args = ap.parse_args()
p = KShellPipeline.synthetic(res=args.res, noise=args.noise)
p.tracer.cost_feedback = False
p.step()
o, d = p._o_t, p._d_t
big = torch.empty(256 << 20, dtype=torch.float32, device="cuda")


def timed(prep, n=10):
    ts = []
    for _ in range(n):
        prep()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(200000)
        a.record()
        p.tracer.trace_all(o, d)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


print(f"trace back to back (warm): {timed(lambda: p.tracer.trace_all(o, d)):.3f} ms")
print(f"trace after a 1 GB fill  (cold): {timed(lambda: big.fill_(1.0)):.3f} ms")
# warm only the tree: read nodes and triangles once
def touch():
    big.fill_(1.0)
    (p.tracer.qnodes.sum() + 0).item() if False else p.tracer.qnodes.sum()
    p.tracer.tris.sum()
print(f"trace after the fill + one streaming read of nodes and triangles: {timed(touch):.3f} ms")
def touch_nodes():
    big.fill_(1.0)
    p.tracer.qnodes.sum()
print(f"trace after the fill + one streaming read of the nodes only: {timed(touch_nodes):.3f} ms")
