"""vsa_trace_q_budgeted: how much is handed over at a given trip budget (items / ray records in the
workspace counters) and the time of the three launches together.
usage: python tools/trace_budget_counts.py [--res 800] [--budgets 8 16 24 32 48 1000]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volsurfs_amd.pipeline import KShellPipeline

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=800)
ap.add_argument("--noise", type=float, default=0.0)
ap.add_argument("--budgets", type=int, nargs="+", default=[0, 8, 16, 24, 32, 48, 64, 96, 1000])
args = ap.parse_args()
p = KShellPipeline.synthetic(res=args.res, noise=args.noise)
p.step()
o, d = p._o_t, p._d_t
rt = p.tracer
rt.cost_feedback = False
pairs = o.shape[0] * rt.nr_meshes
for b in args.budgets:
    rt.round_budget = b
    ts = []
    for _ in range(7):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(200000)
        a.record()
        rt.trace_all(o, d)
        e.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(e))
    ts.sort()
    if b > 0:
        c = rt._ws[0][:8].view(torch.int32).cpu().tolist()
        print(f"budget {b:5d}: {ts[3]:.3f} ms   items {c[0]:8d}  rays handed over {c[1]:8d} ({100.0 * c[1] / pairs:.2f} % of the pairs, "
              f"{c[0] / max(c[1], 1):.1f} items each)")
    else:
        print(f"one pass    : {ts[3]:.3f} ms")
        rt.cost_feedback = True
        ts = []
        for _ in range(9):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(200000)
            a.record()
            rt.trace_all(o, d)
            e.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(e))
        print(f"one pass, launch order from the previous call's cost: first call {ts[0]:.3f} ms, then {sorted(ts[2:])[3]:.3f} ms")
        rt.cost_feedback = False
