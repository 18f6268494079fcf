"""Per candidate side stream: the step time while a step's three flag waits are pending on it (profiles/r05/dp_schedule_one_gpu.txt (5)):
streams that land on another hardware queue than the step's slow it by 0.9 ms, those that share its queue do not (and overlap nothing)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from volsurfs_amd.pipeline import KShellPipeline
from volsurfs_amd.parallel import StepSignals
torch.cuda.set_device(0)
pipe = KShellPipeline.synthetic(K=5, subdiv=6, res=800, device="cuda", seed=42)
for _ in range(3):
    pipe.step()
sg = StepSignals(5, "cuda", wait_mode=int(os.environ.get("DPW", "0")))
pipe.capture_graph_split(dp=sg)
cands = [torch.cuda.Stream() for _ in range(10)] + [torch.cuda.Stream(priority=-1) for _ in range(4)]

def timed(fn, steps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3

def make(side):
    def step():
        pipe.replay_prefix(); pipe.replay_rest()
        if side is not None:
            with torch.cuda.stream(side):
                sg.stream_wait(sg.W, sg.epoch_host); sg.stream_wait(0, sg.epoch_host); sg.stream_wait(1, sg.epoch_host)
    return step
print("none %.4f" % timed(make(None)))
for rnd in range(2):
    for i, c in enumerate(cands):
        print("round %d cand %2d %s %.4f" % (rnd, i, "high" if i >= 10 else "norm", timed(make(c))))
