"""The cost-feedback launch order of the traversal on a MOVING camera: an orbit around the noisy shells
(noise 0.05: the silhouette's bumps change with the view), one trace launch per frame, the order of frame
i taken from the trips measured in frame i-1.  Prints the mean launch time per step size of the orbit,
with the feedback and with the stateless kernel.
usage: python tools/trace_feedback_motion.py [--res 800] [--frames 24]"""
import argparse, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volsurfs_amd import _lib
from volsurfs_amd.camera import pinhole_rays
from volsurfs_amd.mesh import nested_shells
from volsurfs_amd.raytrace import RayTracer

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=800)
ap.add_argument("--frames", type=int, default=24)
ap.add_argument("--noise", type=float, default=0.05)
args = ap.parse_args()
R = args.res
rt = RayTracer(nested_shells(K=5, subdiv=6, noise=args.noise))
o0, d0 = pinhole_rays(R, R, focal=1111.1 * R / 800.0, cam_pos=(0.0, 0.0, -1.5))
o_t, d_t = torch.empty_like(o0), torch.empty_like(d0)


def frame(theta):
    """rays of the camera orbited by theta around y (and nodding a third of it around x), in tile order"""
    c, s = math.cos(theta), math.sin(theta)
    c2, s2 = math.cos(theta / 3), math.sin(theta / 3)
    ry = torch.tensor([[c, 0, s], [0, 1, 0], [-s, 0, c]], device="cuda", dtype=torch.float32)
    rx = torch.tensor([[1, 0, 0], [0, c2, -s2], [0, s2, c2]], device="cuda", dtype=torch.float32)
    m = rx @ ry
    o, d = (o0 @ m.T).contiguous(), (d0 @ m.T).contiguous()
    _lib.call("vsa_tile_order_rays", o, d, None, o_t, d_t, None, R, R, _lib.stream_ptr())
    return o_t, d_t


def run(step_deg, feedback):
    rt.cost_feedback = feedback
    rt._fb = None
    ts = []
    for i in range(args.frames):
        o, d = frame(math.radians(10.0 + step_deg * i))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(200000)
        a.record()
        rt.trace_all(o, d)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts = ts[2:]                       # frame 0 has no feedback, frame 1 allocates
    return sum(ts) / len(ts)


print(f"{R}x{R}, K = 5 noisy shells (noise {args.noise}), {args.frames} frames per orbit step; mean launch ms")
for step in (0.0, 0.25, 1.0, 2.0, 5.0, 15.0):
    off, on = run(step, False), run(step, True)
    print(f"  orbit step {step:5.2f} deg/frame: stateless {off:.3f}   cost feedback {on:.3f}   ({100 * (on / off - 1):+.0f} %)")
