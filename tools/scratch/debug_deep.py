import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from test_raytrace import _chain_mesh
from oracle import raytrace as ort
from volsurfs_amd.mesh import TensorMesh, icosphere
from volsurfs_amd.raytrace import RayTracer
v, f = _chain_mesh()
g = np.random.default_rng(1)
n = 6000
i = g.integers(0, 20, n)
tgt = np.stack([3.0 ** -i, np.zeros(n), np.zeros(n)], 1) + (0.2 * 3.0 ** -i)[:, None] * g.standard_normal((n, 3))
o = np.tile(np.array([[0.2, 0.05, -2.0]]), (n, 1)) + 0.01 * g.standard_normal((n, 3))
d = tgt - o
d /= np.linalg.norm(d, axis=1, keepdims=True)
o, d = o.astype(np.float32), d.astype(np.float32)
ref = ort.trace_bruteforce(v, f, o, d)
for fmt in ("q16", "f32"):
    for leaf in (4, 1, 8):
        rt = RayTracer([TensorMesh(v, f)], node_format=fmt, leaf_size=leaf)
        t, s, uv = rt.trace_all(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda())
        fid = torch.where(s >= 0, rt.slot_face_id[s.clamp(min=0).long()], torch.full_like(s, -1)).cpu().numpy()[0]
        bad = np.nonzero(fid != ref["tri"])[0]
        print(fmt, "leaf", leaf, "depth", rt.max_depth, "mismatches", len(bad))
        for b in bad[:8]:
            print("   ray", b, "cluster", i[b], "got", fid[b], t[0, b].item(), "ref", ref["tri"][b], ref["t"][b],
                  "ref cluster", ref["tri"][b] // 64, "got cluster", fid[b] // 64)
