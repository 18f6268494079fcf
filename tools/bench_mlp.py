"""Micro-benchmark of the fp32 MLP launches at the shape of the background field (models/nerfhash.py:44-56):
vsa_mlp_fwd / vsa_mlp_bwd of 51-64-64-64-65 and 80-64-64-3 over 65 536 x 32 samples.  One line per network.
usage: python tools/bench_mlp.py [--rows 2097152] [--iters 5]"""
import argparse
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from volsurfs_amd.models import MLP          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=65536 * 32)
ap.add_argument("--iters", type=int, default=5)
args = ap.parse_args()
torch.manual_seed(0)
for name, din, dims in (("feat_and_density", 51, [64, 64, 64, 65]), ("rgb", 80, [64, 64, 3])):
    m = MLP(din, dims, last_layer_linear=True).cuda()
    # rows padded to a multiple of 4 floats, as the encoder / the field head hand them over (models.padded_rows)
    x = torch.randn(args.rows, (din + 3) // 4 * 4, device="cuda")[:, :din].detach().requires_grad_(True)
    flops = 2 * sum(a * b for a, b in zip([din] + dims[:-1], dims)) * args.rows
    tf, tb = [], []
    for it in range(args.iters + 2):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        y = m(x)
        e[1].record()
        g = torch.ones_like(y)          # (keeps y's padded rows)
        torch.cuda.synchronize()
        e[1].record()
        y.backward(g)
        e[2].record()
        torch.cuda.synchronize()
        if it >= 2:
            tf.append(e[0].elapsed_time(e[1]))
            tb.append(e[1].elapsed_time(e[2]))
        x.grad = None
    f, b = sorted(tf)[len(tf) // 2], sorted(tb)[len(tb) // 2]
    print(json.dumps({"net": name, "rows": args.rows, "fwd_ms": round(f, 3), "bwd_ms": round(b, 3),
                      "fwd_TFLOPs": round(flops / f / 1e9, 1), "bwd_TFLOPs": round(2 * flops / b / 1e9, 1)}))
