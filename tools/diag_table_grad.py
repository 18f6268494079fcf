#!/usr/bin/env python3
"""Where do the hash-table gradient outliers of tests/test_parity_report.py come from?
(max 4.8e-3 of the tensor's largest entry against the fp32 oracle.)

Splits the kernel's table gradient error into its two stages, on the kernel's own data:
  (a) kernel:   vsa_nt_encode_bwd's fixed-point LDS accumulation of the f16 dF the MLP backward wrote
  (b) exact(dF): the same scatter  grad[idx_c] += w_c * dF  of THOSE f16 dF in float64 on the CPU
so  (a) - (b) = the accumulation's own error (fixed point, rounding of each contribution), and
what is left against the fp32 oracle is already in dF (the fp16 gradient chain of the MLP backward).
Prints per-level maxima of |(a) - (b)| relative to the tensor's largest entry.
Run on the GPU box:  python tools/diag_table_grad.py [K subdiv res]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import tcnn_like  # noqa: E402


def main():
    K, subdiv, res = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (3, 3, 56)
    from test_parity_report import _pipe
    pipe = _pipe(K, subdiv, res)
    bank = pipe.bank
    pipe.step()
    torch.cuda.synchronize()
    gt = bank.tables.grad.cpu().double()
    dF = bank.features_level_major().cpu()            # [type, level, slot, 2] f16: dF * grad_scale after the backward
    seg = bank.seg_start.cpu().numpy()
    slot_xy = bank.slot_xy.cpu()
    geom = tcnn_like.GridGeometry()
    worst = np.zeros(geom.n_levels)
    worst_small = 0.0
    for s in range(K):
        for typ in range(2):
            for d in range(4):
                x = bank.tex_index(s, typ, d)
                if not bank.tex_channels(x):
                    continue
                a, b = seg[s * 4 + d], seg[s * 4 + d + 1]
                ref = torch.zeros(geom.offset[-1], 2, dtype=torch.float64)
                for l in range(geom.n_levels):
                    idx, w = tcnn_like._grid_cells(geom, l, slot_xy[a:b])
                    g = dF[typ, l, a:b].double() / pipe.grad_scale
                    for c in range(4):
                        ref.index_add_(0, geom.offset[l] + idx[c], w[c].double()[:, None] * g)
                err = (gt[x] - ref).abs() / ref.abs().max()
                for l in range(geom.n_levels):
                    worst[l] = max(worst[l], float(err[geom.offset[l]:geom.offset[l + 1]].max()))
    print("accumulation error |kernel - exact scatter of the kernel's f16 dF| / tensor max, per level:")
    print(" ".join("%.1e" % v for v in worst))
    print("max over levels: %.2e" % worst.max())


if __name__ == "__main__":
    main()
