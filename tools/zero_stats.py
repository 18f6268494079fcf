"""Diagnostic: how many slots receive an exactly-zero feature gradient (per shell, degree, type)."""
import sys
import torch
sys.path.insert(0, ".")
from volsurfs_amd.pipeline import KShellPipeline
p = KShellPipeline.synthetic()
for _ in range(2):
    p.step()
torch.cuda.synchronize()
bank = p.bank
seg = bank.seg_start.cpu().tolist()
dF = bank.features_level_major()      # [type, level, slot, 2] now holding dF
nz = (dF != 0).any(dim=3).any(dim=1)   # [type, slot]
for s in range(bank.K):
    for d in range(4):
        a, b = seg[s * 4 + d], seg[s * 4 + d + 1]
        print("shell %d deg %d slots %7d  nonzero rgb %.3f alpha %.3f" % (
            s, d, b - a, nz[0, a:b].float().mean().item(), nz[1, a:b].float().mean().item()))
