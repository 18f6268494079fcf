"""Diagnostic (NT_STAMP build, VSA_NT_MLP_BWD=t): cycles per tile and phase of wave 0 of every
workgroup of nt_mlp_bwd_t_kernel, accumulated over the launches of a few frames."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volsurfs_amd import _lib
from volsurfs_amd.pipeline import KShellPipeline
p = KShellPipeline.synthetic(res=800)
for _ in range(4):
    p.step()
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(16384 * 20, dtype=np.uint64)
L.vsa_debug_read(buf.ctypes.data_as(ctypes.c_void_p))
st = buf[16384 * 12:16384 * 12 + 256 * 16].reshape(256, 16).astype(np.float64)
tiles = st[:, 12].sum()
names = ["wait vmcnt", "stage read, dF store, clear, request", "X image + layer 1", "layer 2 (both)", "output + dOut",
         "dW3", "dH2 (both)", "H1^T + dW2", "dH1 (both)", "dW1 + dX + sum|dF|", "-", "loop edge"]
tot = 0.0
for k, n in enumerate(names):
    c = st[:, k].sum() / tiles     # s_memtime ticks
    tot += c
    print(f"{n:40s} {c:8.0f} ticks/tile")
print(f"{'total':40s} {tot:8.0f} ticks/tile   tiles (wave 0) {int(tiles)}")
