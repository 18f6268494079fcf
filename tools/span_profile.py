"""Diagnostic (NT_SPAN build): busy time along the block index (= along the cost axis)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volsurfs_amd import _lib
from volsurfs_amd.pipeline import KShellPipeline
p = KShellPipeline.synthetic()
for _ in range(3):
    p.step()
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
for tu, ks in (("mlp", {0: ("mlp_fwd", 768), 1: ("mlp_bwd", 256)}),
               ("encode", {0: ("enc_fwd_d", 256), 1: ("enc_fwd_h", 256), 2: ("enc_bwd_d", 256), 3: ("enc_bwd_h", 256)})):
    buf = np.zeros(4 * 2048 * 3, dtype=np.uint64)
    getattr(L, "vsa_span_read_" + tu)(buf.ctypes.data_as(ctypes.c_void_p))
    r = buf.reshape(4, 2048, 3)
    for i, (k, G) in ks.items():
        a = r[i][:G]
        busy = (a[:, 1].astype(np.int64) - a[:, 0].astype(np.int64)) / 100.0
        g = busy.reshape(32, -1).mean(1)
        print(k, "mean %.1f" % busy.mean(), "| 32 bins along the axis:", " ".join("%.0f" % v for v in g))
