"""Diagnostic (NT_SPAN build): busy time of the MLP kernels' workgroups by XCD and by CU."""
import ctypes, os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volsurfs_amd import _lib
from volsurfs_amd.pipeline import KShellPipeline
p = KShellPipeline.synthetic()
for _ in range(3):
    p.step()
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
for tu, ks in (("mlp", ["nt_mlp_fwd", "nt_mlp_bwd"]), ("encode", ["enc_fwd dense", "enc_fwd hashed", "enc_bwd dense", "enc_bwd hashed"])):
    buf = np.zeros(4 * 2048 * 3, dtype=np.uint64)
    getattr(L, "vsa_span_read_" + tu)(buf.ctypes.data_as(ctypes.c_void_p))
    r = buf.reshape(4, 2048, 3)
    for i, k in enumerate(ks):
        a = r[i][(r[i][:, 0] > 0) & (r[i][:, 1] > 0)]
        busy = (a[:, 1].astype(np.int64) - a[:, 0].astype(np.int64)) / 100.0
        start = (a[:, 0].astype(np.int64) - a[:, 0].astype(np.int64).min()) / 100.0
        xcc = (a[:, 2] >> np.uint64(32)).astype(int) & 0xf
        hw = (a[:, 2] & np.uint64(0xffffffff)).astype(int)
        cu = ((hw >> 8) & 15) + 16 * ((hw >> 13) & 7) + 128 * ((hw >> 12) & 1)      # cu_id, se_id, sh_id
        key = xcc * 1000 + cu
        per_cu = collections.Counter(key.tolist())
        print(f"{k}: WGs {len(a)}, distinct CUs {len(per_cu)}, WGs per CU hist {sorted(collections.Counter(per_cu.values()).items())}, start max {start.max():.1f} us")
        print("   busy by XCD:", " ".join(f"{x}:{busy[xcc == x].mean():.1f}" for x in sorted(set(xcc.tolist()))))
        n_on_cu = np.array([per_cu[kk] for kk in key.tolist()])
        print("   busy by #WGs sharing the CU:", " ".join(f"{n}:{busy[n_on_cu == n].mean():.1f}(n={int((n_on_cu == n).sum())})" for n in sorted(set(n_on_cu.tolist()))))
