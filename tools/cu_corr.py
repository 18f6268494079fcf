"""Diagnostic (NT_SPAN build): is a CU that is slow in one persistent kernel slow in the others?"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volsurfs_amd import _lib
from volsurfs_amd.pipeline import KShellPipeline
p = KShellPipeline.synthetic()
L = ctypes.CDLL(_lib.LIB_PATH)


def snap():
    p.step()
    torch.cuda.synchronize()
    out = {}
    for tu, ks in (("mlp", {1: "mlp_bwd"}), ("encode", {0: "enc_fwd_d", 1: "enc_fwd_h", 2: "enc_bwd_d", 3: "enc_bwd_h"})):
        buf = np.zeros(4 * 2048 * 3, dtype=np.uint64)
        getattr(L, "vsa_span_read_" + tu)(buf.ctypes.data_as(ctypes.c_void_p))
        r = buf.reshape(4, 2048, 3)
        for i, k in ks.items():
            a = r[i][:256]
            busy = (a[:, 1].astype(np.int64) - a[:, 0].astype(np.int64)) / 100.0
            xcc = (a[:, 2] >> np.uint64(32)).astype(int) & 0xf
            hw = (a[:, 2] & np.uint64(0xffffffff)).astype(int)
            cu = xcc * 1000 + ((hw >> 8) & 15) + 16 * ((hw >> 13) & 7) + 128 * ((hw >> 12) & 1)
            out[k] = (busy, cu)
    return out


for _ in range(2):
    p.step()
s1, s2 = snap(), snap()
# same kernel, two steps: per-WG (same block index) and per-CU consistency
for k in s1:
    b1, c1 = s1[k]
    b2, c2 = s2[k]
    same_cu = (c1 == c2).mean()
    print(f"{k:10s} step-to-step corr by block index {np.corrcoef(b1, b2)[0, 1]:.2f} (same CU for a block index: {same_cu:.2f})")
    m1 = dict(zip(c1.tolist(), b1.tolist()))
    m2 = dict(zip(c2.tolist(), b2.tolist()))
    common = sorted(set(m1) & set(m2))
    print(f"{'':10s} corr by CU {np.corrcoef([m1[c] for c in common], [m2[c] for c in common])[0, 1]:.2f}")
# across kernels within one step, by CU
ks = list(s1)
for i in range(len(ks)):
    for j in range(i + 1, len(ks)):
        mi = dict(zip(s1[ks[i]][1].tolist(), s1[ks[i]][0].tolist()))
        mj = dict(zip(s1[ks[j]][1].tolist(), s1[ks[j]][0].tolist()))
        common = sorted(set(mi) & set(mj))
        print(f"{ks[i]} vs {ks[j]}: corr by CU {np.corrcoef([mi[c] for c in common], [mj[c] for c in common])[0, 1]:.2f}")
