"""Diagnostic (NT_SPAN build): least-squares fit of per-(type, degree) unit costs of the MLP
kernels from the per-workgroup busy times and the static partition of the cost axis."""
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
buf = np.zeros(4 * 2048 * 3, dtype=np.uint64)
L.vsa_span_read_mlp(buf.ctypes.data_as(ctypes.c_void_p))
r = buf.reshape(4, 2048, 3).astype(np.int64)
bank = p.bank
seg = bank.seg_start.cpu().numpy()
K, D = bank.K, 4
texs = []                                    # (tex, units, class)
for tex in range(K * 8):
    shell, typ, deg = tex // 8, (tex // 4) & 1, tex % 4
    if not bank.tex_channels(tex):
        continue
    n = int(seg[shell * 4 + deg + 1] - seg[shell * 4 + deg])
    if n > 0:
        texs.append((tex, (n + 31) // 32, typ * 4 + deg))


def partition(G, ovh):
    """units of each class (and pieces) per workgroup, as nt_for_each_piece<32> splits them"""
    T = sum(ovh + u for _, u, _ in texs)
    A = np.zeros((G, 9))
    for w in range(G):
        lo, hi = T * w // G, T * (w + 1) // G
        c0 = 0
        for _, u, cls in texs:
            t0 = c0 + ovh
            c0 = t0 + u
            a, b = max(lo - t0, 0), min(hi - t0, u)
            if b > a:
                A[w, cls] += b - a
                A[w, 8] += 1
    return A


for kid, name, G, ovh in ((0, "nt_mlp_fwd", 768, 8), (1, "nt_mlp_bwd", 256, 44)):
    a = r[kid][:G]
    busy = (a[:, 1] - a[:, 0]) / 100.0
    A = partition(G, ovh)
    c, res, *_ = np.linalg.lstsq(A, busy, rcond=None)
    pred = A @ c
    print(name, "busy mean %.1f std %.1f | residual std %.1f" % (busy.mean(), busy.std(), (busy - pred).std()))
    base = c[0] if c[0] > 0 else c[:8][c[:8] > 0].min()
    for cls in range(8):
        if A[:, cls].sum() > 0:
            print("   %s deg %d: %.4f us/unit (x%.2f), units %d" % ("rgb  " if cls < 4 else "alpha", cls % 4, c[cls], c[cls] / base, A[:, cls].sum()))
    print("   per piece: %.2f us" % c[8])
    xcc = (a[:, 2] >> 32) & 0xf
    order = np.argsort(-busy)
    for w in list(order[:6]) + list(order[-3:]):
        print("   WG %4d xcd %d busy %.1f pred %.1f pieces %d units %s" % (w, xcc[w], busy[w], pred[w], A[w, 8], A[w, :8].astype(int).tolist()))

# ---- encode kernels: classes = plane (level, or level x feature) [x degree], + per piece
buf = np.zeros(4 * 2048 * 3, dtype=np.uint64)
L.vsa_span_read_encode(buf.ctypes.data_as(ctypes.c_void_p))
r = buf.reshape(4, 2048, 3).astype(np.int64)
plan = bank.plan
nl = int(plan.n_levels)
lh = next((l for l in range(nl) if int(plan.level_size[l]) >= 32768 and (int(plan.level_res[l]) + 1) ** 2 > int(plan.level_size[l])), nl)
LDS_ENTRIES = 32768
both = [int(plan.level_size[l]) * 2 <= LDS_ENTRIES for l in range(nl)]
etex = []
for tex in range(K * 8):
    shell, typ, deg = tex // 8, (tex // 4) & 1, tex % 4
    if not bank.tex_channels(tex):
        continue
    n = int(seg[shell * 4 + deg + 1] - seg[shell * 4 + deg])
    if n > 0:
        etex.append((tex, (n + 255) // 256, deg))


def epartition(G, ovh, n_planes, by_deg):
    T = sum(ovh + u for _, u, _ in etex)
    ncls = n_planes * (4 if by_deg else 1)
    A = np.zeros((G, ncls + 1))
    total = T * n_planes
    for w in range(G):
        lo, hi = total * w // G, total * (w + 1) // G
        for pl in range(lo // T, (hi - 1) // T + 1):
            c0 = pl * T
            for _, u, deg in etex:
                t0 = c0 + ovh
                c0 = t0 + u
                a, b = max(lo - t0, 0), min(hi - t0, u)
                if b > a:
                    A[w, pl * 4 + deg if by_deg else pl] += b - a
                    A[w, ncls] += 1
    return A


cfgs = ((0, "enc_fwd dense", 8, lh), (1, "enc_fwd hashed", 32, nl - lh),
        (2, "enc_bwd dense", 64, sum(1 if both[l] else 2 for l in range(lh))), (3, "enc_bwd hashed", 64, 2 * (nl - lh)))
for kid, name, ovh, n_planes in cfgs:
    a = r[kid][:256]
    busy = (a[:, 1] - a[:, 0]) / 100.0
    for by_deg in (False, True):
        A = epartition(256, ovh, n_planes, by_deg)
        c, *_ = np.linalg.lstsq(A, busy, rcond=None)
        pred = A @ c
        print(name, "by plane" + (" x degree" if by_deg else ""), "busy mean %.1f std %.1f | residual std %.1f | per piece %.2f us" %
              (busy.mean(), busy.std(), (busy - pred).std(), c[-1]))
        if not by_deg:
            print("   us per unit by plane:", " ".join("%.3f" % v for v in c[:-1]))
        else:
            for pl in range(n_planes):
                print("   plane %2d by degree:" % pl, " ".join("%.3f" % v for v in c[pl * 4:pl * 4 + 4]))
