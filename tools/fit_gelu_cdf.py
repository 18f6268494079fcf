"""Coefficients of csrc/gelu_fast.h: Phi(-u) = 0.5 erfc(u / sqrt 2) = 2^Q(u), u in [0, U], Q a polynomial.

One branch for the whole range (the exact GELU of torch.nn.GELU needs Phi to ABSOLUTE accuracy: GELU = z Phi(z),
GELU' = Phi(z) + z phi(z)), fitted by Lawson-reweighted least squares on Chebyshev nodes with weight Phi (absolute error of
Phi), then checked in emulated fp32 Horner arithmetic.  Run: python tools/fit_gelu_cdf.py"""
import numpy as np
from scipy.special import erfc

U, DEG = 8.0, 9


def fit(deg=DEG, n=6000):
    u = 0.5 * U * (1 - np.cos(np.pi * (np.arange(n) + 0.5) / n))
    phi = 0.5 * erfc(u / np.sqrt(2.0))
    y = np.log2(phi)
    V = np.vander(u, deg + 1, increasing=True)
    lw = np.ones(n)
    for _ in range(80):
        w = phi * lw
        c, *_ = np.linalg.lstsq(V * w[:, None], y * w, rcond=None)
        err = np.abs(np.exp2(V @ c) - phi)
        lw = lw * (0.2 + err / err.max())
        lw /= lw.mean()
    return c, err.max()


def check(c):
    z = np.linspace(-12, 12, 2000001).astype(np.float32)
    u = np.minimum(np.abs(z), np.float32(U))
    c32 = c.astype(np.float32)
    r = np.full_like(u, c32[-1])
    for k in range(len(c) - 2, -1, -1):
        r = (r.astype(np.float64) * u + c32[k]).astype(np.float32)       # fma: one rounding
    e = np.exp2(r.astype(np.float64)).astype(np.float32)
    cdf = np.where(z < 0, e, np.float32(1) - e)
    ref = 0.5 * erfc(-z.astype(np.float64) / np.sqrt(2.0))
    gelu = z.astype(np.float64) * cdf
    return np.abs(cdf - ref).max(), np.abs(gelu - z.astype(np.float64) * ref).max()


if __name__ == "__main__":
    c, e = fit()
    print("fit max abs err (f64):", e)
    print("fp32 Horner: max |Phi err| = %.3e, max |GELU err| = %.3e" % check(c))
    print(", ".join("%.9ef" % v for v in c))
