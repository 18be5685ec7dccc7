"""Developer tool (CPU): the coefficients of csrc/common.h gelu_pk2 — Phi(x) = 1/2 + x Q(x^2) on |x| <= L, Q of degree `deg`, fitted by
iteratively re-weighted least squares (Lawson) so that the maximum ABSOLUTE error of x * Phi(x) is minimised; evaluated in float32 the
way the kernel does (Horner in x^2, x clamped into [-L, L] for Phi only).   python tools/gelu_fit.py [L] [deg]"""
import sys

import numpy as np
from scipy.special import erf


def fit(L, deg, iters=60):
    xs = np.linspace(1e-4, L, 20001)
    s = xs * xs
    target = (0.5 * (1 + erf(xs / np.sqrt(2))) - 0.5) / xs
    V = np.vander(s / L**2, deg + 1, increasing=True)
    w = np.ones_like(xs)
    for _ in range(iters):
        W = np.sqrt(w) * s  # the error of x*Phi is x^2 * dQ
        c, *_ = np.linalg.lstsq(V * W[:, None], target * W, rcond=None)
        err = np.abs((V @ c - target) * s)
        w = w * (err / err.max() + 1e-3)
        w /= w.sum()
    return c / (L**2) ** np.arange(deg + 1)


def check(mono, L, lim=12.0):
    x = np.linspace(-lim, lim, 1200001).astype(np.float32)
    xc = np.clip(x, -L, L).astype(np.float32)
    s = (xc * xc).astype(np.float32)
    q = np.float32(mono[-1]) * np.ones_like(s)
    for c in mono[-2::-1]:
        q = (q * s + np.float32(c)).astype(np.float32)
    g = (x * (xc * q + np.float32(0.5)).astype(np.float32)).astype(np.float32)
    x64 = x.astype(np.float64)
    e = np.abs(g - x64 * 0.5 * (1 + erf(x64 / np.sqrt(2))))
    return e[np.abs(x) <= L].max(), e.max()


if __name__ == "__main__":
    L = float(sys.argv[1]) if len(sys.argv) > 1 else 4.25
    deg = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    m = fit(L, deg)
    inside, anywhere = check(m, L)
    print(f"L = {L}, degree {deg}: max |gelu - exact| {inside:.2e} on |x| <= L, {anywhere:.2e} on |x| <= 12")
    print("coefficients (constant term first):", [float(np.float32(c)) for c in m])
