#!/usr/bin/env python3
"""Feasibility check for an eigen-free staged path ("next"): in observation space the analysis needs
g(M) q for a handful of vectors q = Z x'_v, Z d (13 right-hand sides), M = Z Z^T + c I, with the three smooth functions of
csrc/letkf_staged.hip's apply stage (T: -sqrt(k-1)/(sqrt(c) sqrt(L) (sqrt(c)+sqrt(L))), Pa: -1/(c L), w-bar: 1/L).
cond(M) is small because of the shift c = (k-1)/rho, so a Chebyshev expansion on [c, c + bound(lambda_max)] converges
fast.  Prints the degree that reaches 1e-13 relative for (a) the exact lambda_max, (b) the Frobenius bound |Z Z^T|_F
(free: the Gram stage has the matrix), and the flops against the 9 n^3 per-sweep-count Jacobi."""
import numpy as np
rng = np.random.default_rng(3)

def cheb_apply(M, Q, f, a, b, m):
    n = M.shape[0]
    j = np.arange(m + 1)
    x = np.cos(np.pi * (j + 0.5) / (m + 1))                     # Chebyshev-Gauss nodes
    fx = f(0.5 * (b - a) * x + 0.5 * (b + a))
    cfs = np.array([2.0 / (m + 1) * np.sum(fx * np.cos(np.pi * i * (j + 0.5) / (m + 1))) for i in range(m + 1)])
    cfs[0] *= 0.5
    Mt = (2.0 * M - (a + b) * np.eye(n)) / (b - a)
    T0, T1 = Q, Mt @ Q
    Y = cfs[0] * T0 + cfs[1] * T1
    for i in range(2, m + 1):
        T0, T1 = T1, 2.0 * Mt @ T1 - T0
        Y += cfs[i] * T1
    return Y

for n, k in [(200, 320), (200, 1000), (110, 320)]:
    Y = rng.standard_normal((n, k)) * 2.0; Y -= Y.mean(1, keepdims=True)
    w = np.exp(-0.5 * rng.uniform(0, 1, n) * 12) / 9.0
    Z = np.sqrt(w)[:, None] * Y
    S = Z @ Z.T; c = k - 1.0; M = S + c * np.eye(n)
    lam, U = np.linalg.eigh(M)
    Q = Z @ rng.standard_normal((k, 13))
    gT = lambda L: -np.sqrt(k - 1.0) / (np.sqrt(c) * np.sqrt(L) * (np.sqrt(c) + np.sqrt(L)))
    ref = U @ (gT(lam)[:, None] * (U.T @ Q))
    for name, bound in [("exact", lam[-1] - c), ("frobenius", np.linalg.norm(S, 'fro'))]:
        for m in range(4, 200):
            err = np.abs(cheb_apply(M, Q, gT, c, c + bound, m) - ref).max() / np.abs(ref).max()
            if err < 1e-13: break
        print(f"n={n} k={k} bound={name:9s} cond={(c + bound) / c:5.2f} degree={m:3d} err={err:.1e} "
              f"flops={m * 2 * n * n * 13 / 1e6:6.1f} M  (Jacobi 8 sweeps ~ {8 * 9 * n**3 / 8 / 1e6:6.1f} M)")
