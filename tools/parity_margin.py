"""Parity margins of the das loop body against the oracle (GPU): max error / tolerance per case, and mean sweeps.
Used to judge a change of the eigensolver's stopping rule before the tolerance tests would notice it."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_das as T  # noqa: E402


def margin(k, npts, nobs_tot, n_mean, seed, name, warm_run):
    cfg = T.CONFIGS[name]
    c, ref, got, infl, status, trans, transm = T.run_both(k, 11, npts, nobs_tot, n_mean, seed, cfg, want_trans=True,
                                                          warm_run=warm_run)
    nens = c["nens"]
    g = got.reshape(11, nens, npts)
    e = ref["anal"].reshape(11, nens, npts)
    x = c["gues"].reshape(11, nens, npts)
    worst = 0.0
    for v in range(11):
        scale = max(np.abs(x[v, k]).max(), np.abs(x[v, :k]).max())
        worst = max(worst, np.abs(g[v, :k] - e[v, :k]).max() / (1e-10 * scale))
    Tm = trans.cpu().numpy()
    wt = 0.0
    for p in range(npts):
        if c["beta"][p] == 0.0:
            continue
        den = np.abs(ref["trans"][p]).max()
        wt = max(wt, np.abs(Tm[p].reshape(k, k) - ref["trans"][p].reshape(k, k)).max() / (1e-11 * den))
    return worst, wt, int((status != 0).sum())


if __name__ == "__main__":
    rows = []
    for (k, npts, nobs_tot, n_mean) in [(20, 96, 400, 60), (50, 96, 900, 200), (50, 64, 300, 30), (48, 64, 2000, 600), (33, 64, 500, 100)]:
        for name in ("rtps", "rtpp", "rtps_adaptive_det"):
            for warm in (1, 8):
                a, t, bad = margin(k, npts, nobs_tot, n_mean, 100 + k, name, warm)
                rows.append((k, n_mean, name, warm, a, t, bad))
                print(f"k={k} n~{n_mean} {name} warm_run={warm}: anal err/tol {a:.3g}  trans err/tol {t:.3g}  bad {bad}", flush=True)
    print("worst anal %.3g  worst trans %.3g" % (max(r[4] for r in rows), max(r[5] for r in rows)))
