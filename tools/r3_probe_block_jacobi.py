# Probe of the block Jacobi (orders > 208 through the eigen stage): status / sweeps / error against the oracle as the
# observation errors shrink (cond(A) grows).  Run on the GPU box: python tools/r3_probe_block_jacobi.py
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, torch
from _cases import das_case
import test_gpu_poly as T
from test_gpu_das import CONFIGS
cfg = CONFIGS["rtps"]
for k, n_mean in [(320, 260), (320, 180), (512, 300), (250, 400)]:
    for scale in [1.0, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6]:
        c = das_case(k=k, nv=11, npts=6, nobs_tot=600, n_mean=n_mean, seed=5000 + k, infl0=1.0, vary_n=False)
        c["rdiag"] = c["rdiag"] * scale
        cond = T.point_conds(c, k)
        ref = T.oracle(c, k, 11, cfg)
        a0, i0, s0, w0 = T.run(c, k, 11, cfg, poly=False)
        nens, npts = c["nens"], c["npts"]
        x = c["gues"].reshape(11, nens, npts)
        rel = 0.0
        for v in range(11):
            sc = max(np.abs(x[v, k]).max(), np.abs(x[v, :k]).max())
            rel = max(rel, np.abs(a0.reshape(11, nens, npts)[v, :k] - ref["anal"].reshape(11, nens, npts)[v, :k]).max() / sc)
        print(k, n_mean, scale, "cond %.2e" % cond.max(), "status", s0.tolist(), "sweeps", w0.tolist(),
              "err/scale %.2e" % rel, "err/(cond eps) %.2f" % (rel / (cond.max() * 2.2e-16)), flush=True)
