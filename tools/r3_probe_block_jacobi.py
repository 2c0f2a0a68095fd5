import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, torch
from _cases import das_case
import test_gpu_poly as T
from test_gpu_das import CONFIGS
cfg = CONFIGS["rtps"]
for k, n_mean in [(320, 260), (320, 180)]:
    for scale in [1.0, 1e-2, 1e-3, 1e-4]:
        c = das_case(k=k, nv=11, npts=6, nobs_tot=600, n_mean=n_mean, seed=5000 + k, infl0=1.0, vary_n=False)
        c["rdiag"] = c["rdiag"] * scale
        cond = T.point_conds(c, k)
        a0, i0, s0, w0 = T.run(c, k, 11, cfg, poly=False)
        print(k, n_mean, scale, "cond %.2e" % cond.max(), "status", s0.tolist(), "sweeps", w0.tolist(), flush=True)
