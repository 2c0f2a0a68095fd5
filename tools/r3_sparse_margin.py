"""Which points of the sparse workloads carry the largest analysis error against the oracle, and why:
per point max_v |d xa| / max(|x-bar|, |x'|) for every point of a bench workload, binned by the local observation count."""
import os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench_workload as bw, _oracle
from __graft_entry__ import load_package
name = sys.argv[1] if len(sys.argv) > 1 else "C2-mini-sparse"
pkg = load_package(); dev = torch.device("cuda:0")
ctx = pkg.Context(0, torch.cuda.current_stream().cuda_stream)
w = bw.build(name, dev)
k, nv, npts, nens = w["k"], w["nv"], w["npts"], w["nens"]
ctx.ens_mean(k, nv, npts, w["gues"], 1, npts, npts * nens); ctx.to_perturbations(k, nv, npts, w["gues"], 1, npts, npts * nens)
anal = torch.empty_like(w["gues"]); infl = torch.ones(npts * nv, dtype=torch.float64, device=dev)
status = torch.zeros(npts, dtype=torch.int32, device=dev); nsweep = torch.zeros(npts, dtype=torch.int32, device=dev)
for warm in (0, 1):
    ctx.das_points(k, nv, w["obs_off"], w["obs_idx"], w["rdiag"], w["rloc"], w["ensval"], w["kld"], w["dep"], infl, w["gues"], anal,
                   1, npts, npts * nens, status=status, nsweep=nsweep, relax_alpha_spread=0.95, warm_run=warm)
    torch.cuda.synchronize()
    s = bw.sample_points(w, np.arange(npts))
    prm = _oracle.DasParams(k=k, nv=nv, det_run=0, infl_adaptive=0, relax_to_inflated_prior=0, relax_alpha=0.0, relax_alpha_spread=0.95,
                            q_update_top=0.0, q_sprd_max=0.0, iv_p=4, iv_q_first=5, iv_q_last=10, nthreads=16)
    r = _oracle.das_points(prm, s["off"], s["idx"], s["rdiag"], s["rloc"], w["ensval"].cpu().numpy(), w["dep"].cpu().numpy(), None,
                           np.ones(npts * nv), s["gues"], 1, npts, npts * nens)
    got = anal.cpu().numpy().reshape(nv, nens, npts)[:, :k]; exp = r["anal"].reshape(nv, nens, npts)[:, :k]
    x = s["gues"].reshape(nv, nens, npts)
    err = np.zeros(npts)
    for v in range(nv):
        scale = max(np.abs(x[v, k]).max(), np.abs(x[v, :k]).max())
        err = np.maximum(err, np.abs(got[v] - exp[v]).max(axis=0) / scale)
    n = np.diff(s["off"]); sw = nsweep.cpu().numpy()
    print(f"{name} warm_run={warm}: max err {err.max():.3g} at n={n[err.argmax()]} sweeps={sw[err.argmax()]}; status!=0: {int((status!=0).sum())}")
    for lo, hi in [(0, 1), (1, 5), (5, 15), (15, 30), (30, 49), (49, 51), (51, 100), (100, 1000)]:
        m = (n >= lo) & (n < hi)
        if m.any():
            print(f"  n in [{lo},{hi}): {m.sum():6d} points, max err {err[m].max():.3g}, p99 {np.quantile(err[m], 0.99):.3g}, mean sweeps {sw[m].mean():.2f}")
