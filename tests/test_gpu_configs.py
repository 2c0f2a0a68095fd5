"""The BASELINE.json configurations that round 1 only reached through letkf_core goldens, now at the das_letkf level on
the PRODUCTION instantiations (no k x k outputs requested), against the oracle's restatement of
scale/letkf/letkf_tools.f90:313-527 on the same inputs:
  C3  k = 320, n ~ 200      a slab of the synthetic C3 workload + ragged unrelated points (RTPS, det run, adaptive)
  C4  k = 50,  n ~ 5000     a slab at configs[3]'s observation density
  C5  k = 1000, n ~ 200     a handful of points
Tolerance: |d xa| <= 1e-10 * max(|x-bar|, |x'|) per variable; inflation 1e-12."""
import numpy as np
import pytest
import torch

import _oracle
import bench_workload as bw
from _cases import das_case

pytestmark = pytest.mark.gpu


def check_sample(w, anal, pts, k, nv, relax, nthreads=8):
    s = bw.sample_points(w, pts)
    prm = _oracle.DasParams(k=k, nv=nv, det_run=0, infl_adaptive=0, relax_to_inflated_prior=0,
                            relax_alpha=relax.get("relax_alpha", 0.0),
                            relax_alpha_spread=relax.get("relax_alpha_spread", 0.0), q_update_top=0.0, q_sprd_max=0.0,
                            iv_p=4, iv_q_first=5, iv_q_last=10, nthreads=nthreads)
    ns = s["ns"]
    ref = _oracle.das_points(prm, s["off"], s["idx"], s["rdiag"], s["rloc"], w["ensval"].cpu().numpy(),
                             w["dep"].cpu().numpy(), None, np.ones(ns * nv), s["gues"], 1, ns, ns * w["nens"])
    assert ref["rc"] == 0
    tp = torch.from_numpy(s["pts"]).cuda()
    got = anal.view(nv, w["nens"], w["npts"])[:, :k, tp].cpu().numpy()
    exp = ref["anal"].reshape(nv, w["nens"], ns)[:, :k]
    x = s["gues"].reshape(nv, w["nens"], ns)
    worst = 0.0
    for v in range(nv):
        scale = max(np.abs(x[v, k]).max(), np.abs(x[v, :k]).max())
        err = np.abs(got[v] - exp[v]).max()
        assert np.isfinite(got[v]).all()
        assert err <= 1e-10 * scale, (v, err, scale)
        worst = max(worst, err / scale)
    return worst


def run_workload(name, relax):
    from _gpu import ctx
    w = bw.build(name, torch.device("cuda"))
    k, nv, npts = w["k"], w["nv"], w["npts"]
    c = ctx()
    c.ens_mean(k, nv, npts, w["gues"], w["sp"], w["sm"], w["sv"])
    c.to_perturbations(k, nv, npts, w["gues"], w["sp"], w["sm"], w["sv"])
    anal = torch.full_like(w["gues"], float("nan"))
    infl = torch.ones(npts * nv, dtype=torch.float64, device="cuda")
    status = torch.full((npts,), -1, dtype=torch.int32, device="cuda")
    c.das_points(k, nv, w["obs_off"], w["obs_idx"], w["rdiag"], w["rloc"], w["ensval"], w["kld"], w["dep"], infl,
                 w["gues"], anal, w["sp"], w["sm"], w["sv"], status=status, **relax)
    torch.cuda.synchronize()
    assert int(status.abs().max()) == 0
    return w, anal


def test_c3_slab_k320():
    relax = dict(relax_alpha_spread=0.95)
    w, anal = run_workload("C3-slab", relax)
    assert w["k"] == 320 and 180 < w["n_mean"] < 220
    rng = np.random.default_rng(3)
    pts = np.sort(rng.choice(w["npts"], size=16, replace=False))
    check_sample(w, anal, pts, 320, 11, relax)


@pytest.mark.parametrize("name", ["rtps_adaptive_det", "rtpp", "rtps_qtop"])
def test_c3_ragged_points_k320(name):
    """unrelated points at k = 320 with n from 0 over n < k to n > k, every switch of the loop body"""
    from test_gpu_das import CONFIGS, compare_anal, run_both
    cfg = CONFIGS[name]
    c, ref, got, infl, status, _, _ = run_both(320, 11, 10, 700, 230, seed=41, cfg=cfg)
    assert (status == 0).all(), status
    compare_anal(c, ref, got, 320, 11, bool(cfg.get("det_run", 0)))
    assert np.abs(infl - ref["infl"]).max() <= 1e-12
    n = np.diff(c["obs_off"])
    assert (n == 0).any() and ((n > 0) & (n < 320)).any() and (n > 320).any()


def test_c4_slab_dense_obs():
    relax = dict(relax_alpha_spread=0.95)
    w, anal = run_workload("C4-slab", relax)
    assert w["k"] == 50 and 4500 < w["n_mean"] < 5500, w["n_mean"]
    rng = np.random.default_rng(4)
    pts = np.sort(rng.choice(w["npts"], size=24, replace=False))
    check_sample(w, anal, pts, 50, 11, relax)


def test_c4_ragged_dense_points_det_adaptive():
    """n ~ 5000 with DET_RUN + adaptive inflation on the production instantiation (trans_out = None)"""
    from test_gpu_das import CONFIGS, compare_anal, run_both
    cfg = CONFIGS["rtps_adaptive_det"]
    c, ref, got, infl, status, _, _ = run_both(50, 11, 12, 9000, 2500, seed=43, cfg=cfg)
    assert (status == 0).all(), status
    compare_anal(c, ref, got, 50, 11, True)
    assert np.abs(infl - ref["infl"]).max() <= 1e-12
    assert np.diff(c["obs_off"]).max() > 4000


def test_c5_points_k1000():
    relax = dict(relax_alpha_spread=0.95)
    w, anal = run_workload("C5-slab", relax)
    assert w["k"] == 1000 and 180 < w["n_mean"] < 220
    pts = np.array([0, 77, 300, 767], dtype=np.int64)
    check_sample(w, anal, pts, 1000, 11, relax, nthreads=4)


def test_c5_ragged_points_k1000_det():
    from test_gpu_das import CONFIGS, compare_anal, run_both
    cfg = CONFIGS["rtps_det"]
    c, ref, got, infl, status, _, _ = run_both(1000, 11, 3, 500, 200, seed=47, cfg=cfg)
    assert (status == 0).all(), status
    compare_anal(c, ref, got, 1000, 11, True)


@pytest.mark.parametrize("route", ["columns", "search+points"])
def test_c1_workload(route):
    """BASELINE configs[0] as SURVEY.md section 8(d) specifies it (bench_workload "C1": 40 x 40 x 30 at 15 km, k = 20, 500
    uniformly random conventional observations, vertical localisation in ln p, scale/letkf/letkf_tools.f90:1851-1865): the whole
    domain, EVERY point, against the oracle end to end -- orc_obs_local on a host copy of the tables, then the loop body -- through
    the one-call main loop (what das_letkf_amd calls) and through the column search + letkf_das_points_dev."""
    import _search
    from _gpu import ctx, pkg
    dev = torch.device("cuda")
    w = bw.build("C1", dev)
    k, nv, npts, nens = w["k"], w["nv"], w["npts"], w["nens"]
    assert (k, npts, w["nobs"]) == (20, 48000, 500) and 350 < w["n_mean"] < 450 and w["n_max"] == 500
    nij1, nlev = w["cfg"]["nx"] * w["cfg"]["ny"], w["cfg"]["nz"]
    c = ctx()
    c.ens_mean(k, nv, npts, w["gues"], 1, npts, npts * nens)
    c.to_perturbations(k, nv, npts, w["gues"], 1, npts, npts * nens)
    t_s, keep, order, pts = bw.search_tables(w, pkg, dev)
    ens, dep = w["ensval"][order].contiguous(), w["dep"][order].contiguous()
    rig, rjg = pts[0][:nij1].contiguous(), pts[1][:nij1].contiguous()
    anal = torch.full_like(w["gues"], float("nan"))
    infl = torch.ones(npts * nv, dtype=torch.float64, device=dev)
    status = torch.full((npts,), -1, dtype=torch.int32, device=dev)
    nobs = torch.full((npts,), -1, dtype=torch.int32, device=dev)
    if route == "columns":
        c.das_columns(k, nv, t_s, nij1, nlev, rig, rjg, pts[2], pts[3], ens, w["kld"], dep, infl, w["gues"], anal, 1, npts, npts * nens,
                      nobs_out=nobs, relax_alpha_spread=0.95, status=status)
    else:
        off, idx, rd, rl = c.obs_search_columns(t_s, nij1, nlev, rig, rjg, pts[2], pts[3])
        nobs = (off[1:] - off[:-1]).to(torch.int32)
        c.das_points(k, nv, off, idx, rd, rl, ens, w["kld"], dep, infl, w["gues"], anal, 1, npts, npts * nens, relax_alpha_spread=0.95,
                     status=status, warm_stride=nij1)
    torch.cuda.synchronize()
    assert int(status.abs().max()) == 0
    h, alive = _search.host_struct_from_torch(t_s, keep)
    P = [p.cpu().numpy() for p in pts]
    off_o, idx_o, rd_o, rl_o, tied = _search.oracle_csr(h, P[0], P[1], P[2], P[3])
    assert np.array_equal(np.diff(off_o), nobs.cpu().numpy())
    prm = _oracle.DasParams(k=k, nv=nv, det_run=0, infl_adaptive=0, relax_to_inflated_prior=0, relax_alpha=0.0,
                            relax_alpha_spread=0.95, q_update_top=0.0, q_sprd_max=0.0, iv_p=4, iv_q_first=5, iv_q_last=10, nthreads=8)
    gues = w["gues"].cpu().numpy()
    ref = _oracle.das_points(prm, off_o, idx_o, rd_o, rl_o, ens.cpu().numpy(), dep.cpu().numpy(), None, np.ones(npts * nv), gues, 1,
                             npts, npts * nens)
    assert ref["rc"] == 0
    got = anal.view(nv, nens, npts)[:, :k].cpu().numpy()
    exp = ref["anal"].reshape(nv, nens, npts)[:, :k]
    x = gues.reshape(nv, nens, npts)
    for v in range(nv):
        scale = max(np.abs(x[v, k]).max(), np.abs(x[v, :k]).max())
        assert np.isfinite(got[v]).all()
        assert np.abs(got[v] - exp[v]).max() <= 1e-10 * scale, (v, np.abs(got[v] - exp[v]).max(), scale)
