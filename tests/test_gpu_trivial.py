"""The streaming pass for grid points that have nothing to solve (scale-letkf_amd/csrc/letkf_trivial.hip: beta = 0,
scale/letkf/letkf_tools.f90:333-359, and points without local observations, common/common_letkf.f90:89-107 + the
relaxation / blend / q clamp / inflation class copy of letkf_tools.f90:387-513) against the oracle's restatement of the
loop body, on batches that are MOSTLY such points (a radar disc in a larger domain) -- every switch of the loop body,
both state layouts, the one-wave, two-wave and staged solve paths beside it, a variable-localisation class mask, and the
dynamic run scheduling of the solve kernel (strided runs, ragged run lengths) on the points that remain.
Tolerances as test_gpu_das.py."""
import numpy as np
import pytest
import torch

import _oracle
from _cases import das_case

pytestmark = pytest.mark.gpu

CONFIGS = {
    "rtps": dict(relax_alpha_spread=0.95),
    "rtps_inflated_adaptive_det": dict(relax_alpha_spread=0.8, det_run=1, infl_adaptive=1, relax_to_inflated_prior=1),
    "rtpp_inflated_qclamp": dict(relax_alpha=0.5, relax_to_inflated_prior=1, q_sprd_max=0.05),
    "rtps_qclamp_det": dict(relax_alpha_spread=0.9, q_sprd_max=0.02, det_run=1),
    "norelax_qtop_adaptive": dict(q_update_top=49.0, infl_adaptive=1, det_run=1),
    "rtps_qtop": dict(relax_alpha_spread=0.95, q_update_top=49.0),
}


def _run(k, npts, cfg, seed, layout="ref", mask=0, warm_stride=0, warm_run=0, frac_empty=0.7):
    from _gpu import ctx, dev
    nv = 11
    det = bool(cfg.get("det_run", 0))
    c = das_case(k=k, nv=nv, npts=npts, nobs_tot=600, n_mean=90, seed=seed, det_run=det, infl0=1.0)
    rng = np.random.default_rng(seed + 1)
    # most points lose their observations; beta = 0 / fractional beta on both kinds
    counts = np.diff(c["obs_off"])
    empty = rng.random(npts) < frac_empty
    counts[empty] = 0
    off = np.zeros(npts + 1, dtype=np.int64)
    np.cumsum(counts, out=off[1:])
    keep = np.concatenate([np.arange(c["obs_off"][p], c["obs_off"][p] + counts[p]) for p in range(npts)]).astype(np.int64) \
        if off[-1] else np.zeros(0, dtype=np.int64)
    c["obs_idx"], c["rdiag"], c["rloc"], c["obs_off"] = c["obs_idx"][keep], c["rdiag"][keep], c["rloc"][keep], off
    c["beta"] = np.where(rng.random(npts) < 0.15, 0.0, np.where(rng.random(npts) < 0.2, 0.37, 1.0))
    # an inflation field that differs from variable to variable and point to point (parm of the relaxation, the class
    # copy under adaptive inflation)
    c["infl"] = 1.0 + 0.3 * rng.random(npts * nv)
    # the mean of p (variable 4) straddles q_update_top, so that some points skip the moisture variables
    x = c["gues"].reshape(nv, c["nens"], npts)
    x[4, k] = rng.uniform(30.0, 70.0, npts)
    sp, sm, sv = c["sp"], c["sm"], c["sv"]
    gues = c["gues"]
    if layout == "member":                              # point-major, member-fastest (sm = 1)
        gues = np.ascontiguousarray(x.transpose(2, 0, 1)).reshape(-1)
        sp, sm, sv = nv * c["nens"], 1, c["nens"]
    prm = _oracle.DasParams(k=k, nv=nv, det_run=int(det), infl_adaptive=cfg.get("infl_adaptive", 0),
                            relax_to_inflated_prior=cfg.get("relax_to_inflated_prior", 0),
                            relax_alpha=cfg.get("relax_alpha", 0.0), relax_alpha_spread=cfg.get("relax_alpha_spread", 0.0),
                            q_update_top=cfg.get("q_update_top", 0.0), q_sprd_max=cfg.get("q_sprd_max", 0.0),
                            iv_p=4, iv_q_first=5, iv_q_last=10, nthreads=4, var_mask=mask)
    ref = _oracle.das_points(prm, c["obs_off"], c["obs_idx"], c["rdiag"], c["rloc"], c["ensval"], c["dep"], c["beta"],
                             c["infl"], gues, sp, sm, sv, want_rtps=True)
    assert ref["rc"] == 0
    anal = torch.full((gues.size,), float("nan"), dtype=torch.float64, device="cuda")
    infl = dev(c["infl"])
    status = torch.full((npts,), -1, dtype=torch.int32, device="cuda")
    nsweep = torch.full((npts,), -1, dtype=torch.int32, device="cuda")
    rtps = torch.full((npts * nv,), -7.0, dtype=torch.float64, device="cuda")
    kw = dict(var_mask=mask) if mask else {}
    ctx().das_points(k, nv, dev(c["obs_off"]), dev(c["obs_idx"]), dev(c["rdiag"]), dev(c["rloc"]), dev(c["ensval"]),
                     c["kld"], dev(c["dep"]), infl, dev(gues), anal, sp, sm, sv, beta=dev(c["beta"]), det_run=det,
                     infl_adaptive=cfg.get("infl_adaptive", 0), relax_to_inflated_prior=cfg.get("relax_to_inflated_prior", 0),
                     relax_alpha=cfg.get("relax_alpha", 0.0), relax_alpha_spread=cfg.get("relax_alpha_spread", 0.0),
                     q_update_top=cfg.get("q_update_top", 0.0), q_sprd_max=cfg.get("q_sprd_max", 0.0), iv_p=4,
                     iv_q_first=5, iv_q_last=10, status=status, nsweep=nsweep, rtps_infl_out=rtps, warm_run=warm_run,
                     warm_stride=warm_stride, **kw)
    torch.cuda.synchronize()
    got = anal.cpu().numpy()
    nens = c["nens"]
    view = (lambda a: a.reshape(npts, nv, nens).transpose(1, 2, 0)) if layout == "member" else (lambda a: a.reshape(nv, nens, npts))
    g, e, xg = view(got), view(ref["anal"]), view(gues)
    members = list(range(k)) + ([k + 1] if det else [])
    vars_ = [v for v in range(nv) if (mask >> v) & 1] if mask else list(range(nv))
    for v in range(nv):
        if v in vars_:
            scale = max(np.abs(xg[v, k]).max(), np.abs(xg[v, :k]).max())
            assert np.isfinite(g[v, members]).all()
            assert np.abs(g[v, members] - e[v, members]).max() <= 1e-10 * scale, v
        else:
            assert np.isnan(g[v]).all()                 # a variable outside the class is not written
    assert np.isnan(g[:, k]).all()                      # the mean slot is never written by the path
    st, ns = status.cpu().numpy(), nsweep.cpu().numpy()
    trivial = (counts == 0) | (c["beta"] == 0.0)
    assert (st == 0).all()
    # (staged path, k >= 63: a point is analysed without an eigen stage where that pays and reports -(degree), tests/test_gpu_poly.py)
    solved = ~trivial & (counts >= 2)
    assert (ns[trivial] == 0).all()
    assert (ns[solved] > 0).all() if k < 63 else (ns[solved] != 0).all()
    assert np.abs(infl.cpu().numpy() - ref["infl"]).max() <= 1e-12
    r_got, r_ref = rtps.cpu().numpy().reshape(nv, npts), ref["rtps"].reshape(nv, npts)
    assert np.abs(r_got[vars_] - r_ref[vars_]).max() <= 1e-11 * max(1.0, np.abs(r_ref).max())
    return int(trivial.sum())


@pytest.mark.parametrize("name", list(CONFIGS))
@pytest.mark.parametrize("k", [20, 50])
def test_points_without_observations_every_switch(name, k):
    ntriv = _run(k, 300, CONFIGS[name], seed=100 + k)
    assert ntriv > 150


@pytest.mark.parametrize("k", [50, 100, 144])
def test_points_without_observations_beside_each_solve_path(k):
    """k = 50: one wave per point; 100: two waves per point (the skip sits between workgroup barriers); 144: the staged
    path, which hands them to the same pass"""
    _run(k, 96, CONFIGS["rtps_inflated_adaptive_det"], seed=7 + k)


def test_points_without_observations_member_fastest_layout():
    _run(50, 200, CONFIGS["rtpp_inflated_qclamp"], seed=5, layout="member")
    _run(50, 200, CONFIGS["rtps_inflated_adaptive_det"], seed=6, layout="member")


def test_points_without_observations_class_mask():
    """one variable-localisation class (variables 0-3 and 7): the others are left alone"""
    _run(50, 200, CONFIGS["rtps_inflated_adaptive_det"], seed=8, mask=0b00010001111)
    _run(50, 200, CONFIGS["rtps_qtop"], seed=9, mask=0b11111100000)   # a class of moisture variables only (skipped below q_update_top)


@pytest.mark.parametrize("stride,run", [(0, 0), (25, 0), (25, 7), (0, 5), (100, 0)])
def test_dynamic_run_scheduling_covers_every_point_once(stride, run):
    """runs along the points and up columns of 12 / 3 levels, ragged run lengths, quartered runs: every point is written
    exactly once (a point written twice in a run of another shape would still pass the parity check, a missed one holds NaN)"""
    _run(50, 300, CONFIGS["rtps"], seed=40 + stride + run, warm_stride=stride, warm_run=run, frac_empty=0.3)
