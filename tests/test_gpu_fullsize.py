"""BASELINE.json configs[1] at FULL size (C2: 240 x 240 x 60 = 3 456 000 solves, k = 50, ~200 local obs/point) through
size-independent properties of the transform -- no oracle at this size:
  * Y 1 = 0  =>  1 is an eigenvector of A  =>  T 1 = sqrt(rho) 1: the analysis perturbations stay zero-mean, so the mean
    of the analysis members is exactly  x-bar + X' w-bar  (w-bar from transm_out);
  * RELAX_ALPHA_SPREAD = 1 (scale/letkf/letkf_tools.f90:1971-2002): T^2 = (k-1) Pa makes the relaxed posterior spread
    EQUAL the prior spread, for every point and variable (ties T, Pa-in-RTPS and the transform together);
  * every status 0, sweep counts sane, and the warm-started runs agree with cold starts on a sampled slab.
Tolerances: 1e-10 relative to the variable's scale (FP64)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_c2_full_size_properties():
    import bench_workload as bw
    from _gpu import ctx
    c = ctx()
    w = bw.build("C2", torch.device("cuda"))
    k, nv, npts, nens = w["k"], w["nv"], w["npts"], w["nens"]
    assert npts == 3456000 and k == 50
    sp, sm, sv = w["sp"], w["sm"], w["sv"]
    gues = w["gues"]
    c.ens_mean(k, nv, npts, gues, sp, sm, sv)
    c.to_perturbations(k, nv, npts, gues, sp, sm, sv)
    anal = torch.empty_like(gues)
    infl = torch.ones(npts * nv, dtype=torch.float64, device="cuda")
    st = torch.full((npts,), -1, dtype=torch.int32, device="cuda")
    ns = torch.zeros(npts, dtype=torch.int32, device="cuda")
    wbar = torch.empty(npts, k, dtype=torch.float64, device="cuda")
    c.das_points(k, nv, w["obs_off"], w["obs_idx"], w["rdiag"], w["rloc"], w["ensval"], w["kld"], w["dep"], infl, gues,
                 anal, sp, sm, sv, relax_alpha_spread=1.0, status=st, nsweep=ns, transm_out=wbar)
    torch.cuda.synchronize()
    assert int(st.abs().max()) == 0
    assert 3 <= int(ns.min()) and int(ns.max()) <= 20 and float(ns.double().mean()) < 8.0
    g = gues.view(nv, nens, npts)
    a = anal.view(nv, nens, npts)
    for v in range(nv):
        xb, xp = g[v, k], g[v, :k]
        scale = max(float(xb.abs().max()), float(xp.abs().max()))
        # (1) mean of the analysis members = x-bar + X' w-bar
        want_mean = xb + (xp * wbar.t()).sum(dim=0)
        got_mean = a[v, :k].mean(dim=0)
        assert float((got_mean - want_mean).abs().max()) <= 1e-10 * scale, v
        # (2) RTPS with alpha = 1: posterior spread == prior spread
        sp_a = ((a[v, :k] - got_mean) ** 2).sum(dim=0)
        sp_g = (xp ** 2).sum(dim=0)
        rel = ((sp_a - sp_g).abs() / sp_g).max()
        assert float(rel) <= 1e-9, (v, float(rel))
    # (3) warm-started runs vs cold starts on one level slab
    nij = 240 * 240
    sl = slice(7 * nij, 8 * nij)
    off = (w["obs_off"][7 * nij:8 * nij + 1] - w["obs_off"][7 * nij]).contiguous()
    lo, hi = int(w["obs_off"][7 * nij]), int(w["obs_off"][8 * nij])
    gs = g[:, :, sl].contiguous()
    cold = torch.empty_like(gs)
    c.das_points(k, nv, off, w["obs_idx"][lo:hi].contiguous(), w["rdiag"][lo:hi].contiguous(),
                 w["rloc"][lo:hi].contiguous(), w["ensval"], w["kld"], w["dep"], torch.ones(nij * nv, dtype=torch.float64,
                                                                                         device="cuda"),
                 gs.view(-1), cold.view(-1), 1, nij, nij * nens, relax_alpha_spread=1.0, warm_run=1)
    torch.cuda.synchronize()
    for v in range(nv):
        scale = max(float(gs[v, k].abs().max()), float(gs[v, :k].abs().max()))
        assert float((cold[v, :k] - a[v, :k, sl]).abs().max()) <= 1e-10 * scale, v
