"""GPU parity of the das_letkf set-up passes (C ABI section 7): letkf_relax_beta_dev against the oracle's relax_beta
(scale/letkf/letkf_tools.f90:1911-1948) point by point -- bit exact, the arithmetic is three multiplications -- and
letkf_infl_init_dev against :237-267; then beta from the device drives letkf_das_points_dev like a caller's would."""
import ctypes as C

import numpy as np
import pytest
import torch

import _oracle

pytestmark = pytest.mark.gpu


class OrcBeta(C.Structure):
    _fields_ = [("radar_only", C.c_int), ("radar_zmax", C.c_double), ("vert_local_radar", C.c_double),
                ("boundary_buffer_width", C.c_double), ("dx", C.c_double), ("dy", C.c_double), ("ihalo", C.c_int),
                ("jhalo", C.c_int), ("nlong", C.c_int), ("nlatg", C.c_int)]


def oracle_beta(bp, rig, rjg, hgt, nij1, nlev):
    o = _oracle.oracle()
    out = np.empty(nij1 * nlev)
    for lev in range(nlev):
        for ij in range(nij1):
            out[ij + nij1 * lev] = o.orc_relax_beta(C.byref(bp), C.c_double(rig[ij]), C.c_double(rjg[ij]),
                                                    C.c_double(hgt[ij + nij1 * lev]))
    return out


@pytest.mark.parametrize("radar_only,bw", [(1, 0.0), (0, 5000.0), (1, 7000.0), (0, 0.0)])
def test_relax_beta_matches_oracle(radar_only, bw):
    from _gpu import ctx, dev, pkg
    rng = np.random.default_rng(17 + radar_only)
    nlong, nlatg, ihalo, nlev = 37, 29, 2, 9
    ii, jj = np.meshgrid(np.arange(nlong), np.arange(nlatg))
    rig = (ii.ravel() + 1 + ihalo).astype(np.float64)      # common_mpi_scale.f90:303-308
    rjg = (jj.ravel() + 1 + ihalo).astype(np.float64)
    nij1 = rig.size
    hgt = np.sort(rng.uniform(20.0, 19000.0, size=(nlev, nij1)), axis=0).ravel()
    bp = OrcBeta(radar_only, 10000.0, 2000.0, bw, 1000.0, 1500.0, ihalo, ihalo, nlong, nlatg)
    want = oracle_beta(bp, rig, rjg, hgt, nij1, nlev)
    p = pkg.BetaParams()
    p.radar_only, p.ihalo, p.jhalo, p.nlong, p.nlatg = radar_only, ihalo, ihalo, nlong, nlatg
    p.radar_zmax, p.vert_local_radar, p.boundary_buffer_width, p.dx, p.dy = 10000.0, 2000.0, bw, 1000.0, 1500.0
    beta = torch.full((nij1 * nlev,), -5.0, dtype=torch.float64, device="cuda")
    ctx().relax_beta(p, nij1, nlev, dev(rig), dev(rjg), dev(hgt), beta)
    torch.cuda.synchronize()
    got = beta.cpu().numpy()
    assert np.array_equal(got, want)
    if radar_only:
        assert (want == 0.0).any()
    if bw > 0:
        assert ((want > 0.0) & (want < 1.0)).any()


def test_infl_init_matches_oracle():
    from _gpu import ctx, dev
    o = _oracle.oracle()
    rng = np.random.default_rng(4)
    w0 = rng.uniform(0.7, 1.6, size=5000)
    for mul, mn in [(1.3, 0.0), (-1.0, 0.95), (0.8, 1.0), (-1.0, 0.0)]:
        want = w0.copy()
        o.orc_infl_init(C.c_int64(want.size), want.ctypes.data_as(C.c_void_p), C.c_double(mul), C.c_double(mn))
        t = dev(w0)
        ctx().infl_init(t, mul, mn)
        torch.cuda.synchronize()
        assert np.array_equal(t.cpu().numpy(), want)


def test_device_beta_feeds_the_loop_body():
    """beta computed on the device goes straight into letkf_das_points_dev; result = oracle with the oracle's beta"""
    from _cases import das_case
    from _gpu import ctx, dev, pkg
    k, nv, nij1, nlev = 20, 11, 30, 4
    npts = nij1 * nlev
    c = das_case(k=k, nv=nv, npts=npts, nobs_tot=300, n_mean=40, seed=23)
    rig = 3.0 + np.arange(nij1, dtype=np.float64)
    rjg = np.full(nij1, 9.0)
    hgt = np.repeat(np.array([500.0, 4000.0, 12000.0, 19000.0]), nij1)
    bp = OrcBeta(1, 10000.0, 2000.0, 6000.0, 1000.0, 1000.0, 2, 2, nij1, 20)
    beta_ref = oracle_beta(bp, rig, rjg, hgt, nij1, nlev)
    assert (beta_ref == 0).any() and ((beta_ref > 0) & (beta_ref < 1)).any() and (beta_ref == 1).any()
    prm = _oracle.DasParams(k=k, nv=nv, det_run=0, infl_adaptive=0, relax_to_inflated_prior=0, relax_alpha=0.0,
                            relax_alpha_spread=0.95, q_update_top=0.0, q_sprd_max=0.0, iv_p=4, iv_q_first=5,
                            iv_q_last=10, nthreads=2)
    ref = _oracle.das_points(prm, c["obs_off"], c["obs_idx"], c["rdiag"], c["rloc"], c["ensval"], c["dep"], beta_ref,
                             c["infl"], c["gues"], c["sp"], c["sm"], c["sv"])
    p = pkg.BetaParams()
    p.radar_only, p.ihalo, p.jhalo, p.nlong, p.nlatg = 1, 2, 2, nij1, 20
    p.radar_zmax, p.vert_local_radar, p.boundary_buffer_width, p.dx, p.dy = 10000.0, 2000.0, 6000.0, 1000.0, 1000.0
    beta = torch.empty(npts, dtype=torch.float64, device="cuda")
    ctx().relax_beta(p, nij1, nlev, dev(rig), dev(rjg), dev(hgt), beta)
    anal = torch.zeros(c["gues"].size, dtype=torch.float64, device="cuda")
    ctx().das_points(k, nv, dev(c["obs_off"]), dev(c["obs_idx"]), dev(c["rdiag"]), dev(c["rloc"]), dev(c["ensval"]),
                     c["kld"], dev(c["dep"]), dev(c["infl"]), dev(c["gues"]), anal, c["sp"], c["sm"], c["sv"],
                     beta=beta, relax_alpha_spread=0.95)
    torch.cuda.synchronize()
    g = anal.cpu().numpy().reshape(nv, c["nens"], npts)[:, :k]
    e = ref["anal"].reshape(nv, c["nens"], npts)[:, :k]
    x = c["gues"].reshape(nv, c["nens"], npts)
    for v in range(nv):
        scale = max(np.abs(x[v, k]).max(), np.abs(x[v, :k]).max())
        assert np.abs(g[v] - e[v]).max() <= 1e-10 * scale
