"""Row f3: state_trans / state_trans_inv, the member-field <-> point-major ensemble
re-ordering (grd_to_buf + slot placement) and enssprd_grd.  CPU: the oracle against direct numpy restatements and
its own round trips.  GPU: the kernels against the oracle (bit-exact for the pure data movement)."""
import ctypes as C

import numpy as np
import pytest

import _oracle
from __graft_entry__ import load_package

pkg = load_package()


def field(rng, nlev, nlon, nlat, nv=11):
    """a plausible SCALE restart state v3dg(nlev,nlon,nlat,nv3d), level-fastest, as a flat array (n slowest)"""
    shp = (nv, nlat, nlon, nlev)
    v = np.empty(shp)
    rho = rng.uniform(0.3, 1.2, shp[1:])
    v[0] = rho
    v[1] = rho * rng.normal(5, 10, shp[1:])
    v[2] = rho * rng.normal(0, 10, shp[1:])
    v[3] = rho * rng.normal(0, 1, shp[1:])
    v[4] = rho * rng.uniform(290, 450, shp[1:])          # rho * theta
    v[5] = rng.uniform(0, 0.02, shp[1:])
    for n in range(6, nv):
        v[n] = rng.uniform(-1e-5, 1e-3, shp[1:])          # some slightly negative hydrometeors (clamped by the inverse)
    return np.ascontiguousarray(v).reshape(-1)


def orc_consts(c):
    o = _oracle.oracle()
    return o, C.byref(c)


def test_state_trans_oracle_vs_numpy_and_round_trip():
    rng = np.random.default_rng(5)
    nlev, nlon, nlat, nv = 7, 6, 5, 11
    v = field(rng, nlev, nlon, nlat)
    c = pkg.scale_rm_consts(clamp=False)
    x = v.copy()
    o = _oracle.oracle()
    o.orc_state_trans(C.byref(c), C.c_int(nlev), C.c_int(nlon), C.c_int(nlat), C.c_int(nv), _oracle._dp(x), C.c_int(0))
    a = v.reshape(nv, -1)
    q = a[5:]
    cv = np.array([c.tracer_cv[i] for i in range(6)])
    qdry = 1.0 - q.sum(0)
    cvtot = c.cvdry * qdry + (q * cv[:, None]).sum(0)
    rtot = c.rdry * qdry + c.rvap * q[0]
    pres = c.pre00 * (a[4] * rtot / c.pre00) ** ((cvtot + rtot) / cvtot)
    temp = pres / (a[0] * rtot)
    xr = x.reshape(nv, -1)
    assert np.allclose(xr[4], pres, rtol=1e-13) and np.allclose(xr[3], temp, rtol=1e-13)
    assert np.allclose(xr[0], a[1] / a[0], rtol=1e-15) and np.allclose(xr[2], a[3] / a[0], rtol=1e-15)
    o.orc_state_trans(C.byref(c), C.c_int(nlev), C.c_int(nlon), C.c_int(nlat), C.c_int(nv), _oracle._dp(x), C.c_int(1))
    assert np.allclose(x, v, rtol=1e-12, atol=1e-18)      # inverse(forward) == identity


def test_member_points_and_spread_oracle():
    rng = np.random.default_rng(6)
    nlev, nlon, nlat, nv, k, np_ = 5, 7, 4, 3, 6, 3
    o = _oracle.oracle()
    nxy = nlon * nlat
    fields = [rng.standard_normal(nv * nxy * nlev) for _ in range(k)]
    for rank in range(np_):
        nij1 = (nxy - rank + np_ - 1) // np_
        nens = k + 1
        x = np.full(nv * nens * nij1 * nlev, np.nan)
        sp, sm, sv = 1, nij1 * nlev, nij1 * nlev * nens
        for m in range(k):
            o.orc_member_points(C.c_int(0), C.c_int(nlev), C.c_int(nlon), C.c_int(nlat), C.c_int(nv), C.c_int(np_),
                                C.c_int(rank), C.c_int(m), _oracle._dp(fields[m]), _oracle._dp(x), C.c_int64(nij1),
                                C.c_int64(sp), C.c_int64(sm), C.c_int64(sv))
        xv = x.reshape(nv, nens, nlev, nij1)
        for m in range(k):
            f = fields[m].reshape(nv, nlat, nlon, nlev)
            for i in range(nij1):
                j = rank + np_ * i
                assert np.array_equal(xv[:, m, :, i], f[:, j // nlon, j % nlon, :])
        # and back
        back = np.zeros_like(fields[0])
        o.orc_member_points(C.c_int(1), C.c_int(nlev), C.c_int(nlon), C.c_int(nlat), C.c_int(nv), C.c_int(np_),
                            C.c_int(rank), C.c_int(2), _oracle._dp(back), _oracle._dp(x), C.c_int64(nij1),
                            C.c_int64(sp), C.c_int64(sm), C.c_int64(sv))
        mask = np.zeros(nxy, bool)
        mask[rank::np_] = True
        b = back.reshape(nv, nxy, nlev)
        assert np.array_equal(b[:, mask], fields[2].reshape(nv, nxy, nlev)[:, mask]) and not b[:, ~mask].any()
        # spread
        npts = nij1 * nlev
        o.orc_ensmean(C.c_int(k), C.c_int(nv), C.c_int64(npts), _oracle._dp(x), C.c_int64(sp), C.c_int64(sm),
                      C.c_int64(sv))
        sprd = np.zeros(npts * nv)
        o.orc_ens_spread(C.c_int(k), C.c_int(nv), C.c_int64(npts), _oracle._dp(x), C.c_int64(sp), C.c_int64(sm),
                         C.c_int64(sv), _oracle._dp(sprd))
        xe = x.reshape(nv, nens, npts)
        assert np.allclose(sprd.reshape(nv, npts), xe[:, :k].std(axis=1, ddof=1), rtol=1e-13)


@pytest.mark.gpu
def test_gpu_state_trans_matches_oracle():
    import torch
    from _gpu import ctx, dev
    rng = np.random.default_rng(8)
    nlev, nlon, nlat, nv = 36, 40, 24, 11
    v = field(rng, nlev, nlon, nlat)
    c = pkg.scale_rm_consts(clamp=True)
    o = _oracle.oracle()
    for inverse in (0, 1):
        exp = v.copy()
        if inverse:
            o.orc_state_trans(C.byref(c), C.c_int(nlev), C.c_int(nlon), C.c_int(nlat), C.c_int(nv), _oracle._dp(exp), C.c_int(0))
        src = exp.copy()
        o.orc_state_trans(C.byref(c), C.c_int(nlev), C.c_int(nlon), C.c_int(nlat), C.c_int(nv), _oracle._dp(exp),
                          C.c_int(inverse))
        d = dev(src)
        ctx().state_trans(c, nlev, nlon, nlat, nv, d, inverse=bool(inverse))
        torch.cuda.synchronize()
        got = d.cpu().numpy()
        assert np.allclose(got, exp, rtol=5e-14, atol=0), np.abs(got / np.where(exp == 0, 1, exp) - 1).max()


@pytest.mark.gpu
def test_gpu_member_points_and_spread_match_oracle():
    import torch
    from _gpu import ctx, dev
    rng = np.random.default_rng(9)
    nlev, nlon, nlat, nv, k, np_ = 37, 45, 33, 4, 5, 4
    o = _oracle.oracle()
    nxy = nlon * nlat
    nens = k + 1
    fields = [rng.standard_normal(nv * nxy * nlev) for _ in range(k)]
    for rank in (0, 3):
        nij1 = (nxy - rank + np_ - 1) // np_
        sp, sm, sv = 1, nij1 * nlev, nij1 * nlev * nens
        x = np.zeros(nv * nens * nij1 * nlev)
        xd = torch.zeros(x.size, dtype=torch.float64, device="cuda")
        for m in range(k):
            o.orc_member_points(C.c_int(0), C.c_int(nlev), C.c_int(nlon), C.c_int(nlat), C.c_int(nv), C.c_int(np_),
                                C.c_int(rank), C.c_int(m), _oracle._dp(fields[m]), _oracle._dp(x), C.c_int64(nij1),
                                C.c_int64(sp), C.c_int64(sm), C.c_int64(sv))
            ctx().member_points(0, nlev, nlon, nlat, nv, np_, rank, m, dev(fields[m]), xd, nij1, sp, sm, sv)
        torch.cuda.synchronize()
        assert np.array_equal(xd.cpu().numpy(), x)                       # pure data movement: bit exact
        back = torch.zeros(fields[0].size, dtype=torch.float64, device="cuda")
        ctx().member_points(1, nlev, nlon, nlat, nv, np_, rank, 1, back, xd, nij1, sp, sm, sv)
        eb = np.zeros_like(fields[0])
        o.orc_member_points(C.c_int(1), C.c_int(nlev), C.c_int(nlon), C.c_int(nlat), C.c_int(nv), C.c_int(np_),
                            C.c_int(rank), C.c_int(1), _oracle._dp(eb), _oracle._dp(x), C.c_int64(nij1),
                            C.c_int64(sp), C.c_int64(sm), C.c_int64(sv))
        torch.cuda.synchronize()
        assert np.array_equal(back.cpu().numpy(), eb)
        npts = nij1 * nlev
        ctx().ens_mean(k, nv, npts, xd, sp, sm, sv)
        o.orc_ensmean(C.c_int(k), C.c_int(nv), C.c_int64(npts), _oracle._dp(x), C.c_int64(sp), C.c_int64(sm), C.c_int64(sv))
        sprd = torch.zeros(npts * nv, dtype=torch.float64, device="cuda")
        ctx().ens_spread(k, nv, npts, xd, sp, sm, sv, sprd)
        es = np.zeros(npts * nv)
        o.orc_ens_spread(C.c_int(k), C.c_int(nv), C.c_int64(npts), _oracle._dp(x), C.c_int64(sp), C.c_int64(sm),
                         C.c_int64(sv), _oracle._dp(es))
        torch.cuda.synchronize()
        assert np.array_equal(xd.cpu().numpy(), x)                       # mean: same summation order -> bit exact
        assert np.allclose(sprd.cpu().numpy(), es, rtol=1e-15, atol=0)
