"""GPU parity of the after-the-loop row (f4) through the C ABI against the oracle: monit_dep statistics
(scale/common/common_obs_scale.f90:1851-1895), the additive-inflation update and addinfl_weight
(scale/letkf/letkf_tools.f90:804-929).  Tolerances: counts exact; bias / rmse 1e-13 relative (parallel summation
order; the reference prints ES12.3); inflation update 4 ulp (fused multiply-add); weights 1e-15."""
import ctypes as C

import numpy as np
import pytest
import torch

import _oracle

pytestmark = pytest.mark.gpu

ELEM_UID = np.array([2819, 2820, 3073, 3074, 3330, 3331, 14593, 19999, 4001, 4004, 4002, 4003, 8800, 99991, 99992,
                     99993], dtype=np.int32)   # common_obs_scale.f90:74-77


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


@pytest.mark.parametrize("nn", [0, 1, 1000, 300000])
def test_monit_dep(nn):
    from _gpu import ctx, dev
    rng = np.random.default_rng(nn)
    elm = rng.choice(ELEM_UID[[0, 1, 2, 3, 6, 8, 9, 10]], max(nn, 1)).astype(np.int32)[:nn]
    dep = rng.normal(0.3, 2.0, nn)
    qc = np.where(rng.random(nn) < 0.2, 5, 0).astype(np.int32)
    nid = len(ELEM_UID)
    nobs, bias, rmse = np.zeros(nid, np.int32), np.zeros(nid), np.zeros(nid)
    _oracle.oracle().orc_monit_dep(C.c_int(nid), _p(ELEM_UID, C.c_int32), C.c_int64(nn), _p(elm, C.c_int32),
                                   _p(dep, C.c_double), _p(qc, C.c_int32), _p(nobs, C.c_int32), _p(bias, C.c_double),
                                   _p(rmse, C.c_double))
    g_n, g_b, g_r = ctx().monit_dep(ELEM_UID, dev(elm), dev(dep), dev(qc))
    torch.cuda.synchronize()
    assert np.array_equal(g_n.cpu().numpy(), nobs)
    has = nobs > 0
    assert np.array_equal(g_b.cpu().numpy()[~has], bias[~has]) and np.all(bias[~has] == -9.99e33)
    assert np.allclose(g_b.cpu().numpy()[has], bias[has], rtol=1e-13, atol=1e-13)
    assert np.allclose(g_r.cpu().numpy()[has], rmse[has], rtol=1e-13, atol=0)
    if nn >= 1000:
        assert nobs[3] == 0 and nobs[9] == 0 and nobs[2] > 0 and nobs[8] > 0     # Tv -> T, RE0 -> REF


@pytest.mark.parametrize("q_ratio,ref_only,shuffle", [(False, False, False), (True, True, True)])
def test_additive_inflation(q_ratio, ref_only, shuffle):
    from _gpu import ctx, dev
    rng = np.random.default_rng(7)
    k, nv, nij1, nlev = 6, 11, 90, 4
    npts, nens = nij1 * nlev, k + 1
    sp, sm, sv = 1, npts, npts * nens
    anal = rng.normal(0.0, 1.0, nv * nens * npts)
    add = rng.normal(0.0, 0.1, nv * nens * npts)
    gues = rng.uniform(1e-4, 1e-2, nv * nens * npts)
    rig, rjg = rng.uniform(2.5, 40.5, nij1), rng.uniform(2.5, 40.5, nij1)
    ob_ri, ob_rj = rng.uniform(10.0, 20.0, 37), rng.uniform(10.0, 20.0, 37)
    lib = _oracle.oracle()
    w = None
    if ref_only:
        w = np.zeros(nij1)
        lib.orc_addinfl_weight(C.c_int64(nij1), _p(rig, C.c_double), _p(rjg, C.c_double), C.c_int64(len(ob_ri)),
                               _p(ob_ri, C.c_double), _p(ob_rj, C.c_double), C.c_double(1000.0), C.c_double(1000.0),
                               C.c_double(4000.0), _p(w, C.c_double))
        gw = ctx().addinfl_weight(dev(rig), dev(rjg), dev(ob_ri), dev(ob_rj), 1000.0, 1000.0, 4000.0)
        torch.cuda.synchronize()
        assert np.allclose(gw.cpu().numpy(), w, rtol=1e-15, atol=0)
        assert (w == 0).any() and (w > 0.5).any()
    ishuf = rng.permutation(k).astype(np.int32) if shuffle else None
    qmean = gues[k * sm:] if q_ratio else None            # gues3d(:,:,mmean,:)
    ref = anal.copy()
    lib.orc_additive_inflation(C.c_int(k), C.c_int(nv), C.c_int64(npts), C.c_int64(nij1), _p(ref, C.c_double),
                               _p(add, C.c_double), C.c_int64(sp), C.c_int64(sm), C.c_int64(sv), C.c_double(0.3),
                               _p(w, C.c_double) if w is not None else None,
                               _p(np.ascontiguousarray(qmean), C.c_double) if q_ratio else None, C.c_int64(sp),
                               C.c_int64(sv), C.c_int(5), C.c_int(10), _p(ishuf, C.c_int32) if shuffle else None)
    ga, gg = dev(anal), dev(gues)
    ctx().additive_inflation(k, nv, npts, nij1, ga, dev(add), sp, sm, sv, 0.3, weight=dev(w) if w is not None else None,
                             qmean=gg[k * sm:] if q_ratio else None, q_sp=sp, q_sv=sv, iv_q_first=5, iv_q_last=10,
                             ishuf=dev(ishuf) if shuffle else None)
    torch.cuda.synchronize()
    got = ga.cpu().numpy()
    a3 = got.reshape(nv, nens, npts)
    assert np.array_equal(a3[:, k], anal.reshape(nv, nens, npts)[:, k])     # the mean slot is not touched
    assert np.allclose(got, ref, rtol=1e-15, atol=1e-17)
    assert not np.array_equal(a3[:, :k], anal.reshape(nv, nens, npts)[:, :k])
