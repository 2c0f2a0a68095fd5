"""ctypes bindings to the CHECKERS: oracle/liboracle.so (C restatement) and, where it was
built, oracle/_ref/libletkf_ref.so (the reference's own Fortran).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libletkf_ref.so")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_dp)


def build_oracle():
    src = os.path.join(ORACLE_DIR, "letkf_oracle.c")
    if (not os.path.exists(ORACLE_SO)) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        _oracle = C.CDLL(build_oracle())
        _oracle.orc_letkf_core.restype = C.c_int
        _oracle.orc_mtx_eigen.restype = C.c_int
        _oracle.orc_das_letkf_points.restype = C.c_int
        _oracle.orc_relax_beta.restype = C.c_double
        _oracle.orc_obs_local_cal.restype = C.c_double
        _oracle.orc_pythag.restype = C.c_double
    return _oracle


_ref = None


def ref():
    """The compiled reference, or None when oracle/_ref was not built (no /root/reference)."""
    global _ref
    if _ref is None and os.path.exists(REF_SO):
        _ref = C.CDLL(REF_SO)
    return _ref


def _flags(transm, pao, rdiag_wloc, infl_update, depd, transmd):
    f = 0
    f |= 1 if transm else 0
    f |= 2 if pao else 0
    if rdiag_wloc is not None:
        f |= 4 | (8 if rdiag_wloc else 0)
    if infl_update is not None:
        f |= 16 | (32 if infl_update else 0)
    f |= 64 if depd else 0
    f |= 128 if transmd else 0
    return f


def letkf_core(which, ne, nobs, nobsl, hdxb, rdiag, rloc, dep, parm_infl, want_transm=True,
               want_pao=True, rdiag_wloc=None, infl_update=None, depd=None, want_transmd=False):
    """Run letkf_core through `which` in {"oracle", "ref"}.  hdxb is (nobs, ne) Fortran-ordered.
    Returns dict(trans, transm, pao, transmd, parm_infl, rc)."""
    hdxb = np.asfortranarray(hdxb, dtype=np.float64)
    assert hdxb.shape == (nobs, ne)
    trans = np.zeros((ne, ne), order="F")
    transm = np.zeros(ne)
    pao = np.zeros((ne, ne), order="F")
    transmd = np.zeros(ne)
    infl = C.c_double(parm_infl)
    dd = depd if depd is not None else np.zeros(max(nobs, 1))
    if which == "ref":
        lib = ref()
        assert lib is not None, "oracle/_ref not built"
        fl = _flags(want_transm, want_pao, rdiag_wloc, infl_update, depd is not None, want_transmd)
        lib.ref_letkf_core(C.c_int(ne), C.c_int(nobs), C.c_int(nobsl), _dp(hdxb), _dp(rdiag), _dp(rloc),
                           _dp(dep), C.byref(infl), _dp(trans), _dp(transm), _dp(pao), _dp(dd), _dp(transmd),
                           C.c_int(fl))
        rc = 0
    else:
        lib = oracle()
        wl = C.c_int(1 if rdiag_wloc else 0)
        iu = C.c_int(1 if infl_update else 0)
        rc = lib.orc_letkf_core(
            C.c_int(ne), C.c_int(nobs), C.c_int(nobsl), _dp(hdxb), _dp(rdiag), _dp(rloc), _dp(dep), C.byref(infl),
            _dp(trans), _dp(transm) if want_transm else None, _dp(pao) if want_pao else None,
            C.byref(wl) if rdiag_wloc is not None else None, C.byref(iu) if infl_update is not None else None,
            _dp(depd) if depd is not None else None, _dp(transmd) if want_transmd else None)
    return dict(trans=trans, transm=transm if want_transm else None, pao=pao if want_pao else None,
                transmd=transmd if (want_transmd and depd is not None) else None, parm_infl=infl.value, rc=rc)


def mtx_eigen(which, a):
    n = a.shape[0]
    a = np.asfortranarray(a, dtype=np.float64)
    w = np.zeros(n)
    v = np.zeros((n, n), order="F")
    nr = C.c_int(0)
    if which == "ref":
        ref().ref_mtx_eigen(C.c_int(n), _dp(a), _dp(w), _dp(v), C.byref(nr))
        rc = 0
    else:
        rc = oracle().orc_mtx_eigen(C.c_int(n), _dp(a), _dp(w), _dp(v), C.byref(nr))
    return w, v, nr.value, rc


class DasParams(C.Structure):
    _fields_ = [("k", C.c_int), ("nv", C.c_int), ("det_run", C.c_int), ("infl_adaptive", C.c_int),
                ("relax_to_inflated_prior", C.c_int), ("relax_alpha", C.c_double),
                ("relax_alpha_spread", C.c_double), ("q_update_top", C.c_double), ("q_sprd_max", C.c_double),
                ("iv_p", C.c_int), ("iv_q_first", C.c_int), ("iv_q_last", C.c_int), ("nthreads", C.c_int),
                ("var_mask", C.c_uint)]


def das_points(params, obs_off, obs_idx, rdiag_l, rloc_l, ensval, dep, beta, infl, gues, sp, sm, sv,
               want_trans=False, want_pa=False, want_rtps=False):
    """orc_das_letkf_points on flat numpy buffers.  gues: 1-D float64 buffer holding perturbations+mean(+det).
    Returns dict(anal, infl, trans, transm, pa, status, rc)."""
    k = params.k
    npts = len(obs_off) - 1
    anal = np.zeros_like(gues)
    infl = np.array(infl, dtype=np.float64, copy=True)
    trans = np.zeros((npts, k * k)) if want_trans else None
    transm = np.zeros((npts, k)) if want_trans else None
    pa = np.zeros((npts, k * k)) if want_pa else None
    status = np.zeros(npts, dtype=np.int32)
    obs_off = np.ascontiguousarray(obs_off, dtype=np.int64)
    obs_idx = np.ascontiguousarray(obs_idx, dtype=np.int32)
    kld = ensval.shape[1]
    rtps = np.zeros(npts * params.nv) if want_rtps else None
    oracle().orc_das_letkf_points_diag.restype = C.c_int
    rc = oracle().orc_das_letkf_points_diag(
        C.byref(params), C.c_int64(npts), obs_off.ctypes.data_as(C.POINTER(C.c_int64)),
        obs_idx.ctypes.data_as(C.POINTER(C.c_int32)), _dp(rdiag_l), _dp(rloc_l), _dp(ensval), C.c_int64(kld),
        _dp(dep), _dp(beta), _dp(infl), _dp(gues), _dp(anal), C.c_int64(sp), C.c_int64(sm), C.c_int64(sv),
        _dp(trans), _dp(transm), _dp(pa), status.ctypes.data_as(C.POINTER(C.c_int32)), _dp(rtps))
    return dict(anal=anal, infl=infl, trans=trans, transm=transm, pa=pa, status=status, rc=rc, rtps=rtps)


def das_level1_2d(params, nv2d, n2nc, n2n, nclass, nij1, obs_off, obs_idx, rdiag_l, rloc_l, ensval, dep, beta,
                  work3d, work2d, gues3, s3, gues2, s2):
    """orc_das_letkf_level1_2d (scale/letkf/letkf_tools.f90:313-686): obs_off [nclass][nij1+1] into the concatenated
    per-class lists; s3 / s2 = (sp, sm, sv).  Returns dict(anal3, anal2, work3d, work2d, rc)."""
    anal3, anal2 = np.full_like(gues3, np.nan), np.full_like(gues2, np.nan)
    w3, w2 = np.array(work3d, dtype=np.float64, copy=True), np.array(work2d, dtype=np.float64, copy=True)
    obs_off = np.ascontiguousarray(obs_off, dtype=np.int64).reshape(-1)
    obs_idx = np.ascontiguousarray(obs_idx, dtype=np.int32)
    n2nc = np.ascontiguousarray(n2nc, dtype=np.int32)
    n2n = np.ascontiguousarray(n2n, dtype=np.int32)
    f = oracle().orc_das_letkf_level1_2d
    f.restype = C.c_int
    rc = f(C.byref(params), C.c_int(nv2d), n2nc.ctypes.data_as(C.POINTER(C.c_int32)),
           n2n.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int(nclass), C.c_int64(nij1),
           obs_off.ctypes.data_as(C.POINTER(C.c_int64)), obs_idx.ctypes.data_as(C.POINTER(C.c_int32)), _dp(rdiag_l),
           _dp(rloc_l), _dp(ensval), C.c_int64(ensval.shape[1]), _dp(dep), _dp(beta), _dp(w3), _dp(w2), _dp(gues3),
           _dp(anal3), C.c_int64(s3[0]), C.c_int64(s3[1]), C.c_int64(s3[2]), _dp(gues2), _dp(anal2), C.c_int64(s2[0]),
           C.c_int64(s2[1]), C.c_int64(s2[2]))
    return dict(anal3=anal3, anal2=anal2, work3d=w3, work2d=w2, rc=rc)
