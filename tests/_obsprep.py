"""Synthetic multi-subdomain observation sets for the set_letkf_obs row (f2): every rank's local H(x) table as
obsope leaves it (scale/letkf/letkf_obs.f90:78-360: rows = observations inside the subdomain, member-fastest), the
sorting meshes, and the oracle pipeline departure -> bucket sort -> [all-gather] -> extended-subdomain plan."""
import ctypes as C

import numpy as np

import _oracle

UNDEF = -9.99e33
ID_U, ID_RADAR_REF, ID_RADAR_VR, ID_RAIN = 2819, 4001, 4002, 19999


class QcParams(C.Structure):
    _fields_ = [("member", C.c_int32), ("det_run", C.c_int32), ("use_radar_ref", C.c_int32),
                ("use_radar_vr", C.c_int32), ("min_radar_ref_member", C.c_int32),
                ("min_radar_ref_member_obsref", C.c_int32), ("radar_ref_thres_dbz", C.c_double),
                ("gross_error", C.c_double), ("gross_error_rain", C.c_double), ("gross_error_radar_ref", C.c_double),
                ("gross_error_radar_vr", C.c_double), ("gross_error_radar_prh", C.c_double),
                ("gross_error_tcx", C.c_double), ("gross_error_tcy", C.c_double), ("gross_error_tcp", C.c_double),
                ("h08", C.c_int32), ("h08_min_cld_member", C.c_int32), ("h08_limit_lev", C.c_double),
                ("gross_error_h08", C.c_double), ("h08_bt_min", C.c_double), ("h08_lev", C.c_void_p),
                ("h08_val2", C.c_void_p)]


class Mesh(C.Structure):
    _fields_ = [("nctype", C.c_int32), ("nlon", C.c_int32), ("nlat", C.c_int32), ("ihalo", C.c_int32),
                ("jhalo", C.c_int32), ("rank_i", C.c_int32), ("rank_j", C.c_int32), ("fix_ij_obsgrd", C.c_int32),
                ("ngrd_i", C.c_void_p), ("ngrd_j", C.c_void_p)]


class HaloLayout(C.Structure):
    _fields_ = [("nctype", C.c_int32), ("nprocs", C.c_int32), ("prc_num_x", C.c_int32), ("myrank", C.c_int32),
                ("ngrd_i", C.c_void_p), ("ngrd_j", C.c_void_p), ("ngrdsch_i", C.c_void_p), ("ngrdsch_j", C.c_void_p)]


def fill(struct_cls, **kw):
    s = struct_cls()
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def qc_params(cls, k, det_run, **over):
    d = dict(member=k, det_run=int(det_run), use_radar_ref=1, use_radar_vr=1, min_radar_ref_member=3,
             min_radar_ref_member_obsref=2, radar_ref_thres_dbz=15.0, gross_error=5.0, gross_error_rain=4.0,
             gross_error_radar_ref=3.0, gross_error_radar_vr=2.5, gross_error_radar_prh=5.0, gross_error_tcx=5.0,
             gross_error_tcy=5.0, gross_error_tcp=5.0)
    d.update(over)
    return fill(cls, **d)


def make_world(seed, px=2, py=2, nlon=12, nlat=12, k=10, det_run=True, nobs=3000, ihalo=2,
               ngrd=((4, 4), (6, 6), (3, 3)), ngrdsch=((2, 2), (3, 3), (1, 1))):
    """Global set of `nobs` observations of 3 combined types (0 radar reflectivity, 1 radial velocity, 2 upper-air u)
    scattered over a (px*nlon) x (py*nlat) domain; rank r = (r % px, r // px) owns those inside its subdomain."""
    rng = np.random.default_rng(seed)
    nctype = len(ngrd)
    kld = k + (1 if det_run else 0)
    ctype = rng.integers(0, nctype, nobs).astype(np.int32)
    elm = np.array([ID_RADAR_REF, ID_RADAR_VR, ID_U], dtype=np.int32)[ctype]
    ri = ihalo + 0.5 + rng.uniform(0.0, px * nlon, nobs)
    rj = ihalo + 0.5 + rng.uniform(0.0, py * nlat, nobs)
    dat = np.where(ctype == 0, rng.uniform(5.0, 40.0, nobs), rng.normal(0.0, 4.0, nobs))
    dat[rng.random(nobs) < 0.02] = UNDEF
    err = rng.choice([1.0, 2.0, 3.0], nobs)
    ens = dat[:, None] * (dat[:, None] != UNDEF) + rng.normal(0.0, 3.0, (nobs, kld))
    ens[rng.random(nobs) < 0.05] += 12.0          # gross errors
    low = (ctype == 0) & (rng.random(nobs) < 0.2)  # reflectivity rows where few members see rain
    ens[low] = rng.uniform(0.0, 14.0, (int(low.sum()), kld))
    qc0 = np.where(rng.random(nobs) < 0.1, rng.choice([10, 20, 21, 97], nobs), 0).astype(np.int32)
    ranks = []
    for r in range(px * py):
        pi, pj = r % px, r // px
        inside = ((ri - ihalo - 0.5 > pi * nlon) & (ri - ihalo - 0.5 <= (pi + 1) * nlon) &
                  (rj - ihalo - 0.5 > pj * nlat) & (rj - ihalo - 0.5 <= (pj + 1) * nlat))
        g = np.nonzero(inside)[0]
        ranks.append(dict(rank=r, pi=pi, pj=pj, gidx=g, ctype=ctype[g].copy(), elm=elm[g].copy(), ri=ri[g].copy(),
                          rj=rj[g].copy(), dat=dat[g].copy(), err=err[g].copy(), ensval=ens[g].copy(),
                          qc=qc0[g].copy()))
    gi = np.array([g[0] for g in ngrd], dtype=np.int32)
    gj = np.array([g[1] for g in ngrd], dtype=np.int32)
    si = np.array([s[0] for s in ngrdsch], dtype=np.int32)
    sj = np.array([s[1] for s in ngrdsch], dtype=np.int32)
    return dict(px=px, py=py, nlon=nlon, nlat=nlat, k=k, kld=kld, det_run=det_run, ihalo=ihalo, nctype=nctype,
                ngrd_i=gi, ngrd_j=gj, ngrdsch_i=si, ngrdsch_j=sj, ranks=ranks, nobs=nobs,
                ncell=int((gi.astype(np.int64) * gj).sum()),
                nacx=int(((gi + 2 * si + 1).astype(np.int64) * (gj + 2 * sj)).sum()),
                glob=dict(ctype=ctype, ri=ri, rj=rj))


def mesh_struct(cls, w, rk):
    return fill(cls, nctype=w["nctype"], nlon=w["nlon"], nlat=w["nlat"], ihalo=w["ihalo"], jhalo=w["ihalo"],
                rank_i=rk["pi"], rank_j=rk["pj"], fix_ij_obsgrd=int(w.get("fix_ij_obsgrd", 0)),
                ngrd_i=w["ngrd_i"].ctypes.data, ngrd_j=w["ngrd_j"].ctypes.data)


def layout_struct(cls, w, myrank):
    return fill(cls, nctype=w["nctype"], nprocs=w["px"] * w["py"], prc_num_x=w["px"], myrank=myrank,
                ngrd_i=w["ngrd_i"].ctypes.data, ngrd_j=w["ngrd_j"].ctypes.data,
                ngrdsch_i=w["ngrdsch_i"].ctypes.data, ngrdsch_j=w["ngrdsch_j"].ctypes.data)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def h08_rows(seed, k, det_run, nobs):
    """Rows of a -DH08 build: half Himawari-8 IR (elm 8800: brightness temperatures, cloudy members NEGATIVE, a sensitive
    height per row, a clear-sky BT in val2), half upper-air u; undefined / too-cold observations and rows above the level
    limit mixed in."""
    rng = np.random.default_rng(seed)
    kld = k + (1 if det_run else 0)
    h08 = rng.random(nobs) < 0.5
    elm = np.where(h08, 8800, 2819).astype(np.int32)
    bt = 250.0 + 20.0 * rng.standard_normal((nobs, kld))
    cloudy = rng.random((nobs, kld)) < rng.choice([0.0, 0.02, 0.3], size=(nobs, 1))
    ens = np.where(h08[:, None], np.where(cloudy, -bt, bt), 5.0 * rng.standard_normal((nobs, kld)))
    dat = np.where(h08, 250.0 + 25.0 * rng.standard_normal(nobs), 5.0 * rng.standard_normal(nobs))
    dat[rng.random(nobs) < 0.05] = -9.99e33
    dat[(rng.random(nobs) < 0.05) & h08] = 150.0                      # below H08_BT_MIN = 180
    err = np.where(h08, 3.0, 1.0) * (0.5 + rng.random(nobs))
    lev = np.where(h08, rng.uniform(5000.0, 90000.0, nobs), 50000.0)
    val2 = np.where(h08, 255.0 + 10.0 * rng.standard_normal(nobs), 0.0)
    qc = np.where(rng.random(nobs) < 0.1, 21, 0).astype(np.int32)
    return dict(elm=elm, ens=np.ascontiguousarray(ens), dat=dat, err=err, lev=lev, val2=val2, qc=qc, kld=kld, h08=h08)
def oracle_departure_h08(r, k, det_run, with_val2, **over):
    """The oracle's departure + QC on rows of h08_rows.  Returns (ensval, val, qc, val2)."""
    nobs = len(r["qc"])
    ens, qc, v2, lev = r["ens"].copy(), r["qc"].copy(), r["val2"].copy(), r["lev"].copy()
    val = np.zeros(nobs)
    prm = qc_params(QcParams, k, det_run, **over)
    prm.h08_lev = lev.ctypes.data
    prm.h08_val2 = v2.ctypes.data if with_val2 else None
    _oracle.oracle().orc_obs_departure(C.byref(prm), C.c_int64(nobs), _p(r["elm"], C.c_int32), _p(r["dat"], C.c_double),
                                       _p(r["err"], C.c_double), _p(ens, C.c_double), C.c_int64(r["kld"]),
                                       _p(val, C.c_double), _p(qc, C.c_int32))
    return ens, val, qc, v2


def oracle_rank_stage12(w, rk):
    """departure + QC, then the bucket sort, through the oracle.  Returns dict(ensval, val, qc, n_cell, key)."""
    lib = _oracle.oracle()
    lib.orc_obs_mesh_sort.restype = C.c_int64
    n = len(rk["qc"])
    ens = np.ascontiguousarray(rk["ensval"]).copy()
    val = np.zeros(max(n, 1))
    qc = rk["qc"].copy()
    prm = qc_params(QcParams, w["k"], w["det_run"])
    lib.orc_obs_departure(C.byref(prm), C.c_int64(n), _p(rk["elm"], C.c_int32), _p(rk["dat"], C.c_double),
                          _p(rk["err"], C.c_double), _p(ens, C.c_double), C.c_int64(w["kld"]), _p(val, C.c_double),
                          _p(qc, C.c_int32))
    m = mesh_struct(Mesh, w, rk)
    n_cell = np.zeros(max(w["ncell"], 1), dtype=np.int32)
    key = np.zeros(max(n, 1), dtype=np.int32)
    ns = lib.orc_obs_mesh_sort(C.byref(m), C.c_int64(n), _p(rk["ctype"], C.c_int32), _p(rk["ri"], C.c_double),
                               _p(rk["rj"], C.c_double), _p(qc, C.c_int32), _p(n_cell, C.c_int32),
                               _p(key, C.c_int32))
    return dict(ensval=ens, val=val[:n], qc=qc, n_cell=n_cell[:w["ncell"]], key=key[:ns])


def oracle_plan(w, myrank, n_all, cap):
    lib = _oracle.oracle()
    lib.orc_obs_halo_plan.restype = C.c_int64
    lay = layout_struct(HaloLayout, w, myrank)
    ac_ext = np.zeros(max(w["nacx"], 1), dtype=np.int32)
    src_row = np.zeros(max(cap, 1), dtype=np.int32)
    n_all = np.ascontiguousarray(n_all, dtype=np.int32)
    nt = lib.orc_obs_halo_plan(C.byref(lay), _p(n_all, C.c_int32), _p(ac_ext, C.c_int32), _p(src_row, C.c_int32),
                               C.c_int64(cap))
    return ac_ext[:w["nacx"]], src_row[:max(nt, 0)], nt
