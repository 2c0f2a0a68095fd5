"""Builds the search tables that set_letkf_obs leaves behind (scale/letkf/letkf_obs.f90:655-976: per-ctype sorting
mesh, (ctype, j, i) ordering of obsda_sort, prefix sums ac_ext) for a synthetic single-subdomain case, as numpy arrays
plus the two layout-identical C structs (oracle / C ABI)."""
import ctypes as C
import math

import numpy as np

DIST_ZERO_FAC = float(np.float32(3.651483717))


class SearchTables(C.Structure):
    _fields_ = [("nctype", C.c_int32), ("ngroup", C.c_int32), ("criterion", C.c_int32), ("nlon", C.c_int32),
                ("nlat", C.c_int32), ("limit_hint", C.c_int32), ("dx", C.c_double), ("dy", C.c_double),
                ("i_org", C.c_double), ("j_org", C.c_double), ("rain_base", C.c_double),
                ("group_start", C.c_void_p), ("group_member", C.c_void_p), ("vmode", C.c_void_p),
                ("hori_loc", C.c_void_p), ("vert_loc", C.c_void_p), ("varloc", C.c_void_p), ("max_nobs", C.c_void_p),
                ("ngrd_i", C.c_void_p), ("ngrd_j", C.c_void_p), ("ngrdsch_i", C.c_void_p), ("ngrdsch_j", C.c_void_p),
                ("ngrdext_i", C.c_void_p), ("ngrdext_j", C.c_void_p), ("ac_off", C.c_void_p), ("ac_ext", C.c_void_p),
                ("ob_ri", C.c_void_p), ("ob_rj", C.c_void_p), ("ob_lev", C.c_void_p), ("ob_dat", C.c_void_p),
                ("ob_err", C.c_void_p)]


ARRAY_FIELDS = ["group_start", "group_member", "vmode", "hori_loc", "vert_loc", "varloc", "max_nobs", "ngrd_i",
                "ngrd_j", "ngrdsch_i", "ngrdsch_j", "ngrdext_i", "ngrdext_j", "ac_off", "ac_ext", "ob_ri", "ob_rj",
                "ob_lev", "ob_dat", "ob_err"]


def build_case(seed, nlon=40, nlat=32, dx=1000.0, nobs_per_ctype=(900, 500, 400, 300), max_nobs=(0, 0, 0, 0),
               criterion=1, ihalo=2, npts=160):
    """4 combined obs types: 0 radar reflectivity and 1 radar zero-reflectivity (type 22, z localisation, merged into
    one group like letkf_tools.f90:167-192), 2 upper-air T (ln p), 3 surface pressure (ps value as vertical
    coordinate, larger horizontal scale)."""
    rng = np.random.default_rng(seed)
    nctype = 4
    vmode = np.array([1, 1, 0, 2], dtype=np.int32)
    hori_loc = np.array([4000.0, 3000.0, 6000.0, 8000.0])
    vert_loc = np.array([2000.0, 2000.0, 0.4, 0.3])
    varloc = np.array([1.0, 1.0, 0.8, 1.0])
    groups = [[0, 1], [2], [3]]
    group_start = np.array([0, 2, 3, 4], dtype=np.int32)
    group_member = np.array([0, 1, 2, 3], dtype=np.int32)
    dy = dx
    # mesh (letkf_obs.f90:655-695)
    ngrd_i = np.zeros(nctype, np.int32)
    ngrd_j = np.zeros(nctype, np.int32)
    nsch_i = np.zeros(nctype, np.int32)
    nsch_j = np.zeros(nctype, np.int32)
    for ic in range(nctype):
        spc = hori_loc[ic] * DIST_ZERO_FAC / 6.0
        ngrd_i[ic] = min(math.ceil(dx * nlon / spc), nlon)
        ngrd_j[ic] = min(math.ceil(dy * nlat / spc), nlat)
        nsch_i[ic] = math.ceil(hori_loc[ic] * DIST_ZERO_FAC / (dx * nlon / ngrd_i[ic]))
        nsch_j[ic] = math.ceil(hori_loc[ic] * DIST_ZERO_FAC / (dy * nlat / ngrd_j[ic]))
    next_i = ngrd_i + 2 * nsch_i
    next_j = ngrd_j + 2 * nsch_j
    i_org = ihalo + 0.5   # ri counts from 1 + IHALO at the first interior point (common_mpi_scale.f90:303-308)
    j_org = ihalo + 0.5
    ob = {k: [] for k in ("ri", "rj", "lev", "dat", "err")}
    ac_all, ac_off = [], []
    row0 = 0
    for ic in range(nctype):
        n = nobs_per_ctype[ic]
        # obs also live in the halo of the extended subdomain
        ext_i = nsch_i[ic] * nlon / ngrd_i[ic]
        ext_j = nsch_j[ic] * nlat / ngrd_j[ic]
        ri = i_org + rng.uniform(-0.95 * ext_i, nlon + 0.95 * ext_i, n)
        rj = j_org + rng.uniform(-0.95 * ext_j, nlat + 0.95 * ext_j, n)
        lev = rng.uniform(0.0, 12000.0, n) if vmode[ic] == 1 else rng.uniform(2.0e4, 1.0e5, n)
        dat = rng.uniform(9.0e4, 1.03e5, n)
        err = rng.choice([1.0, 3.0, 5.0], n)
        ogi = np.ceil((ri - i_org) * ngrd_i[ic] / nlon).astype(np.int64) + nsch_i[ic]
        ogj = np.ceil((rj - j_org) * ngrd_j[ic] / nlat).astype(np.int64) + nsch_j[ic]
        ogi = np.clip(ogi, 1, next_i[ic])
        ogj = np.clip(ogj, 1, next_j[ic])
        order = np.lexsort((np.arange(n), ogi, ogj))            # (j, i) major, stable
        cell = (ogj[order] - 1) * next_i[ic] + (ogi[order] - 1)
        counts = np.bincount(cell, minlength=next_i[ic] * next_j[ic]).reshape(next_j[ic], next_i[ic])
        ac = np.zeros((next_j[ic], next_i[ic] + 1), dtype=np.int64)
        run = row0
        for j in range(next_j[ic]):
            ac[j, 0] = run
            ac[j, 1:] = run + np.cumsum(counts[j])
            run = ac[j, -1]
        ac_off.append(sum(a.size for a in ac_all))
        ac_all.append(ac.reshape(-1).astype(np.int32))
        for k, v in zip(("ri", "rj", "lev", "dat", "err"), (ri, rj, lev, dat, err)):
            ob[k].append(v[order])
        row0 += n
    arr = dict(group_start=group_start, group_member=group_member, vmode=vmode, hori_loc=hori_loc, vert_loc=vert_loc,
               varloc=varloc, max_nobs=np.array(max_nobs, dtype=np.int32), ngrd_i=ngrd_i, ngrd_j=ngrd_j,
               ngrdsch_i=nsch_i, ngrdsch_j=nsch_j, ngrdext_i=next_i.astype(np.int32),
               ngrdext_j=next_j.astype(np.int32), ac_off=np.array(ac_off, dtype=np.int64),
               ac_ext=np.concatenate(ac_all), ob_ri=np.concatenate(ob["ri"]), ob_rj=np.concatenate(ob["rj"]),
               ob_lev=np.concatenate(ob["lev"]), ob_dat=np.concatenate(ob["dat"]), ob_err=np.concatenate(ob["err"]))
    scal = dict(nctype=nctype, ngroup=len(groups), criterion=criterion, nlon=nlon, nlat=nlat, dx=dx, dy=dy,
                i_org=i_org, j_org=j_org, rain_base=8.5e4)
    pts = dict(ri=i_org + rng.uniform(0.5, nlon - 0.5, npts), rj=j_org + rng.uniform(0.5, nlat - 0.5, npts),
               rlev=rng.uniform(2.5e4, 1.0e5, npts), rz=rng.uniform(0.0, 12000.0, npts))
    return dict(arr=arr, scal=scal, pts=pts, ctype_rows=np.cumsum([0] + list(nobs_per_ctype)), groups=groups,
                nobs=row0)


def host_struct(case):
    """(struct, keepalive) with numpy-backed pointers, for the oracle."""
    t = SearchTables()
    for k, v in case["scal"].items():
        setattr(t, k, v)
    keep = []
    for f in ARRAY_FIELDS:
        a = np.ascontiguousarray(case["arr"][f])
        keep.append(a)
        setattr(t, f, a.ctypes.data)
    return t, keep


def device_struct(case, dev):
    """(struct, keepalive) with torch device pointers, for the C ABI."""
    import torch
    t = SearchTables()
    for k, v in case["scal"].items():
        setattr(t, k, v)
    keep = []
    for f in ARRAY_FIELDS:
        a = torch.from_numpy(np.ascontiguousarray(case["arr"][f])).to(dev)
        keep.append(a)
        setattr(t, f, a.data_ptr())
    return t, keep


def oracle_lists(case, cap=20000):
    import _oracle
    lib = _oracle.oracle()
    lib.orc_obs_local.restype = C.c_int
    t, keep = host_struct(case)
    p = case["pts"]
    out = []
    idx = np.zeros(cap, dtype=np.int32)
    rd = np.zeros(cap)
    rl = np.zeros(cap)
    ds = np.zeros(cap)
    dp = C.POINTER(C.c_double)
    for i in range(len(p["ri"])):
        n = lib.orc_obs_local(C.byref(t), C.c_double(p["ri"][i]), C.c_double(p["rj"][i]), C.c_double(p["rlev"][i]),
                              C.c_double(p["rz"][i]), C.c_int(cap), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                              rd.ctypes.data_as(dp), rl.ctypes.data_as(dp), ds.ctypes.data_as(dp))
        assert n >= 0
        out.append((idx[:n].copy(), rd[:n].copy(), rl[:n].copy(), ds[:n].copy()))
    return out


def host_struct_from_torch(t, keep):
    """(struct, keepalive): a numpy-backed copy of a SearchTables whose arrays are torch tensors (the dict that
    bench_workload.search_tables returns beside the struct) -- what the oracle's orc_obs_local reads."""
    h = SearchTables()
    for name, _ in SearchTables._fields_:
        if name not in ARRAY_FIELDS:
            setattr(h, name, getattr(t, name))
    alive = []
    for f in ARRAY_FIELDS:
        a = np.ascontiguousarray(keep[f].detach().cpu().numpy())
        alive.append(a)
        setattr(h, f, a.ctypes.data)
    return h, alive


def oracle_csr(h, ri, rj, rlev, rz, cap=1 << 16, nthreads=8):
    """obs_local of the ORACLE (oracle/letkf_oracle.c orc_obs_local = scale/letkf/letkf_tools.f90:1325-1759, every selection
    mode) for the points (ri, rj, rlev, rz): CSR lists (off int64, idx int32, rdiag, rloc) whose entries are rows of the
    table `h` describes, plus tied[i] = 1 where a limited group's selection fell between equal keys (either choice is the
    reference's: its quick-select is unstable).  The points are dealt to a few host threads (ctypes drops the GIL)."""
    import threading
    import _oracle
    lib = _oracle.oracle()
    f = lib.orc_obs_local_tied
    f.restype = C.c_int
    n = len(ri)
    res = [None] * n
    tied = np.zeros(n, dtype=np.int32)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)

    def work(r):
        c = cap
        idx, rd, rl = np.zeros(c, dtype=np.int32), np.zeros(c), np.zeros(c)
        tf = C.c_int(0)
        for i in range(r, n, nthreads):
            while True:
                m = f(C.byref(h), C.c_double(ri[i]), C.c_double(rj[i]), C.c_double(rlev[i]), C.c_double(rz[i]), C.c_int(c),
                      idx.ctypes.data_as(ip), rd.ctypes.data_as(dp), rl.ctypes.data_as(dp), None, C.byref(tf))
                if m >= 0:
                    break
                c *= 2
                idx, rd, rl = np.zeros(c, dtype=np.int32), np.zeros(c), np.zeros(c)
            res[i] = (idx[:m].copy(), rd[:m].copy(), rl[:m].copy())
            tied[i] = tf.value
    th = [threading.Thread(target=work, args=(r,)) for r in range(min(nthreads, max(n, 1)))]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    off = np.zeros(n + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(r[0]) for r in res])
    cat = lambda j, dt: (np.concatenate([r[j] for r in res]) if n else np.zeros(0)).astype(dt)
    return off, cat(0, np.int32), cat(1, np.float64), cat(2, np.float64), tied
