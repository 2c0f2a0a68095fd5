"""das_letkf from a Fortran host on the device: scale-letkf_amd/fortran/letkf_tools_amd.f90 `das_letkf_amd` is what a
maintainer CALLs instead of `das_letkf(gues3d,gues2d,anal3d,anal2d)` (scale/letkf/letkf.f90:196); the module state the
reference's routine reads (scale/letkf/letkf_tools.f90:50-58 and what it USEs) arrives as two derived types.

CPU: every TYPE, BIND(C) of letkf_amd_api.f90 lists the C struct's fields in the C order with the matching kinds, and the
module + driver compile and link.  GPU: a Fortran program runs the whole routine -- set-up helpers, upload, relax_beta,
inflation field, per variable-localisation class obs_local (column search with a limit on a merged radar group) and the
loop body, ensmean_grd -- on a multi-level domain with four combined observation types and two classes, and must reproduce
the oracle's restatement of the same steps (obs_local :1325-1759, the point update :313-527, relax_beta :1911-1948)."""
import ctypes as C
import os
import re
import struct
import subprocess
import tempfile

import numpy as np
import pytest

import _oracle
from __graft_entry__ import PKG_DIR, load_package
from _search import build_case, host_struct

FDIR = os.path.join(PKG_DIR, "fortran")
DRIVER = os.path.join(FDIR, "build", "das_letkf_driver")
HAVE_FC = os.path.exists("/opt/rocm/bin/amdflang")


def build_fortran():
    load_package().build()
    subprocess.check_call(["make", "-C", FDIR], stdout=subprocess.DEVNULL)


def c_structs(hdr):
    """{name: [(kind, field), ...]} for every `typedef struct { ... } name;` of the header; kind in i32 / i64 / f64 / ptr"""
    out = {}
    for m in re.finditer(r"typedef struct \{(.*?)\}\s*(\w+);", hdr, flags=re.S):
        body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            base = "i32" if "int32_t" in decl else "i64" if "int64_t" in decl else "f64" if "double" in decl else None   # (uint32_t: a 32-bit mask)
            assert base, decl
            decl = re.sub(r"\b(const|u?int32_t|int64_t|double)\b", "", decl)
            for part in decl.split(","):
                part = part.strip()
                ptr = "*" in part
                name = part.replace("*", "").strip()
                name = re.sub(r"\[\d+\]", "", name)
                fields.append(("ptr" if ptr else base, name))
        out[m.group(2)] = fields
    return out


def f_types(src):
    out = {}
    for m in re.finditer(r"TYPE, BIND\(C\) :: (\w+)\n(.*?)END TYPE", src, flags=re.S):
        fields = []
        body = re.sub(r"&\s*\n", " ", m.group(2))
        for line in body.splitlines():
            line = line.split("!")[0]
            if "::" not in line:
                continue
            decl, names = line.split("::")
            kind = ("i32" if "c_int32_t" in decl else "i64" if "c_int64_t" in decl else "f64" if "c_double" in decl
                    else "ptr" if "c_ptr" in decl else None)
            assert kind, line
            for n in names.split(","):
                fields.append((kind, re.sub(r"\(\d+\)", "", n.strip())))
        out[m.group(1)] = fields
    return out


@pytest.mark.skipif(not HAVE_FC, reason="amdflang not present")
def test_every_bind_c_type_mirrors_its_c_struct():
    hdr = open(os.path.join(PKG_DIR, "..", "include", "letkf_amd.h")).read()
    src = open(os.path.join(FDIR, "letkf_amd_api.f90")).read()
    cs, fs = c_structs(hdr), f_types(src)
    # every struct of the C ABI has its Fortran mirror ...
    assert set(cs) == set(fs), (sorted(set(cs) - set(fs)), sorted(set(fs) - set(cs)))
    for name in cs:                 # ... field by field, in the C order, with the C kinds
        assert fs[name] == cs[name], (name, fs[name], cs[name])
    # and every device / host entry point of the header is bound by name
    entries = set(re.findall(r"\n(?:int|void)\s+(letkf_\w+)\(", hdr)) - {"letkf_core_c", "letkf_sched_plan_check", "letkf_sched_plan_check_units"}
    bound = set(re.findall(r"BIND\(C, name='(letkf_\w+)'\)", src))
    harness_only = {"letkf_ctx_timing_enable", "letkf_ctx_timing_read", "letkf_ctx_last_path"}
    assert entries - bound <= harness_only, sorted(entries - bound - harness_only)


@pytest.mark.skipif(not HAVE_FC, reason="amdflang not present")
def test_das_letkf_amd_compiles_and_links():
    build_fortran()
    assert os.path.exists(DRIVER)
    src = open(os.path.join(FDIR, "letkf_tools_amd.f90")).read()
    assert "SUBROUTINE das_letkf_amd(" in src
    for call in ("letkf_var_local_classes", "letkf_ctype_merge_groups", "letkf_radar_only", "letkf_ens_to_perturbations_dev",
                 "letkf_infl_init_dev", "letkf_relax_beta_dev", "letkf_das_columns_dev", "letkf_ens_mean_dev"):
        assert call in src, call


class OrcBeta(C.Structure):
    _fields_ = [("radar_only", C.c_int), ("radar_zmax", C.c_double), ("vert_local_radar", C.c_double),
                ("boundary_buffer_width", C.c_double), ("dx", C.c_double), ("dy", C.c_double), ("ihalo", C.c_int),
                ("jhalo", C.c_int), ("nlong", C.c_int), ("nlatg", C.c_int)]


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE_FC, reason="amdflang not present")
@pytest.mark.parametrize("k,det,relax,alpha,bw", [(20, 0, 2, 0.9, 0.0), (50, 1, 1, 0.6, 3000.0)])
def test_das_letkf_amd_from_fortran_matches_the_oracle(k, det, relax, alpha, bw):
    build_fortran()
    rng = np.random.default_rng(700 + k)
    nlon, nlat, nlev, nv, ihalo, dx = 12, 10, 4, 11, 2, 1000.0
    nij1 = nlon * nlat
    npts = nij1 * nlev
    nens = k + 1 + det
    # tables of set_letkf_obs: 4 combined types -- 0 radar reflectivity + 1 zero-reflectivity (type 22, merged, limited to
    # 12 per grid point), 2 upper-air T, 3 surface pressure (limited to 3)
    case = build_case(41 + k, nlon=nlon, nlat=nlat, dx=dx, nobs_per_ctype=(500, 300, 250, 120), max_nobs=(12, 12, 0, 3),
                      criterion=1, ihalo=ihalo, npts=1)
    arr, scal = case["arr"], case["scal"]
    nctype, nobs = 4, case["nobs"]
    elm = np.array([4001, 4004, 3073, 14593], dtype=np.int32)        # common_obs_scale.f90:56-72
    elm_u = np.array([11, 12, 3, 5], dtype=np.int32)
    typ = np.array([22, 22, 1, 2], dtype=np.int32)
    uid_vl = np.array([7, 7, 2, 4], dtype=np.int32)                   # column of var_local per obs element
    nid_obs, nobtype = 16, 24
    merge = np.zeros((nid_obs, nobtype), dtype=np.int32)
    merge[11 - 1, 22 - 1] = merge[12 - 1, 22 - 1] = 1                  # letkf_tools.f90:167-178
    var_local = np.ones((nv, 9))
    var_local[5:, 2 - 1] = 0.5                                         # moisture sees T observations at half weight: 2 classes
    # grid points (first interior point = 1 + IHALO, common_mpi_scale.f90:303-308), heights, the ensemble
    ii, jj = np.meshgrid(np.arange(nlon), np.arange(nlat))
    rig = (ii.ravel() + 1.0 + ihalo)
    rjg = (jj.ravel() + 1.0 + ihalo)
    zlev = np.array([400.0, 1800.0, 4500.0, 9000.0])
    hgt = (zlev[:, None] + rng.uniform(-50.0, 50.0, (nlev, nij1)))     # hgt1(ij, lev) point-fastest
    full = rng.normal(1.0, 1.0, (nv, nens, npts))
    pmean = (1.0e5 * np.exp(-hgt / 7500.0)).ravel()
    full[4] = pmean[None, :] + rng.normal(0.0, 40.0, (nens, npts))
    full[5:] = np.abs(full[5:]) * 1e-3 + 1e-3
    g = full.reshape(-1).copy()
    lib = _oracle.oracle()
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    lib.orc_ensmean(C.c_int(k), C.c_int(nv), C.c_int64(npts), P(g, C.c_double), C.c_int64(1), C.c_int64(npts),
                    C.c_int64(npts * nens))                            # write_ensmean: the mean sits in slot mmean at the CALL
    full_in = g.copy()
    lib.orc_to_perturbations(C.c_int(k), C.c_int(nv), C.c_int64(npts), P(g, C.c_double), C.c_int64(1), C.c_int64(npts),
                             C.c_int64(npts * nens))
    ens = rng.standard_normal((nobs, k + 1))
    ens[:, :k] -= ens[:, :k].mean(axis=1, keepdims=True)
    val = rng.standard_normal(nobs) * 1.5
    infl_mul, infl_min = 1.04, 1.02

    # ---- the oracle's das_letkf: beta, per class obs_local for every point + the loop body
    lib.orc_relax_beta.restype = C.c_double
    bp = OrcBeta(0, 99.0e3, 2000.0, bw, dx, dx, ihalo, ihalo, nlon, nlat)
    beta = np.array([lib.orc_relax_beta(C.byref(bp), C.c_double(rig[p % nij1]), C.c_double(rjg[p % nij1]),
                                        C.c_double(hgt.ravel()[p])) for p in range(npts)])
    classes = [(list(range(0, 5)), 1.0), (list(range(5, nv)), 0.5)]
    lib.orc_obs_local.restype = C.c_int
    gm = g.reshape(nv, nens, npts)
    want = np.full((nv, nens, npts), np.nan)
    counts_first = None
    for vars_, vl_t in classes:
        arr["varloc"] = np.array([1.0, 1.0, vl_t, 1.0])
        t, keep = host_struct(case)
        cap = 4000
        idx, rd, rl, ds = np.zeros(cap, np.int32), np.zeros(cap), np.zeros(cap), np.zeros(cap)
        off, li, lrd, lrl = [0], [], [], []
        for p in range(npts):
            n = lib.orc_obs_local(C.byref(t), C.c_double(rig[p % nij1]), C.c_double(rjg[p % nij1]),
                                  C.c_double(gm[4, k, p]), C.c_double(hgt.ravel()[p]), C.c_int(cap), P(idx, C.c_int32),
                                  P(rd, C.c_double), P(rl, C.c_double), P(ds, C.c_double))
            assert n >= 0
            li.append(idx[:n].copy()); lrd.append(rd[:n].copy()); lrl.append(rl[:n].copy())
            off.append(off[-1] + n)
        if counts_first is None:
            counts_first = np.diff(off)
        mask = sum(1 << v for v in vars_)
        prm = _oracle.DasParams(k=k, nv=nv, det_run=det, infl_adaptive=0, relax_to_inflated_prior=0,
                                relax_alpha=alpha if relax == 1 else 0.0, relax_alpha_spread=alpha if relax == 2 else 0.0,
                                q_update_top=0.0, q_sprd_max=0.0, iv_p=4, iv_q_first=5, iv_q_last=10, nthreads=4, var_mask=mask)
        ref = _oracle.das_points(prm, np.array(off), np.concatenate(li) if off[-1] else np.zeros(0, np.int32),
                                 np.concatenate(lrd) if off[-1] else np.zeros(0), np.concatenate(lrl) if off[-1] else np.zeros(0),
                                 ens, val, beta, np.full(npts * nv, max(infl_mul, infl_min)), g, 1, npts, npts * nens)
        assert ref["rc"] == 0
        ra = ref["anal"].reshape(nv, nens, npts)
        for v in vars_:
            want[v] = ra[v]
    assert counts_first.max() >= 12 + 3 and (counts_first > 0).mean() > 0.5     # the limits bind, most points have observations

    # ---- the Fortran host
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            hdr = [k, det, nij1, nlev, nv, nctype, nobs, k + 1, 1, nlon, nlat, nlon, nlat, ihalo, relax, 0, nid_obs, nobtype,
                   arr["ac_ext"].size]
            f.write(struct.pack("<19i", *hdr))
            f.write(struct.pack("<13d", alpha, infl_mul, infl_min, 0.0, -1.0, bw, 99.0e3, 2000.0, scal["rain_base"], dx, dx,
                                scal["i_org"], scal["j_org"]))
            w = lambda a, dt: f.write(np.ascontiguousarray(a, dtype=dt).tobytes())
            w(var_local.T, "<f8")                     # Fortran var_local(nv3d, 9): column-major
            w(merge.T, "<i4")
            for a in (elm, elm_u, typ, uid_vl, arr["max_nobs"], arr["ngrd_i"], arr["ngrd_j"], arr["ngrdsch_i"], arr["ngrdsch_j"],
                      arr["ngrdext_i"], arr["ngrdext_j"]):
                w(a, "<i4")
            w(arr["ac_off"], "<i8")
            w(arr["ac_ext"], "<i4")
            for a in (arr["hori_loc"], arr["vert_loc"], arr["ob_ri"], arr["ob_rj"], arr["ob_lev"], arr["ob_dat"], arr["ob_err"],
                      ens, val, rig, rjg, hgt, full_in):
                w(a, "<f8")
        r = subprocess.run([DRIVER, fin, fout], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        raw = np.fromfile(fout, dtype="<f8", count=2 * nv * nens * npts + npts * nv)
        nobs_point = np.fromfile(fout, dtype="<i4", offset=8 * (2 * nv * nens * npts + npts * nv))
    got = raw[:nv * nens * npts].reshape(nv, nens, npts)
    pert = raw[nv * nens * npts:2 * nv * nens * npts].reshape(nv, nens, npts)
    work3d = raw[2 * nv * nens * npts:]
    assert np.array_equal(nobs_point, counts_first)                   # obs_local found the same observations at every point
    assert np.array_equal(pert[:, :k], gm[:, :k])                     # the perturbation pass is the reference's subtraction
    assert np.all(work3d == max(infl_mul, infl_min))
    members = list(range(k)) + ([k + 1] if det else [])
    for v in range(nv):
        scale = max(np.abs(gm[v, k]).max(), np.abs(gm[v, :k]).max())
        assert np.abs(got[v, members] - want[v, members]).max() <= 1e-10 * scale, v
    # slot mmean of anal3d = ensmean_grd of the analysis members (letkf.f90:207)
    am = got[:, :k].copy().reshape(-1)
    chk = np.concatenate([got[:, :k], np.zeros((nv, 1, npts))], axis=1).reshape(-1).copy()
    lib.orc_ensmean(C.c_int(k), C.c_int(nv), C.c_int64(npts), P(chk, C.c_double), C.c_int64(1), C.c_int64(npts),
                    C.c_int64(npts * (k + 1)))
    assert np.array_equal(chk.reshape(nv, k + 1, npts)[:, k], got[:, k])
