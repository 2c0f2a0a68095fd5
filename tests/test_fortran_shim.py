"""The Fortran side of the drop-in boundary: scale-letkf_amd/fortran/common_letkf_amd.f90 keeps the reference's
`MODULE common_letkf` / `letkf_core` interface (common/common_letkf.f90:52-68) and forwards to the C ABI.
CPU: it compiles and links with amdflang.  GPU: a Fortran program calling letkf_core with the reference's own
call forms (scale/letkf/letkf_tools.f90:418-436) reproduces the oracle."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

import _oracle
from __graft_entry__ import PKG_DIR, load_package
from _cases import core_case, relerr

FDIR = os.path.join(PKG_DIR, "fortran")
DRIVER = os.path.join(FDIR, "build", "shim_driver")
HAVE_FC = os.path.exists("/opt/rocm/bin/amdflang")


def build_shim():
    load_package().build()
    subprocess.check_call(["make", "-C", FDIR], stdout=subprocess.DEVNULL)
    return DRIVER


@pytest.mark.skipif(not HAVE_FC, reason="amdflang not present")
def test_shim_compiles_and_keeps_the_reference_interface():
    build_shim()
    assert os.path.exists(DRIVER)
    src = open(os.path.join(FDIR, "common_letkf_amd.f90")).read()
    # same module / procedure / dummy-argument names as common/common_letkf.f90:52
    assert "MODULE common_letkf" in src
    for name in ("ne", "nobs", "nobsl", "hdxb", "rdiag", "rloc", "dep", "parm_infl", "trans", "transm", "pao",
                 "rdiag_wloc", "infl_update", "depd", "transmd"):
        assert name in src
    out = subprocess.check_output(["nm", "-D", os.path.join(PKG_DIR, "lib", "libletkf_amd.so")]).decode()
    assert " T letkf_core_c" in out


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE_FC, reason="amdflang not present")
@pytest.mark.parametrize("k,n,flags", [(20, 37, 0b000), (50, 200, 0b111), (50, 0, 0b110), (64, 300, 0b011),
                                       (100, 150, 0b100)])
def test_fortran_caller_matches_oracle(k, n, flags):
    drv = build_shim()
    iupd, det, rtps = bool(flags & 1), bool(flags & 2), bool(flags & 4)
    c = core_case(k, n, seed=900 + k + n, nobs=n + 11, rdiag_wloc=True, infl=1.03, with_det=True)
    exp = _oracle.letkf_core("oracle", k, c["nobs"], n, c["hdxb"], c["rdiag"], c["rloc"], c["dep"], c["infl"],
                             want_transm=True, want_pao=rtps, rdiag_wloc=True, infl_update=iupd,
                             depd=c["depd"] if det else None, want_transmd=det)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<4i", k, c["nobs"], n, flags))
            f.write(struct.pack("<d", c["infl"]))
            for a in (np.asfortranarray(c["hdxb"]).T, c["rdiag"], c["rloc"], c["dep"], c["depd"]):
                f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
        r = subprocess.run([drv, fin, fout], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        raw = np.fromfile(fout, dtype="<f8")
    infl = raw[0]
    trans = raw[1:1 + k * k].reshape(k, k).T
    transm = raw[1 + k * k:1 + k * k + k]
    pao = raw[1 + k * k + k:1 + 2 * k * k + k].reshape(k, k).T
    transmd = raw[1 + 2 * k * k + k:]
    assert relerr(trans, exp["trans"]) <= 1e-11
    assert np.abs(transm - exp["transm"]).max() <= 1e-11 * max(1.0, np.abs(exp["transm"]).max())
    if rtps:
        assert relerr(pao, exp["pao"]) <= 1e-11
    if det:
        assert np.abs(transmd - exp["transmd"]).max() <= 1e-11 * max(1.0, np.abs(exp["transmd"]).max())
    assert abs(infl - exp["parm_infl"]) <= 1e-12


DAS_DRIVER = os.path.join(FDIR, "build", "das_driver")


@pytest.mark.skipif(not HAVE_FC, reason="amdflang not present")
def test_fortran_api_module_mirrors_the_c_struct():
    """letkf_amd_api.f90: TYPE(letkf_das_args), BIND(C) must list the C struct's fields in the C order"""
    import re
    build_shim()
    assert os.path.exists(DAS_DRIVER)
    hdr = open(os.path.join(PKG_DIR, "..", "include", "letkf_amd.h")).read()
    body = hdr[:hdr.index("} letkf_das_args;")]
    body = body[body.rindex("typedef struct {") + len("typedef struct {"):]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    c_fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl or decl.startswith("typedef"):
            continue
        names = decl.replace("*", " ").split(",")
        c_fields.append(names[0].split()[-1])
        c_fields += [n.strip() for n in names[1:]]
    src = open(os.path.join(FDIR, "letkf_amd_api.f90")).read()
    t = src[src.index("TYPE, BIND(C) :: letkf_das_args"):src.index("END TYPE letkf_das_args")]
    f_fields = []
    for line in t.splitlines()[1:]:
        if "::" in line:
            f_fields += [n.strip() for n in line.split("::")[1].split(",")]
    assert f_fields == c_fields


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE_FC, reason="amdflang not present")
@pytest.mark.parametrize("k,det,relax,alpha", [(20, 0, 2, 0.9), (50, 1, 1, 0.6), (50, 0, 0, 0.0)])
def test_fortran_batched_loop_body_matches_oracle(k, det, relax, alpha):
    """INTEGRATION.md level 2 from Fortran: das_driver.f90 uploads a slab, runs ensmean + perturbation pass + ONE
    letkf_das_points_dev call through TYPE(letkf_das_args), BIND(C), and must reproduce the oracle's loop body."""
    import ctypes as C
    from _cases import das_case
    build_shim()
    nv, npts = 11, 37
    c = das_case(k=k, nv=nv, npts=npts, nobs_tot=400, n_mean=70, seed=500 + k, det_run=bool(det), infl0=1.04)
    nens = c["nens"]
    rng = np.random.default_rng(k)
    full = rng.normal(1.0, 1.0, (nv, nens, npts))            # full members; slot k (mean) is overwritten by the driver
    full[4] = rng.uniform(3.0e4, 1.0e5, (nens, npts))
    # the oracle's view: ensmean_grd + perturbation pass, then the loop body
    g = full.reshape(-1).copy()
    lib = _oracle.oracle()
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    lib.orc_ensmean(C.c_int(k), C.c_int(nv), C.c_int64(npts), P(g, C.c_double), C.c_int64(1), C.c_int64(npts),
                    C.c_int64(npts * nens))
    lib.orc_to_perturbations(C.c_int(k), C.c_int(nv), C.c_int64(npts), P(g, C.c_double), C.c_int64(1),
                             C.c_int64(npts), C.c_int64(npts * nens))
    prm = _oracle.DasParams(k=k, nv=nv, det_run=det, infl_adaptive=0, relax_to_inflated_prior=0,
                            relax_alpha=alpha if relax == 1 else 0.0, relax_alpha_spread=alpha if relax == 2 else 0.0,
                            q_update_top=0.0, q_sprd_max=0.0, iv_p=4, iv_q_first=5, iv_q_last=10, nthreads=2)
    ref = _oracle.das_points(prm, c["obs_off"], c["obs_idx"], c["rdiag"], c["rloc"], c["ensval"], c["dep"], c["beta"],
                             c["infl"], g, 1, npts, npts * nens)
    assert ref["rc"] == 0
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<7i", k, nv, npts, c["ensval"].shape[0], c["kld"], det, relax))
            f.write(struct.pack("<d", alpha))
            f.write(np.ascontiguousarray(c["obs_off"], dtype="<i8").tobytes())
            f.write(np.ascontiguousarray(c["obs_idx"], dtype="<i4").tobytes())
            for a in (c["rdiag"], c["rloc"], c["ensval"], c["dep"], c["beta"], c["infl"], full):
                f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
        r = subprocess.run([DAS_DRIVER, fin, fout], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        got = np.fromfile(fout, dtype="<f8").reshape(nv, nens, npts)
    want = ref["anal"].reshape(nv, nens, npts)
    x = g.reshape(nv, nens, npts)
    members = list(range(k)) + ([k + 1] if det else [])
    for v in range(nv):
        scale = max(np.abs(x[v, k]).max(), np.abs(x[v, :k]).max())
        assert np.abs(got[v, members] - want[v, members]).max() <= 1e-10 * scale
