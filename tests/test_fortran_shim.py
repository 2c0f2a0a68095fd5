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
