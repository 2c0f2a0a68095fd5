"""The C restatement against the reference's own Fortran compiled into oracle/_ref.  Runs only where that build
exists (the build container; /root/reference does not travel to the GPU box)."""
import numpy as np
import pytest

import _oracle
from _cases import core_case, relerr

pytestmark = pytest.mark.skipif(_oracle.ref() is None, reason="oracle/_ref not built (no /root/reference here)")


@pytest.mark.parametrize("k,n", [(2, 1), (5, 3), (20, 500), (50, 200), (50, 49), (64, 64), (100, 30)])
@pytest.mark.parametrize("wloc,iu,det", [(True, True, True), (False, False, False)])
def test_letkf_core(k, n, wloc, iu, det):
    c = core_case(k, n, seed=4242 + 7 * k + n, nobs=n + 3, rdiag_wloc=wloc, infl=1.05, with_det=det)
    kw = dict(rdiag_wloc=wloc, infl_update=iu, depd=c["depd"], want_transmd=det)
    a = _oracle.letkf_core("ref", k, c["nobs"], n, c["hdxb"], c["rdiag"], c["rloc"], c["dep"], c["infl"], **kw)
    b = _oracle.letkf_core("oracle", k, c["nobs"], n, c["hdxb"], c["rdiag"], c["rloc"], c["dep"], c["infl"], **kw)
    for key in ("trans", "pao", "transm"):
        assert relerr(b[key], a[key]) <= 1e-13, key
    if det:
        assert relerr(b["transmd"], a["transmd"]) <= 1e-13
    assert abs(a["parm_infl"] - b["parm_infl"]) <= 1e-14


@pytest.mark.parametrize("n", [2, 7, 50, 120])
def test_mtx_eigen(n):
    rng = np.random.default_rng(n)
    m = rng.standard_normal((n, 3 * n))
    a = m @ m.T + n * np.eye(n)
    w1, v1, nr1, _ = _oracle.mtx_eigen("ref", a)
    w2, v2, nr2, rc = _oracle.mtx_eigen("oracle", a)
    assert rc == 0 and nr1 == nr2 == n
    assert np.all(np.diff(w1) <= 0), "descending order (common/common_mtx.f90:93-96)"
    assert relerr(w2, w1) <= 1e-14
    assert relerr(v2, v1) <= 1e-12
