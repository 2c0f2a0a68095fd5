"""The C restatement (oracle/letkf_oracle.c) against the golden vectors produced by the reference's own
letkf_core (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

import _oracle
from _cases import case_sha, golden_case_list, golden_inputs, probes, relerr

CASES = golden_case_list()
# the oracle follows the reference's operation order, so agreement is at the ulp level; the loose bound
# only absorbs FMA-contraction differences between gcc and flang
TOL = 1e-13


def check_against_golden(golden, c, r, tol_t, tol_infl=1e-12):
    """Shared by the oracle test and the GPU parity tests.  Tolerances: SURVEY.md section 8(c)."""
    nm, k = c["name"], c["k"]
    big = k > 100
    pr = probes(k)
    for key in ("trans", "pao"):
        if not c["transm" if key == "trans" else "pao"] and key == "pao":
            continue
        if key == "pao" and not c["pao"]:
            continue
        if big:
            scale = float(golden[f"{nm}/{key}_absmax"][0])
            e1 = np.abs(r[key] @ pr - golden[f"{nm}/{key}_probe"]).max() / (scale * np.sqrt(k))
            e2 = np.abs(np.diag(r[key]) - golden[f"{nm}/{key}_diag"]).max() / scale
            assert max(e1, e2) <= tol_t, (nm, key, e1, e2)
        else:
            assert relerr(r[key], golden[f"{nm}/{key}"]) <= tol_t, (nm, key, relerr(r[key], golden[f"{nm}/{key}"]))
    if c["transm"]:
        g = golden[f"{nm}/transm"]
        # w-bar is compared against the scale of T's rows it is added to (it can be exactly 0 for n = 0)
        den = max(np.abs(g).max(), 1e-300)
        assert np.abs(r["transm"] - g).max() / den <= tol_t or np.abs(r["transm"] - g).max() <= 1e-14, (nm, "transm")
    if c["det"] and c["transm"] is not None and f"{nm}/transmd" in golden.files:
        g = golden[f"{nm}/transmd"]
        den = max(np.abs(g).max(), 1e-300)
        assert np.abs(r["transmd"] - g).max() / den <= tol_t or np.abs(r["transmd"] - g).max() <= 1e-14, (nm, "transmd")
    assert abs(r["parm_infl"] - float(golden[f"{nm}/parm_infl"][0])) <= tol_infl, (nm, "parm_infl")


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_reference_golden(golden, c):
    inp = golden_inputs(c)
    sha = bytes(golden[c["name"] + "/sha"]).hex()
    assert case_sha(inp) == sha, "input generator drifted from the committed fixture"
    r = _oracle.letkf_core("oracle", c["k"], inp["nobs"], c["n"], inp["hdxb"], inp["rdiag"], inp["rloc"], inp["dep"],
                           inp["infl"], want_transm=c["transm"], want_pao=c["pao"], rdiag_wloc=c["rdiag_wloc"],
                           infl_update=c["infl_update"], depd=inp["depd"], want_transmd=c["det"])
    assert r["rc"] == 0
    check_against_golden(golden, c, r, TOL)


def test_golden_covers_matrix(golden):
    assert set(golden["names"]) == {c["name"] for c in CASES}
