"""CPU checks of the oracle's obs_local restatement (scale/letkf/letkf_tools.f90:1325-1759): against a brute-force
search over ALL observations (no sorting mesh), and of the selection semantics against the reference's own
QUICKSELECT_arg / QUICKSELECT_desc_arg (common/common_sort.f90:341,404) compiled into oracle/_ref."""
import ctypes as C

import numpy as np
import pytest

import _oracle
from _search import build_case, oracle_lists

DZ = float(np.float32(3.651483717))
DZ2 = float(np.float32(13.33333333))


def brute(case, i):
    a, s, p = case["arr"], case["scal"], case["pts"]
    ri, rj, rlev, rz = p["ri"][i], p["rj"][i], p["rlev"][i], p["rz"][i]
    res = []
    for g in case["groups"]:
        cand = []
        for ic in g:
            for row in range(case["ctype_rows"][ic], case["ctype_rows"][ic + 1]):
                vm, vl = a["vmode"][ic], a["vert_loc"][ic]
                if vm == 1:
                    ndv = abs(a["ob_lev"][row] - rz) / vl
                elif vm == 2:
                    ndv = abs(np.log(a["ob_dat"][row]) - np.log(rlev)) / vl
                else:
                    ndv = abs(np.log(a["ob_lev"][row]) - np.log(rlev)) / vl
                if ndv > DZ:
                    continue
                ndh = np.hypot((ri - a["ob_ri"][row]) * s["dx"], (rj - a["ob_rj"][row]) * s["dy"]) / a["hori_loc"][ic]
                if ndh > DZ:
                    continue
                nd = ndh * ndh + ndv * ndv
                if nd > DZ2:
                    continue
                rl = a["varloc"][ic] * np.exp(-0.5 * nd)
                cand.append((row, a["ob_err"][row] ** 2 / rl, rl, nd))
        nmax = a["max_nobs"][g[0]]
        if nmax > 0 and len(cand) > nmax:
            c = s["criterion"]
            key = (lambda x: x[3]) if c == 1 else (lambda x: -x[2]) if c == 2 else (lambda x: x[1])
            cand = sorted(cand, key=key)[:nmax]
        res.append(cand)
    return res


@pytest.mark.parametrize("max_nobs,criterion", [((0, 0, 0, 0), 1), ((25, 25, 10, 5), 1), ((25, 25, 10, 5), 2),
                                                ((25, 25, 10, 5), 3)])
def test_oracle_obs_local_vs_brute_force(max_nobs, criterion):
    case = build_case(7, max_nobs=max_nobs, criterion=criterion)
    lists = oracle_lists(case)
    nz = 0
    for i in range(0, len(lists), 7):
        idx, rd, rl, ds = lists[i]
        flat = sorted(c for g in brute(case, i) for c in g)
        assert sorted(idx.tolist()) == [c[0] for c in flat], i
        order = np.argsort(idx)
        assert np.allclose(rl[order], [c[2] for c in flat], rtol=1e-13, atol=0)
        assert np.allclose(rd[order], [c[1] for c in flat], rtol=1e-13, atol=0)
        if max_nobs[0] == 0:
            # no limit: reference order = group, member ctype, mesh row j, table row  ->  ascending row inside a ctype
            for ic in range(4):
                sel = idx[(idx >= case["ctype_rows"][ic]) & (idx < case["ctype_rows"][ic + 1])]
                assert np.all(np.diff(sel) > 0)
        nz += len(flat)
    assert nz > 200


@pytest.mark.skipif(_oracle.ref() is None, reason="oracle/_ref not built")
@pytest.mark.parametrize("desc", [0, 1])
def test_select_matches_reference_quickselect(desc):
    rng = np.random.default_rng(3)
    lib, ref = _oracle.oracle(), _oracle.ref()
    for n, K in [(10, 3), (200, 25), (1000, 100), (57, 56), (5, 5)]:
        a = rng.standard_normal(4 * n)
        x = rng.choice(4 * n, size=n, replace=False).astype(np.int32)
        x_ref = (x + 1).astype(np.int32)      # Fortran indices
        ref.ref_quickselect_arg(C.c_int(a.size), a.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(n),
                                x_ref.ctypes.data_as(C.POINTER(C.c_int)), C.c_int(1), C.c_int(n), C.c_int(K),
                                C.c_int(desc))
        x_orc = x.copy()
        lib.orc_select_arg(a.ctypes.data_as(C.POINTER(C.c_double)), x_orc.ctypes.data_as(C.POINTER(C.c_int32)),
                           C.c_int(n), C.c_int(K), C.c_int(desc))
        assert set((x_ref[:K] - 1).tolist()) == set(x_orc[:K].tolist())
        assert sorted(x_orc.tolist()) == sorted(x.tolist())
